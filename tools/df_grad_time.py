"""Time the density-fitted SCF + analytic gradient of one molecule beside the exact-integral ones.
   python tools/df_grad_time.py benzene cc-pVTZ [B3LYP]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "computational-chemistry-ai_amd/python")
from mi355scf import fixtures, smiles_fixtures   # noqa: E402
from pyscf import gto, scf, dft   # noqa: E402


def _atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))


name, basis = sys.argv[1], sys.argv[2]
xc = sys.argv[3] if len(sys.argv) > 3 else None
atom = fixtures.BENZENE if name == "benzene" else _atoms("CC(C)Cc1ccc(cc1)C(C)C(=O)O")
mol = gto.Mole()
mol.atom, mol.basis, mol.verbose = atom, basis, 0
mol.build()


def run(fit):
    mf = dft.RKS(mol, xc=xc) if xc else scf.RHF(mol)
    if fit:
        mf = mf.density_fit()
    t0 = time.time()
    e = mf.kernel()
    torch.cuda.synchronize()
    t1 = time.time()
    g = mf.nuc_grad_method().kernel()
    torch.cuda.synchronize()
    t2 = time.time()
    return e, g, t1 - t0, t2 - t1, mf


for fit in (True, False, True):
    e, g, ts, tg, mf = run(fit)
    extra = f" naux {mf.with_df.naux} {mf.with_df.grad_timing}" if fit else ""
    print(f"fit={fit}: E {e:.8f}  scf {ts:.2f} s  gradient {tg:.2f} s  |g|max {np.abs(g).max():.5f} sum {np.abs(g.sum(axis=0)).max():.1e}{extra}", flush=True)
    print("   timing", {k: round(v, 3) for k, v in mf.timing.items() if isinstance(v, float)}, "cycles", getattr(mf, "cycles", None), flush=True)
    if fit:
        gf = g
    else:
        print(f"   max |g_fit - g_exact| {np.abs(gf - g).max():.2e}", flush=True)
