"""Time the semi-numerical Hessian (6 N displaced SCF + analytic gradient pairs through ONE scanner object).
   python tools/hessian_time.py benzene cc-pVTZ [B3LYP]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "computational-chemistry-ai_amd/python")
from mi355scf import fixtures, smiles_fixtures   # noqa: E402
from pyscf import gto, scf, dft, hessian   # noqa: E402


def _atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))


name, basis = sys.argv[1], sys.argv[2]
xc = sys.argv[3] if len(sys.argv) > 3 else None
mol = gto.Mole()
mol.atom = fixtures.BENZENE if name == "benzene" else fixtures.H2CO if name == "h2co" else _atoms("CC(C)Cc1ccc(cc1)C(C)C(=O)O")
mol.basis, mol.verbose = basis, 0
mol.build()
mf = dft.RKS(mol, xc=xc) if xc else scf.RHF(mol)
t0 = time.time()
mf.kernel()
t1 = time.time()
h = (hessian.RKS(mf) if xc else hessian.RHF(mf))
H = h.kernel()
torch.cuda.synchronize()
t2 = time.time()
n = mol.natm
Hm = H.transpose(0, 2, 1, 3).reshape(3 * n, 3 * n)
w = np.linalg.eigvalsh(Hm)
print(f"{name}/{basis} {xc or 'HF'}: SCF {t1 - t0:.2f} s, Hessian {t2 - t1:.1f} s for {6 * n} displaced points ({(t2 - t1) / (6 * n):.3f} s each); "
      f"asymmetry {np.abs(Hm - Hm.T).max():.1e}; six smallest |eigenvalues| {np.sort(np.abs(w))[:6]}")
