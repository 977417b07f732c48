#!/usr/bin/env python3
"""Warm RKS cycle time of ibuprofen B3LYP/def2-TZVP (or benzene cc-pVTZ) for the current XC settings (env MI355_XC_BLOCK_GB).
  python tools/rks_cycle.py [ibuprofen|benzene]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from pyscf import gto, dft
from mi355scf import smiles_fixtures, fixtures
name = sys.argv[1] if len(sys.argv) > 1 else "ibuprofen"
if name == "benzene":
    atom, basis = fixtures.BENZENE, "cc-pVTZ"
else:
    sym, xyz = smiles_fixtures.TABLE["CC(C)Cc1ccc(cc1)C(C)C(=O)O"]()
    atom, basis = "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz)), "def2-TZVP"
mol = gto.Mole(); mol.atom = atom; mol.basis = basis; mol.verbose = 0; mol.build()
mf = dft.RKS(mol); mf.xc = "B3LYP"; mf = mf.to_gpu()
e = mf.kernel()
st = mf._start(mf.make_rdm1())
for _ in range(6): mf._step(st)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): mf._step(st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(json.dumps(dict(case=name, xc_block_gb=os.environ.get("MI355_XC_BLOCK_GB", "1.5"), ms_per_cycle=round(dt * 1e3, 3), e_tot=e, e_cycle=st["e_tot"], ngrid=int(mf.grids.size))))
