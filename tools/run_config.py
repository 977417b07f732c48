#!/usr/bin/env python3
"""Run one BASELINE config through the drop-in surface and print timings (used for DESIGN.md numbers).
  python tools/run_config.py benzene cc-pVTZ B3LYP"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from pyscf import gto, scf, dft
from mi355scf import fixtures

name, basis, method = sys.argv[1], sys.argv[2], sys.argv[3]
atom = {"benzene": fixtures.BENZENE, "h2co": fixtures.H2CO, "h2o": fixtures.H2O}[name]
mol = gto.Mole(); mol.atom = atom; mol.basis = basis; mol.verbose = 4; mol.build()
t0 = time.time()
mf = scf.RHF(mol) if method == "HF" else dft.RKS(mol)
if method != "HF":
    mf.xc = method
mf = mf.to_gpu()
e = mf.kernel()
torch.cuda.synchronize()
wall = time.time() - t0
st = mf.engine.stats()
ms = mf.engine.time_jk_kernel(mf._dm, reps=10)
n = mol.nao
alg = 8.0 * st["n_unique_eri"] + 24.0 * n * n
print(json.dumps(dict(config=f"{name} {method}/{basis}", nao=n, e_tot=e, converged=bool(mf.converged), cycles=mf.cycles,
                      wall_s=wall, timing=mf.timing, eri=st, jk_ms=ms, jk_alg_GBps=alg / ms / 1e6,
                      jk_stored_GBps=st["stored_bytes"] / ms / 1e6,
                      ngrids=getattr(getattr(mf, "grids", None), "size", 0))))
