#!/usr/bin/env python3
"""cProfile of the host side of the bench workload's SCF cycle (which Python / ctypes / torch calls the 0.36 ms go to).
  python tools/host_profile.py [cc-pVTZ] [steps]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from pyscf import gto, scf
from mi355scf import fixtures

basis = sys.argv[1] if len(sys.argv) > 1 else "cc-pVTZ"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mol = gto.Mole(); mol.atom = fixtures.BENZENE; mol.basis = basis; mol.verbose = 0; mol.build()
mf = scf.RHF(mol).to_gpu()
mf.kernel()
st = mf._start(mf.make_rdm1())
for _ in range(10):
    mf._step(st)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    mf._step(st)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr)
ps.sort_stats("tottime").print_stats(32)
