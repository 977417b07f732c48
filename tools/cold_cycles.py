#!/usr/bin/env python3
"""Wall time of every SCF cycle in a cold process (first-use costs of libraries show up in the first cycles)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
t00 = time.time()
import torch
from mi355scf.mole import Mole
from mi355scf.scf import RHF
from mi355scf.dft import RKS
from mi355scf import fixtures
print(f"imports {time.time() - t00:.3f} s", flush=True)
method = sys.argv[1] if len(sys.argv) > 1 else "B3LYP"
mol = Mole(atom=fixtures.BENZENE, basis="cc-pVTZ", verbose=0).build()
mf = RHF(mol) if method == "HF" else RKS(mol, xc=method)
t0 = time.time(); mf._setup_once(); torch.cuda.synchronize(); print(f"setup (engine, 1e integrals, ERIs) {time.time() - t0:.3f} s", flush=True)
t0 = time.time(); dm0 = mf.get_init_guess(); torch.cuda.synchronize(); print(f"initial guess {time.time() - t0:.3f} s", flush=True)
t0 = time.time(); st = mf._start(dm0); torch.cuda.synchronize(); print(f"_start (first Fock build) {time.time() - t0:.3f} s", flush=True)
for i in range(10):
    t0 = time.time(); mf._step(st); torch.cuda.synchronize()
    print(f"cycle {i + 1}: {1e3 * (time.time() - t0):8.2f} ms  E = {st['e_tot']:.10f}", flush=True)
