#!/usr/bin/env python3
"""Does the J/K launch time drift with how long the device has been busy?  30-rep timings back to back for ~6 s."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from mi355scf.mole import Mole
from mi355scf.engine import Engine
from mi355scf import fixtures
mol = Mole(atom=fixtures.BENZENE, basis="cc-pVTZ", verbose=0).build()
n = mol.nao
rng = np.random.default_rng(0)
a = rng.normal(size=(n, n)); D = torch.as_tensor(a + a.T, device="cuda")
eng = Engine(mol)
eng.prepare_eri(1e-13)
t0 = time.time()
out = []
while time.time() - t0 < 6.0:
    out.append((round(time.time() - t0, 2), round(eng.time_jk_kernel(D, reps=30), 4)))
print(json.dumps(out[:10] + out[10::10]))
time.sleep(3.0)
print("after 3 s idle:", [round(eng.time_jk_kernel(D, reps=30), 4) for _ in range(5)])
