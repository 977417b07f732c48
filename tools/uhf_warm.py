#!/usr/bin/env python3
"""UHF / UKS restarted from a slightly perturbed converged density (what a geometry step looks like): fast vs plain loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from mi355scf.mole import Mole
from mi355scf.uhf import UHF
from mi355scf.uks import UKS
from mi355scf import fixtures
xc = sys.argv[1] if len(sys.argv) > 1 else None
mol = Mole(atom=fixtures.BENZENE, basis="cc-pVTZ", verbose=0, charge=1, spin=1).build()
mf = UKS(mol) if xc else UHF(mol)
if xc:
    mf.xc = xc
mf.conv_tol = 1e-9
mf.kernel()
dm = mf.make_rdm1()
rng = np.random.default_rng(0)
noise = rng.normal(size=dm.shape) * 2e-3
dm0 = dm + 0.5 * (noise + noise.transpose(0, 2, 1))
for fast in (True, False, True, False):
    mf.fast_loop = fast
    torch.cuda.synchronize(); t0 = time.time()
    e = mf.kernel(dm0=dm0)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"fast_loop={fast}: E = {e:.10f} cycles {mf.cycles} kernel() {dt * 1e3:.1f} ms, loop {mf.timing['loop_seconds'] * 1e3:.1f} ms = {mf.timing['loop_seconds'] / max(mf.cycles, 1) * 1e3:.2f} ms/cycle", flush=True)
