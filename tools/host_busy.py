#!/usr/bin/env python3
"""How much of an SCF cycle of the bench workload the HOST is busy (queueing launches, validating scalars) and how much it
waits for the device: if busy/cycle approaches 1 the cycle is launch bound and a slower host CPU shows up in iter/s.
  python tools/host_busy.py [cc-pVTZ] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from pyscf import gto, scf
from mi355scf import fixtures

basis = sys.argv[1] if len(sys.argv) > 1 else "cc-pVTZ"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mol = gto.Mole(); mol.atom = fixtures.BENZENE; mol.basis = basis; mol.verbose = 0; mol.build()
mf = scf.RHF(mol).to_gpu()
mf.kernel()
st = mf._start(mf.make_rdm1())
for _ in range(10):
    mf._step(st)
wait = [0.0]
orig = torch.cuda.Event.synchronize
def timed(self):
    t = time.perf_counter(); orig(self); wait[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    mf._step(st)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
torch.cuda.Event.synchronize = orig
print("graph", mf.__dict__.get("_fgraph") is not None, "paths", getattr(mf, "path_counts", None), "plan", None if mf._sp2_plan is None else (mf._sp2_plan.shape, mf._sp2_plan_len), "redo", getattr(mf, "n_redo", 0))
print(json.dumps(dict(basis=basis, steps=steps, ms_per_cycle=round(tot / steps * 1e3, 4), host_wait_ms=round(wait[0] / steps * 1e3, 4),
                      host_busy_ms=round((tot - wait[0]) / steps * 1e3, 4), cpus=os.cpu_count(),
                      cpu_model=[l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1])))
