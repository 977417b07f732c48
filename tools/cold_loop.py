#!/usr/bin/env python3
"""Host-side time of every cycle of a COLD benzene/cc-pVTZ RHF without per-cycle device synchronisation (what kernel() runs),
with the purification path each cycle took.  python tools/cold_loop.py [--no-cold-pipeline]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from mi355scf.mole import Mole
from mi355scf.scf import RHF
from mi355scf import fixtures
mol = Mole(atom=fixtures.BENZENE, basis="cc-pVTZ", verbose=0).build()
for rep in range(2):
    mf = RHF(mol)
    if "--no-cold-pipeline" in sys.argv:
        mf.cold_pipeline = False
    if "--no-trace-plan" in sys.argv:
        mf.sp2_trace_plan = False
    for a in sys.argv:
        if a.startswith("--trace-gnorm="):
            mf.sp2_trace_plan_gnorm = float(a.split("=")[1])
    if rep == 1:
        mf._eng = eng      # second object: libraries warm, ERIs resident; still a cold OBJECT (no plan)
    st = mf._start(None)
    eng = mf.engine
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for i in range(9):
        t0 = time.perf_counter(); had_front = "front" in st; mf._step(st)
        print(f"rep {rep} cycle {i + 1}: {1e3 * (time.perf_counter() - t0):7.2f} ms  front={had_front} planned={mf._sp2_planned_pass} iters={mf._sp2_iters} "
              f"redo={getattr(mf, 'n_redo', 0)} |g|={st['gnorm']:.2e} E={st['e_tot']:.10f}", flush=True)
    torch.cuda.synchronize()
    print(f"rep {rep}: 9 cycles {1e3 * (time.perf_counter() - t_all):.2f} ms  plan_len={getattr(mf, '_sp2_plan_len', None)} paths={mf.path_counts}")
