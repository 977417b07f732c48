"""Planned purification: one launch per pass vs one resident launch (HIP events around 200 repetitions).
usage: python tools/sp2_persist_bench.py [cc-pVDZ|cc-pVTZ]"""
import sys, json
import numpy as np, torch
sys.path.insert(0, "computational-chemistry-ai_amd/python")
from mi355scf.mole import Mole
from mi355scf.engine import Engine
from mi355scf import fixtures, sp2plan

basis = sys.argv[1] if len(sys.argv) > 1 else "cc-pVTZ"
eng = Engine(Mole(atom=fixtures.BENZENE, basis=basis, verbose=0).build())
n, nocc = eng.nao, 21
rng = np.random.default_rng(0)
q, _ = np.linalg.qr(rng.normal(size=(n, n)))
e = np.sort(np.concatenate([rng.uniform(-11.3, -0.33, nocc), rng.uniform(0.14, 25.0, n - nocc)]))
F = (q * e) @ q.T
Fd = torch.as_tensor(0.5 * (F + F.T), device=eng.device)
coef = sp2plan.plan(*sp2plan.bounds_from_spectrum(e, nocc))
A = torch.zeros(2, n, n, dtype=torch.float64, device=eng.device); B = torch.zeros_like(A)
tr = torch.zeros(64 * 80, dtype=torch.float64, device=eng.device)
res = {}
for persist in (0, 1, 2, 0, 1, 2):
    eng.set_option("sp2_persist", persist)
    for _ in range(20): eng.sp2_iterate_planned(Fd, A, B, coef, tr, out_scale=2.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(200): eng.sp2_iterate_planned(Fd, A, B, coef, tr, out_scale=2.0)
    b.record(); torch.cuda.synchronize()
    res.setdefault(persist, []).append(round(a.elapsed_time(b) / 200 * 1e3, 1))
print(json.dumps({"basis": basis, "n": n, "passes": int(coef.shape[0]), "per_pass_launch_us": res[0], "resident_us": res[1], "resident_coherent_us": res[2]}))
