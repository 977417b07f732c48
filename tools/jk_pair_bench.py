#!/usr/bin/env python3
"""Two densities per J/K build (UHF / UKS): one pass with two waves per work item (jk_pair=1) vs one pass per density.
   python tools/jk_pair_bench.py [basis ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from mi355scf.mole import Mole
from mi355scf.engine import Engine
from mi355scf import fixtures
for basis in (sys.argv[1:] or ["cc-pVDZ", "cc-pVTZ"]):
    mol = Mole(atom=fixtures.BENZENE, basis=basis, verbose=0).build()
    n = mol.nao
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(n, n)), rng.normal(size=(n, n))
    D2 = torch.as_tensor(np.stack([a + a.T, b + b.T]), device="cuda")
    eng = Engine(mol)
    eng.prepare_eri(1e-13)
    out = {}
    for pair in (0, 1, 0, 1):
        eng.set_option("jk_pair", pair)
        J, K = eng.get_jk(D2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            J, K = eng.get_jk(D2)
        e1.record(); torch.cuda.synchronize()
        out.setdefault(pair, []).append(round(e0.elapsed_time(e1) / 30, 4))
        out[f"sum{pair}"] = (float(J.sum()), float(K.sum()))
    J1, K1 = eng.get_jk(D2[0]); J2, K2 = eng.get_jk(D2[1])
    eng.set_option("jk_pair", 1)
    J, K = eng.get_jk(D2)
    print(json.dumps(dict(basis=basis, two_passes_ms=out[0], pair_ms=out[1], max_dev_J=float(max((J[0] - J1).abs().max(), (J[1] - J2).abs().max())),
                          max_dev_K=float(max((K[0] - K1).abs().max(), (K[1] - K2).abs().max())))), flush=True)
    eng.close()
