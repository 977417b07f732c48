#!/usr/bin/env python3
"""Where one geometry step of config 5 (ibuprofen B3LYP/def2-TZVP) spends its wall time: scanner call on a displaced geometry."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from pyscf import gto, dft
from mi355scf import smiles_fixtures
sym, xyz = smiles_fixtures.TABLE["CC(C)Cc1ccc(cc1)C(C)C(=O)O"]()
mol = gto.Mole(); mol.atom = "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz)); mol.basis = "def2-TZVP"; mol.verbose = 0; mol.build()
mf = dft.RKS(mol); mf.xc = "B3LYP"; mf = mf.to_gpu()
gs = mf.nuc_grad_method().as_scanner(); gs.grad_dtol = 1e-10
t0 = time.time(); e, g = gs(mol); torch.cuda.synchronize(); print(f"first point: {time.time() - t0:.2f} s  (SCF timing {mf.timing})")
R = mol.atom_coords()
rng = np.random.default_rng(1)
for step in range(3):
    m2 = mol.set_geom_(R + 0.01 * rng.standard_normal(R.shape), unit="Bohr", inplace=False)
    t0 = time.time(); e, g = gs(m2); torch.cuda.synchronize(); dt = time.time() - t0
    tm = dict(mf.timing)
    print(f"step {step}: {dt:.2f} s  cycles {mf.cycles}  SCF total {tm.get('total_seconds', 0):.2f} (setup {tm.get('setup_seconds', 0):.2f}, eri {tm.get('eri_seconds', 0):.2f}, loop {tm.get('loop_seconds', 0):.2f})  gradient+rest {dt - tm.get('total_seconds', 0):.2f}  gradient parts {({k: round(v, 3) for k, v in getattr(gs.g, 'timing', {}).items()})}", flush=True)
if "--cprofile" in sys.argv:
    import cProfile, pstats
    m2 = mol.set_geom_(R + 0.01 * rng.standard_normal(R.shape), unit="Bohr", inplace=False)
    pr = cProfile.Profile(); pr.enable(); e, g = gs(m2); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
