#!/usr/bin/env python3
"""Steps / wall of `optimize(mf)` per initial-Hessian model of the internal-coordinate optimiser (`Internals.HESS_MODEL`).
  python tools/opt_variants.py small            # ethanol, acetic acid, benzene: B3LYP/6-31G(d)
  python tools/opt_variants.py ibuprofen        # BASELINE config 5, B3LYP/def2-TZVP
  MODELS=simple,lindh python tools/opt_variants.py ibuprofen"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from pyscf import gto, dft
from pyscf.geomopt.geometric_solver import optimize
from mi355scf import smiles_fixtures, internals

def atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))

which = sys.argv[1] if len(sys.argv) > 1 else "small"
cases = {"small": [("CCO", "6-31G(d)"), ("CC(=O)O", "6-31G(d)"), ("c1ccccc1", "6-31G(d)")],
         "ibuprofen": [("CC(C)Cc1ccc(cc1)C(C)C(=O)O", "def2-TZVP")]}[which]
models = os.environ.get("MODELS", "simple,geometric,lindh").split(",")
for smi, basis in cases:
    for model in models:
        internals.Internals.HESS_MODEL = model
        mol = gto.Mole(); mol.atom = atoms(smi); mol.basis = basis; mol.verbose = 0; mol.build()
        mf = dft.RKS(mol); mf.xc = "B3LYP"; mf = mf.to_gpu()
        mf.kernel()
        cyc, orig = [0], mf.kernel
        def counted(*a, **kw):
            r = orig(*a, **kw); cyc[0] += mf.cycles; return r
        mf.kernel = counted
        steps = []
        t0 = time.time()
        mol_eq = optimize(mf, maxsteps=100, callback=lambda loc: steps.append((loc["step"], float(loc["e_new"]))))
        torch.cuda.synchronize()
        print(json.dumps(dict(molecule=smi, basis=basis, model=model, energy_evals=len(steps), last_step=steps[-1][0],
                              e_final=steps[-1][1], scf_cycles=cyc[0], wall_s=round(time.time() - t0, 1),
                              trace=[round(e - steps[-1][1], 8) for _, e in steps])), flush=True)
        mf._eng = None
        del mf
