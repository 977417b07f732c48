"""Timing experiment: phases of eri_grad_contract skipped one at a time (results wrong by construction)."""
import os, sys, time, json
sys.path.insert(0, "computational-chemistry-ai_amd/python")
import numpy as np, torch
from mi355scf.mole import Mole
from mi355scf import smiles_fixtures
from mi355scf.scf import RHF
sym, xyz = smiles_fixtures.TABLE["CC(C)Cc1ccc(cc1)C(C)C(=O)O"]()
mol = Mole(atom="; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz)), basis="def2-TZVP", verbose=0).build()
mf = RHF(mol); mf.kernel()
D = torch.as_tensor(mf.make_rdm1(), device="cuda")
g = torch.zeros(mol.natm, 3, dtype=torch.float64, device="cuda")
mf.engine.set_option("grad_dtol", 1e-10)
for a, live in ((0, 1), (0, 0), (0, 1), (0, 0), (0, 1), (63, 1), (63, 0)):
    mf.engine.set_option("grad_ablate", a)
    mf.engine.set_option("grad_live", live)
    g.zero_()
    torch.cuda.synchronize(); t0 = time.time(); mf.engine.grad_eri(D, 0.2, g); torch.cuda.synchronize()
    print(f"ablate {a:3d} live {live}: grad_eri {time.time() - t0:.3f} s  sum|g| {float(g.abs().sum()):.10f}", flush=True)
