#!/usr/bin/env python3
"""At a converged Fock matrix: occupied projector by diagonalisation vs the purification paths (difference, traces, energy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from mi355scf.mole import Mole
from mi355scf.scf import RHF
from mi355scf import smiles_fixtures
def _atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))
mol = Mole(atom=_atoms("CC(C)Cc1ccc(cc1)C(C)C(=O)O"), basis="def2-TZVP", verbose=0).build()
mf = RHF(mol); mf.eig_method = "eigh"
st = mf._start()
for _ in range(16):
    mf._step(st)
fo = st["fo"].clone(); nocc = st["nocc"]
e, c = torch.linalg.eigh(fo)
P = c[:, :nocc] @ c[:, :nocc].T
print("gap", float(e[nocc] - e[nocc - 1]), "homo", float(e[nocc - 1]), "lumo", float(e[nocc]), "range", float(e[0]), float(e[-1]))
mf.eig_method = "sp2"
mf._sp2_replan(e, nocc)
print("plan steps", mf._sp2_plan.shape[0] - 1)
X2, tr = mf._sp2_planned_async(fo, nocc)
X = 0.5 * X2
d = X - P
print("planned gemm: |X-P|_F", float(d.norm()), "max", float(d.abs().max()), "tr", tr.cpu().numpy(), "tr(F d)", float((fo * d).sum()), "asym", float((X - X.T).abs().max()))
D2 = mf._density_sp2(fo, nocc, orth=True)
d = 0.5 * D2 - P
print("checked sp2 : |X-P|_F", float(d.norm()), "max", float(d.abs().max()), "tr(F d)", float((fo * d).sum()))
Li = mf._Linv
for label, XX in (("eigh", P), ("planned", X), ("checked", 0.5 * D2)):
    dm = (Li.T @ (2 * XX) @ Li).contiguous()
    es = []
    for _ in range(4):
        part = torch.empty(mf.engine.reduce_blocks, dtype=torch.float64, device=dm.device)
        F, _x = mf._fock_energy(dm, part)
        es.append(float(part.sum()))
    print(label, "E_elec", ["%.11f" % v for v in es])
