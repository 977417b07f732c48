#!/usr/bin/env python3
"""How sparse are the AO values on the XC grid blocks?  Fraction of AOs with max_g sqrt(w_g) |phi(g)| above a cutoff, per block
of consecutive grid points (atom-ordered Becke grid), for several block sizes.   python tools/ao_sparsity.py [ibuprofen|benzene|c60]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from pyscf import gto, dft
from mi355scf import smiles_fixtures, fixtures
name = sys.argv[1] if len(sys.argv) > 1 else "ibuprofen"
key, basis = {"ibuprofen": ("CC(C)Cc1ccc(cc1)C(C)C(=O)O", "def2-TZVP"), "benzene": ("c1ccccc1", "cc-pVTZ"), "c60": ("C60", "6-31G*")}[name]
sym, xyz = smiles_fixtures.TABLE[key]()
mol = gto.Mole(); mol.atom = "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz)); mol.basis = basis; mol.verbose = 0; mol.build()
mf = dft.RKS(mol); mf.xc = "B3LYP"; mf = mf.to_gpu()
mf.grids.build()
eng = mf.engine
coords, weights = mf.grids.coords, mf.grids.weights
ng, n = coords.shape[0], mol.nao
print(name, "nao", n, "grid points", ng)
for B in (32768, 8192, 2048):
    res = {c: [] for c in (1e-6, 1e-8, 1e-10)}
    for p0 in range(0, ng, 32768):
        ao = eng.eval_ao(coords[p0:p0 + 32768], deriv=1)          # [4, nao, npts]
        a = ao.abs().amax(dim=0) * weights[p0:p0 + 32768].abs().sqrt()[None, :]
        for q0 in range(0, a.shape[1], B):
            m = a[:, q0:q0 + B].amax(dim=1)
            for c in res:
                res[c].append(float((m > c).double().mean()))
    print("block", B, {f"{c:g}": (round(float(np.mean(v)), 3), "rms-fraction", round(float(np.sqrt(np.mean(np.square(v)))), 3)) for c, v in res.items()})
