#!/usr/bin/env python3
"""SCF convergence trace (cycles, energies) of one molecule with the step pipeline on / off.  python tools/conv_check.py ibuprofen def2-TZVP HF"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
from mi355scf.mole import Mole
from mi355scf.scf import RHF
from mi355scf.dft import RKS
from mi355scf import smiles_fixtures, fixtures
name, basis, method = sys.argv[1], sys.argv[2], sys.argv[3]
def _atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))
atom = {"benzene": lambda: fixtures.BENZENE, "ibuprofen": lambda: _atoms("CC(C)Cc1ccc(cc1)C(C)C(=O)O")}[name]()
mol = Mole(atom=atom, basis=basis, verbose=0).build()
eng = None
for pipe, eig in ((True, "sp2"), (False, "sp2"), (False, "eigh")):
    mf = RHF(mol) if method == "HF" else RKS(mol, xc=method)
    if eng is not None:
        mf._eng = eng
    mf.pipeline, mf.eig_method, mf.verbose = pipe, eig, 4
    print(f"--- pipeline={pipe} eig={eig}", flush=True)
    e = mf.kernel()
    eng = mf.engine
    print("cycles", mf.cycles, "E", e, "fock builds", getattr(mf, "n_fock_builds", None), flush=True)
