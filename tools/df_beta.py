#!/usr/bin/env python3
"""Fitting error of the generated even-tempered auxiliary basis as a function of its ratio beta.  python tools/df_beta.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
from pyscf import gto, scf
from mi355scf import fixtures, df
for name, atom, basis in (("h2co", fixtures.H2CO, "6-31G(d)"), ("benzene", fixtures.BENZENE, "cc-pVDZ")):
    mol = gto.Mole(); mol.atom, mol.basis, mol.verbose = atom, basis, 0; mol.build()
    e0 = scf.RHF(mol).kernel()
    for beta in (2.0, 1.8, 1.6, 1.45):
        aux = df.even_tempered_aux(mol, beta=beta)
        mf = scf.RHF(mol).density_fit(auxbasis=aux)
        e1 = mf.kernel()
        print(json.dumps(dict(mol=name, basis=basis, beta=beta, naux=mf.with_df.naux, err=e1 - e0)), flush=True)
