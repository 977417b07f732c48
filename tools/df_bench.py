#!/usr/bin/env python3
"""Density-fitted J/K: fitting error of the generated auxiliary basis and GEMM rate.  python tools/df_bench.py [basis ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from pyscf import gto, scf
from mi355scf import fixtures
for name, atom, basis in (("h2co", fixtures.H2CO, "6-31G(d)"), ("benzene", fixtures.BENZENE, "cc-pVDZ"), ("benzene", fixtures.BENZENE, "cc-pVTZ")):
    mol = gto.Mole(); mol.atom, mol.basis, mol.verbose = atom, basis, 0; mol.build()
    e0 = scf.RHF(mol).kernel()
    mf = scf.RHF(mol).density_fit()
    e1 = mf.kernel()
    d = mf.with_df
    n, na = mol.nao, d.naux
    dm = torch.as_tensor(mf.make_rdm1(), device="cuda")
    for _ in range(2):
        d.get_jk(dm)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5):
        d.get_jk(dm)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    flops = 4.0 * n * n * na + 4.0 * n ** 3 * na
    print(json.dumps(dict(mol=name, basis=basis, nao=n, naux=na, e_exact=e0, e_df=e1, err=e1 - e0, df_build_s=mf.timing.get("df_seconds"),
                          jk_ms=dt * 1e3, tflops=flops / dt / 1e12, frac_of_fp64_mfma_peak=flops / dt / 78.6e12)), flush=True)
