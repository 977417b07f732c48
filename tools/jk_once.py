#!/usr/bin/env python3
"""A few J/K launches on a resident tensor, for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE).
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/jk_once.py cc-pVTZ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from mi355scf.mole import Mole
from mi355scf.engine import Engine
from mi355scf import fixtures
basis = sys.argv[1] if len(sys.argv) > 1 else "cc-pVTZ"
mol = Mole(atom=fixtures.BENZENE, basis=basis, verbose=0).build()
n = mol.nao
rng = np.random.default_rng(0)
a = rng.normal(size=(n, n)); D = torch.as_tensor(a + a.T, device="cuda")
eng = Engine(mol)
st = eng.prepare_eri(1e-13)
for _ in range(5):
    J, K = eng.get_jk(D)                      # jk_tiles_kernel<true,true,...>
torch.cuda.synchronize()
for _ in range(5):
    J, _k = eng.get_jk(D, with_k=False)       # jk_tiles_kernel<true,false,...>
torch.cuda.synchronize()
print("stored_bytes", st["stored_bytes"], "alg_bytes J+K", 8 * st["n_unique_eri"] + 24 * n * n, "J only", 8 * st["n_unique_eri"] + 16 * n * n)
