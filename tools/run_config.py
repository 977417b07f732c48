#!/usr/bin/env python3
"""Run one BASELINE config through the drop-in surface and print timings (used for DESIGN.md numbers).
  python tools/run_config.py benzene cc-pVTZ B3LYP"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from pyscf import gto, scf, dft
from mi355scf import fixtures

name, basis, method = sys.argv[1], sys.argv[2], sys.argv[3]
from mi355scf import smiles_fixtures
def _atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))
atom = {"benzene": lambda: fixtures.BENZENE, "h2co": lambda: fixtures.H2CO, "h2o": lambda: fixtures.H2O,
        "ibuprofen": lambda: _atoms("CC(C)Cc1ccc(cc1)C(C)C(=O)O"), "c60": lambda: _atoms("C60")}[name]()
mol = gto.Mole(); mol.atom = atom; mol.basis = basis; mol.verbose = 4; mol.build()
do_grad = "--grad" in sys.argv
shard = [a for a in sys.argv if a.startswith("--shard=")]
t0 = time.time()
mf = scf.RHF(mol) if method == "HF" else dft.RKS(mol)
if method != "HF":
    mf.xc = method
mf = mf.to_gpu()
if shard:
    r, n_ = shard[0].split("=")[1].split("/")
    mf.shard(int(r), int(n_))   # partial Fock on this rank only: timing/memory rehearsal of one shard (energies meaningless)
    mf.max_cycle = 3
e = mf.kernel()
torch.cuda.synchronize()
wall = time.time() - t0
st = mf.engine.stats()
ms = mf.engine.time_jk_kernel(mf._dm, reps=10)
n = mol.nao
alg = 8.0 * st["n_unique_eri"] + 24.0 * n * n
opt = [a for a in sys.argv if a.startswith("--opt=")]
if opt:
    from pyscf.geomopt.geometric_solver import optimize
    mf.verbose = 3; mol.verbose = 3
    t1 = time.time()
    mol_opt = optimize(mf, maxsteps=int(opt[0].split("=")[1]))
    print(f"optimize wall {time.time() - t1:.1f} s")
gt = None
if do_grad:
    t1 = time.time(); g = mf.nuc_grad_method().kernel(); torch.cuda.synchronize(); gt = time.time() - t1
    print("gradient max |g| =", float(abs(g).max()), "sum", g.sum(axis=0))
print(json.dumps(dict(grad_seconds=gt, config=f"{name} {method}/{basis}", nao=n, e_tot=e, converged=bool(mf.converged), cycles=mf.cycles,
                      wall_s=wall, timing=mf.timing, eri=st, jk_ms=ms, jk_alg_GBps=alg / ms / 1e6,
                      jk_stored_GBps=st["stored_bytes"] / ms / 1e6,
                      ngrids=getattr(getattr(mf, "grids", None), "size", 0))))
