#!/usr/bin/env python3
"""UHF / UKS cycle time of a radical cation, fast loop vs plain loop.  python tools/uhf_bench.py benzene cc-pVTZ [B3LYP]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import torch
from mi355scf.mole import Mole
from mi355scf.uhf import UHF
from mi355scf.uks import UKS
from mi355scf import smiles_fixtures, fixtures
name, basis = sys.argv[1], sys.argv[2]
xc = sys.argv[3] if len(sys.argv) > 3 else None
def _atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))
atom = {"ch3": lambda: "C 0 0 0; H 1.079 0 0; H -0.5395 0.934441 0; H -0.5395 -0.934441 0", "benzene": lambda: fixtures.BENZENE, "ibuprofen": lambda: _atoms("CC(C)Cc1ccc(cc1)C(C)C(=O)O")}[name]()
mol = Mole(atom=atom, basis=basis, verbose=0, charge=0 if name == "ch3" else 1, spin=1).build()
# one object, three SCFs from the same guess: the first is cold (plain loop, seeds the purification plans), the others warm
mfw = UKS(mol) if xc else UHF(mol)
if xc:
    mfw.xc = xc
mfw.conv_tol = 1e-9
for i in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    e = mfw.kernel()
    torch.cuda.synchronize()
    print(f"same object, call {i}: E = {e:.10f} cycles {mfw.cycles} loop {mfw.timing['loop_seconds']:.4f} s = {mfw.timing['loop_seconds'] / max(mfw.cycles, 1) * 1e3:.2f} ms/cycle eigh {getattr(mfw, 'n_eigh', 0)} redo {getattr(mfw, 'n_redo', 0)}", flush=True)
eng = mfw.engine
for pair in (0, 1, 0, 1):
    eng.set_option('jk_pair', pair)
    mfp = UKS(mol) if xc else UHF(mol)
    if xc:
        mfp.xc = xc
    mfp._eng = eng; mfp.fast_loop = False; mfp.conv_tol = 1e-9
    e = mfp.kernel()
    print(f'jk_pair={pair}: E = {e:.10f} cycles {mfp.cycles} {mfp.timing["loop_seconds"] / max(mfp.cycles, 1) * 1e3:.2f} ms/cycle', flush=True)
eng.set_option('jk_pair', 1)
for fast in (True, False, True):
    mf = UKS(mol) if xc else UHF(mol)
    if xc:
        mf.xc = xc
    if eng is not None:
        mf._eng = eng
    mf.fast_loop = fast
    mf.conv_tol = 1e-9
    torch.cuda.synchronize(); t0 = time.time()
    e = mf.kernel()
    torch.cuda.synchronize(); dt = time.time() - t0
    eng = mf.engine
    print(f"fast_loop={fast}: E = {e:.10f} converged {mf.converged} cycles {mf.cycles} loop {mf.timing['loop_seconds']:.4f} s = {mf.timing['loop_seconds'] / max(mf.cycles, 1) * 1e3:.2f} ms/cycle, <S^2> = {mf.spin_square()[0]:.6f} eigh {getattr(mf, 'n_eigh', 0)} redo {getattr(mf, 'n_redo', 0)} fock {getattr(mf, 'n_fock_builds', 0)}", flush=True)
