#!/usr/bin/env python3
"""Per-row measurements (SURVEY.md section 8d secondary figures) on one MI355X -> JSON lines:
ERI evaluation (shell quartets/s), J/K kernel (GB/s vs HBM roofline), XC build (FLOP/s of the 4*ng*N^2 dense part),
SCF cycle time, analytic gradient time.   python tools/bench_components.py [case ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
import numpy as np, torch
from mi355scf import fixtures, smiles_fixtures
from mi355scf.mole import Mole
from mi355scf.scf import RHF
from mi355scf.dft import RKS


def atoms(key):
    sym, xyz = smiles_fixtures.TABLE[key]()
    return "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz))


CASES = {
    "benzene/cc-pVDZ": (fixtures.BENZENE, "cc-pVDZ"),
    "benzene/cc-pVTZ": (fixtures.BENZENE, "cc-pVTZ"),
    "ibuprofen/def2-TZVP": (atoms("CC(C)Cc1ccc(cc1)C(C)C(=O)O"), "def2-TZVP"),
}


def sync():
    torch.cuda.synchronize()


for name in (sys.argv[1:] or list(CASES)):
    atom, basis = CASES[name]
    mol = Mole(atom=atom, basis=basis, verbose=0).build()
    n = mol.nao
    out = {"case": name, "n_ao": n, "n_shells": mol.nbas}
    mf = RKS(mol, xc="B3LYP")
    t0 = time.time(); e = mf.kernel(); sync(); out["rks_b3lyp_wall_s"] = round(time.time() - t0, 3)
    st = mf.engine.stats()
    out.update(e_tot=e, cycles=mf.cycles, loop_s=round(mf.timing["loop_seconds"], 4), ms_per_cycle=round(mf.timing["loop_seconds"] / mf.cycles * 1e3, 2))
    out["eri"] = {"shell_quartets": st["n_quartets"], "seconds_prepare": round(st["seconds_eri"], 4),
                  "Mquartets_per_s_incl_setup": round(st["n_quartets"] / st["seconds_eri"] / 1e6, 1), "resident_GB": round(st["stored_bytes"] / 1e9, 3)}
    ms = min(mf.engine.time_jk_kernel(mf._dm, reps=20) for _ in range(3))
    alg = 8.0 * st["n_unique_eri"] + 24.0 * n * n
    out["jk"] = {"ms": round(ms, 4), "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4),
                 "stored_GBps": round(st["stored_bytes"] / ms / 1e6, 1)}
    ng = mf.grids.size
    mf.nr_rks(mf._dm); sync()
    t0 = time.perf_counter()
    for _ in range(5):
        mf.nr_rks(mf._dm)
    sync()
    tx = (time.perf_counter() - t0) / 5
    out["xc"] = {"ngrids": ng, "ms_per_build": round(tx * 1e3, 3), "dense_TFLOPs": round(4.0 * ng * n * n / tx / 1e12, 2),
                 "note": "4*ng*N^2 flop (rho and Vxc contractions) over the whole nr_rks build incl. AO-cache reads, rho, functional, aow"}
    g = mf.nuc_grad_method()
    t0 = time.time(); g.kernel(); sync()
    out["gradient"] = {"seconds": round(time.time() - t0, 3), **{k: round(v, 3) for k, v in g.timing.items()}}
    mh = RHF(mol)
    t0 = time.time(); mh.kernel(); sync()
    out["rhf"] = {"wall_s": round(time.time() - t0, 3), "cycles": mh.cycles, "ms_per_cycle": round(mh.timing["loop_seconds"] / mh.cycles * 1e3, 3)}
    print(json.dumps(out), flush=True)
    mf._eng = None; mh._eng = None
