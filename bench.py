#!/usr/bin/env python3
"""Headline benchmark: SCF iterations/s of benzene RHF/cc-pVTZ -- the configuration BASELINE.json's north_star states its
targets on (">= 70 % of the HBM roofline on the J-build kernel and >= 10x PySCF-CPU wall-clock on benzene/cc-pVTZ RHF") --
on N MI355X, with the J/K digestion kernel's achieved bandwidth against the HBM roofline and a CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step is one SCF cycle exactly as `SCF.kernel` runs it (`SCF._step`): Fock = h + J - K/2 from the HBM-resident ERI tiles
(+ ONE RCCL all-reduce per cycle when sharded), CDIIS, occupied projector, density, energy and orbital gradient.  The
one-off ERI evaluation is outside the timed region (its wall time is reported as `eri_seconds`), exactly as the one-off
integral evaluation of an in-core CPU run is outside its per-cycle figure.  Inputs are synthetic: committed geometry fixture.

Extra legs (not part of `value`): at N = 1 `secondary` = the B3LYP/cc-pVTZ cycle (BASELINE config 3), `roofline_more` = the
J-only kernel and the cache-resident benzene/cc-pVDZ tensor (config 2), `wall_clock` = one COLD-process
`RHF(mol).to_gpu().kernel()` (what the reference's scripts run: templates/calculate_energy.py:145-156) beside the CPU
oracle's in-core total; at every N `scale_leg` = the SCF cycle of ibuprofen RHF/def2-TZVP (config 5's molecule, 103 GB of
tiles sharded over the ranks: 17 ms of J/K per cycle at N = 1, so its scaling is not capped by the replicated 0.25 ms).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
PMC_FILE = os.path.join("profiles", "r02_pmc_jk_traffic.json")


def cpu_share():
    """CPUs this process may really use: min(affinity mask, cgroup v2 quota), at least 1."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(mol, label, budget_s=40.0):
    """CPU oracle, kind "port", IN-CORE: what PySCF's CPU rung (`templates/calculate_energy.py:199-206`) does at this size --
    the 8-fold-unique ERIs are evaluated once into host memory (4.9 GB at N = 264) and every cycle digests that array
    (`orc_jk_incore`, all host cores) + numpy CDIIS / eig.  Bounded sample: if evaluating every row would exceed `budget_s`,
    only the rows of every `stride`-th shell pair are packed and digested and the J/K time is scaled by the stored fraction.
    The same pack time, scaled, is what ONE cycle of a direct (recompute) CPU SCF costs (`direct_*` keys)."""
    import numpy as np
    from oracle import oracle as orc
    # threads = the CPU share this process really has: a GPU box of the pool shows 256 logical CPUs but grants 16 (cgroup
    # cpu.max); 128 OpenMP threads on that quota ran the in-core digestion 4x SLOWER than 32 (0.20 s against 0.046 s)
    share = cpu_share()
    orc.Oracle.set_num_threads(share)
    n = mol.nao
    npair = n * (n + 1) // 2
    total = npair * (npair + 1) // 2
    probe_stride = 48
    t0 = time.time()
    o = orc.Oracle(mol).incore(tol=1e-13, stride=probe_stride, phase=probe_stride // 2)
    t_probe = time.time() - t0
    est_full = t_probe * total / max(o.incore_doubles, 1)
    stride = 1 if est_full <= budget_s else int(np.ceil(est_full / budget_s))
    if stride < probe_stride:
        t0 = time.time()
        o = orc.Oracle(mol).incore(tol=1e-13, stride=stride, phase=0)
        t_pack = time.time() - t0
    else:
        stride, t_pack = probe_stride, t_probe
    frac = o.incore_doubles / total
    S, T, V, _ = o.int1e()
    h = T + V
    nocc = mol.nelectron // 2
    e, c = orc.eig_gen(h, S)
    dm = 2.0 * c[:, :nocc] @ c[:, :nocc].T
    o.jk_incore(dm)                                   # warm-up (page touch, thread pool)
    reps = 3
    t0 = time.time()
    for _ in range(reps):
        J, K = o.jk_incore(dm)
    t_jk = (time.time() - t0) / reps
    if frac < 1.0:   # per-call fixed cost (thread-private N x N accumulators, reduction) must not be scaled by 1/fraction
        oe = orc.Oracle(mol).incore(tol=1e-13, stride=10 ** 9, phase=1)   # no rows at all
        oe.jk_incore(dm)
        t0 = time.time()
        for _ in range(reps):
            oe.jk_incore(dm)
        t_fixed = min((time.time() - t0) / reps, t_jk)
        t_jk_full = t_fixed + (t_jk - t_fixed) / frac
    else:
        t_jk_full = t_jk
    diis = orc.CDIIS()
    try:                                              # N = 264 LAPACK / BLAS calls: a 128-thread pool is 10x SLOWER than 8 threads
        from threadpoolctl import threadpool_limits
    except ImportError:
        import contextlib
        threadpool_limits = lambda limits: contextlib.nullcontext()
    with threadpool_limits(limits=min(8, share)):
        f = diis.update(S, dm, h + J - 0.5 * K)       # (untimed first pass: thread-pool start-up)
        orc.eig_gen(f, S)
        diis = orc.CDIIS()
        t0 = time.time()
        for _ in range(reps):                         # the rest of a cycle at full size (independent of the sample)
            f = diis.update(S, dm, h + J - 0.5 * K)
            e, c = orc.eig_gen(f, S)
            dm2 = 2.0 * c[:, :nocc] @ c[:, :nocc].T
            _ = float(np.sum(dm2 * (h + f)))
        t_rest = (time.time() - t0) / reps
    t_cycle = t_jk_full + t_rest
    t_direct = t_pack / frac + t_rest
    what = "all rows" if stride == 1 else f"rows of every {stride}th shell pair ({100 * frac:.1f} % of the array, J/K time scaled by 1/fraction)"
    return {"value": 1.0 / t_cycle, "unit": "iter/s", "cores": orc.Oracle.num_threads(), "cpus_visible": os.cpu_count(), "kind": "port", "mode": "in-core",
            "sample": f"in-core SCF cycle of {label} (N={n}) with the in-repo CPU oracle (not PySCF): packed 8-fold ERI array "
                      f"{total * 8e-9:.2f} GB, {what}; J/K digestion {t_jk_full:.3f} s + CDIIS/eig/density {t_rest:.3f} s per cycle; "
                      f"one-off ERI evaluation {t_pack / frac:.1f} s (excluded, like eri_seconds on the GPU); CPU work in this leg {t_probe + t_pack + (reps + 1) * t_jk:.0f} s",
            "seconds_per_cycle": t_cycle, "jk_seconds": t_jk_full, "rest_seconds": t_rest, "sample_fraction": frac,
            "direct_value": 1.0 / t_direct, "direct_seconds_per_cycle": t_direct,
            "direct_note": "a direct (recompute-every-cycle) CPU SCF pays the ERI evaluation each cycle: 'port-direct' figure"}


WALL_CHILD = r"""
import json, os, sys, time
t00 = time.time()
sys.path.insert(0, os.path.join(%(root)r, "computational-chemistry-ai_amd", "python"))
import torch
from mi355scf.fixtures import BENZENE
from pyscf import gto
import gpu4pyscf
t_imp = time.time() - t00
t0 = time.time()
mol = gto.Mole(); mol.atom = BENZENE; mol.basis = %(basis)r; mol.verbose = 0; mol.build()
t_mol = time.time() - t0
t0 = time.time()
mf = gpu4pyscf.scf.RHF(mol)
mf.init_guess = "atom"
mf = mf.to_gpu()
e = mf.kernel()
torch.cuda.synchronize()
t_kernel = time.time() - t0
tm = dict(mf.timing)
print("WALL " + json.dumps({"imports_seconds": t_imp, "mole_build_seconds": t_mol, "kernel_seconds": t_kernel, "e_tot": e,
                            "cycles": mf.cycles, "converged": bool(mf.converged),
                            "breakdown": {k: tm.get(k) for k in ("setup_seconds", "eri_seconds", "guess_seconds", "first_fock_seconds",
                                                                 "loop_seconds", "final_seconds", "total_seconds")}}))
"""


def wall_clock_child(basis, runs=2):
    """Cold processes (fresh Python, nothing cached in this process): seconds of ONE `RHF(mol).to_gpu().kernel()` on the
    benchmark molecule, set-up + ERI evaluation + SCF loop + final diagonalisation.  Must run BEFORE this process touches the
    GPU (a child may not be started from a GPU-initialised parent on this pool).  The first child also pays the box's
    first-use costs (file cache, driver); both are reported, `kernel_seconds` of the LAST one is the headline."""
    import subprocess
    out = []
    for _ in range(runs):
        try:
            r = subprocess.run([sys.executable, "-c", WALL_CHILD % {"root": ROOT, "basis": basis}], capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("WALL ")]
            out.append(json.loads(line[-1][5:]) if line else {"error": (r.stderr or r.stdout)[-400:]})
        except Exception as e:   # never let the extra leg take the headline down
            out.append({"error": repr(e)})
    return out


def time_steps(mf, st, steps, warmup, barrier):
    for _ in range(warmup):
        mf._step(st)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        mf._step(st)
    barrier()
    return time.perf_counter() - t0


def pmc_traffic(label, alg_bytes, variant="J+K"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
    --pmc runs as the guide prescribes).  PMC needs the profiler, so this is NOT measured by this process: `traffic_source`
    says where the number comes from; null when no committed pass matches the workload."""
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
        case = pmc["cases"].get(label + " " + variant) or pmc["cases"].get(label)
        if case and abs(case["algorithmic_bytes"] - alg_bytes) < 1e-6 * alg_bytes:
            return case["traffic_bytes"], f"{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/jk_once.py, committed; not measured in this run)"
    except Exception:
        pass
    return None, None


def ibuprofen_leg(args, world, rank, barrier):
    """SCF cycle of ibuprofen RHF/def2-TZVP (N = 573, 103 GB of resident tiles dealt over the ranks by bytes): the workload whose
    scaling over 2/4/8 GPUs is worth plotting -- 17.4 ms of J/K per cycle at N = 1 against ~0.6 ms of replicated algebra, where
    the benzene headline has 0.8 ms against 0.25 ms (Amdahl cap ~3x).  Every rank runs this leg (each cycle holds the Fock
    all-reduce).  Model per cycle: t(N) = 17.4 ms / N + 0.6 ms + t_allreduce(2 x 573^2 x 8 B = 5.3 MB)."""
    import torch
    import torch.distributed as dist
    from mi355scf import smiles_fixtures
    from mi355scf.mole import Mole
    from mi355scf.scf import RHF
    try:
        sym, xyz = smiles_fixtures.TABLE["CC(C)Cc1ccc(cc1)C(C)C(=O)O"]()
        atom = "; ".join(f"{s_} {x:.6f} {y:.6f} {z:.6f}" for s_, (x, y, z) in zip(sym, xyz))
        mol = Mole(atom=atom, basis="def2-TZVP", verbose=0).build()
        mf = RHF(mol)
        mf.eig_method = args.eig
        if world > 1:
            mf.shard(rank, world)
        t0 = time.time()
        mf.kernel()
        torch.cuda.synchronize()
        t_first = time.time() - t0
        if mf._stream_groups > 1:
            return {"workload": "ibuprofen RHF/def2-TZVP", "skipped": "shard does not fit HBM at this N (direct mode)"}
        st = mf._start(mf.make_rdm1())
        for _ in range(3):
            mf._step(st)
        nsteps = max(5, min(args.steps, 20))
        dt = time_steps(mf, st, nsteps, 2, barrier)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        stats = mf.engine.stats()
        ms_jk = mf.engine.time_jk_kernel(st["dm"], reps=20)
        out = {"workload": f"ibuprofen RHF/def2-TZVP SCF cycle (N_ao={mol.nao}, resident tiles sharded x{world})", "value": nsteps / dt,
               "unit": "iter/s", "ms_per_step": dt / nsteps * 1e3, "steps": nsteps, "n_gpus": world, "scaling": "strong",
               "first_scf": {"cycles": mf.cycles, "seconds_incl_eri_and_allocation": t_first, "eri_seconds": mf.timing.get("eri_seconds")},
               "jk_ms_per_launch_this_rank": ms_jk, "stored_bytes_this_rank": stats["stored_bytes"],
               "jk_frac_of_hbm_peak_this_rank": (8.0 * stats["n_unique_eri"] + 24.0 * mol.nao ** 2) / (ms_jk * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "e_tot": st["e_tot"], "model_ms_per_step": 17.4 / world + 0.6 + (0.0 if world == 1 else 0.05)}
        mf.reset()
        from mi355scf import engine as _e
        _e.release_cache()
        return out
    except Exception as e:   # an extra leg must not take the headline down -- but it must not hang the other ranks either
        if world > 1:
            raise
        return {"workload": "ibuprofen RHF/def2-TZVP", "error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--basis", default="cc-pVTZ")
    ap.add_argument("--molecule", default="benzene", choices=["benzene", "c60", "ibuprofen"],
                    help="benzene/cc-pVTZ (default) is the workload BASELINE's targets are stated on; c60 with --basis '6-31G*' is "
                         "config 4, meant for --gpus 8 (63 GB of resident tiles per rank; one GPU falls back to the direct mode); "
                         "ibuprofen with def2-TZVP is config 5's molecule")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=40.0, help="seconds of CPU ERI evaluation the cpu_baseline leg may spend")
    ap.add_argument("--no-extra-legs", "--no-extra-rooflines", dest="no_extra", action="store_true",
                    help="skip the additional N=1 legs (B3LYP cycle, J-only kernel, cc-pVDZ tensor)")
    ap.add_argument("--eig", default="sp2", choices=["sp2", "eigh"], help="projector method inside the SCF step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; 'gloo' only for rehearsing N>1 ranks on a 1-GPU box")
    ap.add_argument("--no-wall-clock", action="store_true", help="skip the cold-process wall-clock leg (N = 1 only)")
    ap.add_argument("--no-scale-leg", action="store_true", help="skip the ibuprofen RHF/def2-TZVP cycle leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # cold-process leg FIRST: children may only be started before this process initialises the GPU
    wall_children = None
    if world == 1 and not args.no_wall_clock and args.molecule == "benzene":
        wall_children = wall_clock_child(args.basis)

    import torch
    import torch.distributed as dist
    from mi355scf.fixtures import BENZENE
    from mi355scf.mole import Mole
    from mi355scf.scf import RHF

    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev = local if args.backend == "nccl" else local % max(ndev, 1)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    if args.molecule == "benzene":
        atom = BENZENE
    else:
        from mi355scf import smiles_fixtures
        sym, xyz = smiles_fixtures.TABLE[{"c60": "C60", "ibuprofen": "CC(C)Cc1ccc(cc1)C(C)C(=O)O"}[args.molecule]]()
        atom = "; ".join(f"{s_} {x:.6f} {y:.6f} {z:.6f}" for s_, (x, y, z) in zip(sym, xyz))
    mol = Mole(atom=atom, basis=args.basis, verbose=0).build()
    mf = RHF(mol)
    mf.eig_method = args.eig
    if world > 1:
        mf.shard(rank, world)
    t0 = time.time()
    # Set-up (untimed, like the ERI evaluation): the FIRST SCF of the object, run to convergence from the atomic guess.  It is a
    # "cold" object: no purification plan yet (a plan needs spectral bounds, and a diagonalisation made only for them is never
    # earned back inside one SCF), so its cycles use the trace-correcting purification; its final diagonalisation (mo_energy)
    # seeds the plan.  The W warm-up and K timed steps are then cycles of the warm object -- what every later SCF of the same
    # object runs (each geometry step of an optimisation, scans, restarts): planned purification, pipelined step.  Every timed
    # step is a complete SCF cycle (J/K, Fock, CDIIS, purification, energy, orbital gradient).  `first_scf` below reports the
    # cold cycles beside it.
    mf.kernel()
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    first_scf = {"cycles": mf.cycles, "loop_seconds": mf.timing.get("loop_seconds"), "converged": bool(mf.converged),
                 "ms_per_cycle_incl_first_use": mf.timing.get("loop_seconds", 0.0) / max(mf.cycles, 1) * 1e3}
    st = mf._start(mf.make_rdm1())
    stats = mf.engine.stats()
    # Untimed settle cycles before the W warm-up steps: the purification plan is taken up once |g| has settled (4 cycles), and the
    # device needs ~50 ms of continuous work after an idle spell before its clocks are steady -- the first J/K launches after the
    # host-side set-up gaps run 4-8 % slower than the rest (tools/jk_drift.py: 0.82-0.85 ms, then 0.784 ms flat).  With the
    # driver's W = 5, K = 20 the whole timed region is 25 ms, i.e. it would sit inside that ramp.  A fixed count (same on every
    # rank: each cycle holds a collective); direct-mode workloads (seconds per cycle) keep the minimum.
    from mi355scf import engine as _engine_mod
    _engine_mod.wait_warm()    # the library warm-up thread (rocSOLVER initialisation, ~0.3 s) must not share the interpreter with the timed loop
    SETTLE = 4 if mf._stream_groups > 1 else 60
    for _ in range(SETTLE):
        mf._step(st)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt = time_steps(mf, st, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline leg: the J/K digestion kernel alone, HIP events on the launch stream (this rank's share of the tiles)
    label = args.molecule + "/" + args.basis
    n = mol.nao
    direct_mode = mf._stream_groups > 1
    roof = None
    if not direct_mode:
        ms = mf.engine.time_jk_kernel(st["dm"], reps=500)   # ~0.4 s of back-to-back launches: a stable average on a part whose clocks wander
        alg_bytes = 8.0 * stats["n_unique_eri"] + 24.0 * n * n
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        traffic, source = pmc_traffic(label, alg_bytes) if world == 1 else (None, None)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": source, "ms_per_launch": ms,
                "kernel": ("jk_tiles_kernel<true,true,true>" if stats["stored_bytes"] > (256 << 20) else "jk_tiles_kernel<true,true,false>"),
                "algorithmic_bytes": alg_bytes, "stored_bytes": stats["stored_bytes"],
                "stored_GBps": stats["stored_bytes"] / (ms * 1e-3) / 1e9,
                "share_of_step": ms / (dt / args.steps * 1e3)}

    more, secondary = [], None
    if world == 1 and not args.no_extra and not direct_mode:
        def leg(engine, nao, est, dm, lab, variants):
            for wj, wk, name, nmat in variants:
                t = engine.time_jk_kernel(dm, reps=200, with_j=wj, with_k=wk)
                b = 8.0 * est["n_unique_eri"] + 8.0 * nmat * nao * nao
                tr, src = pmc_traffic(lab, b, name)
                more.append({"workload": lab, "variant": name, "ms_per_launch": t, "algorithmic_bytes": b, "traffic": tr, "traffic_source": src,
                             "achieved": b / (t * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": b / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, "stored_bytes": est["stored_bytes"]})
        leg(mf.engine, n, stats, st["dm"], label, ((True, False, "J only", 2),))
        if args.molecule == "benzene":
            # BASELINE config 3: one B3LYP/cc-pVTZ cycle (same resident tiles: the engine is shared; J + 0.2 K + XC quadrature)
            from mi355scf.dft import RKS
            ks = RKS(mol, xc="B3LYP")
            ks._eng = mf.engine
            ks.eig_method = args.eig
            ks.kernel()                                  # first SCF of the object (cold), seeds the plan: see above
            st_ks = ks._start(ks.make_rdm1())
            for _ in range(SETTLE):
                ks._step(st_ks)
            dt_ks = time_steps(ks, st_ks, max(10, args.steps // 2), 3, barrier)
            nk = max(10, args.steps // 2)
            secondary = {"workload": f"benzene B3LYP/{args.basis} SCF cycle (BASELINE config 3; level-3 grid, {ks.grids.size} points)",
                         "value": nk / dt_ks, "unit": "iter/s", "ms_per_step": dt_ks / nk * 1e3, "steps": nk, "e_tot": st_ks["e_tot"],
                         "redone_cycles": getattr(ks, "n_redo", 0)}
            if args.basis.lower() != "cc-pvdz":
                from mi355scf.engine import Engine
                mol2 = Mole(atom=BENZENE, basis="cc-pVDZ", verbose=0).build()
                e2 = Engine(mol2)
                st2 = e2.prepare_eri(1e-13)
                g = torch.Generator(device="cpu").manual_seed(0)
                a = torch.randn(mol2.nao, mol2.nao, generator=g, dtype=torch.float64)
                leg(e2, mol2.nao, st2, (a + a.T).cuda(), "benzene/cc-pVDZ", ((True, False, "J only", 2), (True, True, "J+K", 3)))
                e2.close()

    # a second, COLD object in the now warm process (libraries loaded, ERIs resident): the cycles of the one kernel() call the
    # reference's scripts make, without this process's first-use costs (those are in `first_scf` and in `wall_clock`)
    cold_obj = None
    if world == 1 and not direct_mode:
        mf2 = RHF(mol)
        mf2.eig_method = args.eig
        mf2._eng = mf.engine
        mf2.kernel()
        torch.cuda.synchronize()
        cold_obj = {"cycles": mf2.cycles, "loop_seconds": mf2.timing.get("loop_seconds"), "converged": bool(mf2.converged),
                    "ms_per_cycle": mf2.timing.get("loop_seconds", 0.0) / max(mf2.cycles, 1) * 1e3, "e_tot": mf2.e_tot}

    scale_leg = None
    if not args.no_scale_leg and args.molecule == "benzene" and not direct_mode:
        scale_leg = ibuprofen_leg(args, world, rank, barrier)

    if rank == 0:
        value = args.steps / dt
        out = {"metric": "scf_iterations_per_sec", "value": value, "unit": "iter/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{args.molecule} RHF/{args.basis} SCF cycle (N_ao={n}, "
                                      + ("direct mode: tile groups re-evaluated every cycle)" if direct_mode else "resident 8-fold ERI tiles)"),
                          "n_ao": n, "n_unique_eri": stats["n_unique_eri"], "parallelism": f"tile-run shard x{world} (LPT by bytes)",
                          "density_from_fock": args.eig, "object_state": "warm (second and later SCFs of the object; first SCF run in set-up)",
                          "settle_cycles_before_warmup": SETTLE,
                          "redone_cycles": getattr(mf, "n_redo", 0),
                          # the call the reference's scripts make is ONE kernel() on a fresh object: its own cycles, averaged
                          "cold_ms_per_cycle": (cold_obj or first_scf).get("ms_per_cycle", first_scf["ms_per_cycle_incl_first_use"]),
                          "cold_iter_per_s": 1e3 / max((cold_obj or first_scf).get("ms_per_cycle", first_scf["ms_per_cycle_incl_first_use"]), 1e-9),
                          "cold_cycles": (cold_obj or first_scf)["cycles"],
                          "cold_note": "cycles of ONE kernel() on a fresh object (no purification plan: trace-correcting SP2, ~40 passes), "
                                       "averaged over its own cycles, in a process whose libraries are loaded; `first_scf` is the same "
                                       "call as this process's first (incl. first-use costs), `wall_clock` a whole cold process",
                          "workload_note": "default workload is benzene/cc-pVTZ since round 2 (round 1: cc-pVDZ): not comparable with BENCH_r01"},
               "roofline": roof, "roofline_more": more, "secondary": secondary, "e_tot": st["e_tot"],
               "eri_seconds": stats["seconds_eri"], "setup_seconds": setup_s, "first_scf": first_scf, "cold_object_scf": cold_obj, "scale_leg": scale_leg}
        if wall_children and (args.no_cpu_baseline or mol.nao > 300):
            out["wall_clock"] = {"gpu_kernel_seconds": wall_children[-1].get("kernel_seconds"), "gpu_breakdown": wall_children[-1].get("breakdown"),
                                 "gpu_first_process_on_box": wall_children[0], "cpu_total_seconds": None}
        if world == 1 and not args.no_cpu_baseline:
            if mol.nao <= 300:
                cb = cpu_baseline(mol, label, args.cpu_budget)
                cb["gpu_over_cpu"] = value / cb["value"]
                out["cpu_baseline"] = cb
                if wall_children:
                    last = wall_children[-1]
                    cpu_total = cb["direct_seconds_per_cycle"] - cb["rest_seconds"] + (last.get("cycles") or first_scf["cycles"]) * cb["seconds_per_cycle"]
                    out["wall_clock"] = {
                        "what": "ONE cold-process RHF(mol).to_gpu().kernel() on " + label + " (set-up, atomic guess, ERI evaluation, "
                                "SCF loop, final diagonalisation; Python imports excluded, reported beside it) vs the CPU oracle's "
                                "in-core total = one-off ERI evaluation + the same number of cycles",
                        "gpu_kernel_seconds": last.get("kernel_seconds"), "gpu_breakdown": last.get("breakdown"),
                        "gpu_cycles": last.get("cycles"), "gpu_e_tot": last.get("e_tot"), "gpu_imports_seconds": last.get("imports_seconds"),
                        "gpu_first_process_on_box": wall_children[0], "processes": len(wall_children),
                        "cpu_total_seconds": cpu_total, "cpu_eri_seconds": cb["direct_seconds_per_cycle"] - cb["rest_seconds"],
                        "cpu_cores": cb["cores"], "cpu_kind": "port (in-repo oracle, in-core; not PySCF)",
                        "cpu_over_gpu": (cpu_total / last["kernel_seconds"]) if last.get("kernel_seconds") else None}
            elif True:
                out["cpu_baseline"] = {"value": None, "unit": "iter/s", "cores": 0, "kind": "port",
                                       "sample": "skipped: the packed ERI array of this workload exceeds host memory budgets; the default "
                                                 "benzene/cc-pVTZ run carries the CPU baseline"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
