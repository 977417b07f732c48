"""SCF driver behind `pyscf.scf.RHF` / `gpu4pyscf.scf.RHF`.

Host control flow mirrors what the reference's call sites expect of PySCF's `SCF.kernel`
(`templates/calculate_energy.py:125,134,155,177,205`; `templates/optimize_geometry.py:90,108`):
plain settable attributes (`init_guess`, `max_cycle`, `conv_tol`, `xc`, `verbose`), `kernel(dm0=None)`
returning a Python float, `converged` reporting non-convergence (never an exception), NumPy results in
`mo_energy / mo_occ / mo_coeff`, and per-cycle lines on `mol.stdout` at `verbose >= 4`.

All N x N algebra runs on the GPU: J/K from the resident-ERI HIP kernels, DIIS through the HIP
helpers, `eigh`/GEMM through torch (hipSOLVER / rocBLAS).  There is no CPU path.
"""
import sys
import time

import numpy as np
import torch

from . import engine as _engine
from .mole import Mole

AU2DEBYE = 2.541746473  # e*a0 -> Debye [MEM: pyscf.data.nist.AU2DEBYE]

# ground-state configurations: electrons per l for the spherically averaged atom guess
_AUFBAU = [(0, 2), (0, 2), (1, 6), (0, 2), (1, 6), (0, 2), (2, 10), (1, 6)]  # 1s 2s 2p 3s 3p 4s 3d 4p


def _atom_config(z):
    occ = {}
    left = z
    for l, cap in _AUFBAU:
        if left <= 0:
            break
        n = min(cap, left)
        occ.setdefault(l, []).append(n)
        left -= n
    return occ


class DeviceDIIS:
    """Pulay CDIIS with the history on the GPU (row a10): e = (SDF)^T - SDF, subspace 8 [MEM defaults]."""

    MAX_SPACE = 16   # DIIS_MAXM of `diis_solve_kernel` (include/mi355scf.h: mi_diis_solve)

    def __init__(self, eng, space=8, nmat=1):
        n = eng.nao
        space = int(space)
        if not 1 <= space <= self.MAX_SPACE:
            # (the round-1 host solve took any size; the device solve keeps the Pulay system in one wave's registers)
            raise ValueError(f"diis_space = {space}: the device-side Pulay solve supports 1..{self.MAX_SPACE} vectors")
        self.eng, self.space, self.count = eng, space, 0
        shape = (space, n, n) if nmat == 1 else (space, nmat, n, n)    # nmat = 2: the spin-stacked pair of UHF / UKS
        self.F = torch.empty(*shape, dtype=torch.float64, device=eng.device)
        self.E = torch.empty(*shape, dtype=torch.float64, device=eng.device)
        self.B = np.zeros((space, space))
        self._e = torch.empty(n, n, dtype=torch.float64, device=eng.device)

    # --- split form used by the SCF step: `push*` (error-vector Gram row + Pulay solve, all on the device, no sync) then
    # `extrapolate` (combination with the device-resident coefficients): no host round trip inside a cycle ---
    NS = 16   # partial sums per Gram-row entry (DIIS_NS of the kernel), added in index order by the solve kernel

    def _gram_and_solve(self, slot, m):
        if not hasattr(self, "dots_dev"):
            dev = self.F.device
            self.dots_dev = torch.zeros(self.space * self.NS, dtype=torch.float64, device=dev)
            self.B_dev = torch.zeros(self.space, self.space, dtype=torch.float64, device=dev)
            self.coef_dev = torch.zeros(self.space, dtype=torch.float64, device=dev)
        self.eng.diis_dots_dev(self.E, self.E[slot], m, self.dots_dev)
        self.eng.diis_solve(self.dots_dev, m, slot, self.space, self.B_dev, self.coef_dev)
        self._pending = (slot, m)

    def push(self, f, e):
        """Store (F_i, e_i); leaves the Pulay coefficients of the enlarged history in `self.coef_dev[:m]` on the device."""
        slot = self.count % self.space
        self.F[slot].copy_(f)
        self.E[slot].copy_(e)
        self.count += 1
        m = min(self.count, self.space)
        self._gram_and_solve(slot, m)
        return m

    def next_slot(self):
        """History slot the next push will use: the SCF step lets its GEMMs write F' and e straight into it."""
        return self.count % self.space

    def push_inplace(self):
        """`push` for data already written into `F[next_slot()]` / `E[next_slot()]` (no device copies)."""
        slot = self.count % self.space
        self.count += 1
        m = min(self.count, self.space)
        self._gram_and_solve(slot, m)
        return m

    def extrapolate(self, dots=None):
        """F = sum_i c_i F_i with the coefficients the last push solved for (`dots` is ignored: kept for callers of the
        round-1 host-solve signature)."""
        _slot, m = self._pending
        out = torch.empty_like(self.F[0])
        self.eng.diis_combine_dev(self.F, self.coef_dev, m, out)
        return out

    def update(self, s, d, f):
        sdf = s @ d @ f
        self.eng.diis_errvec(sdf, self._e)
        slot = self.count % self.space
        self.F[slot].copy_(f)
        self.E[slot].copy_(self._e)
        self.count += 1
        m = min(self.count, self.space)
        dots = self.eng.diis_dots(self.E, self._e, m)
        self.B[slot, :m] = dots
        self.B[:m, slot] = dots
        A = np.zeros((m + 1, m + 1))
        A[0, 1:] = A[1:, 0] = 1.0
        A[1:, 1:] = self.B[:m, :m]
        rhs = np.zeros(m + 1)
        rhs[0] = 1.0
        try:
            c = np.linalg.solve(A, rhs)
        except np.linalg.LinAlgError:
            c = np.linalg.lstsq(A, rhs, rcond=None)[0]
        out = torch.empty_like(f)
        self.eng.diis_combine(self.F, c[1:], out)
        return out


class SCF:
    conv_tol = 1e-9
    conv_tol_grad = None
    max_cycle = 50
    init_guess = "minao"
    diis_space = 8
    diis_start_cycle = 1
    direct_scf_tol = 1e-13
    conv_check = True
    xc = None  # HF
    # How the occupied-space projector is obtained from the Fock matrix inside the SCF loop:
    #  'sp2'  trace-correcting second-order spectral projection (Niklasson, PRB 66, 155115 (2002)) --
    #         GEMM-only (rocBLAS FP64 MFMA), no host sync; the converged density is the same aufbau
    #         projector the diagonalisation gives.  mo_energy/mo_coeff come from one `eigh` after convergence.
    #  'eigh' hipSOLVER generalised eigenproblem every cycle (what PySCF's `eig` does [MEM]).
    eig_method = "sp2"
    sp2_tol = 1e-11
    sp2_margin = 2     # purification steps kept beyond the first one that met sp2_tol in the previous cycle
    sp2_fused = True   # small N: one fused HIP launch per SP2 step instead of rocBLAS DGEMM + update kernel
    _sp2_iters = 24
    _sp2_validated = False   # True once an iteration count has passed the checked path for this Fock spectrum
    sp2_fused_max = 320  # one fused launch per purification step up to here (single-batch panel loads), rocBLAS DGEMM + update above
    # Planned purification (sp2plan.py): the sequence of quadratics is fixed from bounds of the spectrum and of the HOMO/LUMO
    # taken at the last diagonalisation -- about half the steps of trace-correcting SP2.  The result is validated every cycle;
    # when the spectrum has moved outside the margins the cycle is redone by `eigh`, which also refreshes the bounds.
    sp2_planned = True
    sp2_inner_margin = 0.15   # Hartree the HOMO / LUMO may move towards the gap before the plan fails
    sp2_outer_margin = 2.0    # Hartree the extreme eigenvalues may move outwards
    _sp2_plan = None
    _spin_restricted = True
    # Sharded runs: every rank decides redo / convergence / loop exit from its own replicated algebra.  The own kernels are free
    # of atomics (fixed-order partial sums) and `parallel.blas_atomics_off()` forbids atomics in rocBLAS, so those scalars are
    # bit-identical across ranks -- but that rests on library behaviour (rocBLAS kernel selection, rocSOLVER syevd) that no
    # multi-GPU run has confirmed yet, and a one-ulp difference at a threshold would let one rank leave a loop whose body holds
    # a collective (a hang).  None = auto: with more than one rank, rank 0's packed control scalars of the cycle (a few KB) are
    # broadcast once per cycle and every rank decides from that copy; False = no broadcast (zero per-cycle collectives beside
    # the Fock all-reduce; the world-2 tests assert bit-identical energies in this mode); True = always.
    sync_control = None
    host_cholesky_max = 400   # basis sizes up to which S = L L^T and L^-1 are formed on the host (see _setup)
    level_shift = 0.0    # Hartree; virtual-orbital shift applied to the Fock matrix that is diagonalised / purified

    def __init__(self, mol):
        if not isinstance(mol, Mole):
            raise TypeError("SCF needs a built gto.Mole")
        if not mol._built:
            mol.build()
        if mol.spin != 0 and self._spin_restricted:
            raise NotImplementedError("RHF/RKS need a closed shell (mol.spin = 0); use scf.UHF / dft.UKS for open shells")
        self.mol = mol
        self.verbose = mol.verbose
        self.stdout = mol.stdout
        self.converged = False
        self.e_tot = 0.0
        self.mo_energy = self.mo_coeff = self.mo_occ = None
        self.cycles = 0
        self._eng = None
        self._dm = None
        self._rank, self._nranks, self._pg = 0, 1, None
        self._stream_groups = 1   # >1: direct mode (ERI tile groups recomputed each Fock build)
        self._resident_groups = 0  # direct mode: leading groups kept in HBM on engines of their own
        self._group_engines = []
        self.timing = {}
        self._auto_shard()

    def _auto_shard(self):
        """Under `torchrun` (WORLD_SIZE > 1) every SCF object shards itself over the default process group, so an
        unchanged script (`torchrun --nproc-per-node 8 calculate_energy.py ... --use-gpu`) runs on 8 GPUs: one
        process per GPU, RCCL all-reduce of the Fock contributions.  Set MI355_AUTO_SHARD=0 to disable."""
        import os
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1 or os.environ.get("MI355_AUTO_SHARD", "1") == "0":
            return
        from . import parallel
        import torch.distributed as dist
        if not dist.is_initialized():
            local = int(os.environ.get("LOCAL_RANK", "0"))
            ndev = max(torch.cuda.device_count(), 1)
            backend = os.environ.get("MI355_DIST_BACKEND", "nccl")
            torch.cuda.set_device(local if backend == "nccl" else local % ndev)
            parallel.init(backend, torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else None)
        self.shard(dist.get_rank(), dist.get_world_size())
        if dist.get_rank() != 0:
            self.verbose = min(self.verbose, 0)   # one log stream

    # --- backend selection (row a14) ------------------------------------------------------------
    def to_gpu(self):
        return self

    def to_cpu(self):
        return self

    def _log(self, level, msg):
        if self.verbose >= level:
            out = self.stdout or sys.stdout
            out.write(msg + "\n")

    def shard(self, rank, nranks, process_group=None):
        """Shard the resident-ERI tile runs over `nranks` GPUs; J/K partial sums are all-reduced over
        RCCL each Fock build."""
        self._rank, self._nranks, self._pg = rank, nranks, process_group
        if nranks > 1:
            from . import parallel
            parallel.blas_atomics_off()
        return self

    def _sync_control_on(self):
        """Whether rank 0's control scalars are broadcast every cycle (see `sync_control`)."""
        if self._nranks <= 1:
            return False
        import os
        if self.sync_control is None:
            return os.environ.get("MI355_SYNC_CONTROL") != "0"
        return bool(self.sync_control)

    @property
    def engine(self):
        if self._eng is None or self._eng.mol is not self.mol:
            self._eng = _engine.Engine(self.mol)
        return self._eng

    def reset(self, mol=None):
        if mol is not None:
            self.mol = mol
        self._eng = None
        self._h1 = None
        self._drop_group_engines()
        return self

    def _drop_group_engines(self):
        for g in getattr(self, "_group_engines", []):
            g.close()
        self._group_engines = []

    # --- pieces of the Fock build ---------------------------------------------------------------
    def _setup(self):
        eng = self.engine
        t0 = time.time()
        S, T, V = eng.int1e()
        h1 = T + V
        if self._nranks > 1:
            # one-off per geometry (not per cycle): rank 0's one-electron matrices become everybody's.  V is accumulated with
            # FP64 atomics (one slice per nucleus), so replicated copies differ in the last bits; everything downstream in the
            # SCF loop is deterministic and stays bit-identical across ranks only if it starts from identical inputs.
            from . import parallel
            hs = torch.stack([S, h1])
            parallel.broadcast0(hs, self._pg)
            S, h1 = hs[0].contiguous(), hs[1].contiguous()
        self._S, self._h1 = S, h1
        if eng.nao <= self.host_cholesky_max:
            # S = L L^T and L^-1 on the HOST for small bases (LAPACK, 4-6 ms at N = 264): the first rocSOLVER call of a process costs
            # ~0.13 s of lazy initialisation (tools/wall_profile.py), a third of a cold benzene/cc-pVTZ kernel(); the warm-up
            # thread (engine._warm_libraries) has the device solver ready by the time the final eigh needs it.  Few BLAS
            # threads: an unlimited pool on a many-core host takes 50x longer on these sizes than four threads do.
            try:
                from threadpoolctl import threadpool_limits
            except ImportError:                 # no thread limiter: still correct, possibly slow on a many-core host
                import contextlib
                threadpool_limits = lambda limits: contextlib.nullcontext()
            with threadpool_limits(limits=4):
                Lh = np.linalg.cholesky(S.cpu().numpy())    # raises numpy.linalg.LinAlgError for a linearly dependent basis
                Li = np.linalg.inv(Lh)
            L = torch.from_numpy(np.ascontiguousarray(Lh)).to(eng.device)
            self._Linv = torch.from_numpy(np.ascontiguousarray(np.tril(Li))).to(eng.device)
        else:
            L = torch.linalg.cholesky(S)
            self._Linv = torch.linalg.solve_triangular(L, torch.eye(eng.nao, dtype=torch.float64, device=eng.device), upper=False)
        self._L = L
        if getattr(self, "with_df", None) is not None:
            if self.with_df.mol is not self.mol or self.with_df._B is None:
                self.with_df.mol = self.mol
                t1 = time.time()
                self.with_df.build(eng, self._rank, self._nranks)
                # SCF densities have the rank of the occupied space: the exchange build factorises them (df.pivoted_cholesky)
                self.with_df.rank_hint = (int(self.mol.nelectron) + int(getattr(self.mol, "spin", 0))) // 2
                torch.cuda.synchronize()
                self.timing["df_seconds"] = time.time() - t1
                self._log(4, f"density fitting: {self.with_df.naux} auxiliary functions, tensor "
                             f"{8e-9 * self.with_df.naux * eng.nao ** 2:.2f} GB, built in {self.timing['df_seconds']:.3f} s")
        elif not eng.eri_ready and self._stream_groups <= 1:
            oom = False
            if getattr(self, "_grad_prefetch", False):
                # a gradient follows this SCF (geometry optimisation, scanner): the host half of its set-up runs on a helper
                # thread of the library while the SCF loop keeps the device busy
                eng.set_option("grad_prefetch", 1)
            try:
                st = eng.prepare_eri(self.direct_scf_tol, self._rank, self._nranks)
                self.timing["eri_seconds"] = st["seconds_eri"]
                self._log(4, f"resident ERI store: {st['n_tiles']} tiles, {st['stored_bytes'] / 1e6:.1f} MB, "
                             f"{st['n_quartets']} shell quartets in {st['seconds_eri']:.3f} s")
                need, free = eng.eri_memory()
            except _engine.EngineOutOfMemory as e:   # MI_ERR_NOMEM: sizes come through the ABI (mi_eri_get_memory)
                oom, need, free = True, e.need_bytes, e.free_bytes
            view = getattr(self, "_test_memory_view", None)
            if view is not None:    # tests: pretend this rank saw (oom, need_bytes, free_bytes)
                oom, need, free = view
            if self._nranks > 1:
                # The direct-mode split is (rank * ng + v, nranks * ng) over ONE flat LPT deal of the tile runs, so every rank
                # must use the same ng (and the same number of resident groups): with rank-local values the tile sets of two
                # ranks would be neither disjoint nor exhaustive.  One MAX all-reduce at set-up: direct mode if ANY rank's
                # shard does not fit, sized for the largest shard and the smallest free HBM.
                from . import parallel
                o, n_, f_ = parallel.agree_max([1.0 if oom else 0.0, float(need), -float(free)], self._pg)
                oom, need, free = o > 0.0, n_, -f_
            if oom:
                if eng.eri_ready:
                    eng.release_eri()       # this rank's shard fitted, another rank's did not: everybody goes direct
                need, free = need * 1e-9, free * 1e-9
                self._stream_groups = max(2, int(np.ceil(need / max(0.8 * free, 1.0))))
                if self.direct_resident:
                    self._stream_groups, self._resident_groups = self._plan_direct_groups(need, free, self._stream_groups)
                self._log(3, f"ERI tensor shard ({need:.0f} GB) exceeds free HBM ({free:.0f} GB): direct mode, "
                             f"{self._stream_groups} tile groups, {self._resident_groups} kept resident, the others "
                             "re-evaluated and digested every Fock build")
        self.timing["setup_seconds"] = time.time() - t0

    def get_ovlp(self, mol=None):
        return self.engine.int1e()[0].cpu().numpy()

    def get_hcore(self, mol=None):
        _, T, V = self.engine.int1e()
        return (T + V).cpu().numpy()

    def energy_nuc(self):
        return self.mol.energy_nuc()

    direct_resident = True     # direct mode: keep leading tile groups resident (engines of their own) when that pays
    direct_reserve_gb = 2.0    # ... leaving this much (beyond 15 % of the HBM) beside the streaming buffer; Kohn-Sham classes: 10

    def _plan_direct_groups(self, need_gb, free_gb, ng_min):
        """(number of tile groups, how many of them stay resident) for a tensor of `need_gb` that does not fit `free_gb`.
        Evaluating ONE group costs a fixed part (host planning of the whole tensor 0.17 s + every quartet of the molecule is
        enumerated and the ones of other groups skipped 0.3 s, C60/6-31G*: tools/prepare_laps.py) plus its share of the 1.65 s
        the quartets themselves take, so a Fock build with s streamed groups out of ng costs about s (0.28 + 1 / ng) in units
        of the latter: many small groups pack the HBM better but pay the fixed part too often (measured: 16 groups with 7
        resident 5.3 s per build, 3 groups with none 3.4 s).  C60 on one 288 GB GPU: 4 groups, 1 resident (-14 %)."""
        usable = 0.85 * free_gb - self.direct_reserve_gb
        best = (ng_min * (0.28 + 1.0 / ng_min), ng_min, 0)
        for ng in range(ng_min, ng_min + 6):
            g = 1.02 * need_gb / ng
            if g > usable:
                continue
            r = min(int((usable - g) // g), ng - 1)
            cost = (ng - r) * (0.28 + 1.0 / ng)
            if cost < best[0] - 1e-9:
                best = (cost, ng, r)
        return best[1], best[2]

    def _jk_streamed(self, dm, with_j, with_k):
        """Direct (recompute) mode: the rank's tile runs are cut into `_stream_groups` groups (LPT dealing by bytes inside
        `mi_eri_prepare(rank * ng + v, nranks * ng)`, so the groups are equal shares).  The first `_resident_groups` of them
        are evaluated ONCE per geometry on engines of their own and stay in HBM (as many as fit beside one streaming buffer);
        the others are evaluated by the Rys kernels, digested and discarded every Fock build.  Same kernels as the resident
        mode; (groups, resident) come from `_plan_direct_groups`: C60/6-31G* (499 GB of tiles) on one 288 GB GPU keeps 1 of 4
        groups, so a Fock build re-evaluates 75 % of the tensor in three passes instead of all of it in three."""
        eng = self.engine
        ng = self._stream_groups
        J = K = None

        def add(j, k):
            nonlocal J, K
            J = j if J is None else (J + j if with_j else None)
            K = k if K is None else (K + k if with_k else None)

        ge = self._group_engines
        nres = min(int(getattr(self, "_resident_groups", 0) or 0), ng - 1)
        gstats = self.__dict__.setdefault("_group_stats", {})
        while len(ge) < nres:
            v = len(ge)
            g = _engine.Engine(self.mol, device=eng.device)
            ok = True
            try:
                gstats[v] = g.prepare_eri(self.direct_scf_tol, self._rank * ng + v, self._nranks * ng)
            except _engine.EngineOutOfMemory:
                ok = False
            if ok:
                free, _tot = torch.cuda.mem_get_info(eng.device)
                more = free >= 2.3 * g.stats()["stored_bytes"] + self.direct_reserve_gb * 2 ** 30
            else:
                more = False
            if self._nranks > 1:
                # which groups stay resident decides which (rank * ng + v) shares are streamed by eng below: the group INDEX
                # sets must match on all ranks (each group index is a different share of every rank's runs), so a rank-local
                # failure or memory shortage ends the resident prefix everywhere
                from . import parallel
                bad, stop = parallel.agree_max([0.0 if ok else 1.0, 0.0 if more else 1.0], self._pg)
                if bad > 0.0 and ok:
                    gstats.pop(v, None)
                ok, more = bad == 0.0, stop == 0.0
            if not ok:
                g.close()
                nres = self._resident_groups = len(ge)
                break
            ge.append(g)
            if not more:
                nres = self._resident_groups = len(ge)   # no room for one more AND the streaming buffer of the other groups
        for g in ge[:nres]:
            add(*g.get_jk(dm, with_j, with_k))
        for v in range(nres, ng):
            gstats[v] = eng.prepare_eri(self.direct_scf_tol, self._rank * ng + v, self._nranks * ng)
            add(*eng.get_jk(dm, with_j, with_k))
        return J, K   # pair records / Schwarz data of the last group stay valid (used by the gradient)

    def _jk(self, dm, with_j=True, with_k=True):
        if getattr(self, "with_df", None) is not None:     # fitted integrals: dense GEMMs on this rank's slice of the resident B tensor
            J, K = self.with_df.get_jk(dm, with_j, with_k)
            if self._nranks > 1:
                from . import parallel
                parallel.all_reduce_fused([x for x in (J, K) if x is not None], self._pg)
            return J, K
        if self._stream_groups > 1:
            J, K = self._jk_streamed(dm, with_j, with_k)
        else:
            J, K = self.engine.get_jk(dm, with_j, with_k)
            if self._nranks > 1:
                from . import parallel
                parallel.all_reduce_sum(self.engine.last_jk_buffer, self._pg)   # [J|K] in place, one collective
            return J, K
        if self._nranks > 1:
            from . import parallel
            parallel.all_reduce_fused([x for x in (J, K) if x is not None], self._pg)
        return J, K

    def _jk_into(self, dm, J, K):
        """This rank's PARTIAL J (and K unless None) written into caller-owned views, no collective: the caller all-reduces
        the buffer the views live in (Kohn-Sham: one fused [J|K|Vxc|N|Exc] collective per Fock build)."""
        if getattr(self, "with_df", None) is not None:
            j, k = self.with_df.get_jk(dm, True, K is not None)   # partial sums over this rank's slice of the auxiliary index
            J.copy_(j)
            if K is not None:
                K.copy_(k)
            return
        if self._stream_groups > 1:
            j, k = self._jk_streamed(dm, True, K is not None)
            J.copy_(j)
            if K is not None:
                K.copy_(k)
        else:
            self.engine.get_jk(dm, True, K is not None, out_j=J, out_k=K)

    def get_jk(self, mol=None, dm=None, hermi=1, with_j=True, with_k=True, **kw):
        if dm is None:
            dm = self.make_rdm1()
        self._setup_once()   # integrals + resident ERI tiles (or the direct-mode fallback when the store does not fit)
        J, K = self._jk(dm, with_j, with_k)
        return (J.cpu().numpy() if with_j else None), (K.cpu().numpy() if with_k else None)

    def _veff(self, dm):
        """Returns (vhf, e_two_electron) on device.  RHF: vhf = J - K/2, E2 = 1/2 tr(D vhf)."""
        J, K = self._jk(dm)
        vhf = J - 0.5 * K
        return vhf, 0.5 * torch.sum(dm * vhf)

    def _fock_energy(self, dm, part):
        """F = h + veff(D) on device; `part` receives the fixed-order partial sums of E_elec(D) (fused kernel).  Returns
        (F, extra): `extra` is None or a device tensor whose LAST element is added to the energy (E_xc); RKS passes
        [N_elec on the grid, E_xc] so that the host can also validate the quadrature of the cycle.  Overridden by RKS."""
        if self._fused_fock_ok(dm):
            return self.engine.build_fock(dm, self._h1, 0.5, torch.empty_like(dm), part), None
        J, K = self._jk(dm)
        F = torch.empty_like(J)
        self.engine.fock_energy(self._h1, J, K, None, dm, 0.5, F, part)
        return F, None

    fused_fock = True   # single rank, resident tiles: F and the energy partials straight from the J/K accumulators (mi_build_fock)

    def _fused_fock_ok(self, dm):
        return (self.fused_fock and self._nranks == 1 and self._stream_groups <= 1 and getattr(self, "with_df", None) is None
                and dm.dim() == 2 and dm.is_contiguous())

    @staticmethod
    def _sp2_traces(tr_host):
        """(tr X, tr X^2) from the interleaved partial traces of the fused SP2 kernel (or a plain pair), added in index order."""
        t = np.asarray(tr_host, dtype=np.float64).reshape(-1, 2)
        return float(t[:, 0].sum()), float(t[:, 1].sum())

    def get_veff(self, mol=None, dm=None, **kw):
        if dm is None:
            dm = self.make_rdm1()
        self._setup_once()
        d = torch.as_tensor(np.asarray(dm), dtype=torch.float64, device=self.engine.device)
        return self._veff(d)[0].cpu().numpy()

    def _eig(self, f):
        Li = self._Linv
        e, c = torch.linalg.eigh(Li @ f @ Li.T)
        return e, Li.T @ c

    def _density_sp2(self, f, nocc, orth=False):
        """2 P_occ(F) without diagonalisation: D' in the orthonormal basis if `orth` (then `f` is F'), else
        the AO density.  Returns None if the purification does not converge (e.g. vanishing HOMO-LUMO gap);
        the caller then falls back to `eigh`."""
        Li = self._Linv
        self._sp2_orth = orth
        fo = f if orth else Li @ f @ Li.T
        n = fo.shape[0]
        if nocc == 0 or nocc >= n:
            return None
        eng = self.engine
        if n <= self.sp2_fused_max and self.sp2_fused:
            return self._density_sp2_fused(fo, nocc)
        buf = getattr(self, "_sp2_buf", None)
        if buf is None or buf[0].numel() != 2 + n * n:
            buf = [torch.empty(2 + n * n, dtype=torch.float64, device=fo.device) for _ in range(2)]
            self._sp2_buf = buf
            self._sp2_x2 = torch.empty(n, n, dtype=torch.float64, device=fo.device)
        cur = 0
        X = buf[cur][2:].view(n, n)
        eng.sp2_init(fo.contiguous(), X, buf[1 - cur])
        X2 = self._sp2_x2
        nit = getattr(self, "_sp2_iters", 24)
        target = float(nocc)
        done = 0
        for attempt in range(6):
            for _ in range(nit - done):
                torch.matmul(X, X, out=X2)
                eng.sp2_update(X, X2, target, buf[1 - cur])   # one fused launch: traces, branch, update
                cur = 1 - cur
                X = buf[cur][2:].view(n, n)
            done = nit
            torch.matmul(X, X, out=X2)
            tr = torch.stack([torch.trace(X), torch.trace(X2)]).cpu()
            err = float(tr[0] - tr[1])          # = sum lambda (1 - lambda) >= 0
            if abs(err) < self.sp2_tol and abs(float(tr[0]) - target) < 1e-8:
                self._sp2_iters = nit
                self._sp2_validated = True
                Xs = X + X.T                       # exactly symmetric 2 X (see _sp2_planned_gemm)
                return Xs if self._sp2_orth else Li.T @ Xs @ Li
            nit += 8
        return None

    def _density_sp2_fused(self, fo, nocc):
        """Same SP2 recursion, one fused HIP launch per step (`sp2_fused_kernel`, FP64 MFMA)."""
        eng, Li = self.engine, self._Linv
        n = fo.shape[0]
        ws = getattr(self, "_sp2f", None)
        if ws is None or ws["X"].shape[0] != n:
            mk = lambda *s: torch.empty(*s, dtype=torch.float64, device=fo.device)
            ws = self._sp2f = dict(X=mk(n, n), X2=mk(n, n), work=mk(2 * n * n), tr=mk(64 * 80), b=mk(2 * n))
        nit = min(getattr(self, "_sp2_iters", 24), 72)
        target = float(nocc)
        nbd = (n + 15) // 16
        for attempt in range(5):
            eng.sp2_init(fo.contiguous(), ws["X"], ws["b"])
            off = eng.sp2_iterate(ws["X"], ws["X2"], nit, target, ws["work"], ws["tr"])
            tr = self._sp2_traces(ws["tr"][off:off + 2 * nbd].cpu().numpy())
            err = float(tr[0] - tr[1])
            if abs(err) < self.sp2_tol and abs(float(tr[0]) - target) < 1e-8:
                self._sp2_iters = nit
                self._sp2_validated = True
                if (self.sp2_trace_plan and self.sp2_planned and self._sp2_plan is None and self._sp2_plannable(n)
                        and off == 64 * nit):
                    # cold object: the history of this checked run and the Gershgorin discs give the bounds for a plan
                    # (one more small copy in a path that waits for the device anyway; see _plan_from_traces)
                    h = torch.cat([ws["tr"][:off + 64], ws["b"][:2 * n]]).cpu().numpy()
                    hh = h[:off + 64].reshape(nit + 1, 32, 2)[:, :nbd, :]
                    self._trace_bounds = (hh[:, :, 0].sum(axis=1), hh[:, :, 1].sum(axis=1), float(h[off + 64:off + 64 + n].min()),
                                          float(h[off + 64 + n:].max()), -1)
                return 2.0 * ws["X"] if self._sp2_orth else 2.0 * (Li.T @ ws["X"] @ Li)
            nit = min(nit + 8, 76)
        return None

    def _sp2_fused_async(self, fo, nocc):
        """Optimistic SP2: enqueue the purification with the iteration count that worked last cycle and return
        (D', device traces) WITHOUT a host sync; the caller validates tr(X - X^2) together with the cycle's other
        scalars and redoes the cycle through the checked path if the count was too small."""
        eng = self.engine
        n = fo.shape[0]
        ws = getattr(self, "_sp2f", None)
        if ws is None or ws["X"].shape[0] != n:
            mk = lambda *s: torch.empty(*s, dtype=torch.float64, device=fo.device)
            ws = self._sp2f = dict(X=mk(n, n), X2=mk(n, n), work=mk(2 * n * n), tr=mk(64 * 80), b=mk(2 * n))
        nit = min(self._sp2_iters, 76)
        if n <= self.sp2_fused_max and self.sp2_fused:
            pp = ws.get("pp")
            if pp is None:   # two [X | X2] buffers: the passes ping-pong between them and the result is read where it lands
                pp = ws["pp"] = (torch.empty(2, n, n, dtype=torch.float64, device=fo.device),
                                 torch.empty(2, n, n, dtype=torch.float64, device=fo.device))
            eng.sp2_init(fo.contiguous(), pp[0][0], ws["b"])
            res, off = eng.sp2_iterate_pingpong(pp[0], pp[1], nit, float(nocc), ws["tr"])
            # the partial traces of EVERY step (64 slots per step, 2 ceil(n/16) used): the host validates the last step and
            # reads off the first step at which the projector was already converged (-> iteration count of the next cycle)
            self._sp2_hist_shape = (nit + 1, (n + 15) // 16)
            if self.sp2_trace_plan and self.sp2_planned and self._sp2_plan is None and self._sp2_plannable(n):
                # cold object: the Gershgorin discs of this F' (already on the device for X_0) travel with the traces, so that the
                # host can read spectral bounds for a purification PLAN off this run (sp2plan.bounds_from_traces)
                self._sp2_hist_shape = (nit + 1, (n + 15) // 16, 2 * n)
                return 2.0 * res[0], torch.cat([ws["tr"][:off + 64], ws["b"][:2 * n]])
            return 2.0 * res[0], ws["tr"][:off + 64]
        eng.sp2_init(fo.contiguous(), ws["X"], ws["b"])
        # larger N: rocBLAS DGEMM + fused update kernel per step, still without a host sync
        buf = getattr(self, "_sp2_buf", None)
        if buf is None or buf[0].numel() != 2 + n * n:
            buf = self._sp2_buf = [torch.empty(2 + n * n, dtype=torch.float64, device=fo.device) for _ in range(2)]
        X, X2, cur = ws["X"], ws["X2"], 0
        for _ in range(nit):
            torch.matmul(X, X, out=X2)
            eng.sp2_update(X, X2, float(nocc), buf[cur])
            X = buf[cur][2:].view(n, n)
            cur = 1 - cur
        torch.matmul(X, X, out=X2)
        return X + X.T, torch.stack([torch.trace(X), torch.trace(X2)])   # exactly symmetric 2 X (see _sp2_planned_gemm)

    xc_nelec_rtol = 2e-4   # relative error of the grid electron count a low-rank-factor cycle may show (level-3 grids: ~1e-5)
    _HEAD_MAX = 4096   # doubles reserved in front of the planned-path trace history for [E partials | |g|^2 partials | extra]

    def _sp2_planned_async(self, fo, nocc, scale=2.0):
        """Planned purification, no host sync: (D' = 2 X, partial traces of every pass) -- validated by the caller like the
        optimistic SP2 path.  The traces land behind `_HEAD_MAX` doubles of one persistent buffer whose head the Fock build
        fills afterwards, so the cycle's scalars leave the device as ONE contiguous copy without a gather kernel."""
        eng = self.engine
        n = fo.shape[0]
        if not (n <= self.sp2_fused_max and self.sp2_fused):
            return self._sp2_planned_gemm(fo, nocc, scale)
        ws = getattr(self, "_sp2p", None)
        if ws is None or ws["n"] != n:
            mk = lambda *s: torch.empty(*s, dtype=torch.float64, device=fo.device)
            ws = self._sp2p = dict(n=n, scal=mk(self._HEAD_MAX + 64 * 80), pp=(mk(2, n, n), mk(2, n, n)))
        coef = self._sp2_plan[:self._sp2_plan_len + 1]
        tr = ws["scal"][self._HEAD_MAX:]
        res, off = eng.sp2_iterate_planned(fo.contiguous(), ws["pp"][0], ws["pp"][1], coef, tr, out_scale=scale)
        self._sp2_hist_shape = (coef.shape[0], (n + 15) // 16)
        return res[0], tr[:off + 64]   # a view of the ping-pong buffers: consumed by this cycle's Fock build, before the next pass

    sp2_plan_gnorm = 2e-3      # make the purification plan only once the previous cycle's |g| is below this
    # A diagonalisation made only to obtain the plan's bounds (7 ms at N = 264, 40 ms at 573) is never earned back inside one
    # SCF: the planned path saves 0.3 ms per cycle at N = 264 (0.5 ms at 573).  So by default a cold object runs the
    # trace-correcting purification throughout; the final diagonalisation of kernel() (needed for mo_energy anyway) seeds the
    # plan, and every later kernel() of the object -- geometry steps, scans, restarts -- is planned and pipelined from its first
    # cycle.  True: also plan inside the first SCF, once it has settled (long SCFs).
    sp2_plan_inloop = False
    sp2_planned_gemm = True   # N > sp2_fused_max: the same planned sequence with one rocBLAS DGEMM (addmm) per pass

    def _sp2_plannable(self, n):
        return (n <= self.sp2_fused_max and self.sp2_fused) or self.sp2_planned_gemm

    def _sp2_planned_gemm(self, fo, nocc, scale=2.0):
        """Planned purification for matrices beyond the fused kernel (ibuprofen N = 573, C60 N = 840): X_{k+1} = a X_k^2 + b X_k
        + c I as ONE `addmm` (rocBLAS DGEMM with beta) plus a diagonal shift per pass -- half the passes of the trace-
        correcting recursion of `_sp2_fused_async`, and no branch decisions on the device.  Only the last pass is checked:
        tr X and tr X^2 = |X|_F^2 (X is symmetric) travel to the host with the cycle's other scalars."""
        n = fo.shape[0]
        coef = self._sp2_plan[:self._sp2_plan_len + 1]
        buf = getattr(self, "_sp2g", None)
        if buf is None or buf[0].shape[0] != n:
            buf = self._sp2g = [torch.empty(n, n, dtype=torch.float64, device=fo.device) for _ in range(3)]
        X = torch.mul(fo, float(coef[0, 1]), out=buf[0])
        X.diagonal().add_(float(coef[0, 2]))
        cur, nit = 0, coef.shape[0] - 1
        for k in range(1, nit + 1):
            a, b, c = (float(v) for v in coef[k])
            Y = torch.addmm(X, X, X, beta=b, alpha=a, out=buf[(cur + 1) % 3])
            cur = (cur + 1) % 3
            if c != 0.0:
                Y.diagonal().add_(c)
            if k % 4 == 0 or k == nit:
                # a library GEMM does not return X.X exactly symmetric, and the antisymmetric part A obeys A <- a (SA + AS) + b A:
                # it can double per pass while the gap is being opened (1e-16 -> 1e-12 over 20 passes, measured).  The J/K kernel
                # reads one triangle of D, so an asymmetric D shows up as 1e-9 Ha cycle-to-cycle jitter of a 650 Ha energy
                # (tools/noise_check.py; the fused kernel's mirror stores keep X exactly symmetric)
                Y = torch.add(Y, Y.T, out=buf[(cur + 1) % 3]).mul_(0.5)
                cur = (cur + 1) % 3
            X = Y
        self._sp2_hist_shape = None
        tr = torch.stack([torch.trace(X), torch.sum(X * X)])
        return scale * X, tr

    # Cold object (round 3): the trace-correcting purification of a cycle leaves, for free, an interval inside the HOMO-LUMO gap
    # (sp2plan.gap_from_traces) and Gershgorin bounds outside; once |g| is below `sp2_trace_plan_gnorm` a plan is made from them
    # and the remaining cycles of the FIRST kernel() of an object take the planned, pipelined head too (a warm object's plan
    # comes from its last diagonalisation and is tighter: it replaces this one at the end of the SCF).
    sp2_trace_plan = True
    sp2_first_passes = 48     # passes of the very first (optimistic) purification of an object; 0: checked path
    sp2_trace_plan_gnorm = 2e-2
    _trace_bounds = None
    _sp2_plan_from_traces = False

    def _plan_from_traces(self, nocc):
        from . import sp2plan
        tx, tx2, emin, emax, _cycle = self._trace_bounds
        self._trace_bounds = None
        b = sp2plan.bounds_from_traces(tx, tx2, emin, emax, self.sp2_inner_margin)
        plan = sp2plan.plan(*b) if b is not None else None
        if plan is not None:
            self._sp2_plan = plan
            self._sp2_plan_len = plan.shape[0] - 1
            self._sp2_plan_gen = getattr(self, "_sp2_plan_gen", 0) + 1
            self._sp2_plan_from_traces = True
            self.path_counts["plan_from_traces"] = self.path_counts.get("plan_from_traces", 0) + 1

    def _sp2_replan(self, mo_e, nocc):
        """New plan from the eigenvalues of the (orthonormal-basis) Fock matrix just diagonalised."""
        from . import sp2plan
        e = mo_e.cpu().numpy() if torch.is_tensor(mo_e) else np.asarray(mo_e)
        self._sp2_plan = None
        self._sp2_plan_from_traces = False
        self._sp2_plan_gen = getattr(self, "_sp2_plan_gen", 0) + 1
        if self.sp2_planned and self.eig_method == "sp2" and 0 < nocc < len(e) and self._sp2_plannable(len(e)):
            b = sp2plan.bounds_from_spectrum(e, nocc, self.sp2_inner_margin, self.sp2_outer_margin)
            self._sp2_plan = sp2plan.plan(*b)
            if self._sp2_plan is not None:
                self._sp2_plan_len = self._sp2_plan.shape[0] - 1

    def make_rdm1(self, mo_coeff=None, mo_occ=None):
        if mo_coeff is None:
            if self._dm is not None:
                return self._dm.cpu().numpy()
            mo_coeff, mo_occ = self.mo_coeff, self.mo_occ
        mo_coeff, mo_occ = np.asarray(mo_coeff), np.asarray(mo_occ)
        c = mo_coeff[:, mo_occ > 0]
        return (c * mo_occ[mo_occ > 0]) @ c.T

    # --- initial guess (row a13) ----------------------------------------------------------------
    def get_init_guess(self, mol=None, key=None):
        key = (key or self.init_guess)
        if isinstance(key, np.ndarray):
            return key
        key = str(key).lower()
        if key in ("1e", "hcore"):
            self._setup_once()
            _, c = self._eig(self._h1)
            nocc = self.mol.nelectron // 2
            co = c[:, :nocc]
            return (2.0 * co @ co.T).cpu().numpy()
        return self._init_guess_by_atom()

    def _setup_once(self):
        if getattr(self, "_h1", None) is None or self._eng is None:
            self._setup()

    def _init_guess_by_atom(self):
        """Superposition of spherically averaged, fractionally occupied atomic SCF densities computed
        with this same engine on each distinct element ('atom' guess; 'minao' maps here too)."""
        mol = self.mol
        dm = np.zeros((mol.nao, mol.nao))
        cache = {}
        sl = mol.aoslice_by_atom()
        for ia in range(mol.natm):
            sym = mol.atom_symbol(ia)
            if mol.atom_charge(ia) == 0:
                continue
            if sym not in cache:
                cache[sym] = _atomic_density(sym, mol.basis, mol.atom_charge(ia))
            p0, p1 = sl[ia, 2], sl[ia, 3]
            dm[p0:p1, p0:p1] = cache[sym]
        return dm

    # --- the SCF loop (row a12) -----------------------------------------------------------------
    def _start(self, dm0=None):
        """Prepare integrals/ERIs and the iteration state.  The loop works in the Cholesky-orthogonalised
        basis (F' = L^-1 F L^-T, D' = L^T D L); the DIIS error vector is PySCF's orthonormal-basis C^T (SDF - FDS) C, whose Gram
        matrix equals that of [F', D'], so the extrapolation path is the reference one."""
        mol = self.mol
        self._setup_once()
        eng = self.engine
        self._trace_bounds = None
        t0 = time.time()
        if dm0 is None:
            dm0 = self.get_init_guess()
        self.timing["guess_seconds"] = time.time() - t0
        t0 = time.time()
        dm = torch.as_tensor(np.asarray(dm0), dtype=torch.float64, device=eng.device).contiguous()
        if self._nranks > 1:   # one-off: identical starting density on every rank (the atomic guess is built with atomics per rank)
            from . import parallel
            parallel.broadcast0(dm, self._pg)
        self._fgraph = self._fgraph_seen = None     # a captured head belongs to one SCF (its CDIIS history buffers)
        st = {"nocc": mol.nelectron // 2, "enuc": mol.energy_nuc(), "cycle": 0, "diis": DeviceDIIS(eng, self.diis_space)}
        st["dmo"] = self._L.T @ dm @ self._L
        self._after_density(st, dm, e_last=None, next_cycle=0)
        self.timing["first_fock_seconds"] = time.time() - t0
        return st

    def _after_density(self, st, dm, e_last, next_cycle, sp2_tr=None, nocc=0, hist_shape=None):
        """J/K(+XC) for `dm`, new Fock in the orthonormal basis, commutator error, energy, |g|; pushes
        (F', e) into the DIIS history and fetches all scalars of the cycle with ONE device-to-host copy."""
        ctx = self._after_density_launch(st, dm, next_cycle, sp2_tr, hist_shape)
        return self._after_density_finish(st, ctx, e_last, nocc)

    _PIN_DOUBLES = 16384

    def _after_density_launch(self, st, dm, next_cycle, sp2_tr=None, hist_shape=None, projector=False):
        """Device part of `_after_density`: everything is queued, the scalars of the cycle are on their way to pinned host
        memory (asynchronous copy + event) when this returns -- the caller may queue more work before `_after_density_finish`
        waits for them."""
        Li = self._Linv
        dm = dm.contiguous()
        # `projector`: dm = L^-T (2 X) L^-1 with X (= st["dmo"] / 2) an idempotent of rank n_occ -- lets the XC quadrature of
        # RKS work from occupied-orbital values instead of the full density matrix (dft.RKS._occ_factor)
        self._xc_projector = (dm, st["dmo"], st["nocc"]) if projector else None
        nb = self.engine.reduce_blocks
        self.n_fock_builds = getattr(self, "n_fock_builds", 0) + 1
        # partial sums of [E_elec | |[F',D']|^2]: on the planned path they go right in front of the trace history
        ws = getattr(self, "_sp2p", None)
        inplace = (sp2_tr is not None and ws is not None and 2 * nb <= self._HEAD_MAX
                   and sp2_tr.data_ptr() == ws["scal"].data_ptr() + 8 * self._HEAD_MAX)
        if inplace:
            part = ws["scal"][self._HEAD_MAX - 2 * nb:self._HEAD_MAX]
        else:
            part = torch.empty(2 * nb, dtype=torch.float64, device=dm.device)
        fock, extra = self._fock_energy(dm, part[:nb])
        # PySCF feeds CDIIS only from cycle `diis_start_cycle` on [MEM]: the initial-guess Fock is not stored.  When it is
        # stored, the two GEMM chains write F' and the error vector straight into the history slot (no device copies).
        diis = st["diis"]
        keep = next_cycle >= self.diis_start_cycle
        slot = diis.next_slot()
        fo = torch.matmul(Li @ fock, Li.T, out=diis.F[slot]) if keep else Li @ fock @ Li.T
        m = fo @ st["dmo"]
        # CDIIS error vector in an ORTHONORMAL basis, as PySCF >= 2.1 forms it (scf/diis.py get_err_vec_orth: C^T (SDF - FDS) C
        # with C = `Corth`, the eigenvectors of the first Fock matrix [MEM; the reference pins pyscf 2.8.0]).  With the Cholesky
        # basis X = L^-T instead of C, X^T (SDF - FDS) X = D'F' - F'D' = -[F', D']; the two orthonormal bases differ by an
        # orthogonal matrix, which leaves every <e_i, e_j> -- all that the Pulay system sees -- unchanged.  So the commutator
        # IS the error vector: it is written straight into the history slot (round 1 transformed it to the AO basis, the
        # pre-2.1 definition, with two more GEMMs).  The push also solves the Pulay system on the device.
        eo = diis.E[slot] if keep else torch.empty_like(m)
        self.engine.commutator_norm(m, eo, part[nb:])  # eo = [F', D'] and the partial sums of its squared norm
        if keep:
            diis.push_inplace()
        if inplace and extra is None:
            packed = ws["scal"][self._HEAD_MAX - 2 * nb:self._HEAD_MAX + sp2_tr.numel()]   # already contiguous: no gather kernel
        else:
            parts = [part] + ([extra.reshape(-1)] if extra is not None else []) + ([sp2_tr] if sp2_tr is not None else [])
            packed = torch.cat(parts) if len(parts) > 1 else part
        # Sharded runs: every rank holds the same all-reduced J/K(/Vxc) and the replicated algebra above is free of atomics
        # (fixed-order partial sums), so these scalars should be bit-identical on all ranks; until a multi-GPU run has confirmed
        # that for the library GEMMs in between, rank 0's copy is made authoritative (`sync_control`, one small broadcast).
        if self._sync_control_on():
            from . import parallel
            parallel.broadcast0(packed, self._pg)
        ctx = dict(dm=dm, fock=fock, fo=fo, nb=nb, n_extra=0 if extra is None else extra.numel(), has_tr=sp2_tr is not None,
                   hist_shape=hist_shape, lowrank=self._xc_projector is not None,
                   packed=packed, event=None)
        k = packed.numel()
        if k <= self._PIN_DOUBLES:
            pin = getattr(self, "_pin", None)
            if pin is None:
                pin = self._pin = torch.empty(self._PIN_DOUBLES, dtype=torch.float64).pin_memory()
                self._pin_event = torch.cuda.Event()
            pin[:k].copy_(packed, non_blocking=True)
            self._pin_event.record()
            ctx["event"] = self._pin_event
        return ctx

    def _after_density_finish(self, st, ctx, e_last, nocc=0):
        """Host part: wait for the scalars of the cycle (the only host synchronisation of a cycle), validate the optimistic
        purification, update the state.  False: the purification had not converged -- nothing in `st` was touched."""
        nb = ctx["nb"]
        if ctx["event"] is not None:
            ctx["event"].synchronize()
            vals = self._pin[:ctx["packed"].numel()].numpy().copy()
        else:
            vals = ctx["packed"].cpu().numpy()
        e_el = float(vals[:nb].sum())                  # numpy's pairwise sum: the same order on every rank
        c2 = float(vals[nb:2 * nb].sum())
        pos = 2 * nb
        if ctx["n_extra"]:
            ne = ctx["n_extra"]
            if ne == 2 and ctx["lowrank"]:
                # RKS: electron count of this cycle's quadrature.  The density came from a low-rank factor of the projector
                # (dft.RKS._occ_factor): a failed factorisation (NaN, or a count off by more than the grid error) sends the
                # cycle through the redo path, which uses the full density matrix
                nel = float(vals[pos])
                if not (abs(nel - 2.0 * st["nocc"]) < self.xc_nelec_rtol * 2.0 * st["nocc"]):
                    return False
            e_el += float(vals[pos + ne - 1])
            pos += ne
        if ctx["has_tr"]:
            hist = vals[pos:]
            shape = ctx["hist_shape"]
            discs = None
            if shape is not None and len(shape) > 2 and shape[2] and hist.size == shape[0] * 64 + shape[2]:
                discs, hist = hist[-shape[2]:], hist[:-shape[2]]
            if shape is not None and hist.size == shape[0] * 64:
                h = hist.reshape(shape[0], 32, 2)[:, :shape[1], :]
                tx, tx2 = h[:, :, 0].sum(axis=1), h[:, :, 1].sum(axis=1)     # per step, partials added in index order
                ok = (np.abs(tx - tx2) < self.sp2_tol) & (np.abs(tx - nocc) < 1e-8)
                if not ok[-1]:
                    return False
                self._sp2_validated = True
                first_ok = int(np.argmax(ok))            # steps beyond it were not needed for this Fock matrix
                if getattr(self, "_sp2_planned_pass", False):
                    self._sp2_plan_len = min(self._sp2_plan.shape[0] - 1, max(first_ok + 1, 4))
                else:
                    self._sp2_iters = max(first_ok + self.sp2_margin, 4)
                    if discs is not None:      # trace-correcting run of a cold object: keep what a plan needs (made in _step)
                        nd = discs.size // 2
                        self._trace_bounds = (tx.copy(), tx2.copy(), float(discs[:nd].min()), float(discs[nd:].max()), st["cycle"])
            else:
                trx, trx2 = self._sp2_traces(hist)
                if not (abs(trx - trx2) < self.sp2_tol and abs(trx - nocc) < 1e-8):
                    return False
        e_tot = e_el + st["enuc"]
        n = ctx["fo"].shape[0]
        nvo = max((n - st["nocc"]) * st["nocc"], 1)
        # |g| = |2 F_vo| = |[F',D']|_F / sqrt(2), normalised by sqrt(n_vo) like PySCF's get_grad norm [MEM]
        gnorm = float(np.sqrt(max(c2, 0.0))) / np.sqrt(2.0) / np.sqrt(nvo)
        st.update(dm=ctx["dm"], fock=ctx["fock"], fo=ctx["fo"], e_tot=e_tot, gnorm=gnorm,
                  de=(e_tot - e_last) if e_last is not None else 0.0)
        return True

    cold_pipeline = True   # first SCF of an object: pipelined, optimistic trace-correcting purification (no plan needed)
    cold_margin = 6        # extra purification passes queued while |g| > sp2_plan_gnorm (the needed count still moves)
    pipeline = True   # queue the device-only head of cycle k+1 (extrapolation, purification, density) before waiting for cycle k's scalars

    def _front(self, st):
        """Device-only head of the NEXT cycle, queued speculatively: CDIIS-extrapolated F' (coefficients solved on the device)
        -> planned purification -> AO density.  The GPU works on it while the host reads back and checks the scalars of the
        cycle that just finished; the caller drops it when that cycle turns out converged or invalid.  None when the next
        cycle cannot take the planned path."""
        nocc = st["nocc"]
        n = self._Linv.shape[0]
        if not (self.pipeline and self.eig_method == "sp2" and 0 < nocc < n and not self.level_shift
                and st["cycle"] + 1 >= self.diis_start_cycle and st["diis"].count > 0):
            return None
        use_gnorm = self.sp2_trace_plan_gnorm if self._sp2_plan_from_traces else self.sp2_plan_gnorm
        planned = (self.sp2_planned and self._sp2_plannable(n) and self._sp2_plan is not None
                   and st.get("gnorm", 0.0) <= use_gnorm)
        # a COLD object (first kernel() of the object: no plan yet, see `sp2_plan_inloop`) pipelines too: the trace-correcting
        # purification needs no spectral bounds, only a pass count -- the one the previous cycle needed plus a margin that is
        # generous while the spectrum still moves (a pass costs 7 us, a redone cycle a whole Fock build)
        cold = (not planned and self.cold_pipeline and (self._sp2_validated or self.sp2_first_passes) and n <= self.sp2_fused_max and self.sp2_fused
                and (self._sp2_plan is None or st.get("gnorm", 0.0) > use_gnorm))
        if not (planned or cold):
            return None
        if planned and self.graph_front:
            out = self._front_graphed(st, nocc)
            if out is not None:
                return out
        fo = st["diis"].extrapolate()
        if planned:
            dmo, tr_dev = self._sp2_planned_async(fo, nocc)
        else:
            keep = self._sp2_iters
            self._sp2_iters = min(keep + (self.cold_margin if st.get("gnorm", 0.0) > self.sp2_plan_gnorm else 0), 72)
            if not self._sp2_validated:      # head of cycle 2, queued before cycle 1's first purification has been validated
                self._sp2_iters = self.sp2_first_passes
            dmo, tr_dev = self._sp2_fused_async(fo, nocc)
            self._sp2_iters = keep
        shape, self._sp2_hist_shape = self._sp2_hist_shape, None
        dm = (self._Linv.T @ dmo @ self._Linv).contiguous()
        return dict(fo=fo, dmo=dmo, dm=dm, tr=tr_dev, shape=shape, planned=planned)

    # The planned head of a cycle is ~26 small launches (CDIIS combination, ~21 purification passes, 2-3 GEMMs) queued by the host
    # between the wait for cycle k's scalars and the J/K launch of cycle k+1.  On a loaded host (measured on boxes of the pool:
    # host time per cycle 0.26 ms or 0.6 ms from one run to the next) the device then waits for the host: 1.07 vs 1.25-1.5 ms per
    # cycle.  Once the plan and the CDIIS history length are stable the head is captured ONCE as a HIP graph (torch.cuda.CUDAGraph:
    # our ctypes launches go to torch's current stream, which is the capturing one) and replayed with a single launch call.
    graph_front = True
    _GRAPH_STABLE = 3   # cycles with an unchanged (plan, history length) before the head is captured

    def _front_graphed(self, st, nocc):
        """Replay (or capture) the planned head as a HIP graph; None = not available this cycle (the caller queues it eagerly)."""
        diis = st["diis"]
        m = min(diis.count, diis.space)
        key = (id(diis), m, self._sp2_plan_len, getattr(self, "_sp2_plan_gen", 0), self._Linv.data_ptr())
        g = self.__dict__.get("_fgraph")
        if g is not None and g["key"] == key and g["diis"] is diis:   # (the graph holds `diis` alive: its address cannot be reused)
            g["graph"].replay()
            self._sp2_hist_shape = None
            return dict(fo=g["fo"], dmo=g["dmo"], dm=g["dm"], tr=g["tr"], shape=g["shape"], planned=True)
        seen = self.__dict__.get("_fgraph_seen")
        if seen is None or seen[0] != key:
            self._fgraph_seen = [key, 1]
            return None
        seen[1] += 1
        if seen[1] < self._GRAPH_STABLE or m < diis.space:
            return None
        try:
            def body():
                fo = diis.extrapolate()
                dmo, tr_dev = self._sp2_planned_async(fo, nocc)
                shape, self._sp2_hist_shape = self._sp2_hist_shape, None
                dm = torch.matmul(self._Linv.T @ dmo, self._Linv)
                return fo, dmo, dm, tr_dev, shape
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                body()                       # warm-up outside the capture (library workspaces, lazy kernels)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fo, dmo, dm, tr_dev, shape = body()
            graph.replay()                   # the capture itself executes nothing
            self._fgraph = dict(key=key, diis=diis, graph=graph, fo=fo, dmo=dmo, dm=dm, tr=tr_dev, shape=shape)
            return dict(fo=fo, dmo=dmo, dm=dm, tr=tr_dev, shape=shape, planned=True)
        except Exception as e:   # capture not possible on this stack: stay eager for good
            self.graph_front = False
            self._fgraph = None
            self._log(4, f"HIP-graph capture of the SCF head failed ({e!r}); continuing with eager launches")
            return None

    def _step(self, st, use_diis=True, want_mo=False):
        """One SCF cycle: CDIIS extrapolation -> occupied projector (SP2 or eigh) -> density -> J/K ->
        energy, orbital gradient.  This is the unit bench.py times ("SCF iteration").  With `pipeline` the first three
        stages of a cycle are queued on the device by the PREVIOUS call, before it waited for its own scalars."""
        nocc, Li = st["nocc"], self._Linv
        front = st.pop("front", None)
        if want_mo or not use_diis:
            front = None                                 # final cycle: plain diagonalisation of the last Fock matrix
        paths = self.__dict__.setdefault("path_counts", {})   # diagnostics: purification path per cycle (tools/host_busy.py)

        def took(name):
            paths[name] = paths.get(name, 0) + 1
        tr_dev, hist_shape = None, None
        self._sp2_planned_pass = False
        if front is not None:
            fo, dmo, tr_dev, hist_shape, dm = front["fo"], front["dmo"], front["tr"], front["shape"], front["dm"]
            self._sp2_planned_pass = bool(front.get("planned", True))
            took("front_planned" if self._sp2_planned_pass else "front_cold")
            planned_ok = True
            st.pop("mo_e", None)
        else:
            if use_diis and st["cycle"] >= self.diis_start_cycle:
                fo = st["diis"].extrapolate()
            else:
                fo = st["fo"]
            if self.level_shift and use_diis:
                # PySCF `level_shift`: raise the virtual space of the matrix the new orbitals come from, F' + s (1 - D'/2), D' the
                # current projector x 2; it leaves a converged solution unchanged and is not applied to the final (extra) cycle
                fo = fo + self.level_shift * (torch.eye(fo.shape[0], dtype=fo.dtype, device=fo.device) - 0.5 * st["dmo"])
            use_sp2 = self.eig_method == "sp2" and not want_mo
            n = fo.shape[0]
            planned_ok = (use_sp2 and self.sp2_planned and self._sp2_plannable(n) and 0 < nocc < n and not self.level_shift)
            dmo = None
            # A plan needs spectral bounds, i.e. one diagonalisation (7 ms at N = 264, 40 ms at 573), and holds while HOMO / LUMO
            # move by less than its margins.  From a superposition-of-atoms guess the spectrum moves more than that in the
            # first two or three cycles (measured: a plan made at cycle 1 failed its trace check at cycles 2 and 3, each costing
            # another diagonalisation and a second Fock build), so while the orbital gradient of the previous cycle is above
            # `sp2_plan_gnorm` the trace-correcting purification runs instead -- it needs no bounds -- and the plan is made once,
            # when the SCF has settled; warm starts (dm0 from a nearby geometry) plan at their first cycle.
            early = planned_ok and self._sp2_plan is None and (st.get("gnorm", 0.0) > self.sp2_plan_gnorm or not self.sp2_plan_inloop)
            settled = st.get("gnorm", 0.0) <= (self.sp2_trace_plan_gnorm if self._sp2_plan_from_traces else self.sp2_plan_gnorm)
            if planned_ok and self._sp2_plan is not None and settled and not st.get("_redo"):
                dmo, tr_dev = self._sp2_planned_async(fo, nocc)
                self._sp2_planned_pass = True
                took("planned_nofront")
            elif planned_ok and self._sp2_plan is not None and not settled:
                # a plan exists (seeded by an earlier SCF of this object) but this SCF is still far from its solution -- e.g.
                # kernel() from the atomic guess again: the spectrum is not the planned one yet (measured: three redo cycles,
                # each a diagonalisation and a second Fock build).  Checked purification until the SCF has settled.
                dmo = self._density_sp2(fo, nocc, orth=True)
                took("checked_unsettled_with_plan")
            elif early and st.get("gnorm", 0.0) > self.sp2_plan_gnorm:
                if (self.cold_pipeline and (self._sp2_validated or self.sp2_first_passes) and not st.get("_redo")
                        and n <= self.sp2_fused_max and self.sp2_fused):
                    # optimistic: last count + a generous margin, validated with the cycle's scalars (no host sync here).  The
                    # very first purification of an object has no count yet: `sp2_first_passes` in one go instead of the checked
                    # path's 24 / 32 / 40 with a host round trip each (benzene/cc-pVTZ needs 34); too few -> the redo below
                    keep = self._sp2_iters
                    self._sp2_iters = min(keep + self.cold_margin, 72) if self._sp2_validated else self.sp2_first_passes
                    dmo, tr_dev = self._sp2_fused_async(fo, nocc)
                    self._sp2_iters = keep
                else:
                    # checked purification (validated on the host inside, iteration count adapted there)
                    dmo = self._density_sp2(fo, nocc, orth=True)
            elif early and self._sp2_validated and not st.get("_redo"):
                dmo, tr_dev = self._sp2_fused_async(fo, nocc)     # settled, but no plan on this object yet (see sp2_plan_inloop)
                took("optimistic_noplan")
            elif early:
                dmo = self._density_sp2(fo, nocc, orth=True)
            elif planned_ok:
                # no plan yet: diagonalise below, which also yields the bounds for one.  (Bounds from ~130 Lanczos steps on the
                # projected Fock matrix instead -- HOMO, LUMO with residual bounds, Gershgorin outside -- were tried: as torch
                # vector ops they cost as much as rocSOLVER's syevd at N = 264, 7 ms, and needed redo cycles when a Ritz value
                # had not converged; the plan survives on the object across kernel() calls, so warm starts pay nothing.)
                pass
            elif use_sp2 and self._sp2_validated and 0 < nocc < n and not st.get("_redo"):
                dmo, tr_dev = self._sp2_fused_async(fo, nocc)
            elif use_sp2:
                dmo = self._density_sp2(fo, nocc, orth=True)
            hist_shape, self._sp2_hist_shape = getattr(self, "_sp2_hist_shape", None), None
            if dmo is None:
                took("eigh")
                e, c = torch.linalg.eigh(fo)
                co = c[:, :nocc]
                dmo = 2.0 * co @ co.T
                st.update(mo_e=e, mo_c=Li.T @ c)
                if planned_ok or (want_mo and self.eig_method == "sp2" and not self.level_shift):
                    self._sp2_replan(e, nocc)        # (final cycle: the orbital energies seed the plan of the NEXT kernel())
            else:
                st.pop("mo_e", None)
            dm = Li.T @ dmo @ Li
        saved = (st["dmo"], st["diis"].count)
        st["dmo"] = dmo
        e_prev = st["e_tot"]
        ctx = self._after_density_launch(st, dm, st["cycle"] + 1, sp2_tr=tr_dev, hist_shape=hist_shape, projector=True)
        nxt = self._front(st) if (use_diis and not want_mo) else None
        if (self.sp2_trace_plan and self._sp2_plan is None and self._trace_bounds is not None and use_diis and not want_mo
                and st.get("gnorm", 1.0) <= self.sp2_trace_plan_gnorm):
            # cold object, SCF settling: the plan for the cycles to come is made HERE, from the traces of the last checked
            # purification, while the device is busy with this cycle's Fock build and the head of the next (1 ms of host time)
            self._plan_from_traces(st["nocc"])
        ok = self._after_density_finish(st, ctx, e_prev, nocc)
        if not ok:
            # the optimistic purification had not converged (planned path: the spectrum left the planned bounds): roll the DIIS
            # push back and redo this cycle -- planned path by diagonalisation (fresh bounds), otherwise by the checked SP2
            nxt = None
            self.n_redo = getattr(self, "n_redo", 0) + 1       # diagnostics (bench.py reports it: 0 in a settled loop)
            st["dmo"], st["diis"].count = saved
            st["e_tot"] = e_prev
            self._sp2_validated = False
            st["_redo"] = True
            try:
                dmo = None if self._sp2_planned_pass else self._density_sp2(fo, nocc, orth=True)
                if dmo is None:
                    e, c = torch.linalg.eigh(fo)
                    co = c[:, :nocc]
                    dmo = 2.0 * co @ co.T
                    if self._sp2_planned_pass:
                        self._sp2_replan(e, nocc)
                st["dmo"] = dmo
                self._after_density(st, Li.T @ dmo @ Li, e_last=e_prev, next_cycle=st["cycle"] + 1)
            finally:
                st.pop("_redo", None)
        if nxt is not None:
            st["front"] = nxt
        st["cycle"] += 1
        return st

    def kernel(self, dm0=None, **kw):
        t_start = time.time()
        st = self._start(dm0)
        eng = self.engine
        conv_tol = self.conv_tol
        conv_tol_grad = self.conv_tol_grad if self.conv_tol_grad is not None else np.sqrt(conv_tol)
        self._log(4, f"init E= {st['e_tot']:.15g}")
        self.converged = False
        t_loop = time.time()
        while st["cycle"] < self.max_cycle:
            self._step(st)
            self._log(4, f"cycle= {st['cycle']} E= {st['e_tot']:.15g}  delta_E= {st['de']:.3g}  |g|= {st['gnorm']:.3g}")
            if abs(st["de"]) < conv_tol and st["gnorm"] < conv_tol_grad:
                self.converged = True
                break
        st.pop("front", None)                            # the speculative head of a cycle that will not run
        self.cycles = st["cycle"]
        self.timing["loop_seconds"] = time.time() - t_loop
        t_final = time.time()
        if self.converged and self.conv_check:
            self._step(st, use_diis=False, want_mo=True)
            self._log(4, f"Extra cycle  E= {st['e_tot']:.15g}  delta_E= {st['de']:.3g}")
        if "mo_e" not in st:  # not converged (or max_cycle == 0): orbitals of the last Fock matrix
            e_, c_ = torch.linalg.eigh(st["fo"])
            st["mo_e"], st["mo_c"] = e_, self._Linv.T @ c_
            if self.eig_method == "sp2" and not self.level_shift:
                self._sp2_replan(e_, st["nocc"])
        self._dm, self._vhf = st["dm"], st["fock"] - self._h1
        self.e_tot = float(st["e_tot"])
        self.mo_energy = st["mo_e"].cpu().numpy()
        self.mo_coeff = st["mo_c"].cpu().numpy()
        occ = np.zeros(eng.nao)
        occ[:st["nocc"]] = 2.0
        self.mo_occ = occ
        self.timing["final_seconds"] = time.time() - t_final
        self.timing["total_seconds"] = time.time() - t_start
        if self.converged:
            self._log(3, f"converged SCF energy = {self.e_tot:.15g}")
        else:
            self._log(3, f"SCF not converged.\nSCF energy = {self.e_tot:.15g} after {self.max_cycle} cycles")
        return self.e_tot

    scf = kernel

    def energy_tot(self, dm=None, h1e=None, vhf=None):
        if dm is None:
            return self.e_tot
        d = torch.as_tensor(np.asarray(dm), dtype=torch.float64, device=self.engine.device)
        self._setup_once()
        _, e2 = self._veff(d)
        return float(torch.sum(d * self._h1) + e2) + self.mol.energy_nuc()

    def energy_elec(self, dm=None, h1e=None, vhf=None):
        e = self.energy_tot(dm) - self.mol.energy_nuc()
        return e, None

    # --- properties the templates read (calculate_energy.py:244-254) ----------------------------
    def dip_moment(self, mol=None, dm=None, unit="Debye", verbose=None, **kw):
        mol = mol or self.mol
        if dm is None:
            dm = self.make_rdm1()
        eng = self.engine
        dip = eng.int1e(with_dipole=True)[3]
        d = torch.as_tensor(np.asarray(dm), dtype=torch.float64, device=eng.device)
        el = -(dip * d.unsqueeze(0)).sum(dim=(1, 2)).cpu().numpy()
        nuc = (mol.atom_charges()[:, None] * mol.atom_coords()).sum(axis=0)
        out = el + nuc
        if str(unit).upper().startswith("DEBYE"):
            out = out * AU2DEBYE
            self._log(3, "Dipole moment(X, Y, Z, Debye): %8.5f, %8.5f, %8.5f" % tuple(out))
        else:
            self._log(3, "Dipole moment(X, Y, Z, A.U.): %8.5f, %8.5f, %8.5f" % tuple(out))
        return out

    def density_fit(self, auxbasis=None, with_df=None, only_dfj=False, **kw):
        """PySCF idiom `mf.density_fit()` (never called by the reference): J and K from a fitted
        three-index tensor (`df.DF`) instead of the resident four-centre tiles.  `auxbasis`: None -> generated even-tempered
        set, or a {element: shells} dict.  Returns `self` (PySCF returns a DF-decorated copy; the templates' idiom
        `mf = mf.density_fit()` works with both)."""
        from . import df
        self.with_df = with_df if with_df is not None else df.DF(self.mol, auxbasis)
        self._eng_df_ready = False
        return self

    def nuc_grad_method(self):
        from . import grad
        return grad.Gradients(self)

    Gradients = nuc_grad_method

    def as_scanner(self):
        return _Scanner(self)


class RHF(SCF):
    pass


class _Scanner:
    def __init__(self, mf):
        self.mf = mf

    def __call__(self, mol_or_geom):
        mf = self.mf
        mol = mol_or_geom if isinstance(mol_or_geom, Mole) else mf.mol.set_geom_(mol_or_geom, inplace=False)
        dm0 = mf.make_rdm1() if mf.mo_coeff is not None else None
        mf.reset(mol)
        return mf.kernel(dm0=dm0)


def _atomic_density(symbol, basis, z):
    """Spherically averaged fractional-occupation SCF for one neutral atom, on the GPU engine."""
    atom = Mole(atom=[(symbol, (0.0, 0.0, 0.0))], basis=basis, unit="Bohr", verbose=0)
    atom.spin = z % 2  # only to pass the electron-count check; occupations below are spin-restricted
    atom.build()
    eng = _engine.Engine(atom)
    S, T, V = eng.int1e()
    h1 = (T + V).cpu().numpy()
    S = S.cpu().numpy()
    n = atom.nao
    ls = atom._bas[:, 1]
    loc = atom.ao_loc_nr()
    cfg = _atom_config(z)
    idx_by_l = {l: [loc[s] for s in range(atom.nbas) if ls[s] == l] for l in set(ls.tolist())}

    def new_dm(F):
        D = np.zeros((n, n))
        for l, counts in cfg.items():
            if l not in idx_by_l:
                continue
            base = np.array(idx_by_l[l])
            e, c = _geneig(F[np.ix_(base, base)], S[np.ix_(base, base)])
            P = np.zeros((len(base), len(base)))
            for j, ne in enumerate(counts):
                if j >= len(base):
                    break
                P += (ne / (2 * l + 1.0)) * np.outer(c[:, j], c[:, j])
            for m in range(2 * l + 1):
                D[np.ix_(base + m, base + m)] = P
        return D

    D = new_dm(h1)
    e_last = 0.0
    for it in range(40):
        J, K = eng.get_jk(D)
        F = h1 + (J - 0.5 * K).cpu().numpy()
        e = float(np.sum(D * (h1 + F)) * 0.5)
        Dn = new_dm(F)
        D = 0.5 * D + 0.5 * Dn if it > 0 else Dn
        if abs(e - e_last) < 1e-7:
            break
        e_last = e
    eng.close()
    return D


def _geneig(f, s):
    L = np.linalg.cholesky(s)
    Li = np.linalg.inv(L)
    e, c = np.linalg.eigh(Li @ f @ Li.T)
    return e, Li.T @ c
