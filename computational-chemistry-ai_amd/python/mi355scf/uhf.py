"""Spin-unrestricted SCF behind `pyscf.scf.UHF` / `gpu4pyscf.scf.UHF` and `dft.UKS` (SURVEY.md section 8f
rank 4; call sites `templates/calculate_bde.py:126-128,138-140,189-215`: radicals with `mol.spin = 1`).

Same engine as RHF: both spin densities go through the resident-tile J/K kernel in one call
(`mi_build_jk(n_dm = 2)`; J = J[Da] + J[Db], K_s = K[D_s]), sharded runs all-reduce the [J|K] buffer once.
F_s = h + J - c_x K_s (+ V_xc,s for UKS), E = 1/2 sum_s tr D_s (h + F_s^HF) (+ E_xc).  CDIIS acts on the pair
(F_a, F_b) with the concatenated error vectors F_s D_s S - S D_s F_s; the orbitals of each spin come from one
`eigh` of the Cholesky-orthogonalised Fock matrix per cycle (aufbau occupation, n_alpha >= n_beta).
"""
import time

import numpy as np
import torch

from .scf import SCF


class PairDIIS:
    """Pulay CDIIS on stacked [2, N, N] Fock/error matrices: ring-buffer history on the device, only the new Gram row
    (one fused multiply-reduce; a K = 2 N^2 GEMM with 8x8 output is the shape rocBLAS runs at < 1 TFLOP/s) and the
    (m+1)x(m+1) solve touch the host."""

    def __init__(self, space=8, sync=None):
        self.space, self.count = space, 0
        self.F = self.E = None
        self.B = np.zeros((space, space))
        # sharded runs: rank 0's Gram row is used on every rank.  Replicated FP64 work (eigh, reductions) differs in the last
        # bits between ranks; extrapolation coefficients computed per rank would feed that difference back and amplify it
        # (measured: x8 per cycle) until the ranks' densities disagree at 1e-7.
        self.sync = sync

    def update(self, f, e):
        if self.F is None:
            self.F = torch.empty((self.space,) + tuple(f.shape), dtype=f.dtype, device=f.device)
            self.E = torch.empty_like(self.F)
        slot = self.count % self.space
        self.F[slot].copy_(f)
        self.E[slot].copy_(e)
        self.count += 1
        m = min(self.count, self.space)
        dots = (self.E[:m].reshape(m, -1) * e.reshape(1, -1)).sum(dim=1)
        if self.sync is not None:
            self.sync(dots)
        dots = dots.cpu().numpy()
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
        cw = torch.as_tensor(c[1:], dtype=f.dtype, device=f.device).reshape(m, 1, 1, 1)
        return (cw * self.F[:m]).sum(dim=0)


class UHF(SCF):
    _spin_restricted = False
    sp2_min_nao = 200   # below this a per-cycle `eigh` of each spin is cheaper than the purification

    def __init__(self, mol):
        SCF.__init__(self, mol)
        self.nelec = mol.nelec

    # --- densities ------------------------------------------------------------------------------
    def make_rdm1(self, mo_coeff=None, mo_occ=None):
        if mo_coeff is None:
            if self._dm is not None:
                return self._dm.cpu().numpy()
            mo_coeff, mo_occ = self.mo_coeff, self.mo_occ
        out = []
        for c, o in zip(np.asarray(mo_coeff), np.asarray(mo_occ)):
            co = c[:, o > 0]
            out.append((co * o[o > 0]) @ co.T)
        return np.stack(out)

    def get_init_guess(self, mol=None, key=None):
        """Spin-restricted guess density split by electron count: D_s = D n_s / N (the first diagonalisation
        with n_alpha != n_beta occupations separates the spins)."""
        key = key if key is not None else self.init_guess
        if isinstance(key, np.ndarray) and key.ndim == 3:
            return key
        d = np.asarray(SCF.get_init_guess(self, mol, key))
        if d.ndim == 3:
            return d
        na, nb = self.nelec
        ne = max(na + nb, 1)
        return np.stack([d * (na / ne), d * (nb / ne)])

    # --- Fock pieces ----------------------------------------------------------------------------
    def _fock_pair(self, dm):
        """(F[2,N,N], E_elec) on device for the spin densities dm[2,N,N].  Overridden by UKS."""
        self.n_fock_builds = getattr(self, "n_fock_builds", 0) + 1
        J, K = self._jk(dm)
        Jt = J[0] + J[1]
        F = self._h1.unsqueeze(0) + Jt.unsqueeze(0) - K
        e = 0.5 * torch.sum(dm * (self._h1.unsqueeze(0) + F))
        return F, e

    def get_veff(self, mol=None, dm=None, **kw):
        if dm is None:
            dm = self.make_rdm1()
        self._setup_once()
        d = torch.as_tensor(np.asarray(dm), dtype=torch.float64, device=self.engine.device)
        F, _ = self._fock_pair(d)
        return (F - self._h1.unsqueeze(0)).cpu().numpy()

    def energy_tot(self, dm=None, h1e=None, vhf=None):
        if dm is None:
            return self.e_tot
        self._setup_once()
        d = torch.as_tensor(np.asarray(dm), dtype=torch.float64, device=self.engine.device)
        return float(self._fock_pair(d)[1]) + self.mol.energy_nuc()

    def spin_square(self, mo_coeff=None, s=None):
        """<S^2> and 2S+1 of the UHF determinant: S_z(S_z+1) + n_beta - sum_ij |<i_a|j_b>|^2."""
        mo = np.asarray(self.mo_coeff if mo_coeff is None else mo_coeff)
        occ = np.asarray(self.mo_occ)
        S = self.get_ovlp() if s is None else s
        ca, cb = mo[0][:, occ[0] > 0], mo[1][:, occ[1] > 0]
        na, nb = ca.shape[1], cb.shape[1]
        sab = ca.T @ S @ cb
        sz = 0.5 * (na - nb)
        ss = sz * (sz + 1.0) + nb - float(np.sum(sab * sab))
        return ss, float(np.sqrt(4.0 * ss + 1.0))

    # --- SCF loop -------------------------------------------------------------------------------
    fast_loop = True   # orthonormal-basis loop with device-side pair DIIS, planned purification per spin, pipelined step

    def kernel(self, dm0=None, **kw):
        """Two loops.  The fast one (orthonormal basis, device-side pair DIIS, pipelined step, planned purification per spin
        when plans exist) is the default below `sp2_min_nao` basis functions, where both loops diagonalise per cycle and it is
        5-13 % faster (CH3/cc-pVTZ UHF 2.92 -> 2.54 ms per cycle, UKS B3LYP 2.99 -> 2.83).  Above that size it did not pay in
        the measurements of round 2 (`tools/uhf_bench.py`, `tools/uhf_warm.py`, benzene cation / cc-pVTZ): from a cold object it
        needs diagonalisations for its purification plans (7 ms each), and restarted from a nearby density it ran 2.15 against
        3.1 ms per cycle but took 19 instead of 14 cycles (UKS: 8.6 against 7.2 ms per cycle), so the plain loop stays the default
        there; `fast_loop = "always"` selects the fast loop whenever plans exist (the plain loop's final orbitals seed them)."""
        self._setup_once()
        n = self.engine.nao
        na, nb = self.mol.nelec
        self.nelec = (na, nb)
        small = n < self.sp2_min_nao or self.eig_method != "sp2"
        warm = all(no == 0 or sp.vals["_sp2_plan"] is not None for sp, no in zip(self._spin_states(n), (na, nb)))
        use_fast = self.fast_loop and (small or (self.fast_loop == "always" and warm))
        if use_fast:
            return self._kernel_fast(dm0)
        return self._kernel_plain(dm0)

    # Plain loop (round 3): once a spin's checked, trace-correcting purification has run with |g| below `sp2_trace_plan_gnorm`, its
    # trace history gives the bounds of a purification PLAN for that spin (sp2plan.bounds_from_traces, as in the closed-shell
    # cold path) and the following cycles run the ~21 planned passes instead of the 40-60 of the recursion; the result is checked
    # on the host like the recursion's (this loop synchronises per spin anyway) and a failed check drops the plan.
    def _planned_spin_density(self, fo, no, state, gnorm):
        plan = state.get("plan")
        n = fo.shape[0]
        if plan is None or not (self.sp2_planned and self.sp2_trace_plan and n <= self.sp2_fused_max and self.sp2_fused):
            return None
        keep = (self._sp2_plan, getattr(self, "_sp2_plan_len", 0))
        try:
            self._sp2_plan, self._sp2_plan_len = plan["coef"], plan["len"]
            res, tr_dev = self._sp2_planned_async(fo, no, scale=2.0)
            shape, self._sp2_hist_shape = self._sp2_hist_shape, None
        finally:
            self._sp2_plan, self._sp2_plan_len = keep
        hist = tr_dev.cpu().numpy()
        if shape is None or hist.size != shape[0] * 64:
            state["plan"] = None
            return None
        h = hist.reshape(shape[0], 32, 2)[:, :shape[1], :]
        tx, tx2 = h[:, :, 0].sum(axis=1), h[:, :, 1].sum(axis=1)
        ok = (np.abs(tx - tx2) < self.sp2_tol) & (np.abs(tx - no) < 1e-8)
        if not ok[-1]:
            state["plan"] = None            # the spectrum left the planned window: checked recursion, new plan from its traces
            return None
        plan["len"] = min(plan["coef"].shape[0] - 1, max(int(np.argmax(ok)) + 1, 4))
        self.path_counts = getattr(self, "path_counts", {})
        self.path_counts["planned_spin"] = self.path_counts.get("planned_spin", 0) + 1
        return res.clone()                  # (a view of the ping-pong buffers the other spin is about to reuse)

    def _plan_spin_from_traces(self, state, gnorm):
        tb, self._trace_bounds = self._trace_bounds, None
        if tb is None or not (self.sp2_planned and self.sp2_trace_plan) or gnorm > self.sp2_trace_plan_gnorm:
            return
        from . import sp2plan
        b = sp2plan.bounds_from_traces(tb[0], tb[1], tb[2], tb[3], self.sp2_inner_margin)
        coef = sp2plan.plan(*b) if b is not None else None
        if coef is not None:
            state["plan"] = dict(coef=coef, len=coef.shape[0] - 1)

    def _seed_plans(self, mo_e):
        """Purification plans of both spins from the orbital energies of a finished SCF (for the next kernel() of this object)."""
        n = self.engine.nao
        if self.eig_method != "sp2" or n < self.sp2_min_nao or not self.sp2_planned or self.level_shift:
            return
        for s_, sp in enumerate(self._spin_states(n)):
            if 0 < self.nelec[s_] < n:
                self._with_spin(sp, self._sp2_replan, mo_e[s_], self.nelec[s_])

    scf = kernel

    class _Spin:
        """Purification state of one spin channel (plan, trimmed length, work buffers): swapped into the SCF object around the
        calls of the shared `_sp2_*` helpers."""
        KEYS = ("_sp2_plan", "_sp2_plan_len", "_sp2p", "_sp2g", "_sp2_hist_shape")

        def __init__(self):
            self.vals = {k: None for k in self.KEYS}
            self.vals["_sp2_plan_len"] = 0

    def _spin_states(self, n):
        """The purification plans live on the object: a geometry optimisation (same `mf`, `kernel(dm0=...)` per step) reuses the
        plans of the previous geometry and diagonalises nothing after its first SCF."""
        key = (n, tuple(self.nelec))
        if getattr(self, "_spin_key", None) != key:
            self._spin_key, self._spin_pair = key, [UHF._Spin(), UHF._Spin()]
        return self._spin_pair

    def _with_spin(self, sp, fn, *a, **kw):
        saved = {k: getattr(self, k, None) for k in sp.KEYS}
        for k in sp.KEYS:
            setattr(self, k, sp.vals[k])
        try:
            return fn(*a, **kw)
        finally:
            for k in sp.KEYS:
                sp.vals[k] = getattr(self, k, None)
                setattr(self, k, saved[k])

    def _projector(self, st, s_, fo, nocc_s, allow_plan=True):
        """(X_s, traces or None, hist_shape, planned?) for one spin: planned purification when a plan exists, else `eigh`
        (which also yields the bounds for a plan).  X_s is the occupied projector (occupation 1) in the orthonormal basis."""
        n = fo.shape[0]
        sp = st["spin"][s_]
        if nocc_s == 0:
            return torch.zeros_like(fo), None, None, False
        can_plan = (self.eig_method == "sp2" and self.sp2_planned and n >= self.sp2_min_nao and self._sp2_plannable(n)
                    and 0 < nocc_s < n and not self.level_shift)
        settled = st.get("gnorm") is not None and st["gnorm"] <= self.sp2_plan_gnorm
        if can_plan and allow_plan and sp.vals["_sp2_plan"] is not None and not settled:
            # plans exist but this SCF is still far from its solution (see SCF._step): checked purification, no redo cycles
            d2 = self._density_sp2(fo, nocc_s, orth=True)
            if d2 is not None:
                return 0.5 * d2, None, None, False
        if can_plan and allow_plan and sp.vals["_sp2_plan"] is not None and settled:
            x, tr = self._with_spin(sp, self._sp2_planned_async, fo, nocc_s, 1.0)
            shape, sp.vals["_sp2_hist_shape"] = sp.vals["_sp2_hist_shape"], None
            return x, tr, shape, True
        self.n_eigh = getattr(self, "n_eigh", 0) + 1
        e, c = torch.linalg.eigh(fo)
        co = c[:, :nocc_s]
        if can_plan:
            self._with_spin(sp, self._sp2_replan, e, nocc_s)
        return co @ co.T, None, None, False

    def _ulaunch(self, st, dm, dmo, next_cycle, trs):
        """Device part of a cycle after the densities: Fock pair, F'_s = L^-1 F_s L^-T into the DIIS slot, error vectors
        [F'_s, X_s] (orthonormal-basis CDIIS error of PySCF >= 2.1, spin-stacked) and their norms, Pulay solve, asynchronous
        read-back of the scalars."""
        eng, Li = self.engine, self._Linv
        dm = dm.contiguous()
        # from cycle 1 on the spin densities are projectors X_s of rank n_s: lets UKS take rho_s from a low-rank factor
        self._xc_projector_pair = (dm, dmo, st["nocc"]) if next_cycle > 0 else None
        F, e_el = self._fock_pair(dm)
        diis = st["diis"]
        keep = next_cycle >= self.diis_start_cycle
        slot = diis.next_slot()
        nb = eng.reduce_blocks
        part = torch.empty(2 * nb, dtype=torch.float64, device=dm.device)
        fo = diis.F[slot] if keep else torch.empty_like(diis.F[0])
        eo = diis.E[slot] if keep else torch.empty_like(diis.E[0])
        for s_ in range(2):
            torch.matmul(Li @ F[s_], Li.T, out=fo[s_])
            eng.commutator_norm(fo[s_] @ dmo[s_], eo[s_], part[s_ * nb:(s_ + 1) * nb])
        if keep:
            diis.push_inplace()
        parts = [e_el.reshape(1), part] + [t for t in trs if t is not None]
        packed = torch.cat(parts)
        if self._sync_control_on():     # sharded: rank 0's control scalars are everybody's (see scf.SCF.sync_control)
            from . import parallel
            parallel.broadcast0(packed, self._pg)
        ctx = dict(dm=dm, dmo=dmo, F=F, fo=fo, nb=nb, packed=packed, tr_sizes=[0 if t is None else t.numel() for t in trs], event=None)
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

    def _ufinish(self, st, ctx, shapes, planned, e_last):
        """Host part: wait for the scalars, validate the purifications, update the state.  False: a purification was not
        converged (nothing in `st` touched)."""
        if ctx["event"] is not None:
            ctx["event"].synchronize()
            vals = self._pin[:ctx["packed"].numel()].numpy().copy()
        else:
            vals = ctx["packed"].cpu().numpy()
        nb = ctx["nb"]
        e_el = float(vals[0])
        c2 = float(vals[1:1 + 2 * nb].sum())
        pos = 1 + 2 * nb
        for s_ in range(2):
            k = ctx["tr_sizes"][s_]
            if not k:
                continue
            hist = vals[pos:pos + k]
            pos += k
            nocc_s = st["nocc"][s_]
            shape = shapes[s_]
            if shape is not None and hist.size == shape[0] * 64:
                h = hist.reshape(shape[0], 32, 2)[:, :shape[1], :]
                tx, tx2 = h[:, :, 0].sum(axis=1), h[:, :, 1].sum(axis=1)
                ok = (np.abs(tx - tx2) < self.sp2_tol) & (np.abs(tx - nocc_s) < 1e-8)
                if not ok[-1]:
                    return False
                sp = st["spin"][s_]
                sp.vals["_sp2_plan_len"] = min(sp.vals["_sp2_plan"].shape[0] - 1, max(int(np.argmax(ok)) + 1, 4))
            else:
                trx, trx2 = self._sp2_traces(hist)
                if not (abs(trx - trx2) < self.sp2_tol and abs(trx - nocc_s) < 1e-8):
                    return False
        e_tot = e_el + st["enuc"]
        gnorm = float(np.sqrt(max(c2, 0.0) / 2.0)) / np.sqrt(st["nvo"])
        st.update(dm=ctx["dm"], dmo=ctx["dmo"], F=ctx["F"], fo=ctx["fo"], e_tot=e_tot, gnorm=gnorm,
                  de=(e_tot - e_last) if e_last is not None else 0.0)
        return True

    def _ufront(self, st):
        """Device-only head of the next cycle (pair-extrapolated F', planned purification of both spins, AO densities), queued
        before the host waits for the current cycle's scalars.  None when either spin has no plan."""
        if not (self.pipeline and st["cycle"] + 1 >= self.diis_start_cycle and st["diis"].count > 0 and not self.level_shift):
            return None
        if st.get("gnorm") is None or st["gnorm"] > self.sp2_plan_gnorm:
            return None
        for s_ in range(2):
            if st["nocc"][s_] > 0 and st["spin"][s_].vals["_sp2_plan"] is None:
                return None
        n = self._Linv.shape[0]
        if not (self.eig_method == "sp2" and self.sp2_planned and n >= self.sp2_min_nao and self._sp2_plannable(n)):
            return None
        fo = st["diis"].extrapolate()
        return self._udensities(st, fo)

    def _udensities(self, st, fo, allow_plan=True):
        Li = self._Linv
        xs, trs, shapes, planned = [], [], [], []
        for s_ in range(2):
            x, tr, shape, pl = self._projector(st, s_, fo[s_], st["nocc"][s_], allow_plan)
            xs.append(x); trs.append(tr); shapes.append(shape); planned.append(pl)
        dmo = torch.stack(xs)
        dm = torch.stack([Li.T @ xs[0] @ Li, Li.T @ xs[1] @ Li])
        return dict(fo=fo, dmo=dmo, dm=dm, trs=trs, shapes=shapes, planned=planned)

    def _ustep(self, st):
        front = st.pop("front", None)
        if front is None:
            if st["cycle"] >= self.diis_start_cycle and st["diis"].count > 0:
                fo = st["diis"].extrapolate()
            else:
                fo = st["fo"]
            if self.level_shift:   # F'_s + shift (1 - X_s): virtual space of each spin raised
                eye = torch.eye(fo.shape[-1], dtype=fo.dtype, device=fo.device)
                fo = fo + self.level_shift * (eye.unsqueeze(0) - st["dmo"])
            front = self._udensities(st, fo)
        saved_count = st["diis"].count
        e_prev = st["e_tot"]
        ctx = self._ulaunch(st, front["dm"], front["dmo"], st["cycle"] + 1, front["trs"])
        nxt = self._ufront(st)
        ok = self._ufinish(st, ctx, front["shapes"], front["planned"], e_prev)
        if not ok:
            # a planned purification had not converged (the spectrum left the planned bounds): roll the DIIS push back and
            # redo the cycle by diagonalisation, which also refreshes the plans
            nxt = None
            self.n_redo = getattr(self, "n_redo", 0) + 1
            st["diis"].count = saved_count
            redo = self._udensities(st, front["fo"], allow_plan=False)
            ctx = self._ulaunch(st, redo["dm"], redo["dmo"], st["cycle"] + 1, redo["trs"])
            self._ufinish(st, ctx, redo["shapes"], redo["planned"], e_prev)
        if nxt is not None:
            st["front"] = nxt
        st["cycle"] += 1
        return st

    def _kernel_fast(self, dm0=None):
        t_start = time.time()
        mol = self.mol
        self._setup_once()
        eng = self.engine
        L, Li = self._L, self._Linv
        na, nb = mol.nelec
        self.nelec = (na, nb)
        n = eng.nao
        if dm0 is None:
            dm0 = self.get_init_guess()
        dm0 = np.asarray(dm0)
        if dm0.ndim == 2:
            ne = max(na + nb, 1)
            dm0 = np.stack([dm0 * (na / ne), dm0 * (nb / ne)])
        dm = torch.as_tensor(dm0, dtype=torch.float64, device=eng.device).contiguous()
        if self._nranks > 1:   # one-off: identical starting density on every rank (see SCF._start)
            from . import parallel
            parallel.broadcast0(dm, self._pg)
        from .scf import DeviceDIIS
        st = {"nocc": (na, nb), "enuc": mol.energy_nuc(), "cycle": 0, "diis": DeviceDIIS(eng, self.diis_space, nmat=2),
              "spin": self._spin_states(n), "nvo": max(na * (n - na) + nb * (n - nb), 1), "e_tot": None}
        dmo0 = torch.stack([L.T @ dm[0] @ L, L.T @ dm[1] @ L])
        ctx = self._ulaunch(st, dm, dmo0, 0, [None, None])
        self._ufinish(st, ctx, [None, None], [False, False], None)
        conv_tol = self.conv_tol
        conv_tol_grad = self.conv_tol_grad if self.conv_tol_grad is not None else np.sqrt(conv_tol)
        self._log(4, f"init E= {st['e_tot']:.15g}")
        self.converged = False
        t_loop = time.time()
        while st["cycle"] < self.max_cycle:
            self._ustep(st)
            self._log(4, f"cycle= {st['cycle']} E= {st['e_tot']:.15g}  delta_E= {st['de']:.3g}  |g|= {st['gnorm']:.3g}")
            if abs(st["de"]) < conv_tol and st["gnorm"] < conv_tol_grad:
                self.converged = True
                break
        st.pop("front", None)
        self.cycles = st["cycle"]
        self.timing["loop_seconds"] = time.time() - t_loop

        def orbitals(Fx):
            es, cs = [], []
            for s_ in range(2):
                e_, c_ = torch.linalg.eigh(Li @ Fx[s_] @ Li.T)
                es.append(e_); cs.append(Li.T @ c_)
            return torch.stack(es), torch.stack(cs)

        F, dm, e_tot = st["F"], st["dm"], st["e_tot"]
        mo_e, mo_c = orbitals(F)
        if self.converged and self.conv_check:
            ca, cb = mo_c[0][:, :na], mo_c[1][:, :nb]
            dm = torch.stack([ca @ ca.T, cb @ cb.T])
            F, e_el = self._fock_pair(dm)
            e_new = float(e_el) + st["enuc"]
            self._log(4, f"Extra cycle  E= {e_new:.15g}  delta_E= {e_new - e_tot:.3g}")
            e_tot = e_new
        self._dm, self._fock = dm, F
        self.e_tot = e_tot
        self.mo_energy = mo_e.cpu().numpy()
        self._seed_plans(mo_e)
        self.mo_coeff = mo_c.cpu().numpy()
        occ = np.zeros((2, n))
        occ[0, :na] = 1.0
        occ[1, :nb] = 1.0
        self.mo_occ = occ
        self.timing["total_seconds"] = time.time() - t_start
        if self.converged:
            ss, mult = self.spin_square()
            self._log(3, f"converged SCF energy = {self.e_tot:.15g}  <S^2> = {ss:.8g}  2S+1 = {mult:.8g}")
        else:
            self._log(3, f"SCF not converged.\nSCF energy = {self.e_tot:.15g} after {self.max_cycle} cycles")
        return self.e_tot

    def _kernel_plain(self, dm0=None, **kw):
        t_start = time.time()
        mol = self.mol
        self._setup_once()
        eng = self.engine
        S, L, Li = self._S, self._L, self._Linv
        na, nb = mol.nelec
        self.nelec = (na, nb)
        n = eng.nao
        if dm0 is None:
            dm0 = self.get_init_guess()
        dm0 = np.asarray(dm0)
        if dm0.ndim == 2:
            ne = max(na + nb, 1)
            dm0 = np.stack([dm0 * (na / ne), dm0 * (nb / ne)])
        dm = torch.as_tensor(dm0, dtype=torch.float64, device=eng.device).contiguous()
        if self._nranks > 1:   # one-off: identical starting density on every rank (see SCF._start)
            from . import parallel
            parallel.broadcast0(dm, self._pg)
        enuc = mol.energy_nuc()
        conv_tol = self.conv_tol
        conv_tol_grad = self.conv_tol_grad if self.conv_tol_grad is not None else np.sqrt(conv_tol)
        # sharded runs: the all-reduced J/K(/Vxc) are identical on every rank and the replicated algebra below is made of
        # deterministic library reductions, so the ranks stay bit-identical without broadcasting control scalars;
        # `sync_control` (auto-on for sharded runs, see scf.SCF) makes rank 0's Gram row / scalars authoritative
        sync = None
        if self._sync_control_on():
            from . import parallel
            sync = lambda t: parallel.broadcast0(t, self._pg)
        diis = PairDIIS(self.diis_space, sync)
        F, e_el = self._fock_pair(dm)
        e_tot = float(e_el) + enuc
        self._log(4, f"init E= {e_tot:.15g}")
        self.converged = False
        cycle = 0
        t_loop = time.time()
        mo_e = mo_c = None

        def orbitals(Fx):
            es, cs = [], []
            for s_ in range(2):
                e_, c_ = torch.linalg.eigh(Li @ Fx[s_] @ Li.T)
                es.append(e_); cs.append(Li.T @ c_)
            return torch.stack(es), torch.stack(cs)

        def density(cs):
            ca, cb = cs[0][:, :na], cs[1][:, :nb]
            return torch.stack([ca @ ca.T, cb @ cb.T])

        # larger matrices: occupied projector of each spin by SP2 purification (same GEMM-only path as RHF, `scf.py`)
        # instead of a diagonalisation per cycle; orbitals are then only needed once, after convergence
        use_sp2 = self.eig_method == "sp2" and n >= self.sp2_min_nao
        sp2_state = [dict(_sp2_iters=self._sp2_iters, _sp2_validated=False) for _ in range(2)]

        def new_density(Fx):
            if not use_sp2:
                e_, c_ = orbitals(Fx)
                return density(c_), (e_, c_)
            out = []
            xs = []        # spin projectors X_s in the orthonormal basis (UKS: rho_s from a low-rank factor, `_xc_projector_pair`)
            for s_, no in ((0, na), (1, nb)):
                if no == 0:
                    out.append(torch.zeros(n, n, dtype=torch.float64, device=eng.device))
                    xs.append(out[-1])
                    continue
                fo = (Li @ Fx[s_] @ Li.T).contiguous()
                self._sp2_iters, self._sp2_validated = sp2_state[s_]["_sp2_iters"], sp2_state[s_]["_sp2_validated"]
                g_prev = gnorm if cycle > 0 else 1.0e9          # (no orbital gradient before the first cycle has finished)
                dmo = self._planned_spin_density(fo, no, sp2_state[s_], g_prev)
                if dmo is None:
                    self._trace_bounds = None
                    dmo = self._density_sp2(fo, no, orth=True)
                    self._plan_spin_from_traces(sp2_state[s_], g_prev)
                sp2_state[s_].update(_sp2_iters=self._sp2_iters, _sp2_validated=self._sp2_validated)
                if dmo is None:      # purification did not converge (vanishing gap): diagonalise this spin
                    e_, c_ = torch.linalg.eigh(fo)
                    co = Li.T @ c_[:, :no]
                    out.append(co @ co.T)
                    xs = None
                else:
                    out.append(0.5 * (Li.T @ dmo @ Li))
                    if xs is not None:
                        xs.append(0.5 * dmo)
            self._plain_projectors = xs
            return torch.stack(out), None

        def commutator(Fx, dmx):
            return torch.stack([Fx[s_] @ dmx[s_] @ S - S @ dmx[s_] @ Fx[s_] for s_ in range(2)])

        nvo = max(na * (n - na) + nb * (n - nb), 1)
        de = gnorm = 0.0
        err = commutator(F, dm)
        while cycle < self.max_cycle:
            Fx = diis.update(F, err) if cycle + 1 >= self.diis_start_cycle else F
            if self.level_shift:   # F_s + shift (S - S D_s S): virtual space of each spin raised (AO form of PySCF's level_shift)
                Fx = Fx + self.level_shift * (S.unsqueeze(0) - torch.stack([S @ dm[s_] @ S for s_ in range(2)]))
            self._plain_projectors = None
            dm, mo = new_density(Fx)
            if mo is not None:
                mo_e, mo_c = mo
            # purified spin densities are projectors of known rank: UKS takes rho_s from their low-rank factors (one pass over
            # the AO values instead of an N x N x n_grid product per spin), exactly as in the fast loop
            xs = self._plain_projectors
            self._xc_projector_pair = (dm, xs, (na, nb)) if xs is not None else None
            F, e_el = self._fock_pair(dm)
            if self._xc_projector_pair is not None and not bool(torch.isfinite(e_el).all()):
                self._xc_projector_pair = None          # a failed factorisation poisons itself with NaN: full densities instead
                F, e_el = self._fock_pair(dm)
            self._xc_projector_pair = None
            err = commutator(F, dm)
            # |g| = |F_vo| of both spins = |[F', D']|_F / sqrt(2) in the orthonormal basis (D' is a projector)
            eo = torch.stack([Li @ err[s_] @ Li.T for s_ in range(2)])
            ctrl = torch.stack([e_el.reshape(()), torch.sum(eo * eo)])
            if sync is not None:
                sync(ctrl)
            ctrl = ctrl.cpu().numpy()
            e_new = float(ctrl[0]) + enuc
            gnorm = float(np.sqrt(max(ctrl[1], 0.0) / 2.0)) / np.sqrt(nvo)
            de = e_new - e_tot
            e_tot = e_new
            cycle += 1
            self._log(4, f"cycle= {cycle} E= {e_tot:.15g}  delta_E= {de:.3g}  |g|= {gnorm:.3g}")
            if abs(de) < conv_tol and gnorm < conv_tol_grad:
                self.converged = True
                break
        self.cycles = cycle
        self.timing["loop_seconds"] = time.time() - t_loop
        if self.converged and self.conv_check:
            mo_e, mo_c = orbitals(F)
            dm = density(mo_c)
            F, e_el = self._fock_pair(dm)
            e_el = e_el.reshape(1)
            if sync is not None:
                sync(e_el)
            e_new = float(e_el) + enuc
            self._log(4, f"Extra cycle  E= {e_new:.15g}  delta_E= {e_new - e_tot:.3g}")
            e_tot = e_new
        if mo_e is None or (use_sp2 and not (self.converged and self.conv_check)):
            mo_e, mo_c = orbitals(F)
        self._dm, self._fock = dm, F
        self.e_tot = e_tot
        self.mo_energy = mo_e.cpu().numpy()
        self._seed_plans(mo_e)
        self.mo_coeff = mo_c.cpu().numpy()
        occ = np.zeros((2, n))
        occ[0, :na] = 1.0
        occ[1, :nb] = 1.0
        self.mo_occ = occ
        self.timing["total_seconds"] = time.time() - t_start
        if self.converged:
            ss, mult = self.spin_square()
            self._log(3, f"converged SCF energy = {self.e_tot:.15g}  <S^2> = {ss:.8g}  2S+1 = {mult:.8g}")
        else:
            self._log(3, f"SCF not converged.\nSCF energy = {self.e_tot:.15g} after {self.max_cycle} cycles")
        return self.e_tot

    def dip_moment(self, mol=None, dm=None, unit="Debye", verbose=None, **kw):
        if dm is None:
            dm = self.make_rdm1()
        dm = np.asarray(dm)
        if dm.ndim == 3:
            dm = dm[0] + dm[1]
        return SCF.dip_moment(self, mol, dm, unit, verbose, **kw)

    def nuc_grad_method(self):
        from . import grad
        return grad.UGradients(self)

    Gradients = nuc_grad_method
