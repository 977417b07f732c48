"""Restricted Kohn-Sham on the MI355X engine: `pyscf.dft.RKS` / `gpu4pyscf.dft.RKS` (rows a7-a9, a12).

Reference call sites: `templates/calculate_energy.py:148-149,163-164,202-203` (`mf = RKS(mol); mf.xc = method`),
`templates/optimize_geometry.py:72-73` (xc assigned AFTER `.to_gpu()`).  B3LYP follows libxc's
HYB_GGA_XC_B3LYP (VWN-RPA) as PySCF >= 2.3 does [MEM].  XC quadrature: HIP kernels for AO values,
density, functional (forward-mode dual numbers) and weighted AOs; the two dense contractions per grid
block are rocBLAS DGEMMs (FP64 MFMA) through torch.matmul.
"""
import os

import numpy as np
import torch

from .grids import Grids, _grid_generation
from .scf import RHF

XC_IDS = {"slater": 1, "b88": 2, "vwn_rpa": 3, "vwn5": 4, "lyp": 5, "pbe_x": 6, "pbe_c": 7,
          "tpss_x": 8, "tpss_c": 9, "m062x_x": 10, "m062x_c": 11}
MGGA_KINDS = ("tpss_x", "tpss_c", "m062x_x", "m062x_c")


def parse_xc(name):
    """-> (hyb, [(coef, kind_id)], level): level 0 LDA, 1 GGA, 2 meta-GGA (truthy wherever AO gradients are needed)."""
    key = str(name).upper().replace("-", "").replace("_", "").replace(" ", "")
    table = {
        "HF": (1.0, []),
        "B3LYP": (0.2, [(0.08, "slater"), (0.72, "b88"), (0.19, "vwn_rpa"), (0.81, "lyp")]),
        "B3LYPG": (0.2, [(0.08, "slater"), (0.72, "b88"), (0.19, "vwn_rpa"), (0.81, "lyp")]),
        "B3LYP5": (0.2, [(0.08, "slater"), (0.72, "b88"), (0.19, "vwn5"), (0.81, "lyp")]),
        "PBE": (0.0, [(1.0, "pbe_x"), (1.0, "pbe_c")]),
        "PBE,PBE": (0.0, [(1.0, "pbe_x"), (1.0, "pbe_c")]),
        "PBE0": (0.25, [(0.75, "pbe_x"), (1.0, "pbe_c")]),
        "LDA": (0.0, [(1.0, "slater"), (1.0, "vwn5")]),
        "LDA,VWN": (0.0, [(1.0, "slater"), (1.0, "vwn5")]),
        "SVWN": (0.0, [(1.0, "slater"), (1.0, "vwn5")]),
        "LDA,VWNRPA": (0.0, [(1.0, "slater"), (1.0, "vwn_rpa")]),
        "BLYP": (0.0, [(1.0, "b88"), (1.0, "lyp")]),
        "B88,LYP": (0.0, [(1.0, "b88"), (1.0, "lyp")]),
        # meta-GGAs (templates/calculate_energy.py:263 `--method M06-2X`; templates/calculate_bde.py:105 default).  M06-2X: 54 %
        # exact exchange; parameter tables entered from memory (csrc, "unverified-memory")
        "TPSS": (0.0, [(1.0, "tpss_x"), (1.0, "tpss_c")]),
        "TPSS,TPSS": (0.0, [(1.0, "tpss_x"), (1.0, "tpss_c")]),
        "M062X": (0.54, [(1.0, "m062x_x"), (1.0, "m062x_c")]),
    }
    if key not in table:
        raise NotImplementedError(f"xc functional '{name}' is not implemented on the MI355X engine "
                                  f"(have {sorted(table)})")
    hyb, terms = table[key]
    gga = 2 if any(k in MGGA_KINDS for _c, k in terms) else int(any(k not in ("slater", "vwn_rpa", "vwn5") for _c, k in terms))
    return hyb, [(c, XC_IDS[k]) for c, k in terms], gga


class RKS(RHF):
    xc = "LDA,VWN"
    grid_block = 32768
    cache_ao = True
    direct_reserve_gb = 10.0  # direct mode: room kept for the quadrature (AO blocks, optional AO cache) beside the tile groups
    small_rho_cutoff = 1e-7   # PySCF RKS default [MEM]

    def __init__(self, mol, xc=None):
        super().__init__(mol)
        if xc is not None:
            self.xc = xc
        self.grids = Grids(mol)
        self._nelec_grid = None

    def reset(self, mol=None):
        super().reset(mol)
        old = self.grids
        self.grids = Grids(self.mol)
        self.grids.level, self.grids.prune = old.level, old.prune   # user settings survive scanner / optimize() resets
        self._pruned = False
        self._ao_cache_key = self._ao_cache = None                  # AO values of the old geometry: drop (and free) them
        return self

    def _setup(self):
        super()._setup()
        if self.grids.weights is None or self.grids.mol is not self.mol:
            self.grids.mol = self.mol
            self.grids.build(engine=self.engine)
            self._log(4, f"XC grid: {self.grids.size} points (level {self.grids.level})")

    def nr_rks(self, dm):
        """(N_elec, E_xc, V_xc, hyb) on device for a closed-shell density (numint.nr_rks [MEM]): this rank's share of the
        grid; sharded callers all-reduce."""
        n = self.engine.nao
        vmat = torch.zeros(n, n, dtype=torch.float64, device=self.engine.device)
        tail = torch.zeros(2, dtype=torch.float64, device=self.engine.device)
        hyb = self._nr_rks_raw(dm, vmat, tail)
        return tail[0], tail[1], vmat + vmat.T, hyb

    def _nr_rks_raw(self, dm, vmat, tail):
        """Accumulate the UNsymmetrised XC matrix (V_xc = vmat + vmat^T) into `vmat` and [N_elec, E_xc] into `tail` -- views of
        a caller-owned (zeroed) buffer, e.g. the fused all-reduce buffer of `_fock_energy`.  Returns the exact-exchange
        fraction of the functional."""
        eng = self.engine
        hyb, terms, gga = parse_xc(self.xc)
        n = eng.nao
        coords, weights = self.grids.coords, self.grids.weights
        ng = coords.shape[0]
        lo, hi = self._grid_range(ng)
        # grid block: as large as a ~1.5 GB working set allows (fewer launches for small molecules), at least grid_block
        B = max(self.grid_block, int(self._xc_block_bytes() / (48.0 * n)) // 1024 * 1024)
        if hi - lo <= 1.5 * B:
            B = max(hi - lo, 1)   # no small remainder block: its kernels would be pure launch latency (0.15 ms per build on benzene/cc-pVTZ)
        else:                     # equal blocks instead of full ones plus a short tail
            nblk = (hi - lo + B - 1) // B
            B = ((hi - lo + nblk - 1) // nblk + 1023) // 1024 * 1024
        cache = self._ao_cache_for(n, hi - lo, 4 if gga else 1)
        Zt = self._occ_factor(dm)
        if Zt is not None:   # [nao, ldz] with the orbital index fastest, zero-padded to the kernel's chunk (24 GGA / 32 LDA)
            ch = 24 if gga else 32
            ldz = (Zt.shape[0] + ch - 1) // ch * ch
            Zp = getattr(self, "_zp_buf", None)      # persistent: the padding columns are zeroed once, not every cycle
            if Zp is None or Zp.shape != (n, ldz) or Zp.device != Zt.device or self._zp_nocc != Zt.shape[0]:
                Zp = self._zp_buf = torch.zeros(n, ldz, dtype=torch.float64, device=Zt.device)
                self._zp_nocc = Zt.shape[0]
            Zp[:, :Zt.shape[0]].copy_(Zt.T)
        for ib, p0 in enumerate(range(lo, hi, B)):
            p1 = min(p0 + B, hi)
            c, w = coords[p0:p1], weights[p0:p1]
            if cache is not None and ib < len(cache):
                ao = cache[ib]                       # AO values stay resident in HBM across SCF cycles
            else:
                ao = eng.eval_ao(c, deriv=1 if gga else 0)
                if cache is not None:
                    cache.append(ao)
            if Zt is not None:
                # D = Z Z^T: densities from the occupied orbitals on the grid (nao / n_occ times fewer flops and bytes than D.ao)
                if gga == 2:
                    rho, tau = eng.xc_rho_lowrank(ao, Zp, deriv=1, with_tau=True)
                else:
                    rho, tau = eng.xc_rho_lowrank(ao, Zp, deriv=1 if gga else 0), None
            else:
                C = dm @ ao[0]
                rho = eng.xc_rho(ao, C, deriv=1 if gga else 0)
                tau = eng.xc_tau(ao, dm) if gga == 2 else None
            if gga == 2:
                e, wv = eng.xc_eval_mgga(terms, rho, tau, w)
            else:
                e, wv = eng.xc_eval(terms, rho, w, gga)
            eng.xc_tail(w, (rho[0], e), tail)     # tail[0] += w.rho (N_elec), tail[1] += w.e (E_xc): one deterministic launch
            if os.environ.get("MI355_VMAT_MT") and not getattr(self, "_vmat_mt_set", False):
                eng.set_option("vmat_fold_mt", float(os.environ["MI355_VMAT_MT"]))
                self._vmat_mt_set = True
            if self.xc_vmat_fold or os.environ.get("MI355_XC_FOLD", "0") == "1":
                eng.xc_vmat_fold(ao, wv, gga, vmat)    # vmat += ao0 . (sum_c wv_c ao_c)^T, weighted AOs formed inside the MFMA kernel
            else:
                aow = eng.xc_aow(ao, wv, gga)
                eng.xc_vmat(ao[0], aow, vmat)      # vmat += ao0 . aow^T  (split-K FP64 MFMA kernel)
            if gga == 2:                        # kinetic-energy-density term: sum_k ao_k . (w/4 vtau ao_k)^T
                for k in (1, 2, 3):
                    eng.xc_vmat(ao[k], wv[4] * ao[k], vmat)
        return hyb

    # Working set of one grid block (AO values + weighted AOs).  The per-point kernels are one thread per grid point: a block of
    # 55 k points (1.5 GB at N = 573, the round-1 size) is 216 workgroups for 256 CUs -- one wave per SIMD for a latency-bound
    # kernel.  4 GB (ibuprofen/def2-TZVP: 3 blocks of 108 k points instead of 7): RKS cycle 29.5 -> 28.0 ms; 12 GB: 28.2 ms
    # (tools/rks_cycle.py; env MI355_XC_BLOCK_GB overrides for such sweeps).
    xc_block_gb = 4.0

    def _xc_block_bytes(self):
        import os
        return float(os.environ.get("MI355_XC_BLOCK_GB", self.xc_block_gb)) * 1e9

    xc_vmat_fold = False  # V_xc product with the weighted AOs formed on the fly (round 3 experiment: 0.88-1.8 ms against 0.75 ms for the xc_aow pass + xc_vmat)
    xc_lowrank = True   # inside the SCF loop: rho from occupied-orbital values (D = Z Z^T) instead of D.ao
    xc_lowrank_min_nao = 128   # below this the dozen small launches of the factorisation cost more than the D.ao GEMM (CH3/cc-pVTZ UKS: 2.9 -> 3.4 ms)

    def _occ_factor(self, dm):
        """Z^T [n_occ, nao] with D = Z Z^T when the SCF step declared `dm` a closed-shell projector density (`_xc_projector`:
        D' = 2 X in the orthonormal basis, X idempotent of rank n_occ), else None.  No diagonalisation: W = X G for a fixed
        Gaussian G [nao, n_occ] spans the occupied space, M = G^T X G = W^T W, Cholesky M = R R^T, Z' = W R^-T has orthonormal
        columns (X = Z' Z'^T), Z = sqrt(2) L^-T Z'.  cond(M) is that of a square Gaussian matrix squared (~1e3-1e4), so the
        factorisation is accurate to ~1e-12 (PySCF's numint takes the same shortcut from mo_coeff / mo_occ [MEM: eval_rho2])."""
        proj = getattr(self, "_xc_projector", None)
        if not self.xc_lowrank or proj is None or proj[0] is not dm:
            return None
        _dm, dmo, nocc = proj
        return self._lowrank_factor(dmo, nocc, "rks")

    def _lowrank_factor(self, dmo, nocc, key):
        """Z^T [nocc, nao] with L^-T dmo L^-1 = Z Z^T for a positive semidefinite orthonormal-basis matrix `dmo` of rank nocc
        (2 X for RKS, the spin projectors X_s for UKS); `key` names the warm-start state (one per spin channel)."""
        n = dmo.shape[0]
        if not (0 < nocc < n // 2) or n < self.xc_lowrank_min_nao:
            return None
        st = self.__dict__.setdefault("_nystrom", {})
        G0 = st.get((key, "G0"))
        if G0 is None or G0.shape != (n, nocc) or G0.device != dmo.device:
            g = torch.Generator(device="cpu").manual_seed(20251004)
            G0 = st[(key, "G0")] = torch.randn(n, nocc, generator=g, dtype=torch.float64).to(dmo.device)
            st[(key, "G")] = G0
        # test matrix: the previous cycle's orthonormal occupied basis plus 5 % of the fixed Gaussian one.  Near convergence
        # X G ~ G, so M is close to a multiple of I and the factor is accurate to rounding (a pure Gaussian G gives cond(M) ~
        # 1e3-1e4, i.e. 1e-12 relative noise in rho -- visible as 1e-10 Ha jitter in E_xc of a 650 Ha molecule converged to
        # conv_tol = 1e-10); the Gaussian part keeps M non-singular when the occupied space has changed completely.
        G = st[(key, "G")]
        W = dmo @ G
        M = G.T @ W
        if nocc <= self.engine.NYSTROM_MAX_OCC and W.is_contiguous() and M.is_contiguous():
            Zp_t, _info = self.engine.nystrom_factor(M, W)    # Cholesky + triangular solve in one launch
        else:
            R, _info = torch.linalg.cholesky_ex(M)             # no host sync; a failed factorisation shows up as a wrong N_elec
            Zp_t = torch.linalg.solve_triangular(R, W.T, upper=False)   # R^-1 W^T: dmo = Zp Zp^T
            # `cholesky_ex` returns a finite, partially factored R when it fails: poison the factor so that the cycle's electron
            # count is NaN and the host sends the cycle through the full-density redo path (the fused kernel NaN-fills itself)
            Zp_t = Zp_t + torch.where(_info == 0, 0.0, float("nan")).to(Zp_t.dtype)
        # (a cycle whose projector was not valid -- speculative purification, checked later by the host -- must not poison
        # the warm start: keep the Gaussian matrix unless the factor is finite and the factorisation succeeded)
        # ONE launch (`nystrom_warm_kernel`) for what used to be ~14 elementwise / reduction launches per cycle:
        #   good = isfinite(Zp_t).all() & (info == 0); scale = 1 / |Zp_t[0]|; G <- good ? 0.05 G0 + scale Zp_t^T : G0
        if not Zp_t.is_contiguous():
            Zp_t = Zp_t.contiguous()
        bufs = st.get((key, "Gbuf"))
        if bufs is None or bufs[0].shape != G0.shape or bufs[0].device != G0.device:
            bufs = st[(key, "Gbuf")] = [torch.empty_like(G0), torch.empty_like(G0)]
        Gn = bufs[0] if G.data_ptr() != bufs[0].data_ptr() else bufs[1]     # never the matrix this cycle's W was formed from
        st[(key, "G")] = self.engine.nystrom_warm(Zp_t, _info, G0, Gn)
        return Zp_t @ self._Linv                            # (L^-T Zp)^T

    def _ao_cache_for(self, nao, npts, ncomp):
        """AO values on this rank's grid points are kept resident between SCF cycles when they fit in a quarter
        of the free HBM (benzene/cc-pVTZ 1.2 GB, ibuprofen/def2-TZVP 7.1 GB); invalidated with the grid."""
        key = (self.grids.generation, nao, npts, ncomp)   # generation: bumped whenever the point set changes
        if getattr(self, "_ao_cache_key", None) != key:
            self._ao_cache_key, self._ao_cache = key, None
            need = 8.0 * ncomp * nao * npts
            free, _total = torch.cuda.mem_get_info(self.engine.device)
            if self.cache_ao and need < 0.25 * free:
                self._ao_cache = []
        return self._ao_cache

    def _grid_range(self, ng):
        from . import parallel
        return parallel.split_range(ng, self._rank, self._nranks)

    def _prune_small_rho_grids(self, dm):
        """Drop grid points whose |rho * w| is below small_rho_cutoff / n_grid for the first density the SCF sees
        (PySCF `rks.prune_small_rho_grids_`, applied once in `get_veff` when the grid integrates the electron count to
        1 % [MEM]).  Evaluated on all points on every rank (one-off), so sharded runs prune identically."""
        self._pruned = True
        if not self.small_rho_cutoff or self.small_rho_cutoff <= 1e-20:
            return
        eng = self.engine
        coords, weights = self.grids.coords, self.grids.weights
        ng = coords.shape[0]
        rho = torch.empty(ng, dtype=torch.float64, device=eng.device)
        B = max(self.grid_block, int(1.5e9 / (16.0 * eng.nao)) // 1024 * 1024)
        for p0 in range(0, ng, B):
            p1 = min(p0 + B, ng)
            ao = eng.eval_ao(coords[p0:p1], deriv=0)
            rho[p0:p1] = eng.xc_rho(ao, dm @ ao[0], deriv=0)[0]
        n = float(torch.dot(rho, weights))
        if abs(n - self.mol.nelectron) < 0.01 * n:
            keep = (rho * weights).abs() > self.small_rho_cutoff / ng
            nkeep = int(keep.sum())
            self._log(4, f"Drop grids {ng - nkeep}")
            self.grids.coords = coords[keep].contiguous()
            self.grids.weights = weights[keep].contiguous()
            self.grids.generation = next(_grid_generation)
            if self.grids.atom_of is not None:
                self.grids.atom_of = self.grids.atom_of[keep.cpu().numpy()]

    def _fock_energy(self, dm, part):
        """Kohn-Sham Fock matrix with ONE collective per build: this rank's partial J, K (tile-run
        shard) and unsymmetrised V_xc, N_elec, E_xc (grid shard) live in one flat buffer [J | K | Vxc | N | Exc] that is
        all-reduced once."""
        if not getattr(self, "_pruned", False):
            self._prune_small_rho_grids(dm)
        eng = self.engine
        n = eng.nao
        nn = n * n
        hyb = parse_xc(self.xc)[0]
        with_k = abs(hyb) > 1e-12
        if self._fused_fock_ok(dm):
            # single rank, resident tiles: no collective, so J and K need not exist -- the fused epilogue of the J/K pass adds
            # h, the unsymmetrised XC matrix and its transpose, and writes the energy partials (mi_build_fock)
            buf = torch.zeros(nn + 2, dtype=torch.float64, device=eng.device)
            V, tail = buf[:nn].view(n, n), buf[nn:]
            self._nr_rks_raw(dm, V, tail)
            self._nelec_grid = tail[0]
            F = eng.build_fock(dm, self._h1, 0.5 * hyb, torch.empty_like(dm), part, with_k=with_k, vxc_unsym=V)
            return F, tail[0:2]
        nmat = 3 if with_k else 2
        buf = torch.zeros(nmat * nn + 2, dtype=torch.float64, device=eng.device)
        J = buf[:nn].view(n, n)
        K = buf[nn:2 * nn].view(n, n) if with_k else None
        V = buf[(nmat - 1) * nn:nmat * nn].view(n, n)
        tail = buf[nmat * nn:]
        self._nr_rks_raw(dm, V, tail)
        self._jk_into(dm, J, K)
        if self._nranks > 1:
            from . import parallel
            parallel.all_reduce_sum(buf, self._pg)
        vxc = V + V.T
        self._nelec_grid = tail[0]
        F = torch.empty_like(J)
        eng.fock_energy(self._h1, J, K, vxc, dm, 0.5 * hyb, F, part)
        return F, tail[0:2]      # [N_elec, E_xc]: the last entry is added to the energy, the first validates the quadrature

    def _xc_reduced(self, dm):
        nelec, exc, vxc, hyb = self.nr_rks(dm)
        if self._nranks > 1:
            from . import parallel
            nelec, exc = nelec.reshape(1), exc.reshape(1)
            parallel.all_reduce_fused([vxc, nelec, exc], self._pg)
            nelec, exc = nelec[0], exc[0]
        return nelec, exc, vxc, hyb

    def _veff(self, dm):
        nelec, exc, vxc, hyb = self.nr_rks(dm)
        if self._nranks > 1:
            from . import parallel
            nelec, exc = nelec.reshape(1), exc.reshape(1)
            parallel.all_reduce_fused([vxc, nelec, exc], self._pg)
            nelec, exc = nelec[0], exc[0]
        self._nelec_grid = nelec
        if abs(hyb) > 1e-12:
            J, K = self._jk(dm)
            v = J - (0.5 * hyb) * K + vxc
            e2 = 0.5 * torch.sum(dm * J) - (0.25 * hyb) * torch.sum(dm * K) + exc
        else:
            J, _ = self._jk(dm, with_k=False)
            v = J + vxc
            e2 = 0.5 * torch.sum(dm * J) + exc
        return v, e2
