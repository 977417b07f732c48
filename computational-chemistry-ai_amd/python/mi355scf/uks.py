"""Spin-unrestricted Kohn-Sham behind `pyscf.dft.UKS` / `gpu4pyscf.dft.UKS` (call sites
`templates/calculate_bde.py:128,140,197,215`: `mf = UKS(mol); mf.xc = method` for the radical fragments).

Same grid, AO, density and V_xc HIP kernels as RKS; the functional is evaluated by `mi_xc_eval_spin` (forward-mode
dual numbers over rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb: spin-scaled exchange, VWN/PW92 spin interpolation,
open-shell LYP, PBE correlation with phi(zeta)).  Meta-GGAs (TPSS; the template's default M06-2X, parameter tables unverified-memory) go through `mi_xc_eval_mgga_spin` (dual numbers incl. tau_a, tau_b).
The grid is not pruned by density (PySCF's `small_rho_cutoff` step is RKS-only here).
"""
import numpy as np
import torch

from .dft import RKS, parse_xc
from .grids import Grids
from .uhf import UHF


class UKS(UHF):
    xc = "LDA,VWN"
    grid_block = RKS.grid_block
    cache_ao = True
    direct_reserve_gb = RKS.direct_reserve_gb

    def __init__(self, mol, xc=None):
        UHF.__init__(self, mol)
        if xc is not None:
            self.xc = xc
        self.grids = Grids(mol)
        self._nelec_grid = None

    def reset(self, mol=None):
        UHF.reset(self, mol)
        old = self.grids
        self.grids = Grids(self.mol)
        self.grids.level, self.grids.prune = old.level, old.prune
        self._ao_cache_key = self._ao_cache = None
        return self

    def _setup(self):
        UHF._setup(self)
        if self.grids.weights is None or self.grids.mol is not self.mol:
            self.grids.mol = self.mol
            self.grids.build(engine=self.engine)
            self._log(4, f"XC grid: {self.grids.size} points (level {self.grids.level})")

    _ao_cache_for = RKS._ao_cache_for
    xc_block_gb = RKS.xc_block_gb
    _xc_block_bytes = RKS._xc_block_bytes
    _lowrank_factor = RKS._lowrank_factor
    xc_lowrank, xc_lowrank_min_nao = True, RKS.xc_lowrank_min_nao
    _grid_range = RKS._grid_range

    def nr_uks(self, dm):
        """((N_alpha, N_beta), E_xc, V_xc[2,N,N], hyb) on device for the spin densities dm[2,N,N] (numint.nr_uks [MEM]): this
        rank's share of the grid; sharded callers all-reduce."""
        n = self.engine.nao
        vmat = torch.zeros(2, n, n, dtype=torch.float64, device=self.engine.device)
        tail = torch.zeros(3, dtype=torch.float64, device=self.engine.device)
        hyb = self._nr_uks_raw(dm, vmat, tail)
        return tail[:2], tail[2], vmat + vmat.transpose(1, 2), hyb

    def _nr_uks_raw(self, dm, vmat, tail):
        """Unsymmetrised V_xc,s into `vmat[2,N,N]`, [N_alpha, N_beta, E_xc] into `tail` (views of a zeroed caller buffer)."""
        eng = self.engine
        hyb, terms, gga = parse_xc(self.xc)
        n = eng.nao
        coords, weights = self.grids.coords, self.grids.weights
        ng = coords.shape[0]
        lo, hi = self._grid_range(ng)
        B = max(self.grid_block, int(self._xc_block_bytes() / (64.0 * n)) // 1024 * 1024)
        cache = self._ao_cache_for(n, hi - lo, 4 if gga else 1)
        # spin densities declared projectors by the fast UHF/UKS loop (`_xc_projector_pair`): D_s = Z_s Z_s^T without orbitals
        Zps = [None, None]
        proj = getattr(self, "_xc_projector_pair", None)
        if self.xc_lowrank and proj is not None and proj[0] is dm:
            ch = 24 if gga else 32
            for s_ in range(2):
                Zt = self._lowrank_factor(proj[1][s_], proj[2][s_], ("uks", s_)) if proj[2][s_] > 0 else None
                if Zt is not None:
                    Zp = torch.zeros(n, (Zt.shape[0] + ch - 1) // ch * ch, dtype=torch.float64, device=Zt.device)
                    Zp[:, :Zt.shape[0]] = Zt.T
                    Zps[s_] = Zp
        for ib, p0 in enumerate(range(lo, hi, B)):
            p1 = min(p0 + B, hi)
            c, w = coords[p0:p1], weights[p0:p1]
            if cache is not None and ib < len(cache):
                ao = cache[ib]
            else:
                ao = eng.eval_ao(c, deriv=1 if gga else 0)
                if cache is not None:
                    cache.append(ao)
            rho, tau = [None, None], [None, None]
            for s_ in range(2):
                if Zps[s_] is not None:      # spin density from its low-rank factor (dft.RKS._lowrank_factor), one pass over ao
                    if gga == 2:
                        rho[s_], tau[s_] = eng.xc_rho_lowrank(ao, Zps[s_], deriv=1, with_tau=True)
                    else:
                        rho[s_] = eng.xc_rho_lowrank(ao, Zps[s_], deriv=1 if gga else 0)
                else:
                    rho[s_] = eng.xc_rho(ao, dm[s_] @ ao[0], deriv=1 if gga else 0)
                    if gga == 2:
                        tau[s_] = eng.xc_tau(ao, dm[s_])
            if gga == 2:
                e, wva, wvb = eng.xc_eval_mgga_spin(terms, rho[0], rho[1], tau[0], tau[1], w)
            else:
                e, wva, wvb = eng.xc_eval_spin(terms, rho[0], rho[1], w, gga)
            eng.xc_tail(w, (rho[0][0], rho[1][0], e), tail)   # N_alpha, N_beta, E_xc of the block: one deterministic launch
            for s_, wv in ((0, wva), (1, wvb)):
                eng.xc_vmat(ao[0], eng.xc_aow(ao, wv, gga), vmat[s_])
                if gga == 2:
                    for k in (1, 2, 3):
                        eng.xc_vmat(ao[k], wv[4] * ao[k], vmat[s_])
        return hyb

    def _fock_pair(self, dm):
        """One collective per Fock build: [J(2) | K(2) | Vxc(2) | N_alpha N_beta E_xc] partial sums in one flat buffer."""
        self.n_fock_builds = getattr(self, "n_fock_builds", 0) + 1
        eng = self.engine
        n = eng.nao
        nn = n * n
        hyb = parse_xc(self.xc)[0]
        with_k = abs(hyb) > 1e-12
        nj = 2 if with_k else 1          # pure functionals: one J build for the total density
        nk = 2 if with_k else 0
        buf = torch.zeros((nj + nk + 2) * nn + 3, dtype=torch.float64, device=eng.device)
        J = buf[:nj * nn].view(nj, n, n)
        K = buf[nj * nn:(nj + nk) * nn].view(2, n, n) if with_k else None
        V = buf[(nj + nk) * nn:(nj + nk + 2) * nn].view(2, n, n)
        tail = buf[(nj + nk + 2) * nn:]
        self._nr_uks_raw(dm, V, tail)
        D = dm[0] + dm[1]
        if with_k:
            self._jk_into(dm.contiguous(), J, K)
            Jt = J[0] + J[1]
        else:
            self._jk_into(D.contiguous(), J[0], None)
            Jt = J[0]
        if self._nranks > 1:
            from . import parallel
            parallel.all_reduce_sum(buf, self._pg)
            Jt = J[0] + J[1] if with_k else J[0]
        vxc = V + V.transpose(1, 2)
        self._nelec_grid = tail[:2]
        exc = tail[2]
        h1 = self._h1.unsqueeze(0)
        if with_k:
            F = h1 + Jt.unsqueeze(0) + vxc - hyb * K
            e = torch.sum(D * self._h1) + 0.5 * torch.sum(D * Jt) - 0.5 * hyb * torch.sum(dm * K) + exc
        else:
            F = h1 + Jt.unsqueeze(0) + vxc
            e = torch.sum(D * self._h1) + 0.5 * torch.sum(D * Jt) + exc
        return F, e
