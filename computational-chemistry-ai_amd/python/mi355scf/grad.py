"""Nuclear gradients behind `mf.nuc_grad_method()` (used by `optimize`,
`templates/optimize_geometry.py:99`).

`Gradients` is analytic: dE/dR = 1e term (`mi_grad_1e`: D.dh - W.dS + Hellmann-Feynman) + 2e term
(`mi_grad_eri`: derivative ERIs contracted with the two-particle density) + XC term (HIP AO second
derivatives + device contractions, no grid-weight response, as PySCF's default `grid_response=False` [MEM])
+ nuclear repulsion.  `FDGradients` (central finite differences of the SCF energy) is kept as the
independent check used by the tests.
"""
import numpy as np
import torch

from .mole import Mole


def grad_nuc(mol):
    z = mol.atom_charges().astype(np.float64)
    R = mol.atom_coords()
    g = np.zeros_like(R)
    for i in range(mol.natm):
        for j in range(mol.natm):
            if i != j:
                d = R[i] - R[j]
                g[i] -= z[i] * z[j] * d / np.linalg.norm(d) ** 3
    return g


class Gradients:
    """Analytic RHF/RKS gradient on the MI355X engine."""

    def __init__(self, mf):
        self.base = mf
        self.mol = mf.mol
        self.de = None
        self.verbose = mf.verbose

    def grad_xc(self, dm):
        """-2 sum_{mu on A} D_mu,nu int [v_rho dphi_mu phi_nu + 2 v_sigma grad rho . grad(dphi_mu phi_nu)]"""
        from .dft import parse_xc
        mf = self.base
        eng = mf.engine
        hyb, terms, gga = parse_xc(mf.xc)
        n = eng.nao
        coords, weights = mf.grids.coords, mf.grids.weights
        lo, hi = mf._grid_range(coords.shape[0])
        fmu = torch.zeros(n, 3, dtype=torch.float64, device=eng.device)
        B = max(4096, mf.grid_block // 4)
        pair = {(0, 0): 4, (0, 1): 5, (0, 2): 6, (1, 1): 7, (1, 2): 8, (2, 2): 9}
        for p0 in range(lo, hi, B):
            p1 = min(p0 + B, hi)
            c, w = coords[p0:p1], weights[p0:p1]
            ao = eng.eval_ao(c, deriv=2 if gga else 1)
            C = dm @ ao[0]
            rho = eng.xc_rho(ao, C, deriv=1 if gga else 0)
            if gga == 2:
                _e, wv = eng.xc_eval_mgga(terms, rho, eng.xc_tau(ao, dm), w)
            else:
                _e, wv = eng.xc_eval(terms, rho, w, gga)
            if gga:
                Ck = [dm @ ao[1 + j] for j in range(3)]
                T1 = 2.0 * wv[0] * C
                for j in range(3):
                    T1 += wv[1 + j] * Ck[j]
                for k in range(3):
                    t2 = sum(wv[1 + j] * ao[pair[(min(j, k), max(j, k))]] for j in range(3))
                    fmu[:, k] += -2.0 * ((ao[1 + k] * T1).sum(dim=1) + (t2 * C).sum(dim=1))
                    if gga == 2:   # d tau / d A_x = - sum_k (d_x d_k phi_mu) (D d_k phi)_mu ; wv[4] = w/4 de/dtau
                        fmu[:, k] += -4.0 * sum((ao[pair[(min(j, k), max(j, k))]] * (wv[4] * Ck[j])).sum(dim=1) for j in range(3))
            else:
                T1 = 2.0 * wv[0] * C
                for k in range(3):
                    fmu[:, k] += -2.0 * (ao[1 + k] * T1).sum(dim=1)
        if mf._nranks > 1:
            from . import parallel
            parallel.all_reduce_sum(fmu, mf._pg)
        f = fmu.cpu().numpy()
        sl = mf.mol.aoslice_by_atom()
        return np.array([f[sl[ia, 2]:sl[ia, 3]].sum(axis=0) for ia in range(mf.mol.natm)])

    def kernel(self, mo_energy=None, mo_coeff=None, mo_occ=None, atmlst=None):
        mf = self.base
        if mf._dm is None or not mf.converged:
            mf.kernel()
        eng = mf.engine
        mol = mf.mol
        dm = mf._dm
        fock = mf._h1 + mf._vhf
        W = 0.5 * dm @ fock @ dm
        import time
        tm = {}
        t0 = time.time()
        g = torch.zeros(mol.natm, 3, dtype=torch.float64, device=eng.device)
        eng.grad_1e(dm.contiguous(), W.contiguous(), g)
        torch.cuda.synchronize()
        tm["grad_1e"] = time.time() - t0
        is_ks = getattr(mf, "xc", None) is not None and hasattr(mf, "grids")
        hyb = 1.0
        if is_ks:
            from .dft import parse_xc
            hyb = parse_xc(mf.xc)[0]
        t0 = time.time()
        if getattr(mf, "with_df", None) is not None:
            g2 = _fitted_tensor(mf).grad_jk(dm, hyb, rank=mf._rank, nranks=mf._nranks)
        else:
            g2 = torch.zeros_like(g)
            eng.grad_eri(dm.contiguous(), hyb, g2, rank=mf._rank, nranks=mf._nranks)
        tm["grad_eri"] = time.time() - t0
        if mf._nranks > 1:   # derivative-quartet batches are dealt round-robin to ranks inside mi_grad_eri
            from . import parallel
            parallel.all_reduce_sum(g2, mf._pg)
        de = (g + g2).cpu().numpy() + grad_nuc(mol)
        if is_ks:
            t0 = time.time()
            de = de + self.grad_xc(dm)
            torch.cuda.synchronize()
            tm["grad_xc"] = time.time() - t0
        if mf._nranks > 1:
            from . import parallel
            dt = torch.as_tensor(de, device=eng.device)
            parallel.broadcast0(dt, mf._pg)
            de = dt.cpu().numpy()
        self.timing = tm
        mf._log(4, "gradient timings (s): " + ", ".join(f"{k} {v:.3f}" for k, v in tm.items()))
        self.de = de
        if self.verbose >= 4:
            mf._log(4, "--------------- gradients ---------------")
            for ia in range(mol.natm):
                mf._log(4, "%d %s  %16.10f %16.10f %16.10f" % (ia, mol.atom_pure_symbol(ia), *de[ia]))
        return de

    grad = kernel

    def as_scanner(self):
        return _GradScanner(self)


def _fitted_tensor(mf):
    """The fitted tensor with the WHOLE auxiliary index on this rank: `mf.with_df` itself, or (sharded run: every rank holds a
    slice of the whitened index, and the exchange part of the two-index density couples all slices) one rebuilt here."""
    d = mf.with_df
    if getattr(d, "_nranks_built", 1) == 1 and d._B is not None:
        return d
    from .df import DF
    return DF(mf.mol, d.auxbasis, d.beta).build(mf.engine)


class UGradients(Gradients):
    """Analytic UHF gradient: the restricted pieces with D = Da + Db, W = sum_s Ds Fs Ds, and the exchange part of
    the two-particle density from both spins (`mi_grad_eri_spin`, spin density M = Da - Db)."""

    def kernel(self, mo_energy=None, mo_coeff=None, mo_occ=None, atmlst=None):
        mf = self.base
        if mf._dm is None or not mf.converged:
            mf.kernel()
        eng = mf.engine
        mol = mf.mol
        dma, dmb = mf._dm[0], mf._dm[1]
        F = mf._fock
        D = (dma + dmb).contiguous()
        M = (dma - dmb).contiguous()
        W = (dma @ F[0] @ dma + dmb @ F[1] @ dmb).contiguous()
        g = torch.zeros(mol.natm, 3, dtype=torch.float64, device=eng.device)
        eng.grad_1e(D, W, g)
        is_ks = getattr(mf, "xc", None) is not None and hasattr(mf, "grids")
        hyb = 1.0
        if is_ks:
            from .dft import parse_xc
            hyb = parse_xc(mf.xc)[0]
        if getattr(mf, "with_df", None) is not None:
            g2 = _fitted_tensor(mf).grad_jk([dma, dmb], hyb, rank=mf._rank, nranks=mf._nranks)
        else:
            g2 = torch.zeros_like(g)
            eng.grad_eri(D, hyb, g2, spin_density=M, rank=mf._rank, nranks=mf._nranks)
        if mf._nranks > 1:
            from . import parallel
            parallel.all_reduce_sum(g2, mf._pg)
        de = (g + g2).cpu().numpy() + grad_nuc(mol)
        if is_ks:
            de = de + self.grad_xc_spin(mf._dm)
        if mf._nranks > 1:
            from . import parallel
            dt = torch.as_tensor(de, device=eng.device)
            parallel.broadcast0(dt, mf._pg)
            de = dt.cpu().numpy()
        self.de = de
        if self.verbose >= 4:
            mf._log(4, "--------------- gradients ---------------")
            for ia in range(mol.natm):
                mf._log(4, "%d %s  %16.10f %16.10f %16.10f" % (ia, mol.atom_pure_symbol(ia), *de[ia]))
        return de

    grad = kernel

    def grad_xc_spin(self, dm):
        """XC gradient of UKS: the restricted expression per spin with (D_s, wv_s) from the spin-polarised functional."""
        from .dft import parse_xc
        mf = self.base
        eng = mf.engine
        hyb, terms, gga = parse_xc(mf.xc)
        n = eng.nao
        coords, weights = mf.grids.coords, mf.grids.weights
        lo, hi = mf._grid_range(coords.shape[0])
        fmu = torch.zeros(n, 3, dtype=torch.float64, device=eng.device)
        B = max(4096, mf.grid_block // 4)
        pair = {(0, 0): 4, (0, 1): 5, (0, 2): 6, (1, 1): 7, (1, 2): 8, (2, 2): 9}
        for p0 in range(lo, hi, B):
            p1 = min(p0 + B, hi)
            c, w = coords[p0:p1], weights[p0:p1]
            ao = eng.eval_ao(c, deriv=2 if gga else 1)
            Cs = [dm[s_] @ ao[0] for s_ in range(2)]
            rho = [eng.xc_rho(ao, Cs[s_], deriv=1 if gga else 0) for s_ in range(2)]
            if gga == 2:
                _e, wva, wvb = eng.xc_eval_mgga_spin(terms, rho[0], rho[1], eng.xc_tau(ao, dm[0]), eng.xc_tau(ao, dm[1]), w)
            else:
                _e, wva, wvb = eng.xc_eval_spin(terms, rho[0], rho[1], w, gga)
            for s_, wv in ((0, wva), (1, wvb)):
                C = Cs[s_]
                T1 = 2.0 * wv[0] * C
                if gga:
                    Ck = [dm[s_] @ ao[1 + j] for j in range(3)]
                    for j in range(3):
                        T1 += wv[1 + j] * Ck[j]
                    for k in range(3):
                        t2 = sum(wv[1 + j] * ao[pair[(min(j, k), max(j, k))]] for j in range(3))
                        fmu[:, k] += -2.0 * ((ao[1 + k] * T1).sum(dim=1) + (t2 * C).sum(dim=1))
                        if gga == 2:
                            fmu[:, k] += -4.0 * sum((ao[pair[(min(j, k), max(j, k))]] * (wv[4] * Ck[j])).sum(dim=1) for j in range(3))
                else:
                    for k in range(3):
                        fmu[:, k] += -2.0 * (ao[1 + k] * T1).sum(dim=1)
        if mf._nranks > 1:
            from . import parallel
            parallel.all_reduce_sum(fmu, mf._pg)
        f = fmu.cpu().numpy()
        sl = mf.mol.aoslice_by_atom()
        return np.array([f[sl[ia, 2]:sl[ia, 3]].sum(axis=0) for ia in range(mf.mol.natm)])


class FDGradients:
    step = 2.0e-3  # Bohr

    def __init__(self, mf):
        self.base = mf
        self.mol = mf.mol
        self.de = None
        self.verbose = mf.verbose

    def _energy_at(self, coords, dm0):
        mf = self.base
        mol = mf.mol.set_geom_(coords, unit="Bohr", inplace=False)
        mol.verbose = 0
        clone = mf.__class__(mol)
        for k in ("xc", "conv_tol", "max_cycle", "eig_method", "init_guess", "direct_scf_tol", "level_shift", "diis_space",
                  "diis_start_cycle"):
            if hasattr(mf, k):
                setattr(clone, k, getattr(mf, k))
        if hasattr(mf, "grids"):
            clone.grids.level, clone.grids.prune = mf.grids.level, mf.grids.prune
        clone.verbose = 0
        clone.conv_tol = min(mf.conv_tol, 1e-10)
        e = clone.kernel(dm0=dm0)
        if not clone.converged:
            raise RuntimeError("SCF did not converge at a displaced geometry")
        clone._eng = None
        return e

    def kernel(self, mo_energy=None, mo_coeff=None, mo_occ=None, atmlst=None):
        mf = self.base
        if mf.mo_coeff is None or not mf.converged:
            mf.kernel()
        dm0 = mf.make_rdm1()
        R = mf.mol.atom_coords()
        g = np.zeros_like(R)
        h = self.step
        atoms = range(mf.mol.natm) if atmlst is None else atmlst
        for ia in atoms:
            for x in range(3):
                Rp, Rm = R.copy(), R.copy()
                Rp[ia, x] += h
                Rm[ia, x] -= h
                g[ia, x] = (self._energy_at(Rp, dm0) - self._energy_at(Rm, dm0)) / (2 * h)
        self.de = g
        if self.verbose >= 5:
            mf._log(4, "--------------- gradients (finite difference) ---------------")
            for ia in range(mf.mol.natm):
                mf._log(4, "%d %s  %16.10f %16.10f %16.10f" % (ia, mf.mol.atom_pure_symbol(ia), *g[ia]))
        return g

    grad = kernel

    def as_scanner(self):
        return _GradScanner(self)


class _GradScanner:
    def __init__(self, g):
        self.g = g
        self.base = g.base
        self.mol = g.mol
        self.converged = True

    def __call__(self, mol_or_geom):
        mf = self.g.base
        mol = mol_or_geom if isinstance(mol_or_geom, Mole) else mf.mol.set_geom_(mol_or_geom, unit="Bohr", inplace=False)
        # starting density: the converged AO matrix as it is (the basis functions follow their atoms).  Transporting it in the
        # Cholesky-orthonormal frame instead (D = L_new^-T (L_old^T D L_old) L_new^-1: idempotent and N-electron in the new
        # metric) was measured WORSE -- ibuprofen B3LYP/def2-TZVP optimisation 105 instead of 86 SCF cycles over 12 steps,
        # benzene 20 instead of 14 (tools/opt_variants.py): Gram-Schmidt in AO order drags every later atom's functions along
        dm0 = mf.make_rdm1() if mf.mo_coeff is not None else None
        mf.reset(mol)
        mf._grad_prefetch = True      # the gradient's host-side set-up overlaps this SCF (engine option `grad_prefetch`)
        e = mf.kernel(dm0=dm0)
        self.converged = mf.converged
        self.g.mol = self.mol = mol
        if getattr(self, "grad_dtol", None) is not None:
            mf.engine.set_option("grad_dtol", self.grad_dtol)   # density-weighted screening of the derivative quartets
        de = self.g.kernel()
        return e, de
