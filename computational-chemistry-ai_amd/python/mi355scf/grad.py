"""Nuclear gradients behind `mf.nuc_grad_method()` (used by `optimize`,
`templates/optimize_geometry.py:99`).

Round-1 status: the analytic derivative-integral kernels are not built yet; `Gradients.kernel()`
evaluates dE/dR by CENTRAL FINITE DIFFERENCES of the GPU SCF energy (warm-started from the converged
density, 6 N_atom SCF runs).  It is exact to O(h^2) and serves as the check for the analytic path of
the next round; it is far too slow for ibuprofen-sized systems (BASELINE config 5).
"""
import numpy as np

from .mole import Mole


class Gradients:
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
        for k in ("xc", "conv_tol", "max_cycle", "eig_method", "init_guess", "direct_scf_tol"):
            if hasattr(mf, k):
                setattr(clone, k, getattr(mf, k))
        if hasattr(mf, "grids"):
            clone.grids.level = mf.grids.level
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
        if self.verbose >= 4:
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
        dm0 = mf.make_rdm1() if mf.mo_coeff is not None else None
        mf.reset(mol)
        e = mf.kernel(dm0=dm0)
        self.converged = mf.converged
        self.g.mol = self.mol = mol
        de = self.g.kernel()
        return e, de
