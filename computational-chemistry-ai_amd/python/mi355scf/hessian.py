"""Nuclear Hessian behind `pyscf.hessian.{RHF,RKS,rhf.Hessian,rks.Hessian,...}` and `gpu4pyscf.hessian` (SURVEY.md
section 8f rank 4; call sites `templates/optimize_geometry.py:117-123` (`hessian.RHF(mf)`, `hessian.RKS(mf)`, `.kernel()`)
and `templates/opt-freq.py:392-417` (`gpu_hessian.rks.Hessian(mf_opt).kernel()`, `hessian.rks.Hessian(mf_cpu)`)).

Semi-numerical: central finite differences of the ANALYTIC gradient (HIP derivative-integral kernels, `grad.py`), one SCF +
gradient per displaced geometry (6 N_atom in total, each SCF warm-started from the reference density).  Analytic second
derivatives (CPHF) are not implemented.  Returns the PySCF layout `hess[i, j, x, y] = d2E / dR_ix dR_jy` (Hartree/Bohr^2).
"""
import numpy as np


class Hessian:
    step = 5.0e-3   # Bohr

    def __init__(self, mf):
        self.base = mf
        self.mol = mf.mol
        self.verbose = mf.verbose
        self.de = None

    def _clone_at(self, coords):
        mf = self.base
        mol = mf.mol.set_geom_(coords, unit="Bohr", inplace=False)
        mol.verbose = 0
        clone = mf.__class__(mol)
        for k in ("xc", "max_cycle", "eig_method", "init_guess", "direct_scf_tol", "small_rho_cutoff", "level_shift", "diis_space",
                  "diis_start_cycle"):
            if hasattr(mf, k) and getattr(mf, k) is not None:
                setattr(clone, k, getattr(mf, k))
        if hasattr(mf, "grids") and hasattr(clone, "grids"):
            clone.grids.level, clone.grids.prune = mf.grids.level, mf.grids.prune
        clone.verbose = 0
        clone.conv_tol = min(mf.conv_tol, 1e-10)
        clone.conv_tol_grad = 1e-6
        return clone

    def kernel(self, mo_energy=None, mo_coeff=None, mo_occ=None, atmlst=None):
        mf = self.base
        if mf.mo_coeff is None or not mf.converged:
            mf.kernel()
        dm0 = mf.make_rdm1()
        mol = mf.mol
        R = mol.atom_coords()
        n = mol.natm
        H = np.zeros((n, 3, n, 3))
        dmu = np.zeros((n, 3, 3))          # d mu_c / d R_ix (a.u.), by-product of the same displaced SCFs (IR intensities)
        h = self.step
        for ia in range(n):
            for x in range(3):
                g, mu = [], []
                for sgn in (+1.0, -1.0):
                    Rd = R.copy()
                    Rd[ia, x] += sgn * h
                    c = self._clone_at(Rd)
                    c.kernel(dm0=dm0)
                    if not c.converged:
                        raise RuntimeError("SCF did not converge at a displaced geometry of the Hessian")
                    g.append(c.nuc_grad_method().kernel())
                    mu.append(np.asarray(c.dip_moment(unit="au")))
                    c._eng = None
                H[ia, x] = (g[0] - g[1]) / (2.0 * h)
                dmu[ia, x] = (mu[0] - mu[1]) / (2.0 * h)
            mf._log(4, f"Hessian: atom {ia + 1}/{n} done")
        Hm = H.reshape(3 * n, 3 * n)
        Hm = 0.5 * (Hm + Hm.T)
        self.de = Hm.reshape(n, 3, n, 3).transpose(0, 2, 1, 3).copy()
        self.dipole_deriv = dmu
        return self.de

    hess = kernel
