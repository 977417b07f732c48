"""Nuclear Hessian behind `pyscf.hessian.{RHF,RKS,rhf.Hessian,rks.Hessian,...}` and `gpu4pyscf.hessian` (SURVEY.md
section 8f rank 4; call sites `templates/optimize_geometry.py:117-123` (`hessian.RHF(mf)`, `hessian.RKS(mf)`, `.kernel()`)
and `templates/opt-freq.py:392-417` (`gpu_hessian.rks.Hessian(mf_opt).kernel()`, `hessian.rks.Hessian(mf_cpu)`)).

Semi-numerical: central finite differences of the ANALYTIC gradient (HIP derivative-integral kernels, `grad.py`), one SCF +
gradient per displaced geometry (6 N_atom in total).  ONE working object is moved through all displaced geometries by the
gradient scanner (`mf.reset(mol)`: the resident tile store, the library handles and the purification plans are reused -- a
fresh object per point would allocate its own store, 100 GB for ibuprofen/def2-TZVP), each SCF starts from the density of
the point before.  Analytic second derivatives (CPHF) are not implemented.

Several GPUs: the 6 N displaced points are independent, so `Hessian.distribute(rank, nranks)` (one process per GPU, the base
object NOT sharded) deals the 3 N coordinates round-robin to the ranks -- each rank moves its own working object, with the
whole tile store on its GPU, through its share -- and the rows are summed with one all-reduce at the end.  No collective on
the data path of a point, unlike the tile-sharded SCF (`mf.shard`), whose replicated per-cycle algebra caps the speed-up of
small molecules.  Returns the PySCF layout `hess[i, j, x, y] = d2E / dR_ix dR_jy` (Hartree/Bohr^2).
"""
import numpy as np


class Hessian:
    step = 5.0e-3   # Bohr

    def __init__(self, mf):
        self.base = mf
        self.mol = mf.mol
        self.verbose = mf.verbose
        self.de = None
        self._dist = (0, 1, None)

    def distribute(self, rank, nranks, process_group=None):
        """Deal the displaced coordinates to `nranks` processes (replicas: every rank runs whole, unsharded SCFs + gradients on
        its own GPU); `kernel()` then returns the complete Hessian on every rank."""
        self._dist = (int(rank), int(nranks), process_group)
        return self

    auto_replicas = True   # tile-sharded base object (torchrun auto-shard) whose whole store fits one GPU: replicas instead

    def _replica_plan(self):
        """(rank, nranks, group) of the replica mode, or (0, 1, None): explicit `distribute`, or automatically when the base
        object is tile-sharded over several GPUs although the WHOLE tile store would fit each of them (then dealing the 3 N
        coordinates costs no collective per SCF cycle, where the sharded SCF pays one per Fock build plus replicated algebra)."""
        if self._dist[1] > 1:
            return self._dist
        mf = self.base
        nr = getattr(mf, "_nranks", 1)
        if nr > 1 and self.auto_replicas and getattr(mf, "with_df", None) is None:
            try:
                import torch
                need = float(mf.engine.stats()["stored_bytes"]) * nr * 1.1
                free, total = torch.cuda.mem_get_info(mf.engine.device)
                if need < 0.5 * total:
                    return (mf._rank, nr, getattr(mf, "_pg", None))
            except Exception:
                pass
        return (0, 1, None)

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
        rank, nranks, pg = self._replica_plan()
        work = self._clone_at(R)
        if getattr(mf, "with_df", None) is not None:
            work = work.density_fit(mf.with_df.auxbasis)
        if nranks > 1:
            work.shard(0, 1)                 # replicas: every rank's working object is whole
        elif getattr(mf, "_nranks", 1) > 1:
            work.shard(mf._rank, mf._nranks, getattr(mf, "_pg", None))
        else:
            work.shard(0, 1)
        work.kernel(dm0=dm0)
        scan = work.nuc_grad_method().as_scanner()
        for ia in range(n):
            for x in range(3):
                if (3 * ia + x) % nranks != rank:
                    continue
                g, mu = [], []
                for sgn in (+1.0, -1.0):
                    Rd = R.copy()
                    Rd[ia, x] += sgn * h
                    _e, de = scan(Rd)
                    if not scan.converged:
                        raise RuntimeError("SCF did not converge at a displaced geometry of the Hessian")
                    g.append(np.array(de))
                    mu.append(np.asarray(work.dip_moment(unit="au")))
                H[ia, x] = (g[0] - g[1]) / (2.0 * h)
                dmu[ia, x] = (mu[0] - mu[1]) / (2.0 * h)
            mf._log(4, f"Hessian: atom {ia + 1}/{n} done")
        if nranks > 1:      # rows of the other ranks (zeros here) arrive with one all-reduce
            import torch
            from . import parallel
            dev = work.engine.device
            buf = torch.as_tensor(np.concatenate([H.ravel(), dmu.ravel()]), device=dev)
            parallel.all_reduce_sum(buf, pg)
            buf = buf.cpu().numpy()
            H, dmu = buf[:H.size].reshape(H.shape), buf[H.size:].reshape(dmu.shape)
        Hm = H.reshape(3 * n, 3 * n)
        Hm = 0.5 * (Hm + Hm.T)
        self.de = Hm.reshape(n, 3, n, 3).transpose(0, 2, 1, 3).copy()
        self.dipole_deriv = dmu
        return self.de

    hess = kernel
