"""Python face of the CPU oracle (TEST INFRASTRUCTURE -- see the header of oracle.c).

PARITY UNPINNED against PySCF: the reference's arithmetic lives in un-vendored
pyscf==2.8.0 / gpu4pyscf; nothing here was checked against them in this environment.

The SCF loop restates the control flow PySCF's `scf.hf.kernel` is documented to follow [MEM]
(call site in the reference: `templates/calculate_energy.py:205` `mf.kernel()`):
core guess/dm0 -> get_veff -> loop { Fock (+CDIIS, space 8, start cycle 1) -> eig(F,S) -> aufbau ->
dm -> get_veff -> E_tot ; converge on |dE| < conv_tol and |g|/sqrt(n) < sqrt(conv_tol) } -> one extra
cycle.  numpy + liboracle.so only.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_jk_direct.restype = ctypes.c_long
        _LIB.orc_incore_layout.restype = ctypes.c_long
        _LIB.orc_incore_fill.restype = ctypes.c_long
    return _LIB


def _p(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


class Oracle:
    def __init__(self, mol):
        self.mol = mol
        self.atm = np.ascontiguousarray(mol._atm, dtype=np.int32)
        self.bas = np.ascontiguousarray(mol._bas, dtype=np.int32)
        self.env = np.ascontiguousarray(mol._env, dtype=np.float64)
        self.natm, self.nbas, self.nao = len(self.atm), len(self.bas), mol.nao
        rc = lib().orc_check(_p(self.bas, ctypes.c_int), self.nbas)
        if rc != 0:
            raise ValueError(f"oracle cannot handle this basis (code {rc})")

    def _args(self):
        return (_p(self.atm, ctypes.c_int), self.natm, _p(self.bas, ctypes.c_int), self.nbas, _p(self.env))

    def int1e(self, origin=None):
        n = self.nao
        S, T, V = (np.zeros((n, n)) for _ in range(3))
        dip = np.zeros((3, n, n))
        org = np.zeros(3) if origin is None else np.ascontiguousarray(origin, dtype=np.float64)
        lib().orc_int1e(*self._args(), _p(S), _p(T), _p(V), _p(dip), _p(org))
        return S, T, V, dip

    def schwarz(self):
        q = np.zeros((self.nbas, self.nbas))
        lib().orc_schwarz(*self._args(), _p(q))
        return q

    def eri_full(self):
        n = self.nao
        if n > 80:
            raise MemoryError("eri_full is for small test molecules")
        eri = np.zeros((n, n, n, n))
        lib().orc_eri_full(*self._args(), _p(eri))
        return eri

    def eri_shell(self, i, j, k, l):
        d = [2 * int(self.bas[s, 1]) + 1 for s in (i, j, k, l)]
        out = np.zeros(d)
        lib().orc_eri_shell(*self._args(), int(i), int(j), int(k), int(l), _p(out))
        return out

    def jk(self, dm, tol=1e-13):
        if getattr(self, "_incore", None) is not None:
            return self.jk_incore(dm)
        n = self.nao
        dm = np.ascontiguousarray(dm, dtype=np.float64)
        J, K = np.zeros((n, n)), np.zeros((n, n))
        nq = lib().orc_jk_direct(*self._args(), _p(dm), _p(J), _p(K), ctypes.c_double(tol))
        self.last_nquartets = nq
        return J, K

    def jk_shellblock(self, sa, sb, dm):
        """(J[sa-block, sb-block], K[sa-block, sb-block]) summed over ALL other shells by brute force (full-size spot checks)."""
        da, db = 2 * int(self.bas[sa, 1]) + 1, 2 * int(self.bas[sb, 1]) + 1
        dm = np.ascontiguousarray(dm, dtype=np.float64)
        J, K = np.zeros((da, db)), np.zeros((da, db))
        lib().orc_jk_shellblock(*self._args(), int(sa), int(sb), _p(dm), _p(J), _p(K))
        return J, K

    def incore(self, tol=1e-13, stride=1, phase=0):
        """Packed (ij|kl), ij >= kl (PySCF's cached `int2e` s8 array [MEM]); `stride` > 1 keeps only the rows of every
        stride-th shell pair (bench.py's bounded CPU sample).  Afterwards `jk()` digests the packed array."""
        npair = self.nao * (self.nao + 1) // 2
        row_off = np.zeros(npair, dtype=np.int64)
        need = lib().orc_incore_layout(_p(self.bas, ctypes.c_int), self.nbas, int(stride), int(phase), _p(row_off, ctypes.c_long))
        buf = np.zeros(need)
        nq = lib().orc_incore_fill(*self._args(), ctypes.c_double(tol), int(stride), int(phase), _p(row_off, ctypes.c_long), _p(buf))
        self._incore = (row_off, buf)
        self.incore_nquartets, self.incore_doubles = nq, need
        return self

    def jk_incore(self, dm):
        row_off, buf = self._incore
        n = self.nao
        dm = np.ascontiguousarray(dm, dtype=np.float64)
        J, K = np.zeros((n, n)), np.zeros((n, n))
        lib().orc_jk_incore(_p(self.bas, ctypes.c_int), self.nbas, _p(row_off, ctypes.c_long), _p(buf), _p(dm), _p(J), _p(K))
        return J, K

    @staticmethod
    def num_threads():
        return lib().orc_num_threads()

    @staticmethod
    def set_num_threads(n):
        lib().orc_set_num_threads(int(n))


# ------------------------------------------------------------------------------------------------

class CDIIS:
    """Pulay DIIS, subspace 8 (PySCF `scf.diis.CDIIS` defaults [MEM]).  Error vector: with `Corth` set (what `SCF.kernel` of
    PySCF >= 2.1 does: `_, mf_diis.Corth = mf.eig(fock, s1e)` on the first Fock matrix [MEM: pyscf 2.8 scf/hf.py, scf/diis.py
    get_err_vec_orth]) e = Corth^T (SDF - FDS) Corth in that orthonormal basis; without it the older AO-basis SDF - FDS."""

    def __init__(self, space=8, Corth=None):
        self.space = space
        self.Corth = Corth
        self.f, self.e = [], []

    def update(self, s, d, f):
        sdf = s @ d @ f
        err = sdf.T - sdf
        if self.Corth is not None:
            err = self.Corth.T @ err @ self.Corth
        err = err.ravel()
        self.f.append(f.copy())
        self.e.append(err)
        if len(self.f) > self.space:
            self.f.pop(0)
            self.e.pop(0)
        m = len(self.f)
        B = np.zeros((m + 1, m + 1))
        B[0, 1:] = B[1:, 0] = 1.0
        for i in range(m):
            for j in range(i + 1):
                B[i + 1, j + 1] = B[j + 1, i + 1] = self.e[i] @ self.e[j]
        rhs = np.zeros(m + 1)
        rhs[0] = 1.0
        try:
            c = np.linalg.solve(B, rhs)
        except np.linalg.LinAlgError:
            c = np.linalg.lstsq(B, rhs, rcond=None)[0]
        return sum(ci * fi for ci, fi in zip(c[1:], self.f))


def eig_gen(f, s):
    """FC = SCe by Loewdin-free canonical route: Cholesky of S."""
    L = np.linalg.cholesky(s)
    Li = np.linalg.inv(L)
    e, c = np.linalg.eigh(Li @ f @ Li.T)
    return e, Li.T @ c


def rhf(mol, dm0=None, conv_tol=1e-9, max_cycle=50, veff_fn=None, verbose=False, jk_tol=1e-13, oracle=None):
    """Closed-shell SCF.  `veff_fn(dm) -> (vhf, e_extra_xc_minus_trace_correction)` lets the RKS oracle
    reuse the loop; default is RHF: vhf = J - K/2.  `oracle`: an `Oracle` to reuse (e.g. one holding in-core ERIs)."""
    orc = oracle if oracle is not None else Oracle(mol)
    S, T, V, _ = orc.int1e()
    h = T + V
    enuc = mol.energy_nuc()
    nocc = mol.nelectron // 2

    def make_dm(c):
        co = c[:, :nocc]
        return 2.0 * co @ co.T

    if veff_fn is None:
        def veff_fn(dm):
            J, K = orc.jk(dm, jk_tol)
            return J - 0.5 * K, None

    def energy(dm, vhf, exc):
        e1 = float(np.sum(dm * h))
        if exc is None:
            e2 = 0.5 * float(np.sum(dm * vhf))
        else:
            e2 = exc  # veff_fn returns the full two-electron energy (Coulomb + xc) itself
        return e1 + e2 + enuc

    if dm0 is None:
        e, c = eig_gen(h, S)
        dm = make_dm(c)
    else:
        dm = np.array(dm0, dtype=np.float64)
    vhf, exc = veff_fn(dm)
    e_tot = energy(dm, vhf, exc)
    diis = CDIIS(Corth=eig_gen(h + vhf, S)[1])
    conv_tol_grad = np.sqrt(conv_tol)
    converged = False
    mo_e = mo_c = None
    cycles = 0
    for cycle in range(max_cycle):
        f = h + vhf
        if cycle >= 1:
            f = diis.update(S, dm, f)
        mo_e, mo_c = eig_gen(f, S)
        dm = make_dm(mo_c)
        vhf, exc = veff_fn(dm)
        e_last, e_tot = e_tot, energy(dm, vhf, exc)
        f = h + vhf
        g = 2.0 * mo_c[:, nocc:].T @ f @ mo_c[:, :nocc]
        gnorm = np.linalg.norm(g) / np.sqrt(max(g.size, 1))
        cycles = cycle + 1
        if verbose:
            print(f"oracle cycle {cycles:3d}  E = {e_tot:.12f}  dE = {e_tot - e_last: .3e}  |g| = {gnorm:.3e}")
        if abs(e_tot - e_last) < conv_tol and gnorm < conv_tol_grad:
            converged = True
            break
    if converged:  # one extra cycle (PySCF conv_check)
        mo_e, mo_c = eig_gen(h + vhf, S)
        dm = make_dm(mo_c)
        vhf, exc = veff_fn(dm)
        e_tot = energy(dm, vhf, exc)
    mo_occ = np.zeros(len(mo_e))
    mo_occ[:nocc] = 2.0
    return dict(e_tot=e_tot, converged=converged, mo_energy=mo_e, mo_coeff=mo_c, mo_occ=mo_occ, dm=dm,
                cycles=cycles, S=S, h=h, vhf=vhf)


def uhf(mol, conv_tol=1e-10, max_cycle=100, dm0=None, verbose=False):
    """Spin-unrestricted HF on the CPU oracle integrals (checker for `mi355scf.uhf.UHF`): F_s = h + J[Da+Db] - K[D_s],
    CDIIS on the stacked (F_a, F_b), aufbau occupations n_alpha >= n_beta.  dm0: [2,N,N] or a total density [N,N]
    (split by electron count) or None (core-Hamiltonian guess).  Returns (e_tot, (Da, Db), (ea, eb), (Ca, Cb))."""
    o = Oracle(mol)
    S, T, V, _ = o.int1e()
    h = T + V
    na, nb = mol.nelec
    enuc = mol.energy_nuc()

    def dens(F):
        out, es, cs = [], [], []
        for s_, no in ((0, na), (1, nb)):
            e, c = eig_gen(F[s_], S)
            out.append(c[:, :no] @ c[:, :no].T); es.append(e); cs.append(c)
        return np.stack(out), es, cs

    if dm0 is None:
        dm, _, _ = dens(np.stack([h, h]))
    else:
        dm0 = np.asarray(dm0)
        dm = dm0 if dm0.ndim == 3 else np.stack([dm0 * na / max(na + nb, 1), dm0 * nb / max(na + nb, 1)])

    def fock(dm):
        Ja, Ka = o.jk(dm[0], tol=0.0)
        Jb, Kb = o.jk(dm[1], tol=0.0)
        F = np.stack([h + Ja + Jb - Ka, h + Ja + Jb - Kb])
        return F, 0.5 * float(np.sum(dm * (h[None] + F))) + enuc

    F, e = fock(dm)
    Fh, Eh = [], []
    es = cs = None
    for it in range(max_cycle):
        err = np.stack([F[s_] @ dm[s_] @ S - S @ dm[s_] @ F[s_] for s_ in range(2)])
        Fh.append(F.copy()); Eh.append(err.ravel().copy())
        Fh, Eh = Fh[-8:], Eh[-8:]
        m = len(Fh)
        A = np.zeros((m + 1, m + 1)); A[0, 1:] = A[1:, 0] = 1.0
        A[1:, 1:] = np.array([[a @ b for b in Eh] for a in Eh])
        rhs = np.zeros(m + 1); rhs[0] = 1.0
        c = np.linalg.lstsq(A, rhs, rcond=None)[0]
        Fx = sum(ci * Fi for ci, Fi in zip(c[1:], Fh))
        dm, es, cs = dens(Fx)
        F, e_new = fock(dm)
        if verbose:
            print(f"oracle uhf cycle {it + 1}: E = {e_new:.12f}  dE = {e_new - e:.3e}  |err| = {np.abs(err).max():.3e}")
        done = abs(e_new - e) < conv_tol and np.abs(err).max() < 1e-6
        e = e_new
        if done:
            break
    dm, es, cs = dens(F)
    F, e = fock(dm)
    return e, dm, es, cs


def mp2(mol, r=None):
    """Closed-shell MP2 correlation energy from the oracle's dense ERIs (checker for `mi355scf.mp2.MP2`; small N only)."""
    r = r or rhf(mol, conv_tol=1e-11)
    o = Oracle(mol)
    eri = o.eri_full()
    c, e, occ = r["mo_coeff"], r["mo_energy"], r["mo_occ"]
    co, cv, eo, ev = c[:, occ > 0], c[:, occ == 0], e[occ > 0], e[occ == 0]
    ovov = np.einsum("pqrs,pi,qa,rj,sb->iajb", eri, co, cv, co, cv, optimize=True)
    d = eo[:, None, None, None] - ev[None, :, None, None] + eo[None, None, :, None] - ev[None, None, None, :]
    return float(np.sum(ovov / d * (2 * ovov - ovov.transpose(0, 3, 2, 1))))
