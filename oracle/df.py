"""CPU restatement of density-fitted J/K (TEST INFRASTRUCTURE, like oracle.py; parity with PySCF's `df` module UNPINNED: the
reference never calls `density_fit()` and PySCF is not available here -- SURVEY.md section 8c, 8f rank 3).

(ij|P) and (P|Q) come from the oracle's McMurchie-Davidson four-centre routine `orc_eri_shell` on a combined shell list
[orbital shells | auxiliary shells | unit function], the unit function being an s primitive with exponent 0 and coefficient
sqrt(4 pi) -- exactly how libcint's int3c2e / int2c2e are related to int2e.  Dense numpy algebra afterwards:
    J = (ij|P) [ (P|Q)^-1 (Q|kl) D_kl ],   K_ik = sum_PQ (ij|P) (P|Q)^-1 (Q|kl) D_jl.
"""
import math

import numpy as np

from . import oracle as orc


class _Combined:
    def __init__(self, mol, auxmol):
        n_env = len(mol._env)
        bas_a = auxmol._bas.copy()
        atm_a = auxmol._atm.copy()
        bas_a[:, 5] += n_env
        bas_a[:, 6] += n_env
        atm_a[:, 1] += n_env
        bas_a[:, 0] += mol.natm
        env = np.concatenate([mol._env, auxmol._env, [0.0, math.sqrt(4.0 * math.pi)]])
        pe = n_env + len(auxmol._env)
        unit = np.array([[0, 0, 1, 1, 0, pe, pe + 1, 0]], dtype=np.int32)
        self._atm = np.vstack([mol._atm, atm_a]).astype(np.int32)
        self._bas = np.vstack([mol._bas, bas_a, unit]).astype(np.int32)
        self._env = env
        self.nao = mol.nao + auxmol.nao + 1
        self.nbas_orb, self.nbas_aux = mol.nbas, auxmol.nbas


def integrals(mol, auxmol):
    """((ij|P) [nao, nao, naux], (P|Q) [naux, naux]) by brute force over shell triples."""
    cm = _Combined(mol, auxmol)
    o = orc.Oracle(cm)
    n, na = mol.nao, auxmol.nao
    lo = mol.ao_loc_nr()
    la = auxmol.ao_loc_nr()
    u = cm.nbas_orb + cm.nbas_aux
    j3 = np.zeros((n, n, na))
    for i in range(mol.nbas):
        for j in range(i + 1):
            for p in range(auxmol.nbas):
                blk = o.eri_shell(i, j, cm.nbas_orb + p, u)[..., 0]
                j3[lo[i]:lo[i + 1], lo[j]:lo[j + 1], la[p]:la[p + 1]] = blk
                j3[lo[j]:lo[j + 1], lo[i]:lo[i + 1], la[p]:la[p + 1]] = blk.transpose(1, 0, 2)
    j2 = np.zeros((na, na))
    for p in range(auxmol.nbas):
        for q in range(p + 1):
            blk = o.eri_shell(cm.nbas_orb + p, u, cm.nbas_orb + q, u)[:, 0, :, 0]
            j2[la[p]:la[p + 1], la[q]:la[q + 1]] = blk
            j2[la[q]:la[q + 1], la[p]:la[p + 1]] = blk.T
    return j3, j2


def jk(j3, j2, dm):
    n, _, na = j3.shape
    L = np.linalg.cholesky(j2)
    B = np.linalg.solve(L, j3.reshape(n * n, na).T).reshape(na, n, n)
    rho = np.einsum("pkl,kl->p", B, dm)
    J = np.einsum("pij,p->ij", B, rho)
    K = np.einsum("pij,jl,pkl->ik", B, dm, B)
    return J, K
