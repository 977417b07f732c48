"""Density fitting behind `mf.density_fit()` (the reference never calls it, PySCF idiom [MEM]).

    (ij|kl) ~ sum_PQ (ij|P) (P|Q)^-1 (Q|kl)

Three- and two-index Coulomb integrals come from the HIP Rys kernels (`mi_df_build`: an auxiliary function is the "pair"
(P, unit s function)); the fitted tensor B[P, i, j] = sum_Q L^-1[P, Q] (Q|ij) (L = Cholesky factor of the metric) stays
resident in HBM (8 N^2 N_aux bytes) and every Fock build is a handful of dense FP64 GEMMs (rocBLAS, FP64 MFMA):

    J_ij = sum_P B[P,i,j] (sum_kl B[P,k,l] D_kl)          2 N^2 N_aux flops x 2
    K_ik = sum_P (B_P D B_P)_ik                           4 N^3 N_aux flops

No JKFIT tables exist offline, so the auxiliary basis is generated: even-tempered exponents per element and angular
momentum spanning the products of the orbital primitives (the idea of PySCF's `df.aug_etb` [MEM]; not its exact recipe --
parity with PySCF's fitted energies is UNPINNED), l_aux <= 4 (g auxiliary shells; the two-electron kernels take l <= 4, the Rys tables stop at 8 roots).
"""
import math

import numpy as np
import torch

from . import engine as _engine
from .mole import Mole, split_ghost

LMAX_AUX = 4


def even_tempered_aux(mol, beta=2.0):
    """{element: PySCF-format shell list}: for each l_aux <= min(2 l_max, LMAX_AUX = 4) exponents alpha_k = a_min beta^k covering
    [a_i + a_j] over the orbital primitive pairs with l_i + l_j >= l_aux (|l_i - l_j| <= l_aux)."""
    out = {}
    bas, env = mol._bas, mol._env
    for ia in range(mol.natm):
        sym = split_ghost(mol.atom_symbol(ia))[1]
        if sym in out:
            continue
        shells = [(int(b[1]), env[b[5]:b[5] + b[2]]) for b in bas if b[0] == ia]
        lmax = max(l for l, _ in shells)
        aux = []
        for la in range(0, min(2 * lmax, LMAX_AUX) + 1):
            lo, hi = None, None
            for l1, e1 in shells:
                for l2, e2 in shells:
                    if l1 + l2 >= la and abs(l1 - l2) <= la:
                        lo = min(e1) + min(e2) if lo is None else min(lo, min(e1) + min(e2))
                        hi = max(e1) + max(e2) if hi is None else max(hi, max(e1) + max(e2))
            if lo is None:
                continue
            n = int(math.ceil(math.log(hi / lo) / math.log(beta))) + 1
            for k in range(n):
                aux.append([la, [lo * beta ** k, 1.0]])
        out[sym] = aux
    return out


class DF:
    """Fitted three-index tensor of one molecule on one GPU."""

    def __init__(self, mol, auxbasis=None, beta=2.0):
        self.mol = mol
        self.auxbasis = auxbasis
        self.beta = beta
        self._B = None
        self.auxmol = None

    def build(self, engine, rank=0, nranks=1):
        """Evaluate (ij|P), (P|Q) and keep the whitened tensor B.  `rank`/`nranks`: a sharded run keeps only ITS contiguous
        slice of the whitened auxiliary index, B[i, P_r, j] (1 / nranks of the memory and of the J/K work): rho_P, J and K are
        sums over P, so every rank's partial J, K enter the Fock all-reduce like the partials of the four-centre tile shards.
        (The three-index integrals themselves are evaluated by every rank: the whitening mixes all Q into each P.)"""
        mol = self.mol
        basis = self.auxbasis if isinstance(self.auxbasis, dict) else even_tempered_aux(mol, self.beta)
        if isinstance(self.auxbasis, str) and self.auxbasis:
            basis = self.auxbasis          # a named set, if basis_data has it
        aux = Mole(atom=[(s, xyz) for s, xyz in mol._atom], basis=basis, unit="Bohr", verbose=0, charge=mol.charge, spin=mol.spin)
        aux.build()
        if (aux._bas[:, 1] > LMAX_AUX).any():
            raise NotImplementedError("auxiliary functions beyond g (l > 4) are not supported")
        self.auxmol = aux
        self.naux = aux.nao
        # append the unit function: s primitive, exponent 0, coefficient sqrt(4 pi) (x Y_00 = 1), on atom 0
        env = np.concatenate([aux._env, [0.0, math.sqrt(4.0 * math.pi)]])
        pe = len(aux._env)
        bas = np.vstack([aux._bas, np.array([[0, 0, 1, 1, 0, pe, pe + 1, 0]], dtype=np.int32)])

        class _Packed:      # what Engine needs of a Mole
            pass
        pk = _Packed()
        pk._atm, pk._bas, pk._env, pk.nao = aux._atm, bas, env, aux.nao + 1
        aux_eng = _engine.Engine(pk, device=engine.device)
        n, na = mol.nao, self.naux
        int3c = torch.empty(n, n, na, dtype=torch.float64, device=engine.device)
        int2c = torch.empty(na, na, dtype=torch.float64, device=engine.device)
        engine.df_build(aux_eng, int3c, int2c)
        aux_eng.close()
        self.int2c = int2c
        L = torch.linalg.cholesky(int2c)
        # fitted tensor B[i, P, j] = sum_Q L^-1[P, Q] (Q|ij), stored i-major: the exchange build is then two plain GEMMs
        Linv = torch.linalg.solve_triangular(L, torch.eye(na, dtype=torch.float64, device=engine.device), upper=False)
        from .parallel import split_range
        p0, p1 = split_range(na, rank, nranks) if nranks > 1 else (0, na)
        self.aux_slice = (p0, p1)
        Lr = Linv[p0:p1].contiguous()
        self._B = torch.empty(n, p1 - p0, n, dtype=torch.float64, device=engine.device)                # [i, P (this rank's), j]
        step = max(1, int(1.0e9 / (8.0 * na * n)))
        for i0 in range(0, n, step):      # B[i] = L^-1 (Q|i j)^T as batched GEMMs, a slab of i at a time
            torch.matmul(Lr, int3c[i0:i0 + step].transpose(1, 2), out=self._B[i0:i0 + step])
        del int3c
        self._eng = engine
        return self

    def get_jk(self, dm, with_j=True, with_k=True):
        """J, K of one [N,N] or several [n,N,N] densities from the fitted tensor B[i,P,j]:
             rho_P = sum_ij B[i,P,j] D_ij,  J_ij = sum_P rho_P B[i,P,j]               (two passes over B, bandwidth bound)
             T[i,(P,l)] = sum_j B[i,P,j] D_jl   (GEMM, N P x N x N)
             K_ik = sum_(P,l) T[i,(P,l)] B[k,(P,l)]   (N x N output, contraction length N_aux N: the split-K FP64 MFMA
                                                       kernel `mi_xc_vmat`, rocBLAS has no split-K for this shape)."""
        B = self._B
        n, na, _ = B.shape
        dm = torch.as_tensor(dm, dtype=torch.float64, device=B.device)
        squeeze = dm.dim() == 2
        if squeeze:
            dm = dm.unsqueeze(0)
        J = K = None
        if with_j:
            J = torch.empty_like(dm)
            for s_ in range(dm.shape[0]):
                rho = torch.bmm(B, dm[s_].unsqueeze(2)).sum(dim=0).squeeze(1)      # [P]
                J[s_] = torch.matmul(rho, B)                                      # [i, j] = sum_P rho_P B[i,P,j]
        if with_k:
            K = torch.zeros_like(dm)
            Bf = B.reshape(n, na * n)
            for s_ in range(dm.shape[0]):
                T = torch.matmul(B.reshape(n * na, n), dm[s_]).reshape(n, na * n)  # [i, (P, l)]
                self._eng.xc_vmat(T, Bf, K[s_])                                   # K += T . Bf^T
        if squeeze:
            J = J[0] if J is not None else None
            K = K[0] if K is not None else None
        return J, K
