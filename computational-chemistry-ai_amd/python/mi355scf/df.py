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
import time

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


def tri_inv_lower(L, base=2048):
    """L^-1 of a lower-triangular matrix by recursive 2 x 2 blocking: [[A, 0], [C, D]]^-1 = [[A^-1, 0], [-D^-1 C A^-1, D^-1]]
    (GEMMs above `base` rows; rocBLAS' trsm fails to allocate its workspace for N ~ 6000 right-hand sides)."""
    n = L.shape[0]
    if n <= base:
        return torch.linalg.solve_triangular(L, torch.eye(n, dtype=L.dtype, device=L.device), upper=False)
    h = (n // 2 + 63) // 64 * 64
    if h >= n:
        h = n // 2
    out = torch.zeros_like(L)
    Ai = tri_inv_lower(L[:h, :h].contiguous(), base)
    Di = tri_inv_lower(L[h:, h:].contiguous(), base)
    out[:h, :h] = Ai
    out[h:, h:] = Di
    out[h:, :h] = -(Di @ (L[h:, :h] @ Ai))
    return out


def pivoted_cholesky(dm, rank, rtol=1e-10):
    """L [N, rank] with dm ~ L L^T by `rank` steps of diagonally pivoted Cholesky (no host synchronisation inside the loop), or
    None when the remainder is not negligible (dm is not a positive semi-definite matrix of that rank: a random test density,
    a difference density).  An SCF density is 2 C_occ C_occ^T: rank = number of occupied orbitals."""
    n = dm.shape[0]
    rank = min(int(rank), n)
    d = dm.diagonal().clone()
    floor = (rtol * d.abs().max()).clamp_min(1e-300)       # pivots below it end the factorisation (columns of zeros: beta spin of
    L = torch.zeros(n, rank, dtype=dm.dtype, device=dm.device)   # an open shell has fewer occupied orbitals than the hint)
    zero = torch.zeros((), dtype=dm.dtype, device=dm.device)
    for k in range(rank):
        p = torch.argmax(d).reshape(1)
        piv = d.index_select(0, p)
        col = dm.index_select(1, p).squeeze(1)
        if k:
            col = col - L[:, :k] @ L.index_select(0, p)[0, :k]
        col = torch.where(piv > floor, col * torch.rsqrt(piv.clamp_min(1e-300)), zero)
        L[:, k] = col
        d = d - col * col
    if not bool(((dm - L @ L.t()).abs().max() <= floor).item()):          # the one synchronisation
        return None
    return L


def _psd_factor(ds, rtol=1e-10):
    """F [N, r] with Ds = F F^T for a (numerically) positive semi-definite spin density of low rank, or None when the
    discarded part of the spectrum is not negligible (the caller then takes the dense route)."""
    lam, V = torch.linalg.eigh(ds)
    top = float(lam[-1])
    if top <= 0.0:
        return None
    keep = lam > rtol * top
    if float(lam[~keep].abs().max()) > 1e-9 * top if bool((~keep).any()) else False:
        return None
    return (V[:, keep] * lam[keep].sqrt()).contiguous()


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
        self._aux_packed = pk
        aux_eng = _engine.Engine(pk, device=engine.device)
        n, na = mol.nao, self.naux
        # zeros: shell pairs whose primitive products all vanish (exp(-80)) are skipped by mi_df_build, their blocks stay unwritten
        int3c = torch.zeros(n, n, na, dtype=torch.float64, device=engine.device)
        int2c = torch.zeros(na, na, dtype=torch.float64, device=engine.device)
        engine.df_build(aux_eng, int3c, int2c)
        aux_eng.close()
        self.int2c = int2c
        self._nranks_built = nranks
        L = torch.linalg.cholesky(int2c)
        # fitted tensor B[i, P, j] = sum_Q L^-1[P, Q] (Q|ij), stored i-major: the exchange build is then two plain GEMMs
        Linv = tri_inv_lower(L)
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
            hint = getattr(self, "rank_hint", None)
            for s_ in range(dm.shape[0]):
                # SCF densities are low rank (occupied orbitals): K = sum_P (B^P L)(B^P L)^T costs 4 N^2 N_aux n_occ flops
                # instead of 4 N^3 N_aux; `rank_hint` is set by the SCF driver, any other density takes the dense route
                L = pivoted_cholesky(dm[s_], hint) if hint and 2 * hint <= n else None
                if L is not None:
                    Y = torch.matmul(B.reshape(n * na, n), L).reshape(n, na * L.shape[1])
                    self._eng.xc_vmat(Y, Y, K[s_])                                # K += Y . Y^T
                    self.k_path = "low rank"
                    continue
                self.k_path = "dense"
                T = torch.matmul(B.reshape(n * na, n), dm[s_]).reshape(n, na * n)  # [i, (P, l)]
                self._eng.xc_vmat(T, Bf, K[s_])                                   # K += T . Bf^T
        if squeeze:
            J = J[0] if J is not None else None
            K = K[0] if K is not None else None
        return J, K

    def grad_jk(self, dms, hyb=1.0, rank=0, nranks=1, factorize=True):
        """Nuclear gradient [natm, 3] (device tensor) of the fitted two-electron energy
             E2 = 1/2 g^T V^-1 g - hyb/2 sum_s sum_PQ (ik|P) V^-1_PQ (Q|jl) Ds_ij Ds_kl,     g_P = sum_ij (ij|P) D_ij,
        for the spin densities `dms` = [Da, Db] (closed shell: [D/2, D/2], or one [N,N] total density):
             dE2 = sum_{ij,P} Z3[i,j,P] d(ij|P) + sum_PQ Z2[P,Q] d(P|Q),
             Z3 = c_P D_ij - hyb sum_s Gs^P_ij,   Z2 = -1/2 c c^T + hyb/2 sum_s sum_ij C^P_ij Gs^Q_ij,
             c = V^-1 g,  C^P = sum_Q V^-1_PQ (Q|ij) (fit coefficients),  Gs^P = Ds C^P Ds
        (the idea of pyscf.df.grad.rhf.get_jk [MEM]).  With the whitened tensor B = L^-1 (Q|ij) (V = L L^T): C = L^-T B,
        Gs = L^-T Ws, Ws^P = Ds B^P Ds, so everything is GEMMs on B plus two triangular back-transformations; the derivative
        integrals themselves are contracted on the fly by `mi_df_grad` (no derivative tensor is stored).  Needs the whole
        auxiliary index on this rank (a sharded tensor is rebuilt whole by the caller); `rank`/`nranks` deal the
        derivative-integral batches, the caller sums the partial gradients."""
        if self._B is None or getattr(self, "_nranks_built", 1) != 1:
            raise RuntimeError("DF.grad_jk needs the unsharded fitted tensor: call build(engine) first")
        eng, B = self._eng, self._B
        n, na, _ = B.shape
        dev = B.device
        t0 = time.time()
        if torch.is_tensor(dms) and dms.dim() == 2:
            dms = [0.5 * dms, 0.5 * dms]
        dms = [torch.as_tensor(d, dtype=torch.float64, device=dev).contiguous() for d in dms]
        closed = len(dms) == 2 and (dms[0] is dms[1] or torch.equal(dms[0], dms[1]))
        D = dms[0] + dms[1] if len(dms) == 2 else dms[0]
        L = torch.linalg.cholesky(self.int2c)
        Linv = tri_inv_lower(L)                                                                                # [P', Q]
        rho = torch.einsum("ipj,ij->p", B, D)                                                                  # whitened g
        c = Linv.t() @ rho                                                                                     # V^-1 g
        # three-index density in the layout the kernel reads, Z3[i, j, Q]
        Z3 = torch.empty(n, n, na, dtype=torch.float64, device=dev)
        Z2 = -0.5 * torch.outer(c, c)
        if hyb != 0.0:
            factors = [_psd_factor(ds) for ds in (dms[:1] if closed else dms)] if factorize else None
            if factors is not None and all(f is not None for f in factors):
                # Ds = Fs Fs^T (rank = occupied orbitals): everything through X_s[o, P, o'] = Fs^T B^P Fs, 2 N^2 N_aux n_occ flops
                # where the dense route below costs 2 N^2 N_aux^2
                Z3.zero_()
                for F in factors:
                    no = F.shape[1]
                    Y = torch.matmul(B.reshape(n * na, n), F).reshape(n, na * no)              # Y[i, (P', o')]
                    X = torch.matmul(F.t(), Y).reshape(no, na, no)                             # X[o, P', o']
                    del Y
                    Xt = torch.einsum("pq,opr->oqr", Linv, X).contiguous()                     # back-transformed over the auxiliary index
                    Xm = Xt.permute(1, 0, 2).reshape(na, no * no)
                    Z2 += (1.0 if closed else 0.5) * hyb * (Xm @ Xm.t())
                    U = torch.matmul(F, Xt.reshape(no, na * no)).reshape(n, na, no)             # U[i, Q, o']
                    step = max(1, int(2.0e9 / (8.0 * na * n)))
                    for i0 in range(0, n, step):      # Z3[i, k, Q] -= hyb sum_o' F[k, o'] U[i, Q, o']
                        Z3[i0:i0 + step].add_(torch.matmul(F, U[i0:i0 + step].transpose(1, 2)), alpha=-(2.0 if closed else 1.0) * hyb)
                    del U, X, Xt, Xm
            else:
                BW = torch.zeros(na, na, dtype=torch.float64, device=dev)
                W = torch.zeros(n, na, n, dtype=torch.float64, device=dev)
                for ds in (dms[:1] if closed else dms):
                    T = torch.matmul(B.reshape(n * na, n), ds).reshape(n, na * n)          # T[j, (P, k)] = sum_l B[j, P, l] Ds[l, k]
                    W += torch.matmul(ds, T).reshape(n, na, n)                              # Ws[i, P, k] = sum_j Ds[i, j] T[j, P, k]
                    del T
                if closed:
                    W *= 2.0
                # sum_ik B[i, P, k] W[i, R, k]  (whitened indices), then back-transformed on both sides
                step = max(1, int(2.0e9 / (8.0 * na * n)))
                for i0 in range(0, n, step):
                    BW += torch.einsum("ipk,irk->pr", B[i0:i0 + step], W[i0:i0 + step])
                Z2 += 0.5 * hyb * (Linv.t() @ BW @ Linv)
                for i0 in range(0, n, step):      # Z3[i, k, Q] = -hyb sum_P' W[i, P', k] Linv[P', Q]
                    torch.matmul(W[i0:i0 + step].transpose(1, 2), Linv, out=Z3[i0:i0 + step])
                Z3 *= -hyb
                del W
            Z3 += D.unsqueeze(2) * c.view(1, 1, na)
        else:
            Z3.copy_(D.unsqueeze(2) * c.view(1, 1, na))
        Z2 = (0.5 * (Z2 + Z2.t())).contiguous()      # Z3 is symmetric in (i, j) by construction (B^P, Ds symmetric)
        aux_eng = _engine.Engine(self._aux_packed, device=dev)
        g = torch.zeros(len(eng._atm), 3, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        t1 = time.time()
        try:
            eng.df_grad(aux_eng, Z3, Z2, g, rank=rank, nranks=nranks)
        finally:
            aux_eng.close()
        self.grad_timing = {"densities": t1 - t0, "derivative_integrals": time.time() - t1}
        return g
