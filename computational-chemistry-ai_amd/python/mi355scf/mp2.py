"""Second-order Moller-Plesset correlation energy behind `pyscf.mp.MP2(mf).kernel()` (`templates/calculate_interaction.py:
19,116-120`; `mp.MP2` dispatches on the reference: RHF -> RMP2, UHF -> UMP2).

Small molecules only: the resident ERI tiles are unpacked to a dense (ij|kl) tensor on the device (`mi_eri_unpack`,
8 N^4 bytes) and transformed to (ia|jb) with torch contractions (rocBLAS).  No frozen core (PySCF default), no density
fitting.  Not a hot-path component (SURVEY.md section 8 keeps post-SCF methods out of scope); it exists so that the
interaction-energy template imports and its `--method MP2` branch works for the dimers it is meant for.
"""
import numpy as np
import torch

MAX_NAO = 220   # 8 * 220^4 = 18.7 GB dense tensor


class MP2:
    def __init__(self, mf, frozen=None):
        if frozen:
            raise NotImplementedError("frozen-core MP2 is not implemented")
        self._scf = mf
        self.mol = mf.mol
        self.verbose = mf.verbose
        self.e_corr = None
        self.t2 = None

    @property
    def e_tot(self):
        return self._scf.e_tot + self.e_corr

    def _ovov(self, eri, co, cv):
        # (ia|jb) = sum_pqrs C_pi C_qa (pq|rs) C_rj C_sb, one index at a time
        t = torch.einsum("pqrs,pi->iqrs", eri, co)
        t = torch.einsum("iqrs,qa->iars", t, cv)
        return t

    def kernel(self, mo_energy=None, mo_coeff=None, **kw):
        mf = self._scf
        if mf.mo_coeff is None:
            mf.kernel()
        eng = mf.engine
        n = eng.nao
        if n > MAX_NAO:
            raise NotImplementedError(f"MP2 needs the dense ERI tensor: N_ao = {n} > {MAX_NAO}")
        if mf._nranks > 1:
            raise NotImplementedError("MP2 is single-GPU (unsharded tile store)")
        dev = eng.device
        eri = eng.eri_dense()
        mo_c = np.asarray(mf.mo_coeff if mo_coeff is None else mo_coeff)
        mo_e = np.asarray(mf.mo_energy if mo_energy is None else mo_energy)
        occ = np.asarray(mf.mo_occ)
        T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        if mo_c.ndim == 2:      # restricted
            o, v = occ > 0, occ == 0
            co, cv, eo, ev = T(mo_c[:, o]), T(mo_c[:, v]), T(mo_e[o]), T(mo_e[v])
            iars = self._ovov(eri, co, cv)
            ovov = torch.einsum("iars,rj,sb->iajb", iars, co, cv)
            d = eo[:, None, None, None] - ev[None, :, None, None] + eo[None, None, :, None] - ev[None, None, None, :]
            t2 = ovov / d
            e = float(torch.sum(t2 * (2.0 * ovov - ovov.transpose(1, 3))))
        else:                   # unrestricted: aa, bb (antisymmetrised) and ab
            parts = []
            for s_ in range(2):
                o, v = occ[s_] > 0, occ[s_] == 0
                parts.append((T(mo_c[s_][:, o]), T(mo_c[s_][:, v]), T(mo_e[s_][o]), T(mo_e[s_][v])))
            e = 0.0
            for s_ in range(2):
                co, cv, eo, ev = parts[s_]
                if co.shape[1] == 0:
                    continue
                ovov = torch.einsum("iars,rj,sb->iajb", self._ovov(eri, co, cv), co, cv)
                d = eo[:, None, None, None] - ev[None, :, None, None] + eo[None, None, :, None] - ev[None, None, None, :]
                anti = ovov - ovov.transpose(1, 3)
                e += 0.25 * float(torch.sum(anti * anti / d))
            (coa, cva, eoa, eva), (cob, cvb, eob, evb) = parts
            if cob.shape[1] > 0:
                ovov = torch.einsum("iars,rj,sb->iajb", self._ovov(eri, coa, cva), cob, cvb)
                d = eoa[:, None, None, None] - eva[None, :, None, None] + eob[None, None, :, None] - evb[None, None, None, :]
                e += float(torch.sum(ovov * ovov / d))
            t2 = None
        del eri
        self.e_corr = e
        self.t2 = t2
        mf._log(3, f"E(MP2) = {mf.e_tot + e:.12g}  E_corr = {e:.12g}")
        return self.e_corr, self.t2


RMP2 = UMP2 = MP2
