"""RKS placeholder; replaced by the Becke-grid XC implementation (rows a7-a9)."""
from .scf import RHF


class RKS(RHF):
    xc = "LDA,VWN"

    def kernel(self, dm0=None, **kw):
        raise NotImplementedError("XC quadrature kernels not built yet")
