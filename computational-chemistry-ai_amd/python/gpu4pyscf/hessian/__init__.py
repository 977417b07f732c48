"""`gpu4pyscf.hessian` (`templates/opt-freq.py:392-394`: `gpu_hessian.rks.Hessian(mf_opt).kernel()`)."""
from . import rhf, rks, uhf, uks  # noqa: F401
RHF = rhf.Hessian
RKS = rks.Hessian
UHF = uhf.Hessian
UKS = uks.Hessian
