"""`pyscf.hessian`: `RHF`, `RKS`, `UHF`, `UKS` factories, modules `rhf`/`rks`/`uhf`/`uks` with `Hessian`, and `thermo`
(call sites `templates/optimize_geometry.py:15,117-147`, `templates/opt-freq.py:15,387-417,458,499`).  Semi-numerical:
finite differences of the analytic HIP gradient (`mi355scf/hessian.py`)."""
from . import rhf, rks, uhf, uks, thermo  # noqa: F401
RHF = rhf.Hessian
RKS = rks.Hessian
UHF = uhf.Hessian
UKS = uks.Hessian
