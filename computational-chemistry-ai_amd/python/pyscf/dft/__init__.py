"""`pyscf.dft`: `RKS` (reference call sites `templates/calculate_energy.py:163,202`)."""
from . import rks  # noqa: F401
RKS = rks.RKS
KS = rks.RKS
