"""`pyscf.dft`: `RKS` (reference call sites `templates/calculate_energy.py:163,202`), `UKS` (`templates/calculate_bde.py:140`)."""
from . import rks, uks  # noqa: F401
RKS = rks.RKS
KS = rks.RKS
UKS = uks.UKS
