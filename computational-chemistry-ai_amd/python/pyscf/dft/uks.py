"""`pyscf.dft.uks`: `UKS` (reference call sites `templates/calculate_bde.py:140,215`)."""
from mi355scf.uks import UKS  # noqa: F401
