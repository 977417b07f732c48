"""`gpu4pyscf.dft.uks`: `UKS` (reference call sites `templates/calculate_bde.py:128,197`)."""
from mi355scf.uks import UKS  # noqa: F401
