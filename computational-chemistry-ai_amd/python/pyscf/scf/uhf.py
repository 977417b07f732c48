"""`pyscf.scf.uhf`: `UHF` (reference call sites `templates/calculate_bde.py:138,210`; named in the isinstance
check at `templates/optimize_geometry.py:117`)."""
from mi355scf.uhf import UHF  # noqa: F401
