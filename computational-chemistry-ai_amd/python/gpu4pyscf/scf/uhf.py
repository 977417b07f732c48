"""`gpu4pyscf.scf.uhf`: `UHF` (reference call sites `templates/calculate_bde.py:126,192`)."""
from mi355scf.uhf import UHF  # noqa: F401
