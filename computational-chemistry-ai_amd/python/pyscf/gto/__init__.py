"""`pyscf.gto`: `Mole`, `M` (reference call sites `templates/calculate_energy.py:89-101`)."""
from mi355scf.mole import Mole, M, BOHR  # noqa: F401
from . import mole  # noqa: F401
