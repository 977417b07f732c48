"""`gpu4pyscf.mp` (name imported at `templates/calculate_energy.py:138`, inside its MP2 branch): same dense MP2 as `pyscf.mp`."""
from mi355scf.mp2 import MP2, RMP2, UMP2  # noqa: F401
