"""`pyscf.mp`: `MP2` (`templates/calculate_interaction.py:19,118`).  Dense-tensor MP2 for small molecules on the engine."""
from mi355scf.mp2 import MP2, RMP2, UMP2  # noqa: F401
