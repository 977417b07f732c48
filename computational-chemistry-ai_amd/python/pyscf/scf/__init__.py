"""`pyscf.scf`: `RHF`, modules `hf`/`uhf` (used in `isinstance` at `templates/optimize_geometry.py:117`)."""
from . import hf, uhf  # noqa: F401
RHF = hf.RHF
HF = hf.RHF
UHF = uhf.UHF
