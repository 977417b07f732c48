"""`pyscf.scf`: `RHF`, `UHF`, modules `hf` (alias `rhf`), `uhf`, `rohf` (names used in `isinstance` checks at
`templates/optimize_geometry.py:117` and `templates/calculate_reaction_energy.py:167-169`)."""
from . import hf, uhf, rohf  # noqa: F401
rhf = hf
RHF = hf.RHF
HF = hf.RHF
UHF = uhf.UHF
ROHF = rohf.ROHF
