"""`pyscf.prop`: only `infrared` (`templates/calculate_ir_spectrum.py:17`, `templates/opt-freq.py:17`).  NMR and the other
property modules are not provided."""
from . import infrared  # noqa: F401
