"""Drop-in `pyscf` import surface for the two in-scope reference templates
(`templates/calculate_energy.py:15`, `templates/optimize_geometry.py:15-16`).
Only the symbols those scripts touch exist; everything routes to the MI355X engine (`mi355scf`)."""
__version__ = "2.8.0+mi355x"
from . import lib, gto, scf, dft, hessian, geomopt, mp  # noqa: F401,E402

M = gto.M
