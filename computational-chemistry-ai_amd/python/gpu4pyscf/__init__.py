"""Drop-in `gpu4pyscf` import surface (`templates/calculate_energy.py:46-48`,
`templates/optimize_geometry.py:66-73`, `README.md:182,197`): the GPU-named classes of the same engine."""
__version__ = "1.0+mi355x"
from . import scf, dft, hessian  # noqa: F401,E402
