"""`from pyscf.geomopt.geometric_solver import optimize` (`templates/optimize_geometry.py:16,99`)."""
from mi355scf.geomopt import optimize, kernel  # noqa: F401
