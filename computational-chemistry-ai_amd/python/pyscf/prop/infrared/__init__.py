from mi355scf.infrared import Infrared, RHF, RKS, UHF, UKS  # noqa: F401
from . import rhf, rks  # noqa: F401,E402
