from mi355scf.mole import Mole, M, BOHR  # noqa: F401
