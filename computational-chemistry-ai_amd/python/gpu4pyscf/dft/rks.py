from mi355scf.dft import RKS  # noqa: F401
