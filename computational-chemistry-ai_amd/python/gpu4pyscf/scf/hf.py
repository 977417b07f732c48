from mi355scf.scf import SCF, RHF  # noqa: F401
