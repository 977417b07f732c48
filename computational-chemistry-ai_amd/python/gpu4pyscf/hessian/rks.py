from mi355scf.hessian import Hessian  # noqa: F401
