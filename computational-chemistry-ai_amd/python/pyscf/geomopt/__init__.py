from . import geometric_solver  # noqa: F401
