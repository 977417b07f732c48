from . import rks  # noqa: F401
RKS = rks.RKS
