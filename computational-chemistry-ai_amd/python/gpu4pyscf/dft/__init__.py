from . import rks, uks  # noqa: F401
RKS = rks.RKS
UKS = uks.UKS
