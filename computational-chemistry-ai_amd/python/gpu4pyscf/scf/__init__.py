from . import hf, uhf  # noqa: F401
RHF = hf.RHF
UHF = uhf.UHF
