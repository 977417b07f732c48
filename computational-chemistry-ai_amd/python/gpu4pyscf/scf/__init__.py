from . import hf  # noqa: F401
RHF = hf.RHF
