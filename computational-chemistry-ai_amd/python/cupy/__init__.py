"""Minimal `cupy` STAND-IN: `templates/calculate_energy.py:44-58` only needs the import to succeed and
prints `cupy.__version__` and `cupy.cuda.runtime.runtimeGetVersion()`.  Arrays stay PyTorch-ROCm tensors
inside the engine; results cross the boundary as NumPy (`mo_energy.get()` is therefore never needed)."""
__version__ = "0.0-mi355x-standin"
from . import cuda  # noqa: F401,E402


def asnumpy(a):
    import numpy
    return numpy.asarray(a.cpu() if hasattr(a, "cpu") else a)
