"""Name-only module: `templates/optimize_geometry.py:15` imports `hessian`; the `--freq` branch
(`:112-154`) is out of scope."""
def _unsupported(*a, **k):
    raise NotImplementedError("analytic Hessians are outside the MI355X Fock-build hot path")
RHF = RKS = _unsupported
