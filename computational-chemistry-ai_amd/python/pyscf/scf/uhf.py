class UHF:
    """Open-shell SCF is outside the MI355X hot path; the class exists
    because `templates/optimize_geometry.py:117` names it in an isinstance check."""
    def __init__(self, *a, **k):
        raise NotImplementedError("UHF is not implemented in the MI355X engine (closed-shell RHF/RKS only)")
