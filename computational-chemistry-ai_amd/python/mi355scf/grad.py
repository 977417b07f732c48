"""Analytic nuclear gradients (row a15) -- not built yet."""


class Gradients:
    def __init__(self, mf):
        raise NotImplementedError("analytic gradients are not built yet (SURVEY.md section 8f rank 1)")
