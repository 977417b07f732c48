"""`optimize(mf, maxsteps=...)` (row f-1) -- not built yet."""


def optimize(mf, maxsteps=100, **kw):
    raise NotImplementedError("geometry optimisation is not built yet (SURVEY.md section 8f rank 1)")


kernel = optimize
