"""Stand-in: the fixture geometry already IS the embedded/optimised conformer."""


def EmbedMolecule(mol, randomSeed=-1, **kw):
    return 0


def MMFFOptimizeMolecule(mol, maxIters=200, **kw):
    return 0


def UFFOptimizeMolecule(mol, maxIters=200, **kw):
    return 0
