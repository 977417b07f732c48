_MASS = {"H": 1.008, "C": 12.011, "N": 14.007, "O": 15.999, "F": 18.998}


def MolWt(mol):
    return sum(_MASS[s] for s in mol._sym)
