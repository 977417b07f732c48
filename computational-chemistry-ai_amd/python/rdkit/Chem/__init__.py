import numpy as np

from mi355scf import smiles_fixtures as _fx

_MASS = {"H": 1.008, "C": 12.011, "N": 14.007, "O": 15.999, "F": 18.998}


class _Pos:
    def __init__(self, v):
        self.x, self.y, self.z = (float(t) for t in v)


class _Conf:
    def __init__(self, xyz):
        self._xyz = xyz

    def GetAtomPosition(self, i):
        return _Pos(self._xyz[i])

    def GetPositions(self):
        return np.array(self._xyz)


class _Atom:
    def __init__(self, idx, sym):
        self._i, self._s = idx, sym

    def GetSymbol(self):
        return self._s

    def GetIdx(self):
        return self._i

    def GetAtomicNum(self):
        return {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}[self._s]


class Mol:
    def __init__(self, smiles, sym, xyz, with_h=False):
        self._smiles, self._sym, self._xyz, self._with_h = smiles, sym, xyz, with_h

    def _visible(self):
        return [i for i, s in enumerate(self._sym) if self._with_h or s != "H"]

    def GetAtoms(self):
        return [_Atom(k, self._sym[i]) for k, i in enumerate(self._visible())]

    def GetNumAtoms(self):
        return len(self._visible())

    def GetConformer(self, i=0):
        return _Conf([self._xyz[k] for k in self._visible()])


def MolFromSmiles(smiles):
    got = _fx.lookup(smiles)
    if got is None:
        raise NotImplementedError(
            f"rdkit stand-in: no fixture geometry for SMILES '{smiles}'. Supported: {sorted(_fx.TABLE)} "
            "(real RDKit is not available in this environment)")
    sym, xyz = got
    return Mol(smiles, sym, xyz, with_h=False)


def AddHs(mol):
    return Mol(mol._smiles, mol._sym, mol._xyz, with_h=True)


def MolToSmiles(mol):
    return mol._smiles


class rdMolDescriptors:
    @staticmethod
    def CalcMolFormula(mol):
        from collections import Counter
        c = Counter(mol._sym)
        order = [s for s in ("C", "H") if s in c] + sorted(s for s in c if s not in ("C", "H"))
        return "".join(f"{s}{c[s] if c[s] > 1 else ''}" for s in order)


from . import AllChem, Descriptors  # noqa: E402,F401
