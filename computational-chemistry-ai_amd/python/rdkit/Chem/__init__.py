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
    def __init__(self, idx, sym, props=None):
        self._i, self._s, self._props = idx, sym, dict(props or {})

    def HasProp(self, key):
        return key in self._props

    def GetIntProp(self, key):
        return int(self._props[key])

    def GetSymbol(self):
        return self._s

    def GetIdx(self):
        return self._i

    def GetAtomicNum(self):
        return {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}[self._s]


class _Bond:
    def __init__(self, i, j, kind):
        self._i, self._j, self._kind = i, j, kind

    def GetBeginAtomIdx(self):
        return self._i

    def GetEndAtomIdx(self):
        return self._j

    def GetBondType(self):
        return self._kind   # str(...) gives 'SINGLE' / 'DOUBLE' / 'TRIPLE' / 'AROMATIC' like RDKit's enum names


_Z = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}


def _perceive_bonds(sym, xyz):
    """Bond graph of the fixture geometry (covalent-radius criterion of the optimiser's internal coordinates) with a
    distance-based guess of the bond order -- enough for `templates/calculate_bde.py:256-263` to enumerate bonds."""
    from mi355scf.internals import bond_graph, BOHR, _COV
    x = np.asarray(xyz, dtype=float) / BOHR
    z = [_Z[s] for s in sym]
    out = []
    for i, j in bond_graph(z, x):
        d = np.linalg.norm(x[i] - x[j]) * BOHR
        ratio = d / (_COV[z[i]] + _COV[z[j]])
        kind = "SINGLE"
        if "H" not in (sym[i], sym[j]):
            if sym[i] == sym[j] == "C" and 1.36 < d < 1.43:
                kind = "AROMATIC"
            elif ratio < 0.82:
                kind = "TRIPLE"
            elif ratio < 0.90:
                kind = "DOUBLE"
        out.append((min(i, j), max(i, j), kind))
    return sorted(out)


class Mol:
    def __init__(self, smiles, sym, xyz, with_h=False, bonds=None, origin=None):
        self._smiles, self._sym, self._xyz, self._with_h = smiles, sym, xyz, with_h
        self._bonds_all = bonds          # bonds over ALL atoms of the fixture (None: perceive on demand)
        self._origin = origin            # fragment: index of every atom in the parent molecule

    def _visible(self):
        return [i for i, s in enumerate(self._sym) if self._with_h or s != "H"]

    def _bond_list(self):
        if self._bonds_all is None:
            self._bonds_all = _perceive_bonds(self._sym, self._xyz)
        vis = {i: k for k, i in enumerate(self._visible())}
        return [(vis[i], vis[j], t) for i, j, t in self._bonds_all if i in vis and j in vis]

    def GetAtoms(self):
        props = (lambda i: {"_FromAtomIdx": self._origin[i]}) if self._origin is not None else (lambda i: None)
        return [_Atom(k, self._sym[i], props(i)) for k, i in enumerate(self._visible())]

    def GetAtomWithIdx(self, k):
        return self.GetAtoms()[k]

    def GetBonds(self):
        return [_Bond(i, j, t) for i, j, t in self._bond_list()]

    def GetNumBonds(self):
        return len(self._bond_list())

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
    return Mol(mol._smiles, mol._sym, mol._xyz, with_h=True, bonds=mol._bonds_all, origin=mol._origin)


class EditableMol:
    """`Chem.EditableMol(mol).RemoveBond(i, j).GetMol()` (`templates/calculate_bde.py:291-293`)."""

    def __init__(self, mol):
        self._mol = mol
        self._bonds = list(mol._bond_list())
        if not mol._with_h and "H" in mol._sym:
            raise NotImplementedError("rdkit stand-in: EditableMol needs explicit hydrogens (call Chem.AddHs first)")

    def RemoveBond(self, i, j):
        a, b = min(i, j), max(i, j)
        self._bonds = [t for t in self._bonds if (t[0], t[1]) != (a, b)]

    def GetMol(self):
        m = self._mol
        return Mol(m._smiles, m._sym, m._xyz, with_h=True, bonds=list(self._bonds), origin=m._origin)


def GetMolFrags(mol, asMols=False, sanitizeFrags=True):
    """Connected components of the bond graph; `asMols=True` returns sub-molecules whose atoms carry `_FromAtomIdx`
    (the property `templates/calculate_bde.py:309` looks for)."""
    n = mol.GetNumAtoms()
    vis = mol._visible()
    adj = [[] for _ in range(n)]
    for i, j, _t in mol._bond_list():
        adj[i].append(j); adj[j].append(i)
    seen, comps = [False] * n, []
    for s in range(n):
        if seen[s]:
            continue
        stack, comp = [s], []
        seen[s] = True
        while stack:
            a = stack.pop()
            comp.append(a)
            for b in adj[a]:
                if not seen[b]:
                    seen[b] = True
                    stack.append(b)
        comps.append(tuple(sorted(comp)))
    if not asMols:
        return tuple(comps)
    out = []
    for comp in comps:
        idx = {a: k for k, a in enumerate(comp)}
        bonds = [(idx[i], idx[j], t) for i, j, t in mol._bond_list() if i in idx and j in idx]
        out.append(Mol(mol._smiles, [mol._sym[vis[a]] for a in comp], [mol._xyz[vis[a]] for a in comp], with_h=True,
                       bonds=bonds, origin=[a for a in comp]))
    return tuple(out)


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
