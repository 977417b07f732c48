"""`Mole`: the molecule/basis container behind `pyscf.gto.Mole`.

Mirrors the attribute surface the reference templates touch
(`templates/calculate_energy.py:89-101,317-319`, `templates/optimize_geometry.py:48-54,222`,
`README.md:186-194`): `atom`, `basis`, `charge`, `spin`, `unit`, `verbose`, `output`, `stdout`,
`build()`, `natm`, `nelectron`, `nao`, `atom_coords()` (Bohr), `atom_symbol()`, `set_geom_()`,
`copy()`.  Packs libcint-convention `_atm/_bas/_env` arrays, which is what both the C oracle and the
HIP library consume through the C ABI (`include/mi355scf.h`).
"""
import copy as _copy
import math
import re
import sys

import numpy as np

from . import basis_data

BOHR = 0.52917721092  # Angstrom per Bohr; the value PySCF 2.x uses (CODATA 2010) [MEM]

ELEMENTS = [
    "X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P", "S", "Cl", "Ar",
    "K", "Ca", "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr",
]
_Z = {s.upper(): i for i, s in enumerate(ELEMENTS)}

# libcint slot names
CHARGE_OF, PTR_COORD, NUC_MOD_OF, PTR_ZETA, PTR_FRAC_CHARGE, _RES_ATM = range(6)
ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, KAPPA_OF, PTR_EXP, PTR_COEFF, _RES_BAS = range(8)
PTR_ENV_START = 20


def split_ghost(symbol):
    """PySCF ghost-atom spellings ('Ghost:O', 'GHOST-O', 'ghost_O', 'X-O', 'X:O', also with a trailing label such as
    'Ghost:O1'; `templates/calculate_interaction.py:136,142` writes 'Ghost:' + symbol) -> (is_ghost, element symbol)."""
    m = re.match(r"^(ghost|x)[\s:_\-]+(.+)$", symbol.strip(), flags=re.IGNORECASE)
    if m:
        return True, m.group(2)
    return False, symbol


def charge_of(symbol):
    symbol = split_ghost(symbol)[1]
    s = re.sub(r"[^A-Za-z]", "", symbol).upper()
    if s not in _Z:
        raise ValueError(f"unknown element symbol '{symbol}'")
    return _Z[s]


def gto_norm(l, a):
    """Radial normalisation of r^l exp(-a r^2): 1/sqrt(int r^(2l+2) exp(-2 a r^2) dr)."""
    return 1.0 / math.sqrt(gaussian_int(2 * l + 2, 2.0 * a))


def gaussian_int(n, a):
    """int_0^inf r^n exp(-a r^2) dr = Gamma((n+1)/2) / (2 a^((n+1)/2))."""
    n1 = (n + 1) * 0.5
    return math.gamma(n1) / (2.0 * a ** n1)


def parse_atom(atom):
    """Accepts the 'El x y z; El x y z' / multi-line string form the templates write
    (`calculate_energy.py:85-90`, `README.md:187-192`) or a list of (symbol, (x, y, z))."""
    out = []
    if isinstance(atom, str):
        for line in re.split(r"[;\n]", atom):
            line = line.strip().replace(",", " ")
            if not line:
                continue
            tok = line.split()
            if len(tok) < 4:
                raise ValueError(f"cannot parse atom line '{line}'")
            sym = tok[0]
            if sym.isdigit():
                sym = ELEMENTS[int(sym)]
            out.append((sym, tuple(float(x) for x in tok[1:4])))
    else:
        for item in atom:
            sym, xyz = item[0], item[1] if len(item) == 2 else item[1:4]
            if isinstance(sym, (int, np.integer)):
                sym = ELEMENTS[int(sym)]
            out.append((sym, tuple(float(x) for x in xyz)))
    if not out:
        raise ValueError("empty molecule")
    return out


class Mole:
    verbose = 3
    output = None
    charge = 0
    spin = 0
    unit = "Angstrom"
    basis = "sto-3g"
    cart = False
    symmetry = False
    max_memory = 4000

    def __init__(self, **kw):
        self.atom = []
        self.stdout = sys.stdout
        self._built = False
        self._atom = []
        self._atm = self._bas = self._env = None
        for k, v in kw.items():
            setattr(self, k, v)

    # --- construction -------------------------------------------------------------------------
    def build(self, dump_input=True, parse_arg=True, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        if self.cart:
            raise NotImplementedError("cartesian AOs are not supported (PySCF default cart=False is)")
        if self.output is not None and isinstance(self.output, str):
            self.stdout = open(self.output, "w")
        atoms = parse_atom(self.atom)
        unit = str(self.unit).upper()
        scale = 1.0 if unit.startswith(("B", "AU")) else 1.0 / BOHR
        self._atom = [(s, tuple(x * scale for x in xyz)) for s, xyz in atoms]
        self._pack()
        ne = self.nelectron
        if (ne - self.spin) % 2 != 0 or self.spin < 0 or self.spin > ne:
            raise RuntimeError(f"Electron number {ne} and spin {self.spin} are not consistent")
        self._built = True
        return self

    def _basis_for(self, sym):
        b = self.basis
        pure = re.sub(r"[^A-Za-z]", "", split_ghost(sym)[1])
        pure = pure[0].upper() + pure[1:].lower()
        if isinstance(b, dict):
            b = b.get(sym, b.get(pure, b.get("default")))
            if b is None:
                raise KeyError(f"no basis for {sym}")
        if isinstance(b, str):
            if b.lower().startswith("synthetic:"):
                return basis_data.synthetic_like(b.split(":", 1)[1], pure)
            return basis_data.load(b, pure)
        # explicit PySCF-format list
        shells = []
        for entry in b:
            l, rows = entry[0], entry[1:]
            exps = [r[0] for r in rows]
            for ic in range(len(rows[0]) - 1):
                shells.append((l, exps, [r[1 + ic] for r in rows]))
        shells.sort(key=lambda s: s[0])
        return shells

    def _pack(self):
        natm = len(self._atom)
        env = [0.0] * PTR_ENV_START
        atm = np.zeros((natm, 6), dtype=np.int32)
        bas = []
        for ia, (sym, xyz) in enumerate(self._atom):
            atm[ia, CHARGE_OF] = 0 if split_ghost(sym)[0] else charge_of(sym)
            atm[ia, PTR_COORD] = len(env)
            atm[ia, NUC_MOD_OF] = 1
            env.extend(xyz)
            env.append(0.0)  # zeta slot
        cache = {}
        for ia, (sym, _xyz) in enumerate(self._atom):
            if sym not in cache:
                packed = []
                for (l, exps, coefs) in self._basis_for(sym):
                    pe = len(env)
                    env.extend(exps)
                    cn = [c * gto_norm(l, a) for c, a in zip(coefs, exps)]
                    # contracted normalisation (PySCF `_nomalize_contracted_ao`)
                    s = 0.0
                    for ci, ai in zip(cn, exps):
                        for cj, aj in zip(cn, exps):
                            s += ci * cj * gaussian_int(2 * l + 2, ai + aj)
                    cn = [c / math.sqrt(s) for c in cn]
                    pc = len(env)
                    env.extend(cn)
                    packed.append((l, len(exps), pe, pc))
                cache[sym] = packed
            for (l, npr, pe, pc) in cache[sym]:
                bas.append([ia, l, npr, 1, 0, pe, pc, 0])
        self._atm = atm
        self._bas = np.asarray(bas, dtype=np.int32).reshape(-1, 8)
        self._env = np.asarray(env, dtype=np.float64)
        if (self._bas[:, ANG_OF] > 4).any():
            raise NotImplementedError("angular momentum > g is not supported")

    # --- queries --------------------------------------------------------------------------------
    @property
    def natm(self):
        return len(self._atom)

    @property
    def nbas(self):
        return len(self._bas)

    @property
    def nelectron(self):
        return int(self._atm[:, CHARGE_OF].sum()) - int(self.charge)

    @property
    def nelec(self):
        ne = self.nelectron
        na = (ne + self.spin) // 2
        return na, ne - na

    @property
    def nao(self):
        return int((2 * self._bas[:, ANG_OF] + 1).sum())

    def nao_nr(self):
        return self.nao

    def ao_loc_nr(self):
        dims = 2 * self._bas[:, ANG_OF] + 1
        return np.concatenate([[0], np.cumsum(dims)]).astype(np.int32)

    ao_loc = property(ao_loc_nr)

    def atom_coords(self, unit="Bohr"):
        c = np.array([xyz for _s, xyz in self._atom], dtype=np.float64)
        if str(unit).upper().startswith("ANG"):
            c = c * BOHR
        return c

    def atom_mass_list(self, isotope_avg=False):
        """Atomic masses in amu: isotope-averaged standard weights or the most abundant isotope (PySCF default) [MEM]."""
        avg = {1: 1.008, 2: 4.002602, 3: 6.94, 4: 9.0121831, 5: 10.81, 6: 12.011, 7: 14.007, 8: 15.999, 9: 18.998403163,
               10: 20.1797, 11: 22.98976928, 12: 24.305, 13: 26.9815385, 14: 28.085, 15: 30.973761998, 16: 32.06,
               17: 35.45, 18: 39.948}
        main = {1: 1.00782503223, 2: 4.00260325413, 3: 7.0160034366, 4: 9.012183065, 5: 11.00930536, 6: 12.0,
                7: 14.00307400443, 8: 15.99491461957, 9: 18.99840316273, 10: 19.9924401762, 11: 22.989769282,
                12: 23.985041697, 13: 26.98153853, 14: 27.97692653465, 15: 30.97376199842, 16: 31.9720711744,
                17: 34.968852682, 18: 39.9623831237}
        tab = avg if isotope_avg else main
        return np.array([tab[int(z)] if z else 0.0 for z in self.atom_charges()])

    def atom_charges(self):
        return self._atm[:, CHARGE_OF].astype(np.int64).copy()

    def atom_symbol(self, i):
        return self._atom[i][0]

    def atom_pure_symbol(self, i):
        ghost, el = split_ghost(self._atom[i][0])
        s = re.sub(r"[^A-Za-z]", "", el)
        s = s[0].upper() + s[1:].lower()
        return ("Ghost-" + s) if ghost else s

    def atom_charge(self, i):
        return int(self._atm[i, CHARGE_OF])

    def atom_coord(self, i):
        return np.array(self._atom[i][1])

    @property
    def elements(self):
        return [self.atom_pure_symbol(i) for i in range(self.natm)]

    def aoslice_by_atom(self):
        """[natm, 4]: shell start, shell end, ao start, ao end."""
        loc = self.ao_loc_nr()
        out = np.zeros((self.natm, 4), dtype=np.int64)
        at = self._bas[:, ATOM_OF]
        for ia in range(self.natm):
            idx = np.where(at == ia)[0]
            if len(idx):
                out[ia] = (idx[0], idx[-1] + 1, loc[idx[0]], loc[idx[-1] + 1])
        return out

    def energy_nuc(self):
        z = self.atom_charges().astype(np.float64)
        c = self.atom_coords()
        e = 0.0
        for i in range(self.natm):
            for j in range(i):
                e += z[i] * z[j] / np.linalg.norm(c[i] - c[j])
        return float(e)

    enuc = energy_nuc

    def tot_electrons(self):
        return self.nelectron

    # --- mutation -------------------------------------------------------------------------------
    def set_geom_(self, atoms_or_coords, unit=None, symmetry=None, inplace=True):
        mol = self if inplace else self.copy()
        unit = unit or mol.unit
        if isinstance(atoms_or_coords, np.ndarray) or (
            len(atoms_or_coords) and not isinstance(atoms_or_coords[0][0], str)
            and not isinstance(atoms_or_coords, str)
        ):
            coords = np.asarray(atoms_or_coords, dtype=np.float64).reshape(-1, 3)
            mol.atom = [(s, tuple(c)) for (s, _), c in zip(mol._atom, coords)]
        else:
            mol.atom = atoms_or_coords
        mol.unit = unit
        mol.build()
        return mol

    def set_geom(self, atoms_or_coords, unit=None, symmetry=None):
        return self.set_geom_(atoms_or_coords, unit, symmetry, inplace=False)

    def copy(self):
        new = _copy.copy(self)
        new._atom = list(self._atom)
        new.atom = _copy.deepcopy(self.atom) if not isinstance(self.atom, str) else self.atom
        for k in ("_atm", "_bas", "_env"):
            v = getattr(self, k)
            setattr(new, k, None if v is None else v.copy())
        return new

    def tostring(self, fmt="xyz"):
        c = self.atom_coords() * BOHR
        lines = [f"{self.atom_pure_symbol(i):2s} {c[i,0]:17.8f} {c[i,1]:17.8f} {c[i,2]:17.8f}" for i in range(self.natm)]
        if fmt == "xyz":
            return f"{self.natm}\n\n" + "\n".join(lines)
        return "\n".join(lines)

    # --- integrals (routed to the HIP engine; see engine.py) ---------------------------------------
    def intor(self, name, **kw):
        from . import engine
        return engine.intor(self, name, **kw)

    def intor_symmetric(self, name, **kw):
        return self.intor(name, **kw)


def M(**kw):
    return Mole(**kw).build()
