"""Fixed 3-D geometries for the SMILES strings the BASELINE configs and template docs use
(SURVEY.md section 8b last row, section 8d).  RDKit (ETKDG + MMFF, `templates/calculate_energy.py:62-81`)
is absent here and cannot be reproduced; the `rdkit` stand-in maps known SMILES to these committed
geometries (Angstrom), built from standard bond lengths/angles by a small Z-matrix (NeRF) placer."""
import math

import numpy as np


def _place(a, b, c, r, theta, phi):
    """Position at distance r from a, angle theta (deg) with b, dihedral phi (deg) with c."""
    a, b, c = map(np.asarray, (a, b, c))
    th, ph = math.radians(theta), math.radians(phi)
    bc = (a - b) / np.linalg.norm(a - b)
    n = np.cross(b - c, bc)
    n /= np.linalg.norm(n)
    m = np.cross(n, bc)
    d = np.array([-r * math.cos(th), r * math.sin(th) * math.cos(ph), r * math.sin(th) * math.sin(ph)])
    return a + d[0] * bc + d[1] * m + d[2] * n


class _Builder:
    def __init__(self):
        self.sym, self.xyz = [], []

    def add(self, sym, xyz):
        self.sym.append(sym)
        self.xyz.append(np.asarray(xyz, dtype=float))
        return len(self.sym) - 1

    def z(self, sym, a, b, c, r, theta, phi):
        return self.add(sym, _place(self.xyz[a], self.xyz[b], self.xyz[c], r, theta, phi))

    def methyl_h(self, c, a, b, r=1.09, start=60.0):
        return [self.z("H", c, a, b, r, 109.5, start + 120.0 * k) for k in range(3)]

    def result(self):
        return list(self.sym), np.array(self.xyz)


def _water():
    b = _Builder()
    b.add("O", (0, 0, 0)); b.add("H", (0, -0.757, 0.587)); b.add("H", (0, 0.757, 0.587))
    return b.result()


def _methane():
    b = _Builder()
    b.add("C", (0, 0, 0))
    for s in ((1, 1, 1), (-1, -1, 1), (-1, 1, -1), (1, -1, -1)):
        b.add("H", np.array(s) * 0.629)
    return b.result()


def _formaldehyde():  # reference README.md:187-192
    b = _Builder()
    b.add("C", (0, 0, 0)); b.add("O", (1.2, 0, 0)); b.add("H", (-0.5, 0.9, 0)); b.add("H", (-0.5, -0.9, 0))
    return b.result()


def _ethanol():
    b = _Builder()
    c1 = b.add("C", (0, 0, 0)); c2 = b.add("C", (1.52, 0, 0))
    o = b.add("O", _place(b.xyz[c2], b.xyz[c1], (0, 1, 0), 1.43, 108.0, 180.0))
    b.methyl_h(c1, c2, o, start=60.0)
    b.z("H", c2, c1, o, 1.09, 110.0, 120.0); b.z("H", c2, c1, o, 1.09, 110.0, -120.0)
    b.z("H", o, c2, c1, 0.96, 108.5, 180.0)
    return b.result()


def _acetic_acid():
    b = _Builder()
    c1 = b.add("C", (0, 0, 0)); c2 = b.add("C", (1.50, 0, 0))
    o1 = b.add("O", _place(b.xyz[c2], b.xyz[c1], (0, 1, 0), 1.21, 126.0, 0.0))
    o2 = b.z("O", c2, c1, o1, 1.36, 111.0, 180.0)
    b.methyl_h(c1, c2, o1, start=0.0)
    b.z("H", o2, c2, o1, 0.97, 106.0, 0.0)
    return b.result()


def _benzene(rcc=1.3915, rch=1.0800):
    b = _Builder()
    for k in range(6):
        a = math.pi / 3 * k
        b.add("C", (rcc * math.cos(a), rcc * math.sin(a), 0.0))
    for k in range(6):
        a = math.pi / 3 * k
        b.add("H", ((rcc + rch) * math.cos(a), (rcc + rch) * math.sin(a), 0.0))
    return b.result()


def _ibuprofen():
    """CC(C)Cc1ccc(cc1)C(C)C(=O)O  -> C13H18O2, 33 atoms."""
    b = _Builder()
    R = 1.395
    ring = [b.add("C", (R * math.cos(math.pi / 3 * k), R * math.sin(math.pi / 3 * k), 0.0)) for k in range(6)]
    for k in (1, 2, 4, 5):
        a = math.pi / 3 * k
        b.add("H", ((R + 1.085) * math.cos(a), (R + 1.085) * math.sin(a), 0.0))
    # isobutyl on ring[0]
    ca = b.add("C", (R + 1.51, 0.0, 0.0))
    cb = b.z("C", ca, ring[0], ring[1], 1.54, 113.0, 90.0)
    cc = b.z("C", cb, ca, ring[0], 1.53, 111.0, 180.0)
    cd = b.z("C", cb, ca, ring[0], 1.53, 111.0, -60.0)
    b.z("H", ca, ring[0], cb, 1.095, 108.5, 120.0); b.z("H", ca, ring[0], cb, 1.095, 108.5, -120.0)
    b.z("H", cb, ca, ring[0], 1.10, 108.0, 60.0)
    b.methyl_h(cc, cb, ca, start=60.0)
    b.methyl_h(cd, cb, ca, start=60.0)
    # CH(CH3)COOH on ring[3]
    ce = b.add("C", (-R - 1.52, 0.0, 0.0))
    cf = b.z("C", ce, ring[3], ring[2], 1.53, 112.0, 60.0)
    cg = b.z("C", ce, ring[3], ring[2], 1.52, 109.5, -65.0)
    b.z("H", ce, ring[3], ring[2], 1.10, 107.5, 177.0)
    b.methyl_h(cf, ce, ring[3], start=60.0)
    o1 = b.z("O", cg, ce, ring[3], 1.21, 125.0, 100.0)
    o2 = b.z("O", cg, ce, ring[3], 1.35, 112.0, -80.0)
    b.z("H", o2, cg, ce, 0.97, 106.5, 180.0)
    return b.result()


def _c60(bond=1.43):
    """Truncated icosahedron (I_h) with uniform edge length."""
    phi = (1 + math.sqrt(5)) / 2
    base = [(0, 1, 3 * phi), (1, 2 + phi, 2 * phi), (phi, 2, 2 * phi + 1)]
    pts = set()
    for (x, y, z) in base:
        for sx in (1, -1):
            for sy in (1, -1):
                for sz in (1, -1):
                    v = (sx * x, sy * y, sz * z)
                    for perm in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):  # even permutations
                        pts.add(tuple(round(v[p], 10) for p in perm))
    xyz = np.array(sorted(pts)) * (bond / 2.0)
    assert len(xyz) == 60
    return ["C"] * 60, xyz


_C60_SMILES = "c12c3c4c5c1c1c6c7c2c2c8c3c3c9c4c4c%10c5c5c1c1c6c6c%11c7c2c2c7c8c3c3c8c9c4c4c9c%10c5c5c1c1c6c6c%11c2c2c7c3c3c8c4c4c9c5c1c1c6c2c3c41"

TABLE = {
    "O": _water, "[OH2]": _water,
    "C": _methane,
    "C=O": _formaldehyde, "O=C": _formaldehyde,
    "CCO": _ethanol, "OCC": _ethanol,
    "CC(=O)O": _acetic_acid, "CC(O)=O": _acetic_acid,
    "c1ccccc1": _benzene, "C1=CC=CC=C1": _benzene,
    "CC(C)Cc1ccc(cc1)C(C)C(=O)O": _ibuprofen, "CC(C)CC1=CC=C(C=C1)C(C)C(=O)O": _ibuprofen,
    "C60": _c60, _C60_SMILES: _c60,
}


def lookup(smiles):
    key = smiles.strip()
    if key not in TABLE:
        return None
    return TABLE[key]()
