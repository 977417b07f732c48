"""CPU-side checks of the engine library's host tables (no GPU compute): C-ABI symbols, real solid
harmonics vs scipy and vs the oracle's hand-entered table, Rys roots vs Boys-function moments."""
import ctypes
import re
import os

import numpy as np
import pytest

from conftest import ROOT


def _engine():
    from mi355scf import engine
    engine.build_library()
    return engine


def test_abi_exports_every_declared_symbol():
    eng = _engine()
    hdr = open(os.path.join(ROOT, "include", "mi355scf.h")).read()
    names = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 12
    L = eng.lib()
    for n in sorted(names):
        assert hasattr(L, n), f"libmi355scf.so does not export {n}"
    assert L.mi_abi_version() == 2   # round 2: partial-sum reductions, explicit gradient shard, MI_ERR_NOMEM


@pytest.mark.parametrize("l", [0, 1, 2, 3, 4])
def test_c2s_matches_scipy_spherical_harmonics(l):
    from scipy.special import sph_harm
    eng = _engine()
    c = eng.c2s_table(l)
    rng = np.random.default_rng(l)
    pts = rng.normal(size=(40, 3))
    pts /= np.linalg.norm(pts, axis=1)[:, None]
    x, y, z = pts.T
    theta, phi = np.arctan2(y, x), np.arccos(z)
    mono = []
    for lx in range(l, -1, -1):
        for ly in range(l - lx, -1, -1):
            mono.append(x ** lx * y ** ly * z ** (l - lx - ly))
    mono = np.array(mono).T  # [npts, ncart]
    got = mono @ c
    ref = []
    for m in range(-l, l + 1):
        Y = sph_harm(abs(m), l, theta, phi)
        if m < 0:
            v = np.sqrt(2) * (-1) ** m * Y.imag
        elif m == 0:
            v = Y.real
        else:
            v = np.sqrt(2) * (-1) ** m * Y.real
        ref.append(v)
    ref = np.array(ref).T
    if l == 1:
        ref = ref[:, [2, 0, 1]]  # PySCF p order: x, y, z
    assert np.abs(got - ref).max() < 1e-13


@pytest.mark.parametrize("n", range(1, 9))
def test_rys_roots_reproduce_boys_moments(n):
    """sum_i w_i u_i^k == F_k(x) for k < 2n (Gauss rule exactness), F_k from the oracle's Boys function."""
    eng = _engine()
    from oracle import oracle as orc
    L = orc.lib()
    for x in [0.0, 1e-3, 0.7, 3.3, 9.9, 17.0, 33.3, 39.99, 52.0, 74.9, 80.1, 300.0]:
        r, w = eng.rys_roots(n, x)
        F = np.zeros(2 * n)
        L.orc_boys(2 * n - 1, ctypes.c_double(x), F.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        for k in range(2 * n):
            got = float(np.sum(w * r ** k))
            assert abs(got - F[k]) < 2e-13 * F[0], (n, x, k, got, F[k])
