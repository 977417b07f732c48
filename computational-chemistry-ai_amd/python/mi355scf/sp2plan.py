"""Planned purification: the sequence of quadratics that turns the (orthonormal-basis) Fock matrix into its occupied-space
projector, fixed in advance from four numbers -- outer bounds [lo, hi] of the spectrum and inner bounds of the gap
(homo_in >= HOMO, lumo_in <= LUMO).  Host-side planning for `mi_sp2_iterate_planned` (the density from the
Fock matrix; stands in for the LAPACK `eig` inside PySCF's `SCF.eig`, templates/calculate_energy.py:155 -> kernel()).

Frame: x = (hi - e) / (hi - lo), so the virtual band is V = [0, v1] and the occupied band O = [o0, 1].  One step either folds V
about a point c inside/above it, y = (x - c)^2, or folds O, y = -(x - c)^2, and rescales the result back to V = [0, v1'],
O = [o0', 1]; c is chosen (scan) to maximise the relative gap (o0' - v1') after the step.  Folding a band about its middle
squares its width, so both bands collapse quadratically once the gap has been opened; trace-correcting SP2 (x^2 / 2x - x^2:
the c = 0 / c = 1 members of the same family) needs about twice as many steps because it cannot use the gap bounds.
The caller validates the result (tr(X - X^2), tr X) and falls back to a diagonalisation -- which also refreshes the bounds --
whenever the spectrum has moved outside what was planned for.
"""
import numpy as np

_NSCAN = 256


def _best_step(v1, o0):
    """-> (gain, kind, c, v1', o0', (a, b, c0)) for bands V = [0, v1], O = [o0, 1]; x' = a x^2 + b x + c0."""
    best = None
    # A: y = (x - c)^2, c in [0, o0/2): the virtual band folds
    c = np.linspace(0.0, min(o0 / 2.0, v1 + 0.5 * (o0 - v1)) * (1.0 - 1e-9), _NSCAN)
    nv1 = np.maximum(c * c, (v1 - c) ** 2)
    nv0 = np.where(c <= v1, 0.0, (v1 - c) ** 2)
    no0, no1 = (o0 - c) ** 2, (1.0 - c) ** 2
    span = no1 - nv0
    g = np.where(no0 > nv1, (no0 - nv1) / span, -1.0)
    i = int(np.argmax(g))
    if g[i] > 0:
        best = (g[i], "A", c[i], (nv1[i] - nv0[i]) / span[i], (no0[i] - nv0[i]) / span[i],
                (1.0 / span[i], -2.0 * c[i] / span[i], (c[i] * c[i] - nv0[i]) / span[i]))
    # B: y = -(x - c)^2, c in ((v1 + 1)/2, 1]: the occupied band folds
    c = np.linspace(max((v1 + 1.0) / 2.0, o0 - 0.5 * (o0 - v1)) * (1.0 + 1e-9), 1.0, _NSCAN)
    m = np.maximum((1.0 - c) ** 2, (c - o0) ** 2)
    hi = np.where(c >= o0, 0.0, -(o0 - c) ** 2)
    no0, no1 = -m, hi
    nv0, nv1 = -c * c, -(v1 - c) ** 2
    span = no1 - nv0
    g = np.where(no0 > nv1, (no0 - nv1) / span, -1.0)
    i = int(np.argmax(g))
    if g[i] > 0 and (best is None or g[i] > best[0]):
        best = (g[i], "B", c[i], (nv1[i] - nv0[i]) / span[i], (no0[i] - nv0[i]) / span[i],
                (-1.0 / span[i], 2.0 * c[i] / span[i], (-c[i] * c[i] - nv0[i]) / span[i]))
    return best


def plan(lo, hi, homo_in, lumo_in, tol=1e-14, max_steps=60):
    """Coefficients [nit + 1, 3]: row 0 = (0, b, c) of the affine map X_0 = b F + c I; row k = (a, b, c) of step k.
    Returns None when the bounds leave no gap to work with."""
    if not (lo < homo_in < lumo_in < hi):
        return None
    w = hi - lo
    v1, o0 = (hi - lumo_in) / w, (hi - homo_in) / w
    coef = [(0.0, -1.0 / w, hi / w)]
    for _ in range(max_steps):
        st = _best_step(v1, o0)
        if st is None:
            return None
        _g, _kind, _c, v1, o0, abc = st
        coef.append(abc)
        if v1 < tol and (1.0 - o0) < tol:
            break
    else:
        return None
    return np.ascontiguousarray(coef, dtype=np.float64)


def bounds_from_spectrum(e, nocc, inner_margin=0.15, outer_margin=2.0):
    """(lo, hi, homo_in, lumo_in) from ascending orbital energies: the margins are what the spectrum may move between the
    diagonalisation and the cycles that reuse the plan (a quarter of the gap at most on the inside)."""
    e = np.asarray(e, dtype=np.float64)
    homo, lumo = float(e[nocc - 1]), float(e[nocc])
    d = min(inner_margin, 0.25 * (lumo - homo))
    return float(e[0]) - outer_margin, float(e[-1]) + outer_margin, homo + d, lumo - d


def gap_from_traces(tx, tx2, emin, emax, wmin=1e-9, wmax=0.2):
    """(homo_ub, lumo_lb): an interval INSIDE the HOMO-LUMO gap of the matrix a trace-correcting SP2 run has just purified, from
    the traces it recorded anyway -- tx[i] = tr X_i, tx2[i] = tr X_i^2, X_0 = (emax I - F) / (emax - emin), X_{i+1} = X_i^2 or
    2 X_i - X_i^2.  Every eigenvalue x of X_i lies in [0, 1], so x (1 - x) <= w_i = tr(X_i - X_i^2): the spectrum of X_i avoids
    (u_i, 1 - u_i), u_i = (1 - sqrt(1 - 4 w_i)) / 2.  Both SP2 maps increase on [0, 1], so the hole pulls back through their
    inverses to a hole (a, b) of X_0, i.e. no orbital energy in (emax - b W, emax - a W) (the idea of Rubensson & Niklasson's
    interior-eigenvalue estimates from purification, here with the plain trace instead of the Frobenius norm [MEM]).  Every step
    gives a valid hole; the late ones (small w_i, still above rounding) give the widest.  None if no step qualifies."""
    tx, tx2 = np.asarray(tx, dtype=np.float64), np.asarray(tx2, dtype=np.float64)
    n = len(tx)
    # which map step i -> i + 1 took: the one whose trace reproduces tx[i + 1]
    sq = np.abs(tx2[:-1] - tx[1:]) <= np.abs(2.0 * tx[:-1] - tx2[:-1] - tx[1:])
    w = tx - tx2
    best = None
    for i in range(n):
        if not (wmin < w[i] < wmax):
            continue
        u = 2.0 * w[i] / (1.0 + np.sqrt(1.0 - 4.0 * w[i]))
        a, b = u, 1.0 - u
        for j in range(i - 1, -1, -1):
            if sq[j]:
                a, b = np.sqrt(a), np.sqrt(b)
            else:                                   # y = 2 x - x^2  ->  x = 1 - sqrt(1 - y) = y / (1 + sqrt(1 - y))
                a, b = a / (1.0 + np.sqrt(1.0 - a)), b / (1.0 + np.sqrt(1.0 - b))
        if b > a and (best is None or b - a > best[1] - best[0]):
            best = (a, b)
    if best is None:
        return None
    wd = emax - emin
    return emax - best[1] * wd, emax - best[0] * wd


def bounds_from_traces(tx, tx2, emin, emax, inner_margin=0.15, outer_margin=0.5):
    """(lo, hi, homo_in, lumo_in) for `plan` from a trace-correcting run (see gap_from_traces): Gershgorin bounds outside (they
    move with the matrix, so a small margin), the hole shrunk by `inner_margin` (a quarter of its width at most) inside."""
    g = gap_from_traces(tx, tx2, emin, emax)
    if g is None:
        return None
    homo_ub, lumo_lb = g
    d = min(inner_margin, 0.25 * (lumo_lb - homo_ub))
    return emin - outer_margin, emax + outer_margin, homo_ub + d, lumo_lb - d
