"""`pyscf.geomopt.geometric_solver.optimize(mf, maxsteps=...)` (call sites
`templates/optimize_geometry.py:99`, `README.md:202-203`).

geomeTRIC (pinned 1.1 in the reference image, `.devcontainer/Dockerfile:120`) is absent here, so this is an
independent quasi-Newton optimiser with geomeTRIC's DEFAULT convergence set [MEM]: |dE| < 1e-6 Ha,
RMS/max gradient < 3e-4 / 4.5e-4 Ha/Bohr, RMS/max displacement < 1.2e-3 / 1.8e-3 Angstrom.  Step-for-step
parity with geomeTRIC is impossible; only the converged geometry/energy within those thresholds is comparable.
Default: BFGS in redundant primitive internal coordinates (`internals.py`: stretches, bends, torsions on the
detected bond graph; Lindh model-Hessian guess (diagonal, `Internals.guess_hessian_diag`); trust-radius restricted steps; iterative
back-transformation).  Fallback (linear bends, rank-deficient primitive sets): BFGS in Cartesian coordinates
on the valence-force-field model Hessian B^T K B.  Returns a `Mole` at the optimised geometry, like the PySCF
wrapper.
"""
import itertools

import numpy as np

from .internals import Internals
from .mole import BOHR

CONV = dict(energy=1e-6, grms=3e-4, gmax=4.5e-4, drms=1.2e-3, dmax=1.8e-3)
_COV = {1: 0.31, 2: 0.28, 3: 1.28, 4: 0.96, 5: 0.84, 6: 0.76, 7: 0.71, 8: 0.66, 9: 0.57, 10: 0.58,
        11: 1.66, 12: 1.41, 13: 1.21, 14: 1.11, 15: 1.07, 16: 1.05, 17: 1.02, 18: 1.06}


def _bond(x, i, j):
    return np.linalg.norm(x[i] - x[j])


def _angle(x, i, j, k):
    a, b = x[i] - x[j], x[k] - x[j]
    return np.arccos(np.clip(a @ b / np.linalg.norm(a) / np.linalg.norm(b), -1.0, 1.0))


def _dihedral(x, i, j, k, l):
    b0, b1, b2 = x[i] - x[j], x[k] - x[j], x[l] - x[k]
    b1n = b1 / np.linalg.norm(b1)
    v, w = b0 - (b0 @ b1n) * b1n, b2 - (b2 @ b1n) * b1n
    return np.arctan2(np.cross(b1n, v) @ w, v @ w)


def model_hessian(mol, x):
    """Cartesian model Hessian B^T K B from the bond graph (numerical Wilson B rows)."""
    n = len(x)
    z = mol.atom_charges()
    rc = np.array([_COV.get(int(q), 1.2) for q in z]) / BOHR
    nb = [[] for _ in range(n)]
    bonds = []
    for i in range(n):
        for j in range(i):
            if _bond(x, i, j) < 1.3 * (rc[i] + rc[j]):
                bonds.append((i, j)); nb[i].append(j); nb[j].append(i)
    ics = [(_bond, b, 0.5) for b in bonds]
    for j in range(n):
        for i, k in itertools.combinations(nb[j], 2):
            ics.append((_angle, (i, j, k), 0.2))
    for (j, k) in bonds:
        for i in nb[j]:
            for l in nb[k]:
                if len({i, j, k, l}) == 4:
                    ics.append((_dihedral, (i, j, k, l), 0.1))
    H = np.eye(3 * n) * 0.02
    h = 1e-4
    for fn, idx, kf in ics:
        row = np.zeros(3 * n)
        for a in idx:
            for c in range(3):
                xp, xm = x.copy(), x.copy()
                xp[a, c] += h; xm[a, c] -= h
                d = fn(xp, *idx) - fn(xm, *idx)
                if fn is _dihedral:
                    d = (d + np.pi) % (2 * np.pi) - np.pi
                row[3 * a + c] = d / (2 * h)
        H += kf * np.outer(row, row)
    return H


def _report(e_new, de, g_new, dx):
    disp = dx.reshape(-1, 3) * BOHR
    d_rms = np.sqrt((disp ** 2).sum(axis=1).mean())
    d_max = np.sqrt((disp ** 2).sum(axis=1)).max()
    gm = g_new.reshape(-1, 3)
    g_rms = np.sqrt((gm ** 2).sum(axis=1).mean())
    g_max = np.sqrt((gm ** 2).sum(axis=1)).max()
    done = (abs(de) < CONV["energy"] and g_rms < CONV["grms"] and g_max < CONV["gmax"]
            and d_rms < CONV["drms"] and d_max < CONV["dmax"])
    return done, f"{e_new:16.10f}  {de:+.2e}  {g_rms:.3e}  {g_max:.3e}  {d_rms:.3e}  {d_max:.3e}"


def _bfgs(H, s, y):
    sy = s @ y
    if sy > 1e-10:
        Hs = H @ s
        H = H + np.outer(y, y) / sy - np.outer(Hs, Hs) / (s @ Hs)
    return H


def _restricted_step(H, g, trust):
    """Minimiser of the quadratic model inside |s| <= trust (level-shifted Newton step; H is made positive)."""
    w, v = np.linalg.eigh(H)
    gv = v.T @ g
    w = np.maximum(np.abs(w), 1e-4)
    s = -(v * (1.0 / w)) @ gv
    if np.linalg.norm(s) <= trust:
        return s
    lo, hi = 0.0, 1e3
    for _ in range(100):  # bisection on the shift: |s(mu)| decreases monotonically
        mu = 0.5 * (lo + hi)
        if np.linalg.norm(gv / (w + mu)) > trust:
            lo = mu
        else:
            hi = mu
    return -(v * (1.0 / (w + hi))) @ gv


def optimize_cartesian(energy_grad, mol, maxsteps=100, log=lambda m: None, callback=None):
    """BFGS in Cartesian coordinates on a valence-force-field model Hessian (fallback for molecules whose
    primitive set is incomplete, e.g. with linear bends)."""
    x = mol.atom_coords().ravel().copy()
    H = model_hessian(mol, x.reshape(-1, 3))
    trust = 0.3
    e, g = energy_grad(mol)
    g = g.ravel()
    converged = False
    for step in range(1, maxsteps + 1):
        dx = _restricted_step(H, g, trust)
        nrm = np.linalg.norm(dx)
        mol_new = mol.set_geom_((x + dx).reshape(-1, 3), unit="Bohr", inplace=False)
        e_new, g_new = energy_grad(mol_new)
        g_new = g_new.ravel()
        de = e_new - e
        pred = g @ dx + 0.5 * dx @ H @ dx
        ratio = de / pred if abs(pred) > 1e-14 else 1.0
        if de > 1e-5 and trust > 0.01:       # uphill: shrink and retry from the old point
            trust *= 0.5
            log(f"{step:4d}  step rejected (dE = {de:+.2e}); trust radius -> {trust:.3f}")
            continue
        if ratio > 0.75 and nrm > 0.8 * trust:
            trust = min(trust * 1.5, 0.5)
        elif ratio < 0.25:
            trust = max(trust * 0.5, 0.02)
        H = _bfgs(H, dx, g_new - g)
        done, line = _report(e_new, de, g_new, dx)
        log(f"{step:4d}  {line}")
        x, e, g, mol = x + dx, e_new, g_new, mol_new
        if callback is not None:
            callback(locals())
        if done:
            converged = True
            break
    return mol, converged, step


def optimize_internal(energy_grad, mol, maxsteps=100, log=lambda m: None, callback=None):
    """BFGS in redundant primitive internal coordinates (stretches, bends, torsions).

    Per step: g_q = G^- B g_x; Newton step on the projected Hessian P H P + 1000 (1 - P) restricted to the
    trust radius; iterative back-transformation to Cartesians; BFGS update with the ACHIEVED internal step.
    Returns (mol, converged, steps) or None when the primitive set cannot span the 3N-6 internal motions."""
    x = mol.atom_coords().copy()
    n = len(x)
    ic = Internals(mol.atom_charges(), x)
    B = ic.bmatrix(x)
    Ginv, P, rank = ic.ginv(B)
    if n < 2 or ic.has_linear or rank < max(3 * n - 6, 1):
        return None
    H = np.diag(ic.guess_hessian_diag(x))
    trust = 0.3
    e, g = energy_grad(mol)
    gq = Ginv @ (B @ g.ravel())
    converged = False
    for step in range(1, maxsteps + 1):
        Hp = P @ H @ P + 1000.0 * (np.eye(ic.nq) - P)
        dq = _restricted_step(Hp, P @ gq, trust)
        x_new, dq_got = ic.to_cartesian(x, dq)
        dx = (x_new - x).ravel()
        mol_new = mol.set_geom_(x_new, unit="Bohr", inplace=False)
        e_new, g_new = energy_grad(mol_new)
        de = e_new - e
        pred = gq @ dq_got + 0.5 * dq_got @ H @ dq_got
        ratio = de / pred if abs(pred) > 1e-14 else 1.0
        if de > 1e-5 and trust > 0.01:
            trust *= 0.5
            log(f"{step:4d}  step rejected (dE = {de:+.2e}); trust radius -> {trust:.3f}")
            continue
        nrm = np.linalg.norm(dq_got)
        if ratio > 0.75 and nrm > 0.8 * trust:
            trust = min(trust * 1.5, 0.5)
        elif ratio < 0.25:
            trust = max(trust * 0.5, 0.02)
        B_new = ic.bmatrix(x_new)
        Ginv_new, P_new, rank_new = ic.ginv(B_new)
        gq_new = Ginv_new @ (B_new @ g_new.ravel())
        H = _bfgs(H, dq_got, gq_new - gq)
        done, line = _report(e_new, de, g_new.ravel(), dx)
        log(f"{step:4d}  {line}")
        x, e, g, gq, mol, B, Ginv, P = x_new, e_new, g_new, gq_new, mol_new, B_new, Ginv_new, P_new
        if callback is not None:
            callback(locals())
        if done:
            converged = True
            break
        if rank_new < 3 * n - 6:   # a bend went linear / the primitive set lost rank: finish in Cartesians
            log("      primitive internal coordinates lost rank; continuing in Cartesian coordinates")
            mol2, conv2, st2 = optimize_cartesian(energy_grad, mol, maxsteps - step, log, callback)
            return mol2, conv2, step + st2
    return mol, converged, step


def optimize(mf, maxsteps=100, callback=None, coordsys="internal", **kw):
    """Drop-in for `pyscf.geomopt.geometric_solver.optimize(mf, maxsteps=...)`: returns a `Mole` at the
    optimised geometry.  `coordsys="internal"` (default; redundant primitive internals, Cartesian fallback) or
    "cart"."""
    gs = mf.nuc_grad_method().as_scanner()
    # forces inside the optimisation loop: derivative quartets screened at 1e-10 instead of 1e-13.  Measured on ibuprofen /
    # def2-TZVP with one density (tools/eri_bench.py, GRAD_DTOLS): 1e-13 1.30 s; 1e-11 1.03 s, forces change by 1.5e-7 Ha/Bohr;
    # 1e-10 0.87 s, 1.3e-6 (0.3 % of geomeTRIC's 4.5e-4 max-gradient criterion); 1e-9 0.72 s, 1.5e-5 (too close to it).
    # `optimize(mf, grad_dtol=...)` overrides; single gradients (`mf.nuc_grad_method().kernel()`) keep 1e-13.
    gs.grad_dtol = kw.pop("grad_dtol", 1e-10)
    mol = mf.mol
    log = lambda msg: mf._log(3, msg)
    log("Step    Energy (Ha)        dE         RMS grad    max grad    RMS disp(A)  max disp(A)")
    res = None
    if coordsys.lower() in ("internal", "ric", "tric", "prim", "dlc"):
        res = optimize_internal(gs, mol, maxsteps, log, callback)
    if res is None:
        res = optimize_cartesian(gs, mol, maxsteps, log, callback)
    mol_opt, converged, step = res
    mol_opt._opt_converged, mol_opt._opt_steps = bool(converged), int(step)   # (PySCF's wrapper returns only the Mole)
    log("Geometry optimization " + (f"converged in {step} steps" if converged else f"NOT converged in {maxsteps} steps"))
    mol_opt.verbose = mf.mol.verbose
    return mol_opt


kernel = optimize
