"""`pyscf.geomopt.geometric_solver.optimize(mf, maxsteps=...)` (call sites
`templates/optimize_geometry.py:99`, `README.md:202-203`).

geomeTRIC (pinned 1.1 in the reference image, `.devcontainer/Dockerfile:120`) is absent here, so this is an
independent quasi-Newton optimiser with geomeTRIC's DEFAULT convergence set [MEM]: |dE| < 1e-6 Ha,
RMS/max gradient < 3e-4 / 4.5e-4 Ha/Bohr, RMS/max displacement < 1.2e-3 / 1.8e-3 Angstrom.  Step-for-step
parity with geomeTRIC is impossible; only the converged geometry/energy within those thresholds is comparable.
BFGS in Cartesian coordinates, trust-radius limited, Hessian guess 0.5 Ha/Bohr^2 on the diagonal.
Returns a `Mole` at the optimised geometry, like the PySCF wrapper.
"""
import numpy as np

from .mole import BOHR

CONV = dict(energy=1e-6, grms=3e-4, gmax=4.5e-4, drms=1.2e-3, dmax=1.8e-3)


def optimize(mf, maxsteps=100, callback=None, **kw):
    gs = mf.nuc_grad_method().as_scanner()
    mol = mf.mol
    x = mol.atom_coords().ravel().copy()
    n = x.size
    H = np.eye(n) * 0.5
    trust = 0.3
    e, g = gs(mol)
    g = g.ravel()
    log = lambda msg: mf._log(3, msg)
    log("Step    Energy (Ha)        dE         RMS grad    max grad    RMS disp(A)  max disp(A)")
    converged = False
    for step in range(1, maxsteps + 1):
        w, v = np.linalg.eigh(H)
        w = np.maximum(np.abs(w), 1e-3)
        dx = -(v * (1.0 / w)) @ (v.T @ g)
        nrm = np.linalg.norm(dx)
        if nrm > trust:
            dx *= trust / nrm
        x_new = x + dx
        mol_new = mol.set_geom_(x_new.reshape(-1, 3), unit="Bohr", inplace=False)
        e_new, g_new = gs(mol_new)
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
        s, y = dx, g_new - g
        sy = s @ y
        if sy > 1e-10:
            Hs = H @ s
            H = H + np.outer(y, y) / sy - np.outer(Hs, Hs) / (s @ Hs)
        disp = dx.reshape(-1, 3) * BOHR
        d_rms = np.sqrt((disp ** 2).sum(axis=1).mean())
        d_max = np.sqrt((disp ** 2).sum(axis=1)).max()
        gm = g_new.reshape(-1, 3)
        g_rms = np.sqrt((gm ** 2).sum(axis=1).mean())
        g_max = np.sqrt((gm ** 2).sum(axis=1)).max()
        log(f"{step:4d}  {e_new:16.10f}  {de:+.2e}  {g_rms:.3e}  {g_max:.3e}  {d_rms:.3e}  {d_max:.3e}")
        x, e, g, mol = x_new, e_new, g_new, mol_new
        if callback is not None:
            callback(locals())
        if (abs(de) < CONV["energy"] and g_rms < CONV["grms"] and g_max < CONV["gmax"]
                and d_rms < CONV["drms"] and d_max < CONV["dmax"]):
            converged = True
            break
    log("Geometry optimization " + (f"converged in {step} steps" if converged else f"NOT converged in {maxsteps} steps"))
    mol.verbose = mf.mol.verbose
    return mol


kernel = optimize
