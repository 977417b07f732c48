"""CPU oracle for the DFT half of the hot path.  TEST INFRASTRUCTURE.

PARITY UNPINNED against PySCF/libxc (absent here).  Restates, in plain numpy:
  * atom-centred grids the way `pyscf.dft.gen_grid.Grids.build` is documented to build them [MEM]:
    Treutler-Ahlrichs M4 radial quadrature, Lebedev spheres (scipy.integrate.lebedev_rule), NWChem-style
    pruning, Becke fuzzy cells (3x iterated polynomial) with Treutler's sqrt(Bragg radius) adjustment;
  * AO values and gradients on the grid;
  * the closed-shell energy densities of Slater, VWN-RPA, VWN5, B88, LYP, PBE (spin-resolved textbook
    forms evaluated at rho_a = rho_b), with d/drho and d/dsigma from the COMPLEX-STEP derivative -- a
    different technique from the forward-mode dual numbers of the HIP kernel, so the two cross-check;
  * `nr_rks`: (N_elec, E_xc, V_xc) like `numint.nr_rks` [MEM], reached in the reference through
    `mf.xc = 'B3LYP'` (templates/calculate_energy.py:149; templates/optimize_geometry.py:73).
B3LYP = 0.2 HF + 0.08 Slater + 0.72 B88 + 0.19 VWN-RPA + 0.81 LYP (libxc HYB_GGA_XC_B3LYP) [MEM].
"""
import ctypes

import numpy as np

from . import oracle as orc

BOHR = 0.52917721092
_BRAGG = {0: 0.35, 1: 0.35, 2: 1.40, 3: 1.45, 4: 1.05, 5: 0.85, 6: 0.70, 7: 0.65, 8: 0.60, 9: 0.50, 10: 1.50,
          11: 1.80, 12: 1.50, 13: 1.25, 14: 1.10, 15: 1.00, 16: 1.00, 17: 1.00, 18: 1.80}
_TA_XI = {0: 1.0, 1: 0.8, 2: 0.9, 3: 1.8, 4: 1.4, 5: 1.3, 6: 1.1, 7: 0.9, 8: 0.9, 9: 0.9, 10: 0.9,
          11: 1.4, 12: 1.3, 13: 1.3, 14: 1.2, 15: 1.1, 16: 1.0, 17: 1.0, 18: 1.0}
_RAD = [(10, 15, 20), (30, 40, 50), (40, 60, 65), (50, 75, 80), (60, 90, 95), (70, 105, 110)]
_ANG = [(50, 86, 110), (110, 194, 194), (194, 302, 302), (302, 302, 434), (434, 590, 590), (590, 770, 770)]
_LEB_NGRID = [1, 6, 14, 26, 38, 50, 74, 86, 110, 146, 170, 194, 230, 266, 302, 350, 434, 590, 770, 974]
_LEB_DEGREE = {6: 3, 14: 5, 26: 7, 38: 9, 50: 11, 74: 13, 86: 15, 110: 17, 146: 19, 170: 21, 194: 23, 230: 25,
               266: 27, 302: 29, 350: 31, 434: 35, 590: 41, 770: 47, 974: 53}


def _period(z):
    return 0 if z <= 2 else (1 if z <= 10 else 2)


def lebedev(n):
    from scipy.integrate import lebedev_rule
    x, w = lebedev_rule(_LEB_DEGREE[n])
    assert x.shape[1] == n
    return x.T.copy(), w / (4 * np.pi)


def treutler_ahlrichs(n, z):
    xi = _TA_XI[z]
    i = np.arange(1, n + 1)
    step = np.pi / (n + 1)
    x = np.cos(i * step)
    ln2 = xi / np.log(2)
    r = -ln2 * (1 + x) ** 0.6 * np.log((1 - x) / 2)
    dr = step * np.sin(i * step) * ln2 * (1 + x) ** 0.6 * (-0.6 / (1 + x) * np.log((1 - x) / 2) + 1 / (1 - x))
    return r[::-1].copy(), dr[::-1].copy()


def nwchem_prune(z, rads, n_ang):
    alphas = np.array(((0.25, 0.5, 1.0, 4.5), (0.1667, 0.5, 0.9, 3.5), (0.1, 0.4, 0.8, 2.5)))
    leb = np.array(_LEB_NGRID[4:])
    if n_ang < 50:
        return np.repeat(n_ang, len(rads))
    if n_ang == 50:
        leb_l = np.array([1, 2, 2, 2, 1])
    else:
        idx = int(np.where(leb == n_ang)[0][0])
        leb_l = np.array([1, 3, idx - 1, idx, idx - 1])
    r_atom = _BRAGG[z] / BOHR + 1e-200
    row = 0 if z <= 2 else (1 if z <= 10 else 2)
    place = ((rads / r_atom).reshape(-1, 1) > alphas[row]).sum(axis=1)
    return leb[leb_l[place]]


def build_grids(mol, level=3):
    coords_all, w_all = [], []
    zs = mol.atom_charges()
    R = mol.atom_coords()
    natm = mol.natm
    rad = np.sqrt(np.array([_BRAGG[int(z)] / BOHR for z in zs]))   # index 0: ghost centre (first-period grid, radius 0.35 A)
    rr = rad[:, None] / rad[None, :]
    a = 0.25 * (rr.T - rr)
    a = np.clip(a, -0.5, 0.5)
    dist = np.linalg.norm(R[:, None] - R[None, :], axis=2)
    for ia in range(natm):
        z = int(zs[ia])
        n_rad, n_ang = _RAD[level][_period(z)], _ANG[level][_period(z)]
        r, dr = treutler_ahlrichs(n_rad, z)
        rw = 4 * np.pi * r * r * dr
        angs = nwchem_prune(z, r, n_ang)
        cs, ws = [], []
        for n in sorted(set(angs.tolist())):
            x, w = lebedev(n)
            idx = np.where(angs == n)[0]
            cs.append(np.einsum("i,jk->jik", r[idx], x).reshape(-1, 3))
            ws.append(np.einsum("i,j->ji", rw[idx], w).ravel())
        c = np.vstack(cs) + R[ia]
        vol = np.hstack(ws)
        gd = np.linalg.norm(c[None, :, :] - R[:, None, :], axis=2)  # [natm, ng]
        pb = np.ones((natm, len(c)))
        for i in range(natm):
            for j in range(i):
                g = (gd[i] - gd[j]) / dist[i, j]
                g = g + a[i, j] * (1 - g * g)
                for _ in range(3):
                    g = (3 - g * g) * g * 0.5
                pb[i] *= 0.5 * (1 - g)
                pb[j] *= 0.5 * (1 + g)
        coords_all.append(c)
        w_all.append(vol * pb[ia] / pb.sum(axis=0))
    return np.vstack(coords_all), np.hstack(w_all)


# ---------------------------------------------------------------------------------------------
def _c2s(l):
    out = np.zeros((15, 9))      # [NCART_MAX][NSPH_MAX] of oracle.c
    orc.lib().orc_c2s.restype = None
    orc.lib().orc_c2s(ctypes.c_int(l), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out[:(l + 1) * (l + 2) // 2, :2 * l + 1]


def eval_ao(mol, coords, deriv=1):
    """ao[4 or 1][ng][nao]: values and d/dx, d/dy, d/dz of the real-spherical contracted AOs."""
    ng = len(coords)
    nao = mol.nao
    out = np.zeros((4 if deriv else 1, ng, nao))
    loc = mol.ao_loc_nr()
    for ish in range(mol.nbas):
        ia, l, npr, _, _, pe, pc, _ = mol._bas[ish]
        exps, cs = mol._env[pe:pe + npr], mol._env[pc:pc + npr]
        d = coords - mol._env[mol._atm[ia, 1]:mol._atm[ia, 1] + 3]
        r2 = (d * d).sum(axis=1)
        e = np.exp(-np.outer(r2, exps))
        rad = e @ cs
        drad = e @ (-2.0 * exps * cs)
        c2s = _c2s(l)
        x, y, z = d.T
        k = 0
        poly = np.zeros((ng, c2s.shape[0]))
        dpoly = np.zeros((3, ng, c2s.shape[0]))
        for lx in range(l, -1, -1):
            for ly in range(l - lx, -1, -1):
                lz = l - lx - ly
                poly[:, k] = x ** lx * y ** ly * z ** lz
                if lx:
                    dpoly[0, :, k] = lx * x ** (lx - 1) * y ** ly * z ** lz
                if ly:
                    dpoly[1, :, k] = ly * x ** lx * y ** (ly - 1) * z ** lz
                if lz:
                    dpoly[2, :, k] = lz * x ** lx * y ** ly * z ** (lz - 1)
                k += 1
        s = poly @ c2s
        sl = slice(loc[ish], loc[ish + 1])
        out[0][:, sl] = rad[:, None] * s
        if deriv:
            for c in range(3):
                out[1 + c][:, sl] = (drad * d[:, c])[:, None] * s + rad[:, None] * (dpoly[c] @ c2s)
    return out


# ---------------------------------------------------------------------------------------------
# Energy densities per volume, spin-resolved textbook forms; complex-safe (no abs / max).
# ---------------------------------------------------------------------------------------------
def _slater(ra, rb):
    cx = 1.5 * (3.0 / (4 * np.pi)) ** (1.0 / 3)
    return -cx * (ra ** (4.0 / 3) + rb ** (4.0 / 3))


def _b88(ra, rb, saa, sbb):
    beta = 0.0042

    def one(r, s):
        x = np.sqrt(s) / r ** (4.0 / 3)
        asinh = np.log(x + np.sqrt(x * x + 1))
        return -beta * r ** (4.0 / 3) * x * x / (1 + 6 * beta * x * asinh)
    return _slater(ra, rb) + one(ra, saa) + one(rb, sbb)


def _vwn(r, A, x0, b, c):
    rs = (3.0 / (4 * np.pi * r)) ** (1.0 / 3)
    x = np.sqrt(rs)
    X = x * x + b * x + c
    X0 = x0 * x0 + b * x0 + c
    Q = np.sqrt(4 * c - b * b)
    at = np.arctan(Q / (2 * x + b))  # numpy's complex arctan keeps the O(h) imaginary part (complex step)
    eps = A * (np.log(x * x / X) + 2 * b / Q * at
               - b * x0 / X0 * (np.log((x - x0) ** 2 / X) + 2 * (b + 2 * x0) / Q * at))
    return r * eps


def _vwn_rpa(r):
    return _vwn(r, 0.0310907, -0.409286, 13.0720, 42.7198)


def _vwn5(r):
    return _vwn(r, 0.0310907, -0.10498, 3.72744, 12.9352)


def _lyp(ra, rb, saa, sab, sbb):
    a, b, c, d = 0.04918, 0.132, 0.2533, 0.349
    r = ra + rb
    s = saa + 2 * sab + sbb
    t = r ** (-1.0 / 3)
    D = 1 + d * t
    w = np.exp(-c * t) / D * r ** (-11.0 / 3)
    dl = c * t + d * t / D
    cf = 0.3 * (3 * np.pi ** 2) ** (2.0 / 3)
    t1 = -4 * a / D * ra * rb / r
    br = ra * rb * (2 ** (11.0 / 3) * cf * (ra ** (8.0 / 3) + rb ** (8.0 / 3)) + (47.0 / 18 - 7 * dl / 18) * s
                    - (2.5 - dl / 18) * (saa + sbb) - (dl - 11) / 9 * (ra / r * saa + rb / r * sbb))
    br = br - 2.0 / 3 * r * r * s + (2.0 / 3 * r * r - ra * ra) * sbb + (2.0 / 3 * r * r - rb * rb) * saa
    return t1 - a * b * w * br


def _pw92_mod(rs):
    A, a1, b1, b2, b3, b4 = 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294
    A = 0.031090690869654895  # PW_MOD, as libxc's GGA_C_PBE uses [MEM]
    x = np.sqrt(rs)
    return -2 * A * (1 + a1 * rs) * np.log(1 + 1 / (2 * A * (b1 * x + b2 * rs + b3 * rs * x + b4 * rs * rs)))


def _pbe_x(r, s):
    kappa, mu = 0.804, 0.06672455060314922 * np.pi ** 2 / 3
    kf = (3 * np.pi ** 2 * r) ** (1.0 / 3)
    s2 = s / (4 * kf * kf * r * r)
    ex = -0.75 * (3 / np.pi) ** (1.0 / 3) * r ** (4.0 / 3)
    return ex * (1 + kappa - kappa / (1 + mu * s2 / kappa))


def _pbe_c(r, s):
    beta, gamma = 0.06672455060314922, (1 - np.log(2)) / np.pi ** 2
    rs = (3.0 / (4 * np.pi * r)) ** (1.0 / 3)
    ec = _pw92_mod(rs)
    kf = (3 * np.pi ** 2 * r) ** (1.0 / 3)
    ks2 = 4 * kf / np.pi
    t2 = s / (4 * ks2 * r * r)
    Aa = beta / gamma / (np.exp(-ec / gamma) - 1)
    H = gamma * np.log(1 + beta / gamma * t2 * (1 + Aa * t2) / (1 + Aa * t2 + Aa * Aa * t2 * t2))
    return r * (ec + H)


def parse_xc(name):
    """-> (hyb, [(coef, kind)])."""
    key = str(name).upper().replace("-", "").replace("_", "").replace(" ", "")
    if key in ("HF", ""):
        return 1.0, []
    if key == "B3LYP":
        return 0.2, [(0.08, "slater"), (0.72, "b88"), (0.19, "vwn_rpa"), (0.81, "lyp")]
    if key == "PBE":
        return 0.0, [(1.0, "pbe_x"), (1.0, "pbe_c")]
    if key in ("LDA", "LDA,VWN", "SVWN", "LDA,VWN5", "SVWN5"):
        return 0.0, [(1.0, "slater"), (1.0, "vwn5")]
    if key in ("BLYP", "B88,LYP"):
        return 0.0, [(1.0, "b88"), (1.0, "lyp")]
    if key == "PBE0":
        return 0.25, [(0.75, "pbe_x"), (1.0, "pbe_c")]
    raise NotImplementedError(f"xc functional '{name}' is not implemented in the oracle")


def energy_density(terms, rho, sigma):
    """Closed-shell e(rho, sigma) per volume; works on complex arguments."""
    ra = rb = rho * 0.5
    s4 = sigma * 0.25
    e = 0.0
    for coef, kind in terms:
        if kind == "slater":
            e = e + coef * _slater(ra, rb)
        elif kind == "b88":
            e = e + coef * _b88(ra, rb, s4, s4)
        elif kind == "vwn_rpa":
            e = e + coef * _vwn_rpa(rho)
        elif kind == "vwn5":
            e = e + coef * _vwn5(rho)
        elif kind == "lyp":
            e = e + coef * _lyp(ra, rb, s4, s4, s4)
        elif kind == "pbe_x":
            e = e + coef * _pbe_x(rho, sigma)
        elif kind == "pbe_c":
            e = e + coef * _pbe_c(rho, sigma)
        else:
            raise KeyError(kind)
    return e


def eval_xc(terms, rho, sigma):
    """-> (e per volume, de/drho, de/dsigma) with complex-step derivatives."""
    h = 1e-30
    e = energy_density(terms, rho, sigma)
    vr = np.imag(energy_density(terms, rho + 1j * h, sigma + 0j)) / h
    vs = np.imag(energy_density(terms, rho + 0j, sigma + 1j * h)) / h
    return np.real(e), vr, vs


def nr_rks(mol, coords, weights, xc, dm, block=20000, rho_cut=1e-10):
    hyb, terms = parse_xc(xc)
    nao = mol.nao
    nelec = exc = 0.0
    vmat = np.zeros((nao, nao))
    for p0 in range(0, len(coords), block):
        c, w = coords[p0:p0 + block], weights[p0:p0 + block]
        ao = eval_ao(mol, c, 1)
        c0 = ao[0] @ dm
        rho = np.einsum("gi,gi->g", ao[0], c0)
        grad = np.array([2 * np.einsum("gi,gi->g", ao[1 + k], c0) for k in range(3)])
        sigma = (grad * grad).sum(axis=0)
        ok = rho > rho_cut
        rr = np.where(ok, rho, 1.0)
        ss = np.where(ok, sigma, 0.0)
        e, vr, vs = eval_xc(terms, rr, ss)
        e, vr, vs = np.where(ok, e, 0), np.where(ok, vr, 0), np.where(ok, vs, 0)
        nelec += float(w @ rho)
        exc += float(w @ e)
        aow = ao[0] * (0.5 * w * vr)[:, None]
        for k in range(3):
            aow += ao[1 + k] * (2 * w * vs * grad[k])[:, None]
        vmat += ao[0].T @ aow
    return nelec, exc, vmat + vmat.T, hyb


def rks(mol, xc="B3LYP", level=3, dm0=None, conv_tol=1e-9, max_cycle=50, verbose=False, small_rho_cutoff=1e-7, oracle=None):
    coords, weights = build_grids(mol, level)
    o = oracle if oracle is not None else orc.Oracle(mol)
    info = {}
    grid = {"c": coords, "w": weights, "pruned": small_rho_cutoff <= 1e-20}

    def veff(dm):
        if not grid["pruned"]:
            # PySCF rks.prune_small_rho_grids_ [MEM]: once, with the first density, if the grid integrates N_elec to 1 %
            grid["pruned"] = True
            c, w = grid["c"], grid["w"]
            rho = np.concatenate([np.einsum("gi,ij,gj->g", a, dm, a) for a in
                                  (eval_ao(mol, c[p:p + 20000], 0)[0] for p in range(0, len(c), 20000))])
            n = float(rho @ w)
            if abs(n - mol.nelectron) < 0.01 * n:
                keep = np.abs(rho * w) > small_rho_cutoff / len(w)
                grid["c"], grid["w"] = c[keep], w[keep]
        coords, weights = grid["c"], grid["w"]
        info["ngrids"] = len(weights)
        n, exc, vxc, hyb = nr_rks(mol, coords, weights, xc, dm)
        J, K = o.jk(dm)
        info["nelec"] = n
        v = J + vxc
        e2 = 0.5 * float(np.sum(dm * J)) + exc
        if hyb:
            v = v - 0.5 * hyb * K
            e2 -= 0.25 * hyb * float(np.sum(dm * K))
        return v, e2

    r = orc.rhf(mol, dm0=dm0, conv_tol=conv_tol, max_cycle=max_cycle, veff_fn=veff, verbose=verbose, oracle=o)
    r["nelec_grid"] = info.get("nelec")
    r["ngrids"] = info.get("ngrids", len(weights))
    return r


# ---------------------------------------------------------------------------------------------
# Spin-polarised functionals and UKS (checker for `mi355scf.uks.UKS`)
# ---------------------------------------------------------------------------------------------
def _vwn_eps(x, A, x0, b, c):
    X = x * x + b * x + c
    X0 = x0 * x0 + b * x0 + c
    Q = np.sqrt(4 * c - b * b)
    at = np.arctan(Q / (2 * x + b))
    return A * (np.log(x * x / X) + 2 * b / Q * at - b * x0 / X0 * (np.log((x - x0) ** 2 / X) + 2 * (b + 2 * x0) / Q * at))


def _fzeta(z):
    return ((1 + z) ** (4.0 / 3) + (1 - z) ** (4.0 / 3) - 2) / (2 ** (4.0 / 3) - 2)


_FPP0 = 4.0 / (9.0 * (2 ** (1.0 / 3) - 1))   # f''(0)


def _zeta(ra, rb):
    z = (ra - rb) / (ra + rb)
    lim = 1.0 - 1e-10
    return np.where(np.real(z) > lim, lim, np.where(np.real(z) < -lim, -lim, z))


def _vwn_rpa_spin(ra, rb):
    """libxc LDA_C_VWN_RPA [MEM]: eps_P + (eps_F - eps_P) f(zeta) with the RPA parameter sets."""
    r = ra + rb
    x = np.sqrt((3.0 / (4 * np.pi * r)) ** (1.0 / 3))
    eP = _vwn_eps(x, 0.0310907, -0.409286, 13.0720, 42.7198)
    eF = _vwn_eps(x, 0.01554535, -0.743294, 20.1231, 101.578)
    return r * (eP + (eF - eP) * _fzeta(_zeta(ra, rb)))


def _vwn5_spin(ra, rb):
    """Vosko-Wilk-Nusair 1980, eq. 4.4 interpolation with the Ceperley-Alder fits (para, ferro, spin stiffness)."""
    r = ra + rb
    x = np.sqrt((3.0 / (4 * np.pi * r)) ** (1.0 / 3))
    eP = _vwn_eps(x, 0.0310907, -0.10498, 3.72744, 12.9352)
    eF = _vwn_eps(x, 0.01554535, -0.32500, 7.06042, 18.0578)
    ac = _vwn_eps(x, -1.0 / (6 * np.pi ** 2), -0.0047584, 1.13107, 13.0045)
    z = _zeta(ra, rb)
    fz, z4 = _fzeta(z), z ** 4
    return r * (eP + ac * fz / _FPP0 * (1 - z4) + (eF - eP) * fz * z4)


def _pw92_g(rs, A, a1, b1, b2, b3, b4):
    x = np.sqrt(rs)
    return -2 * A * (1 + a1 * rs) * np.log(1 + 1 / (2 * A * (b1 * x + b2 * rs + b3 * rs * x + b4 * rs * rs)))


def _pbe_c_spin(ra, rb, saa, sab, sbb):
    """Perdew-Burke-Ernzerhof 1996 eqs. 3, 7, 8 with the Perdew-Wang 1992 spin interpolation (PW_MOD constants)."""
    beta, gamma = 0.06672455060314922, (1 - np.log(2)) / np.pi ** 2
    r = ra + rb
    s = saa + 2 * sab + sbb
    rs = (3.0 / (4 * np.pi * r)) ** (1.0 / 3)
    e0 = _pw92_g(rs, 0.031090690869654895, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294)
    e1 = _pw92_g(rs, 0.015545345434827448, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517)
    mac = _pw92_g(rs, 0.016886863940389627, 0.11125, 10.357, 3.6231, 0.88026, 0.49671)
    z = _zeta(ra, rb)
    fz, z4 = _fzeta(z), z ** 4
    ec = e0 - mac * fz / _FPP0 * (1 - z4) + (e1 - e0) * fz * z4
    phi = 0.5 * ((1 + z) ** (2.0 / 3) + (1 - z) ** (2.0 / 3))
    kf = (3 * np.pi ** 2 * r) ** (1.0 / 3)
    ks2 = 4 * kf / np.pi
    t2 = s / (4 * phi ** 2 * ks2 * r * r)
    Aa = beta / gamma / (np.exp(-ec / (gamma * phi ** 3)) - 1)
    H = gamma * phi ** 3 * np.log(1 + beta / gamma * t2 * (1 + Aa * t2) / (1 + Aa * t2 + Aa * Aa * t2 * t2))
    return r * (ec + H)


def energy_density_spin(terms, ra, rb, saa, sab, sbb):
    """e(rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb) per volume; complex-safe.  Exchange by spin scaling."""
    tiny = 1e-300
    e = 0.0
    for coef, kind in terms:
        if kind == "slater":
            e = e + coef * _slater(ra, rb)
        elif kind == "b88":
            # per spin channel; an empty channel contributes nothing (evaluated at a dummy density and masked)
            for r_, s_ in ((ra, saa), (rb, sbb)):
                ok = np.real(r_) > 1e-20
                one = _b88(np.where(ok, r_, 1.0), 0.0 * r_ + 1.0, np.where(ok, s_, 0.0), 0.0 * s_) - _slater(0.0 * r_, 0.0 * r_ + 1.0)
                e = e + coef * np.where(ok, one, 0.0)
        elif kind == "vwn_rpa":
            e = e + coef * _vwn_rpa_spin(ra, rb)
        elif kind == "vwn5":
            e = e + coef * _vwn5_spin(ra, rb)
        elif kind == "lyp":
            e = e + coef * _lyp(ra, rb, saa, sab, sbb)
        elif kind == "pbe_x":
            for r_, s_ in ((ra, saa), (rb, sbb)):
                ok = np.real(r_) > 1e-20
                e = e + coef * np.where(ok, 0.5 * _pbe_x(2 * np.where(ok, r_, 1.0), 4 * np.where(ok, s_, 0.0)), 0.0)
        elif kind == "pbe_c":
            e = e + coef * _pbe_c_spin(ra, rb, saa, sab, sbb)
        else:
            raise KeyError(kind)
    return e


def eval_xc_spin(terms, ra, rb, saa, sab, sbb):
    """-> (e per volume, [de/dra, de/drb, de/dsaa, de/dsab, de/dsbb]) with complex-step derivatives."""
    h = 1e-30
    args = [np.asarray(a, dtype=complex) for a in (ra, rb, saa, sab, sbb)]
    e = np.real(energy_density_spin(terms, *args))
    ders = []
    for i in range(5):
        a2 = [a.copy() for a in args]
        a2[i] = a2[i] + 1j * h
        ders.append(np.imag(energy_density_spin(terms, *a2)) / h)
    return e, ders


def nr_uks(mol, coords, weights, xc, dm, block=20000, rho_cut=1e-10):
    """(N_alpha, N_beta), E_xc, (V_a, V_b), hyb for the spin densities dm[2,N,N]."""
    hyb, terms = parse_xc(xc)
    nao = mol.nao
    nel = np.zeros(2)
    exc = 0.0
    vmat = np.zeros((2, nao, nao))
    for p0 in range(0, len(coords), block):
        c, w = coords[p0:p0 + block], weights[p0:p0 + block]
        ao = eval_ao(mol, c, 1)
        rho, grad = [], []
        for s_ in range(2):
            c0 = ao[0] @ dm[s_]
            rho.append(np.maximum(np.einsum("gi,gi->g", ao[0], c0), 0.0))
            grad.append(np.array([2 * np.einsum("gi,gi->g", ao[1 + k], c0) for k in range(3)]))
        ok = rho[0] + rho[1] > rho_cut
        ra, rb = np.where(ok, rho[0], 0.5), np.where(ok, rho[1], 0.5)
        saa = np.where(ok, (grad[0] * grad[0]).sum(axis=0), 0.0)
        sab = np.where(ok, (grad[0] * grad[1]).sum(axis=0), 0.0)
        sbb = np.where(ok, (grad[1] * grad[1]).sum(axis=0), 0.0)
        e, d = eval_xc_spin(terms, ra, rb, saa, sab, sbb)
        e = np.where(ok, e, 0.0)
        d = [np.where(ok, x, 0.0) for x in d]
        nel += [float(w @ rho[0]), float(w @ rho[1])]
        exc += float(w @ e)
        for s_, (vr, vss, g_own, g_oth) in enumerate(((d[0], d[2], grad[0], grad[1]), (d[1], d[4], grad[1], grad[0]))):
            aow = ao[0] * (0.5 * w * vr)[:, None]
            for k in range(3):
                aow += ao[1 + k] * (w * (2 * vss * g_own[k] + d[3] * g_oth[k]))[:, None]
            m = ao[0].T @ aow
            vmat[s_] += m + m.T
    return nel, exc, vmat, hyb


def uks(mol, xc="B3LYP", level=3, dm0=None, conv_tol=1e-10, max_cycle=100, verbose=False):
    """Spin-unrestricted Kohn-Sham on the oracle integrals and grid (no small-rho pruning).  dm0 as in oracle.uhf."""
    coords, weights = build_grids(mol, level)
    o = orc.Oracle(mol)
    S, T, V, _ = o.int1e()
    h = T + V
    na, nb = mol.nelec
    enuc = mol.energy_nuc()
    info = {}

    def dens(F):
        out = []
        for s_, no in ((0, na), (1, nb)):
            e, c = orc.eig_gen(F[s_], S)
            out.append(c[:, :no] @ c[:, :no].T)
        return np.stack(out)

    def fock(dm):
        nel, exc, vxc, hyb = nr_uks(mol, coords, weights, xc, dm)
        info["nelec"] = nel
        Ja, Ka = o.jk(dm[0], tol=0.0)
        Jb, Kb = o.jk(dm[1], tol=0.0)
        J = Ja + Jb
        F = np.stack([h + J + vxc[0] - hyb * Ka, h + J + vxc[1] - hyb * Kb])
        D = dm[0] + dm[1]
        e = float(np.sum(D * h)) + 0.5 * float(np.sum(D * J)) - 0.5 * hyb * float(np.sum(dm[0] * Ka) + np.sum(dm[1] * Kb)) + exc
        return F, e + enuc

    dm0 = np.asarray(dm0)
    dm = dm0 if dm0.ndim == 3 else np.stack([dm0 * na / max(na + nb, 1), dm0 * nb / max(na + nb, 1)])
    F, e = fock(dm)
    Fh, Eh = [], []
    for it in range(max_cycle):
        err = np.stack([F[s_] @ dm[s_] @ S - S @ dm[s_] @ F[s_] for s_ in range(2)])
        Fh.append(F.copy()); Eh.append(err.ravel().copy())
        Fh, Eh = Fh[-8:], Eh[-8:]
        m = len(Fh)
        A = np.zeros((m + 1, m + 1)); A[0, 1:] = A[1:, 0] = 1.0
        A[1:, 1:] = np.array([[a @ b for b in Eh] for a in Eh])
        rhs = np.zeros(m + 1); rhs[0] = 1.0
        c = np.linalg.lstsq(A, rhs, rcond=None)[0]
        dm = dens(sum(ci * Fi for ci, Fi in zip(c[1:], Fh)))
        F, e_new = fock(dm)
        if verbose:
            print(f"oracle uks cycle {it + 1}: E = {e_new:.12f}  dE = {e_new - e:.3e}")
        done = abs(e_new - e) < conv_tol and np.abs(err).max() < 1e-6
        e = e_new
        if done:
            break
    dm = dens(F)
    F, e = fock(dm)
    return {"e_tot": e, "dm": dm, "nelec_grid": info["nelec"], "ngrids": len(weights)}


# ---------------------------------------------------------------------------------------------
# meta-GGA functionals (checker for `mi_xc_eval_mgga*`): numpy restatement of the published forms, complex-step safe.
#   TPSS   : Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 (2003), eqs. 10 (exchange) and 11-14 (revPKZB correlation).
#   M06-2X : Zhao, Truhlar, Theor. Chem. Acc. 120, 215 (2008).  PARAMETERS ENTERED FROM MEMORY ("unverified-memory"); what
#            the tests pin are the uniform-gas sums (a0 + X = c0 + d0 = 1) and the one-electron limits.
# ---------------------------------------------------------------------------------------------
M062X_A = (4.600000e-01, -2.206052e-01, -9.431788e-02, 2.164494e+00, -2.556466e+00, -1.422133e+01,
           1.555044e+01, 3.598078e+01, -2.722754e+01, -3.924093e+01, 1.522808e+01, 1.522227e+01)
M062X_CSS = (3.097855e-01, -5.528642e+00, 1.347420e+01, -3.213623e+01, 2.846742e+01)
M062X_CAB = (8.833596e-01, 3.357972e+01, -7.043548e+01, 4.978271e+01, -1.852891e+01)
M062X_DSS = (6.902145e-01, 9.847204e-02, 2.214797e-01, -1.968264e-03, -6.775479e-03, 0.0)
M062X_DAB = (1.166404e-01, -9.120847e-02, -6.726189e-02, 6.720580e-05, 8.448011e-04, 0.0)
M062X_HYB = 0.54
_CF6 = 0.6 * (6 * np.pi ** 2) ** (2.0 / 3)


def _pw92_eps_spin(ra, rb):
    r = ra + rb
    rs = (3.0 / (4 * np.pi * r)) ** (1.0 / 3)
    e0 = _pw92_g(rs, 0.031090690869654895, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294)
    e1 = _pw92_g(rs, 0.015545345434827448, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517)
    mac = _pw92_g(rs, 0.016886863940389627, 0.11125, 10.357, 3.6231, 0.88026, 0.49671)
    z = _zeta(ra, rb)
    fz, z4 = _fzeta(z), z ** 4
    return e0 - mac * fz / _FPP0 * (1 - z4) + (e1 - e0) * fz * z4


def _clip_hi(x, hi):
    return np.where(np.real(x) > hi, hi + 0 * x, x)


def _clip_lo(x, lo):
    return np.where(np.real(x) < lo, lo + 0 * x, x)


def _tpss_x_unpol(r, s, tau):
    kappa, b, c, e, mu = 0.804, 0.40, 1.59096, 1.537, 0.21951
    c3 = (3 * np.pi ** 2) ** (2.0 / 3)
    p = s / (4 * c3 * r ** (8.0 / 3))
    tw = s / (8 * r)
    z = _clip_hi(tw / tau, 1.0)
    alpha = _clip_lo((tau - tw) / (0.3 * c3 * r ** (5.0 / 3)), 0.0)
    qb = 0.45 * (alpha - 1) / np.sqrt(1 + b * alpha * (alpha - 1)) + 2 * p / 3
    x = ((10.0 / 81 + c * z * z / (1 + z * z) ** 2) * p + 146.0 / 2025 * qb * qb
         - 73.0 / 405 * qb * np.sqrt(0.5 * (0.6 * z) ** 2 + 0.5 * p * p + 1e-300) + (10.0 / 81) ** 2 / kappa * p * p
         + 2 * np.sqrt(e) * (10.0 / 81) * (0.6 * z) ** 2 + e * mu * p ** 3) / (1 + np.sqrt(e) * p) ** 2
    ex = -0.75 * (3 / np.pi) ** (1.0 / 3) * r ** (4.0 / 3)
    return ex * (1 + kappa - kappa / (1 + x / kappa))


def _channel(ok, fn, *args):
    """Evaluate `fn` only where the spin channel holds density (dummy arguments elsewhere), 0 otherwise."""
    safe = [np.where(ok, a, 1.0 + 0 * a) for a in args]
    return np.where(ok, fn(*safe), 0.0)


def _tpss_x_spin(ra, rb, saa, sbb, ta, tb):
    e = 0.0
    for r_, s_, t_ in ((ra, saa, ta), (rb, sbb, tb)):
        ok = (np.real(r_) > 1e-12) & (np.real(t_) > 1e-14)
        e = e + _channel(ok, lambda r, s, t: 0.5 * _tpss_x_unpol(2 * r, 4 * s, 2 * t), r_, s_, t_)
    return e


def _tpss_c_spin(ra, rb, saa, sab, sbb, ta, tb):
    d = 2.8
    r, s, tau = ra + rb, saa + 2 * sab + sbb, ta + tb
    z = _clip_hi(s / (8 * r) / tau, 1.0)
    zeta = _zeta(ra, rb)
    gz2 = _clip_lo(4 * (rb * rb * saa - 2 * ra * rb * sab + ra * ra * sbb) / r ** 4, 0.0)
    xi2 = gz2 / (4 * (3 * np.pi ** 2 * r) ** (2.0 / 3))
    C0 = 0.53 + 0.87 * zeta ** 2 + 0.50 * zeta ** 4 + 2.26 * zeta ** 6
    C = C0 / (1 + xi2 * 0.5 * ((1 + zeta) ** (-4.0 / 3) + (1 - zeta) ** (-4.0 / 3))) ** 4
    epbe = _pbe_c_spin(ra, rb, saa, sab, sbb) / r
    ets = []
    for r_, s_ in ((ra, saa), (rb, sbb)):
        ok = np.real(r_) > 1e-12
        one = _channel(ok, lambda rr, ss: _pbe_c_spin(rr, 0 * rr, ss, 0 * ss, 0 * ss) / rr, r_, s_)
        ets.append(np.where(ok & (np.real(one) >= np.real(epbe)), one, epbe))
    erev = epbe * (1 + C * z * z) - (1 + C) * z * z * (ra / r * ets[0] + rb / r * ets[1])
    return r * erev * (1 + d * erev * z ** 3)


def _m06_g(x2, cc, gamma):
    u = gamma * x2 / (1 + gamma * x2)
    return sum(c * u ** i for i, c in enumerate(cc))


def _m06_h(x2, z, dc, alpha):
    g = 1 + alpha * (x2 + z)
    return dc[0] / g + (dc[1] * x2 + dc[2] * z) / g ** 2 + (dc[3] * x2 * x2 + dc[4] * x2 * z + dc[5] * z * z) / g ** 3


def _m062x_x_spin(ra, rb, saa, sbb, ta, tb):
    def chan(r, s, t):
        tl = 0.3 * (6 * np.pi ** 2) ** (2.0 / 3) * r ** (5.0 / 3)
        w = (tl / t - 1) / (tl / t + 1)
        return 0.5 * _pbe_x(2 * r, 4 * s) * sum(a * w ** i for i, a in enumerate(M062X_A))
    e = 0.0
    for r_, s_, t_ in ((ra, saa, ta), (rb, sbb, tb)):
        ok = (np.real(r_) > 1e-12) & (np.real(t_) > 1e-14)
        e = e + _channel(ok, chan, r_, s_, t_)
    return e


def _m062x_c_spin(ra, rb, saa, sbb, ta, tb):
    oka = (np.real(ra) > 1e-12) & (np.real(ta) > 1e-14)
    okb = (np.real(rb) > 1e-12) & (np.real(tb) > 1e-14)
    sa = [np.where(oka, a, 1.0 + 0 * a) for a in (ra, saa, ta)]
    sb = [np.where(okb, a, 1.0 + 0 * a) for a in (rb, sbb, tb)]
    out = 0.0
    xs, zs, ess = [], [], []
    for ok, (r, s, t) in ((oka, sa), (okb, sb)):
        x2 = s / r ** (8.0 / 3)
        z = 2 * t / r ** (5.0 / 3) - _CF6
        e_ss = r * _pw92_eps_spin(r, 0 * r)
        D = _clip_lo(1 - x2 / (4 * (z + _CF6)), 0.0)
        out = out + np.where(ok, e_ss * (_m06_g(x2, M062X_CSS, 0.06) + _m06_h(x2, z, M062X_DSS, 0.00515088)) * D, 0.0)
        xs.append(x2); zs.append(z); ess.append(e_ss)
    both = oka & okb
    eab = (sa[0] + sb[0]) * _pw92_eps_spin(sa[0], sb[0]) - ess[0] - ess[1]
    out = out + np.where(both, eab * (_m06_g(xs[0] + xs[1], M062X_CAB, 0.0031) + _m06_h(xs[0] + xs[1], zs[0] + zs[1], M062X_DAB, 0.00304966)), 0.0)
    return out


MGGA_TERMS = {"tpss_x": lambda ra, rb, saa, sab, sbb, ta, tb: _tpss_x_spin(ra, rb, saa, sbb, ta, tb),
              "tpss_c": _tpss_c_spin,
              "m062x_x": lambda ra, rb, saa, sab, sbb, ta, tb: _m062x_x_spin(ra, rb, saa, sbb, ta, tb),
              "m062x_c": lambda ra, rb, saa, sab, sbb, ta, tb: _m062x_c_spin(ra, rb, saa, sbb, ta, tb)}


def parse_xc_mgga(name):
    key = str(name).upper().replace("-", "").replace("_", "").replace(" ", "")
    if key in ("TPSS", "TPSS,TPSS"):
        return 0.0, [(1.0, "tpss_x"), (1.0, "tpss_c")]
    if key == "M062X":
        return M062X_HYB, [(1.0, "m062x_x"), (1.0, "m062x_c")]
    return None


def energy_density_mgga_spin(terms, ra, rb, saa, sab, sbb, ta, tb):
    e = 0.0
    for coef, kind in terms:
        if kind in MGGA_TERMS:
            e = e + coef * MGGA_TERMS[kind](ra, rb, saa, sab, sbb, ta, tb)
        else:
            e = e + energy_density_spin([(coef, kind)], ra, rb, saa, sab, sbb)
    return e


def eval_xc_mgga(terms, rho, sigma, tau):
    """closed shell: (e, de/drho, de/dsigma, de/dtau) by complex steps."""
    h = 1e-30

    def f(r, s, t):
        return energy_density_mgga_spin(terms, r / 2, r / 2, s / 4, s / 4, s / 4, t / 2, t / 2)
    e = np.real(f(rho + 0j, sigma + 0j, tau + 0j))
    vr = np.imag(f(rho + 1j * h, sigma + 0j, tau + 0j)) / h
    vs = np.imag(f(rho + 0j, sigma + 1j * h, tau + 0j)) / h
    vt = np.imag(f(rho + 0j, sigma + 0j, tau + 1j * h)) / h
    return e, vr, vs, vt


def eval_xc_mgga_spin(terms, ra, rb, saa, sab, sbb, ta, tb):
    """(e, [7 partial derivatives]) by complex steps."""
    h = 1e-30
    args = [np.asarray(a, dtype=complex) for a in (ra, rb, saa, sab, sbb, ta, tb)]
    e = np.real(energy_density_mgga_spin(terms, *args))
    ds = []
    for i in range(7):
        a2 = [a + (1j * h if j == i else 0) for j, a in enumerate(args)]
        ds.append(np.imag(energy_density_mgga_spin(terms, *a2)) / h)
    return e, ds


def nr_rks_mgga(mol, coords, weights, xc, dm, block=20000, rho_cut=1e-10):
    hyb, terms = parse_xc_mgga(xc)
    nao = mol.nao
    nelec = exc = 0.0
    vmat = np.zeros((nao, nao))
    for p0 in range(0, len(coords), block):
        c, w = coords[p0:p0 + block], weights[p0:p0 + block]
        ao = eval_ao(mol, c, 1)
        c0 = ao[0] @ dm
        ck = [ao[1 + k] @ dm for k in range(3)]
        rho = np.einsum("gi,gi->g", ao[0], c0)
        grad = np.array([2 * np.einsum("gi,gi->g", ao[1 + k], c0) for k in range(3)])
        tau = 0.5 * sum(np.einsum("gi,gi->g", ao[1 + k], ck[k]) for k in range(3))
        sigma = (grad * grad).sum(axis=0)
        ok = rho > rho_cut
        e, vr, vs, vt = eval_xc_mgga(terms, np.where(ok, rho, 1.0), np.where(ok, sigma, 0.0), np.where(ok, tau, 1.0))
        e, vr, vs, vt = (np.where(ok, x, 0) for x in (e, vr, vs, vt))
        nelec += float(w @ rho)
        exc += float(w @ e)
        aow = ao[0] * (0.5 * w * vr)[:, None]
        for k in range(3):
            aow += ao[1 + k] * (2 * w * vs * grad[k])[:, None]
        vmat += ao[0].T @ aow
        for k in range(3):
            vmat += ao[1 + k].T @ (ao[1 + k] * (0.25 * w * vt)[:, None])
    return nelec, exc, vmat + vmat.T, hyb


def rks_mgga(mol, xc="TPSS", level=3, dm0=None, conv_tol=1e-9, max_cycle=50, verbose=False):
    """Closed-shell meta-GGA Kohn-Sham on the oracle integrals (no small-density grid pruning: compare with
    `mf.small_rho_cutoff = 0`)."""
    coords, weights = build_grids(mol, level)
    o = orc.Oracle(mol)
    info = {}

    def veff(dm):
        n, exc, vxc, hyb = nr_rks_mgga(mol, coords, weights, xc, dm)
        J, K = o.jk(dm)
        info["nelec"] = n
        v = J + vxc
        e2 = 0.5 * float(np.sum(dm * J)) + exc
        if hyb:
            v = v - 0.5 * hyb * K
            e2 -= 0.25 * hyb * float(np.sum(dm * K))
        return v, e2

    r = orc.rhf(mol, dm0=dm0, conv_tol=conv_tol, max_cycle=max_cycle, veff_fn=veff, verbose=verbose, oracle=o)
    r["nelec_grid"] = info.get("nelec")
    r["ngrids"] = len(weights)
    return r
