"""`pyscf.hessian.thermo`: `harmonic_analysis(mol, hess)` and `thermo(model, freq_au, temperature, pressure)` (call sites
`templates/optimize_geometry.py:125-147`, `templates/opt-freq.py:15,458,499-506`).  Host NumPy; rigid-rotor /
harmonic-oscillator / ideal-gas formulas as in PySCF's module [MEM]: results are dicts of `(value, unit)` tuples
(`thermo_results['ZPE'][0]` is how the template reads them).  Imaginary modes are returned as NEGATIVE real numbers
(the templates test `frequencies < 0`).
"""
import numpy as np

# CODATA 2018
HARTREE2J = 4.3597447222071e-18
BOHR_M = 5.29177210903e-11
AMU_KG = 1.66053906660e-27
ME_KG = 9.1093837015e-31
KB = 1.380649e-23
PLANCK = 6.62607015e-34
C_LIGHT = 299792458.0
AU2HZ = HARTREE2J / PLANCK                     # omega in Eh/hbar -> nu = omega_au * Eh / h
AMU2AU = AMU_KG / ME_KG


def _projector_trans_rot(mass, coords):
    """Orthonormal basis (mass-weighted Cartesians) of the translations and rotations; returns (basis[3N, k], rotor type)."""
    n = len(mass)
    sm = np.sqrt(mass)
    com = (mass[:, None] * coords).sum(axis=0) / mass.sum()
    r = coords - com
    vecs = []
    for x in range(3):
        v = np.zeros((n, 3)); v[:, x] = sm
        vecs.append(v.ravel())
    inertia = np.zeros((3, 3))
    for m, ri in zip(mass, r):
        inertia += m * (np.dot(ri, ri) * np.eye(3) - np.outer(ri, ri))
    w, ax = np.linalg.eigh(inertia)
    for k in range(3):
        v = np.cross(ax[:, k][None, :], r) * sm[:, None]
        vecs.append(v.ravel())
    V = np.array(vecs).T
    # orthonormalise, dropping null vectors (atoms: no rotations; linear molecules: one rotation missing)
    u, s, _ = np.linalg.svd(V, full_matrices=False)
    keep = s > 1e-6 * max(s.max(), 1e-30)
    if n == 1:
        rotor = "ATOM"
    elif w[0] < 1e-6 * max(w[2], 1e-30):
        rotor = "LINEAR"
    else:
        rotor = "NONLINEAR"
    return u[:, keep], rotor, w


def harmonic_analysis(mol, hess, exclude_trans=True, exclude_rot=True, imaginary_freq=True, mass=None):
    if mass is None:
        # isotope-averaged masses, like `thermo()` below and PySCF's `harmonic_analysis` default [MEM; parity unpinned]
        mass = mol.atom_mass_list(isotope_avg=True)
    mass = np.asarray(mass, dtype=float)
    if np.any(mass <= 0.0):
        raise ValueError("harmonic_analysis: massless centres (ghost atoms) cannot be mass-weighted; pass `mass=` explicitly")
    n = mol.natm
    coords = mol.atom_coords()
    H = np.asarray(hess)
    if H.ndim == 4:
        H = H.transpose(0, 2, 1, 3).reshape(3 * n, 3 * n)
    sm = np.repeat(np.sqrt(mass), 3)
    Hm = H / np.outer(sm, sm)                  # Eh / (Bohr^2 amu)
    B, rotor, _ = _projector_trans_rot(mass, coords)
    if exclude_trans or exclude_rot:
        P = np.eye(3 * n) - B @ B.T
        Hm = P @ Hm @ P
    w, v = np.linalg.eigh(0.5 * (Hm + Hm.T))
    # drop the projected-out (zero) modes: the len(B.T) eigenvalues of smallest magnitude
    order = np.argsort(np.abs(w))
    nzero = B.shape[1] if (exclude_trans or exclude_rot) else 0
    keep = np.sort(order[nzero:])
    w, v = w[keep], v[:, keep]
    idx = np.argsort(w)
    w, v = w[idx], v[:, idx]
    force_const_au = w / AMU2AU                 # Eh / (Bohr^2 m_e)
    freq_au = np.sign(force_const_au) * np.sqrt(np.abs(force_const_au))
    freq_wn = freq_au * AU2HZ / C_LIGHT / 100.0
    modes = (v / sm[:, None]).T.reshape(-1, n, 3)          # Cartesian displacements
    red_mass = 1.0 / np.einsum("kix,kix->k", modes, modes)
    modes = modes * np.sqrt(red_mass)[:, None, None]        # normalised Cartesian modes, as PySCF prints them
    return {"freq_error": 0, "freq_au": freq_au, "freq_wavenumber": freq_wn, "norm_mode": modes,
            "reduced_mass": red_mass, "vib_temperature": freq_au * HARTREE2J / KB,
            "force_const_au": force_const_au, "force_const_dyne": red_mass * (2 * np.pi * freq_wn * 100 * C_LIGHT) ** 2 * AMU_KG * 1e-2,
            "rotor_type": rotor}


def rotational_symmetry_number(mol, tol=1e-3, max_atoms=40):
    """Number of proper rotations mapping the nuclear framework onto itself (1 if nothing is found or the molecule is too
    large for the brute-force search); linear molecules: 2 with an inversion centre, else 1."""
    z = np.asarray(mol.atom_charges())
    mass = mol.atom_mass_list(isotope_avg=True)
    R = mol.atom_coords()
    n = len(z)
    if n == 1 or n > max_atoms:
        return 1
    com = (mass[:, None] * R).sum(axis=0) / mass.sum()
    r = R - com

    def maps_onto_itself(U):
        rr = r @ U.T
        for i in range(n):
            d = np.linalg.norm(r - rr[i], axis=1)
            j = np.argmin(d)
            if d[j] > tol * max(1.0, np.abs(r).max()) or z[j] != z[i]:
                return False
        return True

    _, rotor, _ = _projector_trans_rot(mass, R)
    if rotor == "LINEAR":
        return 2 if maps_onto_itself(-np.eye(3)) else 1
    cand = [v for v in r if np.linalg.norm(v) > 1e-6]
    for i in range(n):
        for j in range(i):
            m = 0.5 * (r[i] + r[j])
            if np.linalg.norm(m) > 1e-6:
                cand.append(m)
            c = np.cross(r[i], r[j])
            if np.linalg.norm(c) > 1e-6:
                cand.append(c)
    axes = []
    for v in cand:
        v = v / np.linalg.norm(v)
        if not any(abs(abs(v @ a) - 1.0) < 1e-6 for a in axes):
            axes.append(v)
    ops = [np.eye(3)]
    for a in axes:
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        for order in (2, 3, 4, 5, 6):
            for k in range(1, order):
                t = 2 * np.pi * k / order
                U = np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)
                if maps_onto_itself(U) and not any(np.abs(U - O).max() < 1e-6 for O in ops):
                    ops.append(U)
    return len(ops)


def thermo(model, freq, temperature=298.15, pressure=101325):
    """RRHO thermochemistry of an SCF model at its (stationary) geometry: dict of (value, unit) tuples [MEM: PySCF keys]."""
    mol = getattr(model, "mol", model)
    T, P = float(temperature), float(pressure)
    kT = KB * T / HARTREE2J                                    # Eh
    mass = mol.atom_mass_list(isotope_avg=True)
    coords = mol.atom_coords()
    _, rotor, inertia = _projector_trans_rot(mass, coords)
    e0 = float(getattr(model, "e_tot", 0.0) or 0.0)
    out = {"temperature": (T, "K"), "pressure": (P, "Pa"), "E0": (e0, "Eh")}
    # electronic
    mult = getattr(mol, "spin", 0) + 1
    s_elec = KB / HARTREE2J * np.log(mult)
    # translation
    m_kg = mass.sum() * AMU_KG
    q_trans = (2 * np.pi * m_kg * KB * T / PLANCK ** 2) ** 1.5 * KB * T / P
    s_trans = KB / HARTREE2J * (2.5 + np.log(q_trans))
    e_trans = 1.5 * kT
    # rotation (moments in amu Bohr^2 -> rotational constants in Eh)
    sigma = rotational_symmetry_number(mol)
    if rotor == "ATOM":
        e_rot = s_rot = 0.0
        rot_const = np.zeros(0)
    else:
        I = inertia[inertia > 1e-6 * inertia.max()] * AMU2AU   # m_e Bohr^2
        rot_const = 0.5 / I                                   # Eh
        if rotor == "LINEAR":
            e_rot = kT
            s_rot = KB / HARTREE2J * (1.0 + np.log(kT / (sigma * rot_const[-1])))
        else:
            e_rot = 1.5 * kT
            s_rot = KB / HARTREE2J * (1.5 + np.log(np.sqrt(np.pi) / sigma * np.prod(np.sqrt(kT / rot_const))))
    # vibration (real modes only)
    f = np.asarray(freq, dtype=float)
    f = f[f > 0]
    zpe = 0.5 * f.sum()
    x = f / kT if T > 0 else np.full_like(f, np.inf)
    e_vib = zpe + float(np.sum(f / np.expm1(x))) if len(f) else 0.0
    s_vib = KB / HARTREE2J * float(np.sum(x / np.expm1(x) - np.log1p(-np.exp(-x)))) if len(f) else 0.0
    s_tot = s_elec + s_trans + s_rot + s_vib
    e_tot = e0 + e_trans + e_rot + e_vib
    h_tot = e_tot + kT
    out.update({"ZPE": (zpe, "Eh"), "E_0K": (e0 + zpe, "Eh"),
                "S_elec": (s_elec, "Eh/K"), "S_trans": (s_trans, "Eh/K"), "S_rot": (s_rot, "Eh/K"), "S_vib": (s_vib, "Eh/K"),
                "S_tot": (s_tot, "Eh/K"), "E_trans": (e_trans, "Eh"), "E_rot": (e_rot, "Eh"), "E_vib": (e_vib, "Eh"),
                "E_tot": (e_tot, "Eh"), "H_tot": (h_tot, "Eh"), "G_tot": (h_tot - T * s_tot, "Eh"),
                "sym_number": (sigma, ""), "rot_const": (rot_const * HARTREE2J / PLANCK / 1e9, "GHz")})
    return out
