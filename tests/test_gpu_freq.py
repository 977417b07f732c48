"""Vibrational analysis (call sites `templates/optimize_geometry.py:112-147`,
`templates/opt-freq.py:387-417,458,499-506`): semi-numerical Hessian from the analytic HIP gradient, harmonic analysis and
RRHO thermochemistry.  Checked against second differences of the ENERGY, rigid-body invariances, and the experimental-scale
sanity of water's RHF/6-31G(d) frequencies (literature HF/6-31G(d): 1827, 4070, 4189 cm-1 [MEM], +-1 %)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_water_hessian_frequencies_and_thermo():
    from pyscf import gto, scf, hessian
    from pyscf.hessian import thermo
    from pyscf.geomopt.geometric_solver import optimize
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", "6-31G(d)", 0
    mol.build()
    mf = scf.RHF(mol).to_gpu()
    mol_eq = optimize(mf, maxsteps=50)
    mf = scf.RHF(mol_eq)
    mf.conv_tol = 1e-11
    mf.kernel()
    h = hessian.RHF(mf)                       # optimize_geometry.py:117-118
    hess = h.kernel()
    n = mol_eq.natm
    assert hess.shape == (n, n, 3, 3)
    H = hess.transpose(0, 2, 1, 3).reshape(3 * n, 3 * n)
    assert np.abs(H - H.T).max() < 1e-10
    # translational invariance: sum over atoms of each row block vanishes
    assert np.abs(hess.sum(axis=1)).max() < 2e-5
    # one diagonal element against a second difference of the SCF energy
    R = mol_eq.atom_coords()
    d = 1e-2
    es = []
    for s in (-1, 0, 1):
        Rd = R.copy(); Rd[0, 2] += s * d
        m2 = mol_eq.set_geom_(Rd, unit="Bohr", inplace=False); m2.verbose = 0
        f2 = scf.RHF(m2); f2.conv_tol = 1e-12
        es.append(f2.kernel())
    assert abs((es[0] - 2 * es[1] + es[2]) / d ** 2 - hess[0, 0, 2, 2]) < 2e-4
    info = thermo.harmonic_analysis(mf.mol, hess)      # optimize_geometry.py:125
    freq = info["freq_wavenumber"]
    assert len(freq) == 3 and np.all(freq > 0)
    for got, ref in zip(freq, (1827.0, 4070.0, 4189.0)):
        assert abs(got - ref) < 0.012 * ref, freq
    assert info["norm_mode"].shape == (3, n, 3)
    res = thermo.thermo(mf, info["freq_au"], 298.15, 101325)    # opt-freq.py:499
    zpe = res["ZPE"][0]
    assert abs(zpe - 0.5 * info["freq_au"].sum()) < 1e-12 and 0.020 < zpe < 0.026
    assert res["sym_number"][0] == 2
    assert abs(res["H_tot"][0] - (res["E_tot"][0] + 298.15 * 3.166811563e-6)) < 1e-9
    s_cal = res["S_tot"][0] * 627.509 * 1000.0
    assert 44.0 < s_cal < 46.5                 # experimental S(H2O, g, 298 K) = 45.1 cal/mol/K
    assert abs(res["G_tot"][0] - (res["H_tot"][0] - 298.15 * res["S_tot"][0])) < 1e-12


def test_gpu4pyscf_hessian_surface_rks():
    """`gpu_hessian.rks.Hessian(mf_opt).kernel()` (opt-freq.py:392-395) on H2 B3LYP: one stretching mode."""
    import gpu4pyscf
    from gpu4pyscf import hessian as gpu_hessian
    from pyscf import gto, hessian
    from pyscf.hessian import thermo
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = "H 0 0 0; H 0 0 0.743", "6-31G(d,p)", 0
    mol.build()
    mf = gpu4pyscf.dft.RKS(mol).to_gpu()
    mf.xc = "B3LYP"
    mf.kernel()
    hess = gpu_hessian.rks.Hessian(mf).kernel()
    assert hessian.rks.Hessian is gpu_hessian.rks.Hessian
    info = thermo.harmonic_analysis(mol, hess)
    assert info["rotor_type"] == "LINEAR" and len(info["freq_wavenumber"]) == 1
    assert 4300.0 < info["freq_wavenumber"][0] < 4600.0      # B3LYP H2 stretch ~ 4450 cm-1


def test_opt_freq_template_dipole_derivative_call_sequence():
    """The numerical IR fallback of `templates/opt-freq.py:186-262`: `mol.atom_coords(unit='Bohr')`, `mol.copy()`,
    `mol_plus.set_geom_(coords, unit='Bohr')`, `isinstance(mf, (dft.rks.RKS, dft.uks.UKS))`, `dft.RKS(mol_plus)`,
    `mf_plus.kernel(dm0=dm0)`, `dip_moment(unit='au')` -- one displacement pair on water B3LYP/6-31G(d); the O-H stretch
    direction must change the dipole (d mu / d z_H ~ 0.1-0.4 e)."""
    from pyscf import gto, dft
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", "6-31G(d)", 0
    mol.build()
    mf = dft.RKS(mol)
    mf.xc = "B3LYP"
    mf.kernel()
    assert isinstance(mf, (dft.rks.RKS, dft.uks.UKS))
    mu0 = mf.dip_moment(unit="au")
    assert mu0.shape == (3,) and 0.7 < np.linalg.norm(mu0) < 0.95          # ~2.1 Debye
    coords = mol.atom_coords(unit="Bohr")
    dm0 = mf.make_rdm1()
    delta = 0.001
    mus = []
    for sgn in (+1, -1):
        c2 = coords.copy()
        c2[1, 2] += sgn * delta
        m2 = mol.copy()
        m2.set_geom_(c2, unit="Bohr")
        assert np.allclose(m2.atom_coords(), c2) and np.allclose(mol.atom_coords(), coords)
        f2 = dft.RKS(m2)
        f2.xc = mf.xc
        f2.verbose = 0
        f2.kernel(dm0=dm0)
        mus.append(f2.dip_moment(unit="au"))
    dmu = (mus[0] - mus[1]) / (2 * delta)
    assert np.all(np.isfinite(dmu)) and 0.05 < abs(dmu[2]) < 0.6 and abs(dmu[0]) < 1e-6


def test_infrared_module_water_intensities():
    """`infrared.RHF(mf).kernel()`, `.summary()`, `.freq_info`, `.ir_intensity` (calculate_ir_spectrum.py:90-105).  Water
    RHF/6-31G(d): literature harmonic IR intensities 107 / 18 / 58 km/mol for the bend / symmetric / antisymmetric stretch
    [MEM, CCCBDB HF/6-31G*], +-15 %."""
    from pyscf import gto, scf
    from pyscf.prop import infrared
    from pyscf.geomopt.geometric_solver import optimize
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", "6-31G(d)", 0
    mol.build()
    mol_eq = optimize(scf.RHF(mol).to_gpu(), maxsteps=50)
    mf = scf.RHF(mol_eq)
    mf.conv_tol = 1e-11
    mf.kernel()
    ir = infrared.RHF(mf)
    ir.kernel()
    ir.summary()
    freq = ir.freq_info["freq_wavenumber"]
    inten = ir.ir_intensity
    assert len(freq) == 3 and inten.shape == (3,)
    for got, ref in zip(inten, (107.0, 18.0, 58.0)):
        assert abs(got - ref) < 0.15 * ref + 1.0, (freq, inten)


def test_benzene_b3lyp_frequencies_ir_and_thermo_against_literature():
    """Benzene B3LYP/6-31G(d), optimised with `optimize()`: harmonic frequencies (literature 414 / 622 ... 3212 cm-1), the
    dominant IR band (out-of-plane C-H bend, 694 cm-1, ~75 km/mol), ZPE 63.3 kcal/mol, sigma = 12, S(298 K) = 64.3 cal/mol/K
    [MEM].  Exercises the whole chain: RKS, analytic gradient, internal-coordinate optimiser, semi-numerical Hessian,
    dipole derivatives, harmonic analysis, RRHO."""
    from pyscf import gto, dft
    from pyscf.hessian import thermo
    from pyscf.prop import infrared
    from pyscf.geomopt.geometric_solver import optimize
    from mi355scf import fixtures
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = fixtures.BENZENE, "6-31G*", 0
    mol.build()
    mf = dft.RKS(mol)
    mf.xc = "B3LYP"
    mol_eq = optimize(mf, maxsteps=50)
    mf = dft.RKS(mol_eq)
    mf.xc, mf.conv_tol = "B3LYP", 1e-11
    mf.kernel()
    ir = infrared.RKS(mf)
    info = ir.kernel()
    f = info["freq_wavenumber"]
    assert len(f) == 30 and (f > 0).all()
    assert abs(f[0] - 414.0) < 6.0 and abs(f[2] - 622.0) < 6.0 and abs(f[-1] - 3212.0) < 12.0
    k = int(np.argmax(ir.ir_intensity))
    assert abs(f[k] - 694.0) < 8.0 and 60.0 < ir.ir_intensity[k] < 95.0
    t = thermo.thermo(mf, info["freq_au"], 298.15, 101325)
    assert abs(t["ZPE"][0] * 627.509 - 63.3) < 0.5
    assert t["sym_number"][0] == 12
    assert abs(t["S_tot"][0] * 627509 - 64.3) < 1.0


def test_water_hessian_matches_second_differences_of_the_oracle_energy():
    """Parity (not self-consistency): every element of the semi-numerical HIP Hessian of H2O RHF/6-31G against second
    differences of the CPU ORACLE's SCF energy (McMurchie-Davidson integrals, numpy SCF: a different code path end to end).
    H_ij = [E(+i+j) - E(+i-j) - E(-i+j) + E(-i-j)] / 4 h^2 with h = 0.005 Bohr (truncation ~h^2 E'''' / 6 ~ 2e-5 -- h = 0.02 gave
    3e-4 -- energy noise 1e-12 / h^2 ~ 4e-8), diagonal by the three-point formula; tolerance 5e-5 Hartree/Bohr^2 on elements of
    order 0.1-0.7."""
    from pyscf import gto, scf, hessian
    from oracle import oracle as orc
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = "O 0 0 0.1; H 0 -0.76 0.59; H 0 0.74 0.60", "6-31G", 0   # no symmetry: every element distinct
    mol.build()
    mf = scf.RHF(mol)
    mf.conv_tol = 1e-12
    mf.kernel()
    hess = hessian.RHF(mf).kernel()
    n = mol.natm
    H = hess.transpose(0, 2, 1, 3).reshape(3 * n, 3 * n)
    R = mol.atom_coords().ravel()
    h = 0.005

    def e_orc(dx):
        m2 = mol.set_geom_((R + dx).reshape(-1, 3), unit="Bohr", inplace=False)
        m2.verbose = 0
        r = orc.rhf(m2, conv_tol=1e-13, max_cycle=100)
        assert r["converged"]
        return r["e_tot"]

    e0 = e_orc(np.zeros_like(R))
    assert abs(e0 - mf.e_tot) < 1e-8
    Ho = np.zeros_like(H)
    ep, em = np.zeros(3 * n), np.zeros(3 * n)
    for i in range(3 * n):
        d = np.zeros_like(R); d[i] = h
        ep[i], em[i] = e_orc(d), e_orc(-d)
        Ho[i, i] = (ep[i] - 2.0 * e0 + em[i]) / h ** 2
    for i in range(3 * n):
        for j in range(i):
            di = np.zeros_like(R); di[i] = h
            dj = np.zeros_like(R); dj[j] = h
            Ho[i, j] = Ho[j, i] = (e_orc(di + dj) - e_orc(di - dj) - e_orc(dj - di) + e_orc(-di - dj)) / (4.0 * h ** 2)
    assert np.abs(Ho).max() > 0.3
    assert np.abs(H - Ho).max() < 5e-5, np.abs(H - Ho).max()
