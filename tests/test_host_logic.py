"""CPU tests of the host-side logic of the product (no GPU, no engine calls): Mole parsing/packing, basis
normalisation, XC name parsing, grid tables, SMILES fixtures, optimiser model Hessian, loud failure without a GPU."""
import numpy as np
import pytest

from conftest import MOLECULES


def test_mole_parsing_units_and_counts():
    from mi355scf.mole import Mole, BOHR
    m1 = Mole(atom="O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", basis="6-31G(d)").build()
    m2 = Mole(atom="""
        O  0.0  0.0   0.0
        H  0.0 -0.757 0.587
        H  0.0  0.757 0.587
    """, basis="6-31g*", unit="Angstrom").build()
    assert m1.nao == m2.nao == 2 + 2 + 14 and m1.nelectron == 10 and m1.natm == 3
    assert np.allclose(m1.atom_coords(), m2.atom_coords()) and np.allclose(m1.atom_coords()[1] * BOHR, [0, -0.757, 0.587])
    m3 = m1.set_geom_(m1.atom_coords() * 1.01, unit="Bohr", inplace=False)
    assert np.allclose(m3.atom_coords(), m1.atom_coords() * 1.01) and m3 is not m1
    with pytest.raises(RuntimeError):
        Mole(atom="O 0 0 0; H 0 0 1", basis="sto-3g").build()          # odd electron count with spin 0
    with pytest.raises(KeyError):
        Mole(atom="O 0 0 0", basis="no-such-basis").build()
    # BASELINE config sizes
    from mi355scf import fixtures
    assert Mole(atom=fixtures.BENZENE, basis="cc-pVTZ").build().nao == 264
    assert Mole(atom=fixtures.H2CO, basis="6-31G(d)").build().nao == 32


def test_contracted_functions_are_normalised():
    """<chi|chi> = 1 for every packed shell (radial integral with the stored coefficients)."""
    from mi355scf.mole import Mole, gaussian_int
    mol = Mole(atom=MOLECULES["h2co"], basis="cc-pvtz").build()
    for b in mol._bas:
        l, n, pe, pc = int(b[1]), int(b[2]), int(b[5]), int(b[6])
        e, c = mol._env[pe:pe + n], mol._env[pc:pc + n]
        s = sum(ci * cj * gaussian_int(2 * l + 2, ai + aj) for ci, ai in zip(c, e) for cj, aj in zip(c, e))
        assert abs(s - 1.0) < 1e-12


def test_xc_name_parsing():
    from mi355scf.dft import parse_xc
    hyb, terms, gga = parse_xc("B3LYP")
    assert hyb == 0.2 and gga and abs(sum(c for c, _ in terms) - (0.08 + 0.72 + 0.19 + 0.81)) < 1e-15
    assert parse_xc("pbe")[0] == 0.0 and parse_xc("PBE0")[0] == 0.25 and not parse_xc("lda,vwn")[2]
    hyb, terms, level = parse_xc("M06-2X")          # templates/calculate_energy.py:263; calculate_bde.py:105 default
    assert hyb == 0.54 and level == 2 and parse_xc("TPSS")[2] == 2 and parse_xc("B3LYP")[2] == 1
    with pytest.raises(NotImplementedError):
        parse_xc("SCAN")


def test_oracle_meta_gga_limits():
    """The numpy restatement of TPSS / M06-2X (oracle/dft.py; the M06-2X tables are unverified-memory) obeys what the papers
    guarantee: both reduce to LSDA (+ their share of exact exchange) for the uniform gas; their correlation vanishes for any
    one-electron density; TPSS exchange gives the exact -0.3125 Ha for the hydrogen atom."""
    from oracle import dft as od
    r = np.array([0.05, 0.3, 1.0, 5.0])
    tu = 0.3 * (3 * np.pi ** 2) ** (2.0 / 3) * r ** (5.0 / 3)
    for name in ("TPSS", "M06-2X"):
        hyb, terms = od.parse_xc_mgga(name)
        e = od.eval_xc_mgga(terms, r, 0 * r, tu)[0]
        lsda = (1 - hyb) * (-0.75 * (3 / np.pi) ** (1.0 / 3) * r ** (4.0 / 3)) + r * od._pw92_eps_spin(r / 2, r / 2)
        assert np.abs(e / lsda - 1).max() < 1e-12
    assert abs(od.M062X_A[0] + od.M062X_HYB - 1) < 1e-12 and abs(od.M062X_CSS[0] + od.M062X_DSS[0] - 1) < 1e-7 \
        and abs(od.M062X_CAB[0] + od.M062X_DAB[0] - 1) < 1e-7
    rr = np.linspace(1e-4, 30, 200001)
    dr = rr[1] - rr[0]
    rho = np.exp(-2 * rr) / np.pi
    sig = (2 * rho) ** 2
    tau = sig / (8 * rho)                              # one electron: tau = tau_W
    vol = 4 * np.pi * rr ** 2 * dr
    assert abs(np.sum(vol * od._tpss_x_spin(rho, 0 * rho, sig, 0 * sig, tau, 0 * tau)) + 0.3125) < 2e-6
    assert abs(np.sum(vol * od._tpss_c_spin(rho, 0 * rho, sig, 0 * sig, 0 * sig, tau, 0 * tau))) < 1e-12
    assert abs(np.sum(vol * od._m062x_c_spin(rho, 0 * rho, sig, 0 * sig, tau, 0 * tau))) < 1e-12
    ex = np.sum(vol * od._m062x_x_spin(rho, 0 * rho, sig, 0 * sig, tau, 0 * tau)) + od.M062X_HYB * (-0.3125)
    assert -0.3125 - 0.01 < ex < -0.3125 + 0.01


def test_grid_tables():
    from mi355scf import grids
    r, dr = grids.radial_treutler_ahlrichs(75, 6)
    assert len(r) == 75 and np.all(np.diff(r) > 0) and np.all(dr > 0)
    # radial rule integrates exp(-r^2) r^2 -> sqrt(pi)/4
    assert abs(np.sum(np.exp(-r * r) * r * r * dr) - np.sqrt(np.pi) / 4) < 1e-8
    for n in (50, 86, 266, 302):
        x, w = grids.lebedev(n)
        assert x.shape == (n, 3) and abs(w.sum() - 1) < 1e-13 and np.abs(np.linalg.norm(x, axis=1) - 1).max() < 1e-13
        assert abs((w * x[:, 0] ** 2 * x[:, 1] ** 2).sum() - 1.0 / 15) < 1e-13      # exact for degree-4 polynomials
    angs = grids.prune_nwchem(6, r, 302)
    assert set(angs.tolist()) <= {50, 86, 266, 302} and angs[0] == 50 and 302 in angs


def test_smiles_fixtures_and_rdkit_standin():
    from rdkit import Chem
    from rdkit.Chem import Descriptors
    m = Chem.MolFromSmiles("CC(C)Cc1ccc(cc1)C(C)C(=O)O")
    assert Chem.rdMolDescriptors.CalcMolFormula(m) == "C13H18O2" and abs(Descriptors.MolWt(m) - 206.28) < 0.05
    mh = Chem.AddHs(m)
    assert mh.GetNumAtoms() == 33 and m.GetNumAtoms() == 15
    c60 = Chem.AddHs(Chem.MolFromSmiles("C60")).GetConformer().GetPositions()
    d = np.linalg.norm(c60[:, None] - c60[None], axis=2) + np.eye(60) * 9
    assert c60.shape == (60, 3) and abs(d.min() - 1.43) < 1e-6 and np.all((d < 1.44).sum(axis=1) == 3)
    with pytest.raises(NotImplementedError):
        Chem.MolFromSmiles("N#N")


def test_model_hessian_is_positive_and_invariant():
    from mi355scf.geomopt import model_hessian
    from mi355scf.mole import Mole
    mol = Mole(atom=MOLECULES["c2h4"], basis="sto-3g").build()
    x = mol.atom_coords()
    H = model_hessian(mol, x)
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() > 0
    t = np.tile([1.0, 0, 0], mol.natm)                   # a rigid translation only feels the small diagonal shift
    assert abs(t @ (H - 0.02 * np.eye(len(t))) @ t) < 1e-6


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pyscf import gto, scf
    from mi355scf.engine import EngineError
    mol = gto.M(atom="H 0 0 0; H 0 0 0.74", basis="sto-3g", verbose=0)
    with pytest.raises(EngineError):
        scf.RHF(mol).kernel()


def _fixture_mol(smiles):
    from mi355scf import smiles_fixtures as sf
    from mi355scf.mole import Mole
    sym, xyz = sf.TABLE[smiles]()
    return Mole(atom="; ".join(f"{s} {a} {b} {c}" for s, (a, b, c) in zip(sym, xyz)), basis="sto-3g", verbose=0).build()


def test_internal_coordinates_bmatrix_and_rank():
    """Wilson B rows against central differences; the primitive set spans all 3N-6 internal motions; the
    iterative back-transformation reproduces a projected internal step."""
    from mi355scf.internals import Internals
    rng = np.random.default_rng(1)
    for smiles in ("CC(C)Cc1ccc(cc1)C(C)C(=O)O", "C=O", "c1ccccc1", "O"):
        mol = _fixture_mol(smiles)
        x = mol.atom_coords() + rng.normal(scale=0.02, size=(mol.natm, 3))
        ic = Internals(mol.atom_charges(), x)
        B = ic.bmatrix(x)
        Bn = np.zeros_like(B)
        h = 1e-5
        for a in range(x.size):
            xp = x.copy().ravel(); xm = xp.copy()
            xp[a] += h; xm[a] -= h
            Bn[:, a] = ic.diff(ic.values(xp.reshape(-1, 3)), ic.values(xm.reshape(-1, 3))) / (2 * h)
        assert np.abs(B - Bn).max() < 1e-8
        Ginv, P, rank = ic.ginv(B)
        assert rank == 3 * mol.natm - 6
        assert np.abs(P @ P - P).max() < 1e-8
        dq = P @ rng.normal(scale=0.02, size=ic.nq)
        x2, got = ic.to_cartesian(x, dq)
        assert np.abs(got - dq).max() < 2e-3
        assert np.abs(ic.diff(ic.values(x2), ic.values(x)) - got).max() < 1e-12


def test_optimizers_reach_the_same_minimum_on_a_model_surface():
    """Internal-coordinate and Cartesian BFGS on a toy valence force field (ethanol): same minimum, both within
    the geomeTRIC-default thresholds; linear molecules fall back to Cartesians."""
    from mi355scf import geomopt
    from mi355scf.internals import Internals
    from mi355scf.mole import Mole
    mol = _fixture_mol("CCO")
    ic = Internals(mol.atom_charges(), mol.atom_coords())
    rng = np.random.default_rng(3)
    q0 = ic.values(mol.atom_coords())
    kf = np.array([{"bond": 0.6, "angle": 0.25, "dihedral": 0.02}[k] for k in ic.kinds])
    q0 = q0 + np.where(np.array(ic.kinds) == "dihedral", 0.0, rng.normal(scale=0.06, size=ic.nq))

    def energy(x):
        d = ic.diff(ic.values(x), q0)
        return 0.5 * float(np.sum(kf * d * d))

    calls = [0]

    def energy_grad(m):
        calls[0] += 1
        x = m.atom_coords()
        g = np.zeros_like(x)
        for a in range(x.shape[0]):
            for c in range(3):
                xp = x.copy(); xm = x.copy()
                xp[a, c] += 1e-5; xm[a, c] -= 1e-5
                g[a, c] = (energy(xp) - energy(xm)) / 2e-5
        return energy(x), g

    m1, ok1, n1 = geomopt.optimize_internal(energy_grad, mol, 100)
    m2, ok2, n2 = geomopt.optimize_cartesian(energy_grad, mol, 100)
    assert ok1 and ok2 and n1 <= n2 + 2
    assert abs(energy(m1.atom_coords()) - energy(m2.atom_coords())) < 2e-6
    co2 = Mole(atom="O 0 0 -1.16; C 0 0 0; O 0 0 1.16", basis="sto-3g", verbose=0).build()
    assert geomopt.optimize_internal(energy_grad, co2, 5) is None


def test_initial_hessian_models_of_the_optimiser():
    """`Internals.guess_hessian_diag`: Lindh's model (CPL 241, 423 (1995)) at known distances, the out-of-plane primitives of
    three-coordinate centres, and -- on a model surface with soft rotors (torsion constants of 0.005 a.u.) -- no more steps
    than the stiff round-1 constants need (on ibuprofen B3LYP/def2-TZVP: 12 instead of 22)."""
    from mi355scf import geomopt
    from mi355scf.internals import Internals
    mol = _fixture_mol("CC(=O)O")
    x = mol.atom_coords()
    ic = Internals(mol.atom_charges(), x)
    kinds = np.array(ic.kinds)
    lin, simple, geo = (ic.guess_hessian_diag(x, m) for m in ("lindh", "simple", "geometric"))
    assert set(np.unique(simple)) == {0.5, 0.2, 0.1} and set(np.unique(geo)) == {0.35, 0.16, 0.023}
    z = mol.atom_charges()
    per = lambda q: 0 if q <= 2 else 1
    alpha = {(0, 0): 1.0, (0, 1): 0.3949, (1, 0): 0.3949, (1, 1): 0.28}
    rref = {(0, 0): 1.35, (0, 1): 2.10, (1, 0): 2.10, (1, 1): 2.87}
    for q, (k, a) in enumerate(zip(ic.kinds, ic.atoms)):
        if k == "bond":
            i, j = a
            key = (per(z[i]), per(z[j]))
            r2 = float(((x[i] - x[j]) ** 2).sum())
            assert abs(lin[q] - 0.45 * np.exp(alpha[key] * (rref[key] ** 2 - r2))) < 1e-12
    assert len(ic.oop) == 1                                            # the carboxyl carbon
    tors = [q for q in range(ic.nq) if kinds[q] == "dihedral" and q not in ic.oop]
    assert lin[tors].max() < 0.03 and lin[tors].min() >= 1e-3           # soft torsions
    assert all(0.02 < lin[q] < 0.2 for q in ic.oop)                     # out-of-plane: 0.045 x bond factors, not a torsion value
    assert 0.3 < lin[kinds == "bond"].min() and lin[kinds == "bond"].max() < 1.3 and 0.1 < lin[kinds == "angle"].min()

    big = _fixture_mol("CCO")
    icb = Internals(big.atom_charges(), big.atom_coords())
    rng = np.random.default_rng(5)
    q0 = icb.values(big.atom_coords())
    kb = np.array(icb.kinds)
    kf = np.where(kb == "bond", 0.55, np.where(kb == "angle", 0.2, 0.005))
    q0 = q0 + np.where(kb == "dihedral", rng.normal(scale=0.25, size=icb.nq), rng.normal(scale=0.04, size=icb.nq))

    def energy_grad(m):
        xx = m.atom_coords()
        B = icb.bmatrix(xx)
        d = icb.diff(icb.values(xx), q0)
        return 0.5 * float(np.sum(kf * d * d)), (B.T @ (kf * d)).reshape(-1, 3)

    steps = {}
    old = Internals.HESS_MODEL
    try:
        for model in ("simple", "lindh"):
            Internals.HESS_MODEL = model
            m, ok, n = geomopt.optimize_internal(energy_grad, big, 100)
            assert ok
            steps[model] = n
    finally:
        Internals.HESS_MODEL = old
    assert steps["lindh"] <= steps["simple"], steps


def test_direct_mode_group_plan_cost_model():
    """`SCF._plan_direct_groups`: (tile groups, resident groups) for an ERI tensor that does not fit the HBM.  C60/6-31G* on one
    288 GB GPU (500 GB of tiles, 304 GB free): 4 groups with 1 resident beat 3 streamed groups; with less free memory or the
    Kohn-Sham reserve nothing can stay; the plan never keeps every group and never makes a group larger than what fits."""
    import numpy as np
    from mi355scf.scf import RHF
    from mi355scf.dft import RKS

    def plan(cls, need, free):
        obj = cls.__new__(cls)                      # the method only reads class attributes
        return obj._plan_direct_groups(need, free, int(np.ceil(need / (0.8 * free))))

    assert plan(RHF, 500.0, 304.0) == (4, 1)
    assert plan(RHF, 500.0, 283.0) == (3, 0)
    assert plan(RKS, 500.0, 304.0)[1] == 0          # 10 GB kept for the quadrature: the second 125 GB store no longer fits
    for need, free in ((300.0, 280.0), (1000.0, 280.0), (108.0, 60.0), (2000.0, 250.0)):
        ng, r = plan(RHF, need, free)
        assert 0 <= r < ng and ng >= int(np.ceil(need / (0.8 * free)))
        assert (r + 1) * 1.02 * need / ng <= 0.85 * free          # resident groups + the streaming buffer fit


def test_thermo_symmetry_numbers_and_diatomic_model():
    """`pyscf.hessian.thermo` host logic: rotational symmetry numbers by brute-force rotation search, harmonic analysis of
    a model diatomic Hessian (omega = sqrt(k/mu)), RRHO identities."""
    from pyscf.hessian import thermo
    from mi355scf import fixtures
    from mi355scf.mole import Mole
    cases = (("O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", 2), (fixtures.BENZENE, 12),
             ("C 0 0 0; H 0.629 0.629 0.629; H -0.629 -0.629 0.629; H -0.629 0.629 -0.629; H 0.629 -0.629 -0.629", 12),
             ("N 0 0 0.1; H 0.94 0 -0.27; H -0.47 0.814 -0.27; H -0.47 -0.814 -0.27", 3),
             ("O 0 0 -1.16; C 0 0 0; O 0 0 1.16", 2), ("H 0 0 0; F 0 0 0.92", 1))
    for atom, sigma in cases:
        assert thermo.rotational_symmetry_number(Mole(atom=atom, basis="sto-3g", verbose=0).build()) == sigma
    m = Mole(atom="H 0 0 0; F 0 0 0.92", basis="sto-3g", verbose=0).build()
    k = 0.62
    H = np.zeros((2, 2, 3, 3))
    H[0, 0, 2, 2] = H[1, 1, 2, 2] = k
    H[0, 1, 2, 2] = H[1, 0, 2, 2] = -k
    info = thermo.harmonic_analysis(m, H)
    m1, m2 = m.atom_mass_list(isotope_avg=True)   # the default masses of harmonic_analysis (same as thermo())
    mu = m1 * m2 / (m1 + m2) * 1822.888486209
    assert info["rotor_type"] == "LINEAR" and len(info["freq_au"]) == 1
    assert abs(info["freq_au"][0] - np.sqrt(k / mu)) < 1e-9
    assert abs(info["freq_wavenumber"][0] - np.sqrt(k / mu) * 219474.6313632) < 1e-3
    res = thermo.thermo(m, info["freq_au"], 298.15, 101325)
    assert abs(res["ZPE"][0] - 0.5 * info["freq_au"][0]) < 1e-14
    assert abs(res["G_tot"][0] - (res["H_tot"][0] - 298.15 * res["S_tot"][0])) < 1e-15
    # Sackur-Tetrode check: S_trans of a 20.006 amu ideal gas at 298.15 K, 1 atm = 34.9 cal/mol/K (HF: 34.96)
    assert abs(res["S_trans"][0] * 627.509474 * 1000 - 34.94) < 0.1


def test_diffuse_pople_sets():
    """6-31+G / 6-31++G family (`templates/opt-freq.py:86` defaults to '6-31+G**'): shell counts and name folding."""
    from mi355scf.mole import Mole
    w = "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587"
    assert Mole(atom=w, basis="6-31+G**", verbose=0).build().nao == 28      # O: 4s3p1d = 18, H: 2s1p = 5
    assert Mole(atom=w, basis="6-31++G(d,p)", verbose=0).build().nao == 30
    assert Mole(atom=w, basis="6-31+G(d)", verbose=0).build().nao == 22
    m = Mole(atom=w, basis="6-31+g*", verbose=0).build()
    exps = sorted({float(m._env[m._bas[i, 5]]) for i in range(m.nbas) if m._bas[i, 2] == 1})
    assert abs(exps[0] - 0.0845) < 1e-12                                     # the diffuse sp exponent of oxygen


def test_planned_purification_reaches_the_projector_in_about_twenty_quadratics():
    """`sp2plan.plan` (host logic of row a11): the planned sequence of folded quadratics applied in numpy to a matrix with a
    molecule-like spectrum (cores at -20, valence to -0.33, virtuals from +0.14 to +40) gives the aufbau projector of `eigh` to
    1e-12 with ~20 matrix products, and fails loudly (no plan) when the bounds leave no gap."""
    from mi355scf import sp2plan
    rng = np.random.default_rng(0)
    n, nocc = 240, 28
    e = np.concatenate([np.linspace(-20.6, -11.2, 8), np.linspace(-1.5, -0.33, nocc - 8), np.sort(rng.uniform(0.14, 40.0, n - nocc))])
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    F = (Q * e) @ Q.T
    F = 0.5 * (F + F.T)
    w, V = np.linalg.eigh(F)
    P = V[:, :nocc] @ V[:, :nocc].T
    lo, hi, homo_in, lumo_in = sp2plan.bounds_from_spectrum(w, nocc, 0.15, 2.0)
    assert lo < w[0] and hi > w[-1] and w[nocc - 1] < homo_in < lumo_in < w[nocc]
    coef = sp2plan.plan(lo, hi, homo_in, lumo_in)
    assert coef is not None and 12 <= coef.shape[0] - 1 <= 24
    X = coef[0, 1] * F + coef[0, 2] * np.eye(n)
    for a, b, c in coef[1:]:
        X = a * (X @ X) + b * X + c * np.eye(n)
    assert np.abs(X - P).max() < 1e-12 and abs(np.trace(X) - nocc) < 1e-10 and abs(np.trace(X) - np.sum(X * X)) < 1e-10
    # the spectrum may move inside the margins without breaking the plan ...
    F2 = (Q * (e + np.where(np.arange(n) < nocc, 0.1, -0.1))) @ Q.T
    X = coef[0, 1] * F2 + coef[0, 2] * np.eye(n)
    for a, b, c in coef[1:]:
        X = a * (X @ X) + b * X + c * np.eye(n)
    assert np.abs(X - P).max() < 1e-10
    # ... and bounds without a gap give no plan
    assert sp2plan.plan(lo, hi, 0.2, 0.1) is None


def test_gap_interval_from_sp2_traces_lies_inside_the_true_gap():
    """sp2plan.gap_from_traces: from the (tr X_i, tr X_i^2) sequence of a trace-correcting SP2 run alone, an energy interval that
    contains no orbital energy and brackets the Fermi level -- also with a degenerate HOMO / LUMO -- and a plan made from it
    purifies the matrix (the cold-object path of SCF._plan_from_traces)."""
    from mi355scf import sp2plan
    rng = np.random.default_rng(7)
    for n, nocc, degenerate in ((60, 12, False), (80, 21, True), (40, 5, False)):
        occ = np.sort(rng.uniform(-11.0, -0.35, nocc))
        vir = np.sort(rng.uniform(0.12, 30.0, n - nocc))
        if degenerate:
            occ[-2] = occ[-1]
            vir[1] = vir[0]
        e = np.concatenate([occ, vir])
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        f = q @ np.diag(e) @ q.T
        r = np.abs(f).sum(axis=1) - np.abs(np.diag(f))
        emin, emax = (np.diag(f) - r).min(), (np.diag(f) + r).max()
        x = (emax * np.eye(n) - f) / (emax - emin)
        tx, tx2 = [np.trace(x)], [np.trace(x @ x)]
        for _ in range(60):
            x2 = x @ x
            x = x2 if abs(np.trace(x2) - nocc) < abs(2 * np.trace(x) - np.trace(x2) - nocc) else 2 * x - x2
            tx.append(np.trace(x)); tx2.append(np.trace(x @ x))
        assert abs(tx[-1] - nocc) < 1e-8
        homo_ub, lumo_lb = sp2plan.gap_from_traces(tx, tx2, emin, emax)
        assert occ[-1] <= homo_ub < lumo_lb <= vir[0], (occ[-1], homo_ub, lumo_lb, vir[0])
        assert lumo_lb - homo_ub > 0.5 * (vir[0] - occ[-1])          # and it is not uselessly narrow
        b = sp2plan.bounds_from_traces(tx, tx2, emin, emax)
        coef = sp2plan.plan(*b)
        assert coef is not None and coef.shape[0] - 1 <= 45
        y = coef[0][1] * f + coef[0][2] * np.eye(n)
        for a_, b_, c_ in coef[1:]:
            y = a_ * (y @ y) + b_ * y + c_ * np.eye(n)
        p = q[:, :nocc] @ q[:, :nocc].T
        assert np.abs(y - p).max() < 1e-9
    assert sp2plan.gap_from_traces([3.0, 3.0], [1.0, 1.0], -1.0, 1.0) is None      # no step with a small enough tr(X - X^2)


def test_density_fitting_host_helpers():
    """df.pivoted_cholesky (rank-revealing factor of an SCF density, None for anything else) and df.tri_inv_lower (blocked
    inverse of a Cholesky factor) on the CPU."""
    import torch
    from mi355scf.df import pivoted_cholesky, tri_inv_lower
    g = torch.Generator().manual_seed(3)
    c = torch.randn(90, 11, generator=g, dtype=torch.float64)
    d = 2.0 * c @ c.T
    for rank in (11, 14):                     # exact rank, and a larger hint (beta spin of an open shell)
        lf = pivoted_cholesky(d, rank)
        assert lf is not None and lf.shape == (90, rank) and (lf @ lf.T - d).abs().max() < 1e-11
    assert pivoted_cholesky(d, 9) is None                                   # rank too small: remainder not negligible
    a = torch.randn(90, 90, generator=g, dtype=torch.float64)
    assert pivoted_cholesky(a + a.T, 11) is None                            # indefinite
    s = a @ a.T + 90.0 * torch.eye(90, dtype=torch.float64)
    lo = torch.linalg.cholesky(s)
    assert (tri_inv_lower(lo, base=16) @ lo - torch.eye(90, dtype=torch.float64)).abs().max() < 1e-12
