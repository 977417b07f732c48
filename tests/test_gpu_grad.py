"""Analytic gradient pieces vs central finite differences of the corresponding energy terms evaluated by
the (already oracle-checked) GPU integral kernels at displaced geometries (SURVEY.md 8c golden check vii)."""
import numpy as np
import pytest
import torch

from conftest import MOLECULES

pytestmark = pytest.mark.gpu
H = 1e-4


def _mol(name, basis):
    from mi355scf.mole import Mole
    return Mole(atom=MOLECULES[name], basis=basis, verbose=0).build()


def _fd(mol, fn):
    R = mol.atom_coords()
    g = np.zeros_like(R)
    for ia in range(mol.natm):
        for x in range(3):
            Rp, Rm = R.copy(), R.copy()
            Rp[ia, x] += H
            Rm[ia, x] -= H
            g[ia, x] = (fn(mol.set_geom_(Rp, unit="Bohr", inplace=False)) - fn(mol.set_geom_(Rm, unit="Bohr", inplace=False))) / (2 * H)
    return g


@pytest.mark.parametrize("name,basis", [("h2o", "cc-pvdz"), ("h2co", "6-31g(d)"), ("h2o", "cc-pvtz")])
def test_grad_1e_matches_finite_difference(name, basis):
    from mi355scf.engine import Engine
    mol = _mol(name, basis)
    n = mol.nao
    rng = np.random.default_rng(1)
    a, b = rng.normal(size=(n, n)), rng.normal(size=(n, n))
    D, W = a + a.T, b + b.T
    eng = Engine(mol)
    dD, dW = torch.as_tensor(D, device=eng.device), torch.as_tensor(W, device=eng.device)
    g = torch.zeros(mol.natm, 3, dtype=torch.float64, device=eng.device)
    eng.grad_1e(dD, dW, g)

    def f(m):
        S, T, V = (x.cpu().numpy() for x in Engine(m).int1e())
        return float(np.sum(D * (T + V)) - np.sum(W * S))
    ref = _fd(mol, f)
    assert np.abs(g.cpu().numpy() - ref).max() < 2e-6 * max(1.0, np.abs(ref).max()), (g.cpu().numpy(), ref)
    assert np.abs(g.cpu().numpy().sum(axis=0)).max() < 1e-8 * max(1.0, np.abs(ref).max())  # translational invariance


@pytest.mark.parametrize("name,basis,hyb", [("h2o", "6-31g", 1.0), ("h2o", "cc-pvdz", 1.0), ("h2co", "6-31g(d)", 0.2),
                                            ("h2o", "cc-pvtz", 1.0)])
def test_grad_eri_matches_finite_difference(name, basis, hyb):
    from mi355scf.engine import Engine
    mol = _mol(name, basis)
    n = mol.nao
    rng = np.random.default_rng(4)
    c = rng.normal(size=(n, 4)) * 0.4
    D = 2 * c @ c.T
    eng = Engine(mol)
    eng.prepare_eri(1e-14)
    dD = torch.as_tensor(D, device=eng.device)
    g = torch.zeros(mol.natm, 3, dtype=torch.float64, device=eng.device)
    eng.grad_eri(dD, hyb, g)
    g = g.cpu().numpy()

    def f(m):
        e = Engine(m)
        e.prepare_eri(1e-14)
        J, K = e.get_jk(D)
        return float(0.5 * (torch.as_tensor(D, device=J.device) * (J - 0.5 * hyb * K)).sum())
    ref = _fd(mol, f)
    assert np.abs(g - ref).max() < 5e-6 * max(1.0, np.abs(ref).max()), (g, ref)
    assert np.abs(g.sum(axis=0)).max() < 1e-8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name,basis,method", [("h2o", "cc-pvdz", "HF"), ("h2co", "6-31g(d)", "HF"), ("h2o", "cc-pvtz", "HF")])
def test_total_rhf_gradient_matches_finite_difference(name, basis, method):
    from pyscf import gto, scf
    from mi355scf.grad import FDGradients
    mol = gto.Mole()
    mol.atom = MOLECULES[name]
    mol.basis = basis
    mol.verbose = 0
    mol.build()
    mf = scf.RHF(mol)
    mf.conv_tol = 1e-11
    mf.kernel()
    g = mf.nuc_grad_method().kernel()
    fd = FDGradients(mf)
    fd.step = 5e-4
    ref = fd.kernel()
    assert np.abs(g - ref).max() < 1e-6, (g, ref)
    assert np.abs(g.sum(axis=0)).max() < 1e-7


@pytest.mark.parametrize("xc", ["LDA,VWN", "B3LYP", "PBE"])
def test_xc_gradient_matches_frozen_weight_finite_difference(xc):
    """d/dR of E_xc[D] on a grid FROZEN in space (points and weights) while the basis functions move ==
    the analytic XC gradient without grid response (what PySCF computes by default [MEM])."""
    from pyscf import gto, dft
    mol = gto.Mole()
    mol.atom = MOLECULES["h2o"]
    mol.basis = "cc-pvdz"
    mol.verbose = 0
    mol.build()
    mf = dft.RKS(mol)
    mf.xc = xc
    mf.kernel()
    dm = mf._dm
    g = mf.nuc_grad_method().grad_xc(dm)
    coords0, w0, owner = mf.grids.coords.clone(), mf.grids.weights.clone(), mf.grids.atom_of
    R = mol.atom_coords()
    ref = np.zeros_like(R)
    for ia in range(mol.natm):
        for x in range(3):
            vals = []
            for sgn in (+1, -1):
                Rn = R.copy()
                Rn[ia, x] += sgn * H
                m2 = mol.set_geom_(Rn, unit="Bohr", inplace=False)
                mf2 = dft.RKS(m2)
                mf2.xc = xc
                mf2._setup_once()
                mf2.grids.coords, mf2.grids.weights = coords0, w0
                vals.append(float(mf2.nr_rks(dm)[1]))
            ref[ia, x] = (vals[0] - vals[1]) / (2 * H)
    assert np.abs(g - ref).max() < 2e-6, (g, ref)


def test_total_b3lyp_gradient_close_to_finite_difference():
    """Total RKS gradient vs full finite differences; the difference is the neglected grid-weight response."""
    from pyscf import gto, dft
    from mi355scf.grad import FDGradients
    mol = gto.Mole()
    mol.atom = MOLECULES["h2o"]
    mol.basis = "6-31g(d)"
    mol.verbose = 0
    mol.build()
    mf = dft.RKS(mol)
    mf.xc = "B3LYP"
    mf.conv_tol = 1e-11
    mf.kernel()
    g = mf.nuc_grad_method().kernel()
    ref = FDGradients(mf).kernel()
    assert np.abs(g - ref).max() < 2e-4, (g, ref)


def test_rhf_gradient_matches_cpu_oracle_finite_difference():
    """Independent check: GPU analytic gradient vs central differences of the CPU ORACLE's RHF energy."""
    from pyscf import gto, scf
    from oracle import oracle as orc
    mol = gto.Mole()
    mol.atom = MOLECULES["h2o"]
    mol.basis = "6-31g*"
    mol.verbose = 0
    mol.build()
    mf = scf.RHF(mol)
    mf.conv_tol = 1e-11
    mf.kernel()
    g = mf.nuc_grad_method().kernel()
    R = mol.atom_coords()
    h = 1e-3
    ref = np.zeros_like(R)
    for ia in range(mol.natm):
        for x in range(3):
            e = []
            for sgn in (+1, -1):
                Rn = R.copy()
                Rn[ia, x] += sgn * h
                e.append(orc.rhf(mol.set_geom_(Rn, unit="Bohr", inplace=False), conv_tol=1e-12)["e_tot"])
            ref[ia, x] = (e[0] - e[1]) / (2 * h)
    assert np.abs(g - ref).max() < 2e-6, (g, ref)


@pytest.mark.parametrize("spin", [0, 1])
def test_gradient_paths_agree(spin):
    """The two-electron gradient through its three routes for the mid / high classes -- row kernel on the live-quartet list
    (default), hand-over pipeline on the live list, hand-over pipeline screening per wave (round-2 path) -- on benzene/cc-pVTZ
    (f shells, all 55 class pairs) with a converged density; closed shell and the spin-density form of the cation."""
    from mi355scf import fixtures
    from mi355scf.mole import Mole
    from mi355scf.scf import RHF
    from mi355scf.uhf import UHF
    mol = Mole(atom=fixtures.BENZENE, basis="cc-pVTZ", verbose=0, charge=spin, spin=spin).build()
    mf = (UHF if spin else RHF)(mol)
    mf.conv_tol = 1e-10
    mf.kernel()
    eng = mf.engine
    if spin:
        D = (mf._dm[0] + mf._dm[1]).contiguous()
        M = (mf._dm[0] - mf._dm[1]).contiguous()
    else:
        D, M = mf._dm.contiguous(), None
    out = {}
    for name, opts in (("rows+live", {"grad_rows": 1, "grad_live": 1}), ("pipeline+live", {"grad_rows": 0, "grad_live": 1}),
                       ("pipeline", {"grad_rows": 0, "grad_live": 0}), ("rows", {"grad_rows": 1, "grad_live": 0})):
        for k, v in opts.items():
            eng.set_option(k, v)
        for dtol in (1e-13, 1e-10):
            eng.set_option("grad_dtol", dtol)
            g = torch.zeros(mol.natm, 3, dtype=torch.float64, device=eng.device)
            eng.grad_eri(D, 1.0, g, spin_density=M)
            out[(name, dtol)] = g.cpu().numpy()
    eng.set_option("grad_rows", 1); eng.set_option("grad_live", 1); eng.set_option("grad_dtol", 1e-13)
    ref = out[("pipeline", 1e-13)]
    assert np.abs(ref).max() > 1.0
    for (name, dtol), g in out.items():
        # same quartets, different summation order (1e-13); the looser threshold drops quartets worth < 1e-10 each (4e-7 in all here)
        assert np.abs(g - out[("pipeline", dtol)]).max() < 1e-10, (name, dtol, np.abs(g - out[("pipeline", dtol)]).max())
        assert np.abs(g - ref).max() < (1e-10 if dtol == 1e-13 else 1e-6), (name, dtol)
