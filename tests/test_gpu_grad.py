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
