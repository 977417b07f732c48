"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on the same inputs.
Tolerances: FP64 path; integrals and J/K agree to 1e-10 absolute (values are O(1)-O(10)); the
north_star's energy tolerance is 1e-6 Ha (tests/test_gpu_scf.py)."""
import numpy as np
import pytest

from conftest import MOLECULES

pytestmark = pytest.mark.gpu

CASES = [("h2o", "sto-3g"), ("h2o", "6-31g*"), ("h2co", "6-31g(d)"), ("h2o", "cc-pvdz"), ("ch4", "cc-pvdz"),
         ("h2o", "cc-pvtz")]


def _mol(name, basis):
    from mi355scf.mole import Mole
    return Mole(atom=MOLECULES[name], basis=basis).build()


def _sym_density(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, n))
    return (a + a.T) * 0.5


@pytest.mark.parametrize("name,basis", CASES)
def test_int1e_matches_oracle(name, basis):
    from mi355scf.engine import Engine
    from oracle import oracle as orc
    mol = _mol(name, basis)
    eng = Engine(mol)
    S, T, V, dip = (x.cpu().numpy() for x in eng.int1e(with_dipole=True, origin=[0.1, -0.2, 0.3]))
    So, To, Vo, dipo = orc.Oracle(mol).int1e(origin=[0.1, -0.2, 0.3])
    assert np.abs(S - So).max() < 1e-12
    assert np.abs(T - To).max() < 1e-11
    assert np.abs(V - Vo).max() < 1e-10
    assert np.abs(dip - dipo).max() < 1e-11


@pytest.mark.parametrize("name,basis", CASES)
def test_jk_matches_oracle(name, basis):
    from mi355scf.engine import Engine
    from oracle import oracle as orc
    mol = _mol(name, basis)
    eng = Engine(mol)
    st = eng.prepare_eri(1e-13)
    assert st["n_tiles"] > 0 and st["n_unique_eri"] > 0
    D = _sym_density(mol.nao, 7)
    J, K = eng.get_jk(D)
    Jo, Ko = orc.Oracle(mol).jk(D, tol=0.0)
    assert np.abs(J.cpu().numpy() - Jo).max() < 1e-10, np.abs(J.cpu().numpy() - Jo).max()
    assert np.abs(K.cpu().numpy() - Ko).max() < 1e-10, np.abs(K.cpu().numpy() - Ko).max()
    # J-only and K-only variants of the kernel
    J2, _ = eng.get_jk(D, with_k=False)
    _, K2 = eng.get_jk(D, with_j=False)
    assert np.abs((J2 - J).cpu().numpy()).max() < 1e-11
    assert np.abs((K2 - K).cpu().numpy()).max() < 1e-11


def test_jk_linearity_and_symmetry_benzene_sized():
    """Size-independent properties at the bench size (benzene/cc-pVDZ, N=114): J,K symmetric, linear in D."""
    from mi355scf.engine import Engine
    from mi355scf.fixtures import BENZENE
    from mi355scf.mole import Mole
    mol = Mole(atom=BENZENE, basis="cc-pvdz").build()
    assert mol.nao == 114
    eng = Engine(mol)
    eng.prepare_eri(1e-13)
    D1, D2 = _sym_density(114, 1), _sym_density(114, 2)
    J1, K1 = eng.get_jk(D1)
    J2, K2 = eng.get_jk(D2)
    J3, K3 = eng.get_jk(D1 + 2.0 * D2)
    assert (J1 - J1.T).abs().max() < 1e-10 and (K1 - K1.T).abs().max() < 1e-10
    assert (J3 - J1 - 2 * J2).abs().max() < 1e-9
    assert (K3 - K1 - 2 * K2).abs().max() < 1e-9
    # trace identities: tr(D1 J2) == tr(D2 J1), tr(D1 K2) == tr(D2 K1)
    import torch
    d1, d2 = torch.as_tensor(D1, device=J1.device), torch.as_tensor(D2, device=J1.device)
    assert abs(float((d1 * J2).sum() - (d2 * J1).sum())) < 1e-8
    assert abs(float((d1 * K2).sum() - (d2 * K1).sum())) < 1e-8


def test_against_committed_golden_vectors():
    """HIP path vs tests/golden/h2co_631gd_vectors.npz (made by tests/golden/make_golden_vectors.py)."""
    import os
    import torch
    from mi355scf.engine import Engine
    from mi355scf.dft import RKS
    from mi355scf.fixtures import H2CO
    from mi355scf.mole import Mole
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "h2co_631gd_vectors.npz"))
    mol = Mole(atom=H2CO, basis="6-31g(d)", verbose=0).build()
    eng = Engine(mol)
    S, T, V, dip = (x.cpu().numpy() for x in eng.int1e(with_dipole=True))
    assert np.abs(S - g["S"]).max() < 1e-12 and np.abs(T - g["T"]).max() < 1e-11
    assert np.abs(V - g["V"]).max() < 1e-10 and np.abs(dip - g["dip"]).max() < 1e-11
    J, K = eng.get_jk(g["D"])
    assert np.abs(J.cpu().numpy() - g["J"]).max() < 1e-10 and np.abs(K.cpu().numpy() - g["K"]).max() < 1e-10
    mf = RKS(mol, xc="B3LYP")
    mf.grids.level = 1
    mf._setup_once()
    assert mf.grids.size == int(g["ngrid"]) and abs(float(mf.grids.weights.sum()) - float(g["wsum"])) < 1e-8
    n, exc, vxc, hyb = mf.nr_rks(torch.as_tensor(g["Docc"], device=eng.device))
    assert abs(float(n) - float(g["nelec"])) < 1e-9 and abs(float(exc) - float(g["exc"])) < 1e-9
    assert np.abs(vxc.cpu().numpy() - g["vxc"]).max() < 1e-9 and hyb == float(g["hyb"])


@pytest.mark.parametrize("name,basis", [("h2o", "cc-pvtz"), ("ch4", "cc-pvdz")])
def test_pair_kernel_two_densities_in_one_pass_equals_two_passes(name, basis):
    """`mi_build_jk(n_dm = 2)` through `jk_tiles_pair_kernel` (two waves per work item, one per density; default for stores beyond
    16 GB, forced here with `jk_pair = 1`) against one pass per density, J+K, J-only and K-only: 1e-11 (ragged last blocks included:
    N = 58 and N = 34 are not multiples of 8)."""
    import torch
    from mi355scf.engine import Engine
    mol = _mol(name, basis)
    eng = Engine(mol)
    eng.prepare_eri(1e-13)
    n = mol.nao
    D2 = torch.as_tensor(np.stack([_sym_density(n, 1), _sym_density(n, 2)]), device="cuda")
    eng.set_option("jk_pair", 0)
    Jr, Kr = eng.get_jk(D2)
    eng.set_option("jk_pair", 1)
    J, K = eng.get_jk(D2)
    assert float((J - Jr).abs().max()) < 1e-11 and float((K - Kr).abs().max()) < 1e-11
    Jj, _ = eng.get_jk(D2, with_k=False)
    _, Kk = eng.get_jk(D2, with_j=False)
    assert float((Jj - Jr).abs().max()) < 1e-11 and float((Kk - Kr).abs().max()) < 1e-11
