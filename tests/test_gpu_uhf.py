"""UHF (reference call sites `templates/calculate_bde.py:126,138,192,210`): the HIP path
against the CPU oracle's UHF from the SAME initial density, plus known answers (H atom: exact STO-3G / cc-pVDZ
values; closed shell: UHF == RHF)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OH = "O 0 0 0; H 0 0 0.97"
CH3 = "C 0 0 0; H 1.079 0 0; H -0.5395 0.934441 0; H -0.5395 -0.934441 0"


def _mol(atom, basis, spin):
    from pyscf import gto
    m = gto.Mole()
    m.atom, m.basis, m.spin, m.verbose = atom, basis, spin, 0
    m.build()
    return m


@pytest.mark.parametrize("atom,basis,spin", [(OH, "6-31G*", 1), (CH3, "cc-pVDZ", 1), ("O 0 0 0; O 0 0 1.2", "6-31G", 2)])
def test_uhf_matches_oracle(atom, basis, spin):
    from pyscf import scf
    from oracle import oracle as orc
    mol = _mol(atom, basis, spin)
    mf = scf.UHF(mol).to_gpu()
    mf.conv_tol = 1e-10
    dm0 = mf.get_init_guess()
    e = mf.kernel(dm0=dm0)
    assert mf.converged
    e_ref, dm_ref, _, _ = orc.uhf(mol, dm0=dm0, conv_tol=1e-11)
    assert abs(e - e_ref) < 1e-7
    dm = mf.make_rdm1()
    assert dm.shape == (2, mol.nao, mol.nao)
    # (spatially degenerate radicals such as OH have equivalent solutions rotated into each other: densities are
    # compared through the energy functional only)
    S = mf.get_ovlp()
    na, nb = mol.nelec
    assert abs(np.trace(dm[0] @ S) - na) < 1e-8 and abs(np.trace(dm[1] @ S) - nb) < 1e-8
    ss, mult = mf.spin_square()
    sz = 0.5 * (na - nb)
    assert ss >= sz * (sz + 1) - 1e-8 and ss < sz * (sz + 1) + 0.2   # mild spin contamination only
    assert mf.mo_energy.shape == (2, mol.nao) and mf.mo_occ.sum() == mol.nelectron


def test_uhf_known_answers():
    from pyscf import scf
    import gpu4pyscf
    for basis, ref in (("sto-3g", -0.46658185), ("cc-pVDZ", -0.49927840)):
        mf = gpu4pyscf.scf.UHF(_mol("H 0 0 0", basis, 1))
        assert abs(mf.kernel() - ref) < 2e-8
        assert abs(mf.spin_square()[0] - 0.75) < 1e-10
    h2o = _mol("O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", "6-31G", 0)
    e_u = scf.UHF(h2o).kernel()
    e_r = scf.RHF(h2o).kernel()
    assert abs(e_u - e_r) < 1e-8 and abs(e_r + 75.98394849812) < 1e-7   # PySCF test-suite value [MEM]


def test_uhf_gradient_matches_finite_differences():
    """Analytic UHF gradient (open-shell two-particle density in `mi_grad_eri_spin`) vs central differences of the
    UHF energy; closed shell: UHF gradient == RHF gradient."""
    from pyscf import scf
    from mi355scf.grad import FDGradients
    mol = _mol("O 0 0 0; H 0.1 0.05 0.97", "6-31G*", 1)
    mf = scf.UHF(mol).to_gpu()
    mf.conv_tol = 1e-11
    mf.kernel()
    g = mf.nuc_grad_method().kernel()
    fd = FDGradients(mf)
    fd.step = 5e-4
    g_fd = fd.kernel()
    assert np.abs(g - g_fd).max() < 2e-6
    assert np.abs(g.sum(axis=0)).max() < 1e-8           # translational invariance
    h2o = _mol("O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", "6-31G*", 0)
    gu = scf.UHF(h2o).nuc_grad_method().kernel()
    gr = scf.RHF(h2o).nuc_grad_method().kernel()
    assert np.abs(gu - gr).max() < 1e-6


def test_uhf_sp2_path_matches_diagonalisation():
    """N >= 200: the per-spin SP2 purification (no diagonalisation inside the loop) gives the same UHF solution as `eigh`
    per cycle (benzene radical cation, cc-pVTZ, N = 264)."""
    import time
    from pyscf import gto, scf
    from mi355scf import fixtures
    mol = gto.Mole()
    mol.atom, mol.basis, mol.charge, mol.spin, mol.verbose = fixtures.BENZENE, "cc-pVTZ", 1, 1, 0
    mol.build()
    res = {}
    for method in ("sp2", "eigh"):
        mf = scf.UHF(mol).to_gpu()
        mf.eig_method = method
        mf.conv_tol = 1e-10
        t0 = time.time()
        e = mf.kernel()
        res[method] = (e, mf.cycles, time.time() - t0, mf.spin_square()[0])
        assert mf.converged
    assert abs(res["sp2"][0] - res["eigh"][0]) < 1e-8, res
    assert abs(res["sp2"][3] - res["eigh"][3]) < 1e-5
    print("UHF C6H6+ cc-pVTZ:", res)


def test_uhf_fast_loop_with_planned_purification_equals_the_plain_loop():
    """The orthonormal-basis UHF/UKS loop (device-side pair DIIS, planned purification per spin, pipelined step) against the
    plain loop: benzene cation / cc-pVDZ with the purification forced on (sp2_min_nao = 0).  The first kernel() of an object is
    cold (plain loop, seeds the plans from its final orbitals), the second runs the fast loop with those plans."""
    import gpu4pyscf
    from pyscf import gto
    from mi355scf import fixtures
    mol = gto.Mole()
    mol.atom, mol.basis, mol.charge, mol.spin, mol.verbose = fixtures.BENZENE, "cc-pVDZ", 1, 1, 0
    mol.build()
    ref = gpu4pyscf.scf.UHF(mol).to_gpu()
    ref.fast_loop, ref.conv_tol = False, 1e-10
    e_ref = ref.kernel()
    assert ref.converged
    mf = gpu4pyscf.scf.UHF(mol).to_gpu()
    mf.sp2_min_nao, mf.conv_tol, mf.fast_loop = 0, 1e-10, "always"
    e_cold = mf.kernel()                               # plain loop + plans
    assert mf.converged and abs(e_cold - e_ref) < 1e-9
    assert all(sp.vals["_sp2_plan"] is not None for sp in mf._spin_pair)
    dm = mf.make_rdm1()
    e_warm = mf.kernel(dm0=dm)                         # fast loop, planned purification of both spins from the first cycle
    assert mf.converged and abs(e_warm - e_ref) < 1e-9 and mf.cycles <= 4
    assert abs(mf.spin_square()[0] - ref.spin_square()[0]) < 1e-6
    ks_ref = gpu4pyscf.dft.UKS(mol).to_gpu()
    ks_ref.xc, ks_ref.fast_loop, ks_ref.conv_tol = "B3LYP", False, 1e-10
    ek_ref = ks_ref.kernel()
    ks = gpu4pyscf.dft.UKS(mol).to_gpu()
    ks.xc, ks.sp2_min_nao, ks.conv_tol, ks.fast_loop = "B3LYP", 0, 1e-10, "always"
    ks.kernel()
    ek = ks.kernel(dm0=ks.make_rdm1())
    assert ks.converged and abs(ek - ek_ref) < 1e-8, (ek, ek_ref)
