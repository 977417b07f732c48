"""`templates/calculate_interaction.py` call sequences: counterpoise correction with 'Ghost:' atoms (`:127-157`) and the
`--method MP2` branch (`mp.MP2(mf).kernel()`, `:116-120`)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W1 = [("O", (0.0, 0.0, 0.0)), ("H", (0.0, -0.757, 0.587)), ("H", (0.0, 0.757, 0.587))]
W2 = [("O", (0.0, 0.0, 2.95)), ("H", (0.0, -0.757, 3.537)), ("H", (0.0, 0.757, 3.537))]


def _mol(atoms, basis):
    from pyscf import gto
    m = gto.Mole()
    m.atom = [[a, c] for a, c in atoms]
    m.basis, m.verbose = basis, 0
    m.build()
    return m


def test_mp2_known_answer_and_oracle():
    """H2O/cc-pVDZ MP2 correlation energy of PySCF's own example/test: -0.204019967288338 Ha [MEM]; UMP2 on the same
    closed shell gives the same number; 6-31G value against the CPU oracle."""
    from pyscf import scf, mp
    from oracle import oracle as orc
    mol = _mol(W1, "cc-pVDZ")
    mf = scf.RHF(mol)
    mf.conv_tol = 1e-11
    mf.kernel()
    pt = mp.MP2(mf)
    e_corr, _t2 = pt.kernel()
    assert abs(e_corr + 0.204019967288338) < 2e-8
    assert abs(pt.e_tot - (mf.e_tot + e_corr)) < 1e-14
    mu = scf.UHF(mol)
    mu.conv_tol = 1e-11
    mu.kernel()
    assert abs(mp.MP2(mu).kernel()[0] - e_corr) < 1e-7
    small = _mol(W1, "6-31G")
    ms = scf.RHF(small)
    ms.conv_tol = 1e-11
    ms.kernel()
    assert abs(mp.MP2(ms).kernel()[0] - orc.mp2(small)) < 1e-8


def test_counterpoise_with_ghost_atoms_matches_oracle():
    """Monomer energies in the dimer basis: ghost centres carry basis functions, no charge, no grid points
    (calculate_interaction.py:136-145).  RHF against the oracle; B3LYP/6-31+G* BSSE of the water dimer is positive and
    a few tenths of a kcal/mol."""
    from pyscf import scf, dft
    from oracle import oracle as orc
    ghost2 = [("Ghost:" + a, c) for a, c in W2]
    m_g = _mol(W1 + ghost2, "6-31G")
    assert m_g.nelectron == 10 and m_g.natm == 6 and m_g.nao == 26
    mf = scf.RHF(m_g)
    mf.conv_tol = 1e-10
    e_g = mf.kernel()
    assert abs(e_g - orc.rhf(m_g, dm0=mf.get_init_guess(), conv_tol=1e-10)["e_tot"]) < 1e-8
    e_1 = scf.RHF(_mol(W1, "6-31G")).kernel()
    assert -3e-3 < e_g - e_1 < -1e-5                          # basis-set superposition lowers the monomer energy
    # RKS with ghost centres: they carry a first-period grid and take part in the Becke partition (PySCF convention [MEM])
    from oracle import dft as od
    ks = dft.RKS(m_g)
    ks.xc, ks.conv_tol, ks.small_rho_cutoff = "B3LYP", 1e-10, 0
    e_ks = ks.kernel()
    ref = od.rks(m_g, "B3LYP", level=3, dm0=ks.get_init_guess(), conv_tol=1e-10, small_rho_cutoff=0)
    assert ks.grids.size == ref["ngrids"] and abs(e_ks - ref["e_tot"]) < 2e-7
    assert abs(float(ks._nelec_grid) - 10.0) < 2e-4

    def energy(atoms):                                         # calculate_interaction.py:92-123 with method = 'B3LYP'
        mol = _mol(atoms, "6-31+G*")
        f = dft.RKS(mol) if mol.spin == 0 else dft.UKS(mol)
        f.xc = "B3LYP"
        f.kernel()
        assert f.converged
        return f.e_tot
    ghost1 = [("Ghost:" + a, c) for a, c in W1]
    bsse = (energy(W1 + ghost2) - energy(W1)) + (energy(ghost1 + W2) - energy(W2))
    e_int = energy(W1 + W2) - energy(W1) - energy(W2)
    assert -2e-3 < bsse < -1e-5
    assert -0.008 < e_int - bsse < -0.001                      # stacked (not hydrogen-bonded) water dimer, R(OO) = 2.95 A: ~ -2 kcal/mol
