"""`optimize(mf)` on the README example (reference README.md:181-205): H2CO RHF/6-31G(d)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_h2co_optimize_readme_example():
    import gpu4pyscf
    from pyscf import gto
    from pyscf.geomopt.geometric_solver import optimize
    mol = gto.Mole()
    mol.atom = '''
        C  0.0  0.0  0.0
        O  1.2  0.0  0.0
        H -0.5  0.9  0.0
        H -0.5 -0.9  0.0
    '''
    mol.basis = '6-31G(d)'
    mol.verbose = 0
    mol.build()
    mf = gpu4pyscf.scf.RHF(mol).to_gpu()
    e0 = mf.kernel()
    mol_opt = optimize(mf, maxsteps=40)
    c = mol_opt.atom_coords() * 0.529177
    assert c.shape == (4, 3)
    from pyscf import scf
    e1 = scf.RHF(mol_opt).kernel()
    assert e1 < e0 - 1e-4
    rco = np.linalg.norm(c[0] - c[1])
    rch = np.linalg.norm(c[0] - c[2])
    assert 1.17 < rco < 1.20 and 1.08 < rch < 1.10     # HF/6-31G(d) H2CO: r(CO) ~1.184 A, r(CH) ~1.092 A
    g = scf.RHF(mol_opt)
    g.verbose = 0
    g.kernel()
    assert np.abs(g.nuc_grad_method().kernel()).max() < 1e-3


def test_water_hf_631gd_equilibrium_geometry_and_dipole_against_literature():
    """RHF/6-31G(d) water: r(OH) = 0.947 A, angle 105.5 deg, dipole 2.20 D (standard textbook / G2 reference data [MEM]).
    The textbook energy -76.01075 Ha is for six Cartesian d functions; with PySCF's default five spherical d functions
    (what this engine implements) the energy is ~1.4 mHa higher, so only a bracket is asserted for it."""
    import numpy as np
    from pyscf import gto, scf
    from pyscf.geomopt.geometric_solver import optimize
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", "6-31G(d)", 0
    mol.build()
    mol_eq = optimize(scf.RHF(mol).to_gpu(), maxsteps=50)
    x = mol_eq.atom_coords() * 0.52917721092
    r1, r2 = np.linalg.norm(x[1] - x[0]), np.linalg.norm(x[2] - x[0])
    ang = np.degrees(np.arccos(np.dot(x[1] - x[0], x[2] - x[0]) / r1 / r2))
    assert abs(r1 - 0.947) < 2e-3 and abs(r2 - 0.947) < 2e-3 and abs(ang - 105.5) < 0.3
    mf = scf.RHF(mol_eq)
    e = mf.kernel()
    assert -76.0108 < e < -76.0085
    d = np.linalg.norm(mf.dip_moment(unit="Debye"))
    assert abs(d - 2.20) < 0.03
