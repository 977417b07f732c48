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
