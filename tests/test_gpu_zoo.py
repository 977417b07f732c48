"""A wider set of small molecules (charged species, linear molecules, a far-separated dimer, N and F tables, def2-TZVP):
RHF energies of the HIP path against the CPU oracle from the same initial density, and geometry optimisation of linear
molecules through the Cartesian fallback of `optimize()`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ZOO = [("NH3", "N 0 0 0.1; H 0.94 0 -0.27; H -0.47 0.814 -0.27; H -0.47 -0.814 -0.27", "cc-pVDZ", 0),
       ("CO", "C 0 0 0; O 0 0 1.128", "cc-pVTZ", 0),
       ("CO2", "O 0 0 -1.16; C 0 0 0; O 0 0 1.16", "6-31G*", 0),
       ("C2H2", "H 0 0 -1.66; C 0 0 -0.6; C 0 0 0.6; H 0 0 1.66", "cc-pVDZ", 0),
       ("HF", "H 0 0 0; F 0 0 0.917", "cc-pVDZ", 0),
       ("H3O+", "O 0 0 0.08; H 0.93 0 -0.2; H -0.465 0.805 -0.2; H -0.465 -0.805 -0.2", "6-31G**", 1),
       ("OH-", "O 0 0 0; H 0 0 0.97", "6-31G*", -1),
       ("H2O2", "O 0 0.7 0; O 0 -0.7 0; H 0.8 0.9 0.5; H -0.8 -0.9 0.5", "def2-TZVP", 0),
       ("H2O diffuse", "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587", "6-31++G**", 0),
       ("ethanol diffuse", "C -0.748 -0.015 0.024; C 0.558 0.420 -0.278; O 1.505 -0.604 0.023; H -1.489 0.772 -0.198; H -0.994 -0.906 -0.566; H -0.789 -0.268 1.091; H 0.622 0.669 -1.345; H 0.822 1.326 0.282; H 1.455 -0.850 0.961", "6-31+G*", 0),
       ("far H2O dimer", "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587; O 0 0 8.0; H 0 -0.757 8.587; H 0 0.757 8.587", "6-31G", 0)]


def _mol(atom, basis, charge=0):
    from pyscf import gto
    m = gto.Mole()
    m.atom, m.basis, m.charge, m.verbose = atom, basis, charge, 0
    m.build()
    return m


@pytest.mark.parametrize("name,atom,basis,charge", ZOO, ids=[z[0] for z in ZOO])
def test_rhf_energy_matches_oracle(name, atom, basis, charge):
    from pyscf import scf
    from oracle import oracle as orc
    mol = _mol(atom, basis, charge)
    mf = scf.RHF(mol).to_gpu()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged
    ref = orc.rhf(mol, dm0=mf.get_init_guess(), conv_tol=1e-10)
    assert abs(e - ref["e_tot"]) < 1e-8


def test_known_rhf_energies_at_experimental_geometries():
    """Literature RHF energies [MEM, 4 decimals]: CO/cc-pVTZ -112.7804, HF/cc-pVDZ -100.0194, C2H2/cc-pVDZ -76.8259."""
    from pyscf import scf
    for atom, basis, ref in (("C 0 0 0; O 0 0 1.128", "cc-pVTZ", -112.7804), ("H 0 0 0; F 0 0 0.917", "cc-pVDZ", -100.0194),
                             ("H 0 0 -1.66; C 0 0 -0.6; C 0 0 0.6; H 0 0 1.66", "cc-pVDZ", -76.8259)):
        assert abs(scf.RHF(_mol(atom, basis)).kernel() - ref) < 2e-4


def test_linear_molecules_optimise_through_the_cartesian_fallback():
    """CO2 and acetylene have linear bends: `optimize()` falls back to Cartesian BFGS; B3LYP/6-31G(d) bond lengths
    r(C=O) = 1.169 A, r(C-H) = 1.067 A, r(CC) = 1.205 A [MEM, 3 decimals]."""
    from pyscf import dft
    from pyscf.geomopt.geometric_solver import optimize
    out = {}
    for name, atom in (("CO2", "O 0 0 -1.2; C 0 0 0.02; O 0 0 1.13"), ("C2H2", "H 0 0 -1.7; C 0 0 -0.62; C 0 0 0.6; H 0 0 1.66")):
        mf = dft.RKS(_mol(atom, "6-31G*")).to_gpu()
        mf.xc = "B3LYP"
        x = optimize(mf, maxsteps=60).atom_coords() * 0.52917721092
        out[name] = x
    co2 = out["CO2"]
    assert abs(np.linalg.norm(co2[0] - co2[1]) - 1.169) < 3e-3 and abs(np.linalg.norm(co2[2] - co2[1]) - 1.169) < 3e-3
    c2h2 = out["C2H2"]
    assert abs(np.linalg.norm(c2h2[0] - c2h2[1]) - 1.067) < 3e-3 and abs(np.linalg.norm(c2h2[1] - c2h2[2]) - 1.205) < 3e-3
