"""The two in-scope reference templates import and build molecules UNCHANGED against the drop-in packages
(`pyscf`, `gpu4pyscf`, `cupy`, `rdkit` stand-ins).  Runs here (CPU container, reference mounted); skipped on
the GPU box where /root/reference does not exist.  The SCF itself needs a GPU (tests/test_gpu_template_flow.py)."""
import importlib.util
import io
import os
import sys

import numpy as np
import pytest

TEMPLATES = "/root/reference/templates"
pytestmark = pytest.mark.skipif(not os.path.isdir(TEMPLATES), reason="reference not mounted")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(TEMPLATES, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_calculate_energy_template_imports_and_builds_mol(capsys):
    mod = _load("calculate_energy")
    assert mod.GPU4PYSCF_AVAILABLE is True          # cupy + gpu4pyscf imports succeeded
    atoms, coords = mod.smiles_to_xyz("C=O")
    assert atoms == ["C", "O", "H", "H"] and coords.shape == (4, 3)
    buf = io.StringIO()
    mol = mod.create_pyscf_mol(atoms, coords, "6-31G(d)", 0, 0, output_stream=buf)
    assert (mol.natm, mol.nelectron, mol.nao) == (4, 16, 32)     # BASELINE config 1
    atoms, coords = mod.smiles_to_xyz("c1ccccc1")
    mol = mod.create_pyscf_mol(atoms, coords, "cc-pVDZ")
    assert (mol.natm, mol.nelectron, mol.nao) == (12, 42, 114)   # BASELINE config 2
    # analyze_orbitals works on NumPy results
    class _MF:
        mo_energy = np.array([-1.0, -0.5, 0.2])
        mo_occ = np.array([2.0, 2.0, 0.0])
    info = mod.analyze_orbitals(_MF(), mol)
    assert info["homo_idx"] == 1 and abs(info["gap"] - 0.7) < 1e-12


def test_optimize_geometry_template_imports_and_builds_mol():
    mod = _load("optimize_geometry")
    atoms, coords = mod.smiles_to_xyz("CC(C)Cc1ccc(cc1)C(C)C(=O)O")
    assert len(atoms) == 33
    mol = mod.create_pyscf_mol(atoms, coords, "def2-TZVP")
    assert (mol.nelectron, mol.nao) == (112, 573)                # BASELINE config 5
    assert mol.atom_coords().shape == (33, 3)
    assert callable(mod.optimize)


def test_calculate_bde_template_imports_enumerates_bonds_and_fragments():
    """`templates/calculate_bde.py` imports unchanged; its RDKit-side helpers run on the
    stand-in: bond enumeration, homolytic fragmentation into two radicals with coordinates, open-shell `Mole`s."""
    mod = _load("calculate_bde")
    bonds, rd = mod.get_all_bonds("CCO")
    assert len(bonds) == 8 and rd.GetNumAtoms() == 9
    kinds = {(a, b) for _i, _j, _t, a, b in bonds}
    assert kinds == {("C", "C"), ("C", "H"), ("C", "O"), ("O", "H")}
    i, j = next((i, j) for i, j, _t, a, b in bonds if (a, b) == ("C", "C"))
    a1, c1, a2, c2 = mod.create_radical_fragments("CCO", i, j)
    assert sorted(a1) == ["C", "H", "H", "H"] and sorted(a2) == sorted(["C", "O", "H", "H", "H"])
    assert c1.shape == (4, 3) and c2.shape == (5, 3)
    atoms, coords = mod.smiles_to_xyz("CCO")
    # fragment coordinates are the parent's coordinates of the same atoms (the `_FromAtomIdx` path of the template)
    assert any(np.allclose(c1[0], coords[k]) for k in range(len(atoms)))
    m1 = mod.create_pyscf_mol(a1, c1, "6-31G(d)", charge=0, spin=1)       # methyl radical
    m2 = mod.create_pyscf_mol(a2, c2, "6-31G(d)", charge=0, spin=1)       # CH2OH radical
    assert m1.nelectron == 9 and m1.spin == 1 and m2.nelectron == 17
    from pyscf import scf, dft
    assert scf.UHF.__name__ == "UHF" and dft.UKS.__name__ == "UKS"
    ring = mod.get_all_bonds("c1ccccc1")[0]
    assert sum(1 for b in ring if b[2] == "AROMATIC") == 6


def test_opt_freq_template_imports():
    """`templates/opt-freq.py` imports unchanged: `pyscf.hessian.thermo`, the gpu4pyscf probe,
    `dft.rks.RKS` / `dft.uks.UKS` for its isinstance checks, and the Hessian factories it calls are all present."""
    spec = importlib.util.spec_from_file_location("opt_freq", os.path.join(TEMPLATES, "opt-freq.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.GPU4PYSCF_AVAILABLE is True
    assert callable(mod.numerical_ir_intensities) and callable(mod.thermo.harmonic_analysis) and callable(mod.thermo.thermo)
    from pyscf import dft, hessian
    from gpu4pyscf import hessian as gpu_hessian
    assert hessian.rks.Hessian is gpu_hessian.rks.Hessian
    assert isinstance(dft.rks.RKS, type) and isinstance(dft.uks.UKS, type)


def test_interaction_and_reaction_templates_import():
    """`templates/calculate_interaction.py` (needs `pyscf.mp`, 'Ghost:' atoms, 6-31+G*) and
    `templates/calculate_reaction_energy.py` import unchanged; the counterpoise molecules they build parse."""
    inter = _load("calculate_interaction")
    _load("calculate_reaction_energy")
    atoms1, coords1 = inter.smiles_to_xyz("O")
    ghost = ["Ghost:" + a for a in atoms1]
    coords = np.vstack([coords1, coords1 + np.array([0.0, 0.0, 3.0])])
    mol = inter.create_pyscf_mol(list(atoms1) + ghost, coords, "6-31+G*")
    assert mol.nelectron == 10 and mol.natm == 6 and list(mol.atom_charges()) == [8, 1, 1, 0, 0, 0]
    from pyscf import mp
    assert callable(mp.MP2)


def test_ir_spectrum_template_imports():
    """`templates/calculate_ir_spectrum.py` imports unchanged (`pyscf.prop.infrared`, `pyscf.hessian`)."""
    mod = _load("calculate_ir_spectrum")
    assert callable(mod.calculate_ir_spectrum)
    from pyscf.prop import infrared
    assert callable(infrared.RHF) and callable(infrared.RKS)


def test_reaction_energy_template_surface():
    """`templates/calculate_reaction_energy.py`: `gto.M(...)`, `scf.rhf.RHF`, `scf.rohf.ROHF`, `scf.uhf.UHF`, `dft.rks.RKS` in its
    isinstance dispatch (`:167-174`), `hessian.{RHF,UHF,RKS,UKS}`, `thermo`."""
    mod = _load("calculate_reaction_energy")
    from pyscf import gto, scf, dft, hessian
    h = gto.M(atom="H 0 0 0", basis="sto-3g", charge=0, spin=1)      # calculate_reaction_energy.py:86
    assert h.nelectron == 1 and h.spin == 1
    assert scf.rhf.RHF is scf.RHF and isinstance(scf.rohf.ROHF, type) and scf.uhf.UHF is scf.UHF
    assert all(callable(x) for x in (hessian.RHF, hessian.UHF, hessian.RKS, hessian.UKS))
    assert isinstance(dft.rks.RKS, type)
    assert callable(mod.thermo.harmonic_analysis)
