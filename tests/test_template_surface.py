"""The reference templates' import surface resolves against the drop-in packages (`pyscf`, `gpu4pyscf`, `cupy`, `rdkit`
stand-ins) -- checked STATICALLY: the scripts under /root/reference/templates are parsed with `ast` (read as text, never
imported or executed, nothing is written next to them), every `import` / `from ... import ...` and every dotted attribute
chain rooted at an imported chemistry module (`gto.Mole`, `scf.hf.RHF`, `gpu4pyscf.dft.rks.RKS`, `Chem.AddHs`, ...) is
collected and resolved against this repo's packages.  Runs here (CPU container, reference mounted); skipped on the GPU box
where /root/reference does not exist.  The call sequences themselves are replayed on the GPU in
tests/test_gpu_template_flow.py."""
import ast
import importlib
import os

import numpy as np
import pytest

TEMPLATES = "/root/reference/templates"
pytestmark = pytest.mark.skipif(not os.path.isdir(TEMPLATES), reason="reference not mounted")

OURS = ("pyscf", "gpu4pyscf", "cupy", "rdkit")   # top-level packages this repo stands in for


def _surface(name):
    """-> (imports, chains): imports = [(module, symbol|None)], chains = dotted names rooted at an alias of one of OURS."""
    with open(os.path.join(TEMPLATES, name + ".py"), "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read())
    imports, alias = [], {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split(".")[0] in OURS:
                    imports.append((a.name, None))
                    alias[a.asname or a.name.split(".")[0]] = a.name if a.asname else a.name.split(".")[0]
        elif isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in OURS:
            for a in node.names:
                imports.append((node.module, a.name))
                alias[a.asname or a.name] = node.module + "." + a.name
    chains = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            parts, cur = [], node
            while isinstance(cur, ast.Attribute):
                parts.append(cur.attr)
                cur = cur.value
            if isinstance(cur, ast.Name) and cur.id in alias:
                chains.add((alias[cur.id], tuple(reversed(parts))))
    return imports, chains


def _resolve(dotted):
    """Import the longest module prefix of `dotted`, then getattr the rest."""
    parts = dotted.split(".")
    for cut in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:cut]))
        except ImportError:
            continue
        for p in parts[cut:]:
            obj = getattr(obj, p)
        return obj
    raise ImportError(dotted)


def _check(name, max_depth=3):
    imports, chains = _surface(name)
    assert imports, f"{name}: no chemistry imports found"
    for mod, sym in imports:
        _resolve(mod if sym is None else mod + "." + sym)
    seen = 0
    for root, attrs in sorted(chains):
        obj = _resolve(root)
        # walk the chain while it stays on modules / classes / functions of the stand-ins: attributes of INSTANCES
        # (mf.e_tot, mol.natm, conformer positions ...) are covered by the GPU replay, not by a static check
        for a in attrs[:max_depth]:
            if not (isinstance(obj, type) or callable(obj) or type(obj).__name__ == "module"):
                break
            if isinstance(obj, type) or (callable(obj) and type(obj).__name__ != "module"):
                break   # a class or function: what follows is a call result
            assert hasattr(obj, a), f"{name}: {root}.{'.'.join(attrs)} -- '{a}' missing on {obj!r}"
            obj = getattr(obj, a)
            seen += 1
    return imports, chains, seen


@pytest.mark.parametrize("name", ["calculate_energy", "optimize_geometry"])
def test_in_scope_templates_resolve(name):
    """the two in-scope callers of the hot path."""
    imports, chains, seen = _check(name)
    mods = {m for m, _s in imports}
    assert "pyscf" in {m.split(".")[0] for m in mods} and "rdkit" in {m.split(".")[0] for m in mods}
    assert seen > 5


@pytest.mark.parametrize("name", ["calculate_bde", "opt-freq", "calculate_interaction", "calculate_reaction_energy",
                                  "calculate_ir_spectrum"])
def test_next_row_templates_resolve(name):
    """SURVEY.md section 8(f) rank 4 callers (UHF/UKS, Hessian, thermo) and the CPU drivers around the same SCF."""
    _check(name)


def test_surface_symbols_named_by_survey():
    """The symbol list of SURVEY.md section 8(b), spelled out."""
    import cupy
    import gpu4pyscf
    from gpu4pyscf.dft import rks as gpu_rks
    from gpu4pyscf.scf import hf as gpu_hf
    from pyscf import dft, gto, hessian, scf
    from pyscf.geomopt.geometric_solver import optimize
    from rdkit import Chem
    from rdkit.Chem import AllChem, Descriptors
    assert callable(gto.Mole) and callable(gto.M) and callable(optimize)
    assert scf.hf.RHF is scf.RHF and scf.rhf.RHF is scf.RHF and scf.uhf.UHF is scf.UHF and isinstance(scf.rohf.ROHF, type)
    assert isinstance(dft.rks.RKS, type) and isinstance(dft.uks.UKS, type) and dft.RKS is dft.rks.RKS
    assert gpu4pyscf.scf.RHF is gpu_hf.RHF and gpu4pyscf.dft.RKS is gpu_rks.RKS
    assert all(callable(x) for x in (hessian.RHF, hessian.UHF, hessian.RKS, hessian.UKS, hessian.thermo.harmonic_analysis))
    assert isinstance(cupy.__version__, str) and isinstance(cupy.cuda.runtime.runtimeGetVersion(), int)
    assert all(callable(x) for x in (Chem.MolFromSmiles, Chem.AddHs, AllChem.EmbedMolecule, AllChem.MMFFOptimizeMolecule,
                                     Chem.rdMolDescriptors.CalcMolFormula, Descriptors.MolWt))


def test_benchmark_molecules_build_with_the_sizes_of_the_survey():
    """SMILES stand-in -> `gto.Mole` exactly as `create_pyscf_mol` assembles it (calculate_energy.py:83-103: atom string of
    'El x y z' with 6 decimals joined by '; ', Angstrom), sizes of SURVEY.md section 8's config table."""
    from pyscf import gto
    from rdkit import Chem
    from rdkit.Chem import AllChem

    def build(smiles, basis, charge=0, spin=0):
        m = Chem.AddHs(Chem.MolFromSmiles(smiles))
        AllChem.EmbedMolecule(m, randomSeed=42)
        AllChem.MMFFOptimizeMolecule(m)
        conf = m.GetConformer()
        atoms = [a.GetSymbol() for a in m.GetAtoms()]
        xyz = np.array([[conf.GetAtomPosition(i).x, conf.GetAtomPosition(i).y, conf.GetAtomPosition(i).z] for i in range(len(atoms))])
        mol = gto.Mole()
        mol.atom = "; ".join(f"{a} {x:.6f} {y:.6f} {z:.6f}" for a, (x, y, z) in zip(atoms, xyz))
        mol.basis, mol.charge, mol.spin, mol.verbose = basis, charge, spin, 0
        mol.build()
        return mol

    for smiles, basis, natm, nelec, nao in (("C=O", "6-31G(d)", 4, 16, 32), ("c1ccccc1", "cc-pVDZ", 12, 42, 114),
                                            ("c1ccccc1", "cc-pVTZ", 12, 42, 264),
                                            ("CC(C)Cc1ccc(cc1)C(C)C(=O)O", "def2-TZVP", 33, 112, 573)):
        mol = build(smiles, basis)
        assert (mol.natm, mol.nelectron, mol.nao) == (natm, nelec, nao)
        assert mol.atom_coords().shape == (natm, 3)
    from mi355scf import smiles_fixtures
    sym, xyz = smiles_fixtures.TABLE["C60"]()
    c60 = gto.Mole()
    c60.atom = "; ".join(f"{a} {x:.6f} {y:.6f} {z:.6f}" for a, (x, y, z) in zip(sym, xyz))
    c60.basis, c60.verbose = "6-31G*", 0
    c60.build()
    assert (c60.natm, c60.nao, c60.nbas) == (60, 840, 360)       # BASELINE config 4
    ghost = gto.Mole()
    ghost.atom = "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587; Ghost:O 0 0 3; Ghost:H 0 -0.757 3.587; Ghost:H 0 0.757 3.587"
    ghost.basis, ghost.verbose = "6-31+G*", 0
    ghost.build()
    assert ghost.nelectron == 10 and list(ghost.atom_charges()) == [8, 1, 1, 0, 0, 0]   # calculate_interaction.py:127-157
