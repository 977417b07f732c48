"""Replays, on the GPU, the exact call sequence of `templates/calculate_energy.py:105-254` and
`templates/optimize_geometry.py:58-108` (the scripts themselves live under /root/reference, which does not
exist on the GPU box; tests/test_template_surface.py covers the unchanged-import side)."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class MultiWriter:
    def __init__(self, streams):
        self.streams = streams

    def write(self, m):
        for s in self.streams:
            s.write(m)

    def flush(self):
        pass


def _mol_from_smiles(smiles, basis, stream=None):
    from rdkit import Chem
    from rdkit.Chem import AllChem
    from pyscf import gto
    m = Chem.AddHs(Chem.MolFromSmiles(smiles))
    AllChem.EmbedMolecule(m, randomSeed=42)
    AllChem.MMFFOptimizeMolecule(m)
    conf = m.GetConformer()
    atom_str = ""
    for a in m.GetAtoms():
        p = conf.GetAtomPosition(a.GetIdx())
        atom_str += f"{a.GetSymbol()} {p.x:.6f} {p.y:.6f} {p.z:.6f}; "
    mol = gto.Mole()
    mol.atom = atom_str
    mol.basis = basis
    mol.charge = 0
    mol.spin = 0
    mol.verbose = 4
    if stream is not None:
        mol.output = None
        mol.stdout = stream
    mol.build()
    return mol


@pytest.mark.parametrize("method", ["HF", "B3LYP", "PBE"])
def test_calculate_energy_flow(method):
    import cupy, gpu4pyscf  # noqa: F401
    from gpu4pyscf.dft import rks as gpu_rks
    from gpu4pyscf.scf import hf as gpu_hf
    from pyscf import scf, dft
    log = io.StringIO()
    mol = _mol_from_smiles("C=O", "6-31G(d)", MultiWriter([log]))
    # GPU rung 1 (calculate_energy.py:145-156)
    mf = gpu_hf.RHF(mol) if method == "HF" else gpu_rks.RKS(mol)
    if method != "HF":
        mf.xc = method
    mf.init_guess = "atom"
    mf = mf.to_gpu()
    energy = mf.kernel()
    assert isinstance(energy, float)
    assert "converged SCF energy" in log.getvalue() and "cycle= 1" in log.getvalue()
    # hybrid rung 2 (calculate_energy.py:157-178)
    mf_cpu = scf.RHF(mol) if method == "HF" else dft.RKS(mol)
    if method != "HF":
        mf_cpu.xc = method
    mf_cpu.max_cycle = 5
    mf_cpu.kernel()
    dm = mf_cpu.make_rdm1()
    mf2 = gpu_hf.RHF(mol) if method == "HF" else gpu_rks.RKS(mol)
    if method != "HF":
        mf2.xc = method
    mf2 = mf2.to_gpu()
    e2 = mf2.kernel(dm0=dm)
    assert abs(e2 - energy) < 1e-8
    # analyze_orbitals (calculate_energy.py:208-242)
    mo_energy, mo_occ = mf.mo_energy, mf.mo_occ
    if hasattr(mo_energy, "get"):
        mo_energy = mo_energy.get()
    homo = np.where(mo_occ > 0)[0][-1]
    assert homo == 7 and mo_energy[homo + 1] > mo_energy[homo]
    # calculate_dipole (calculate_energy.py:244-254)
    mf_c = mf.to_cpu() if hasattr(mf, "to_cpu") else mf
    dip = mf_c.dip_moment(mf_c.mol, mf_c.make_rdm1(), unit="Debye")
    assert 1.5 < np.linalg.norm(dip) < 3.5


def test_optimize_geometry_flow():
    import torch
    import gpu4pyscf
    from pyscf import scf
    from pyscf.geomopt.geometric_solver import optimize
    assert torch.cuda.is_available() and isinstance(torch.cuda.get_device_name(0), str)
    mol = _mol_from_smiles("O", "6-31G")
    mol.verbose = 0
    mf = gpu4pyscf.scf.RHF(mol).to_gpu()
    e_init = mf.kernel()
    mol_opt = optimize(mf, maxsteps=30)
    mf_opt = scf.RHF(mol_opt)
    e_opt = mf_opt.kernel()
    assert e_opt <= e_init + 1e-9
    opt_coords = mol_opt.atom_coords() * 0.529177
    assert opt_coords.shape == (3, 3)
    roh = np.linalg.norm(opt_coords[0] - opt_coords[1])
    assert 0.93 < roh < 0.97      # RHF/6-31G water r(OH) ~0.95 A


def test_bde_template_call_sequence_methane():
    """The call sequence of `templates/calculate_bde.py:181-236` (gpu4pyscf.dft.RKS / UKS by `spin`, `mf.xc = method`,
    `conv_tol = 1e-6`, `max_cycle = 100`, `optimize(mf, maxsteps=100)`, `mf.__class__(mol_eq)`, `mf_opt.xc = mf.xc`) for
    CH4 -> CH3 + H with B3LYP/6-31G(d) (the template's default M06-2X is a meta-GGA and stays out of scope)."""
    import gpu4pyscf
    from pyscf import gto
    from pyscf.geomopt.geometric_solver import optimize

    def create_pyscf_mol(atoms, coords, basis, charge=0, spin=0):   # calculate_bde.py:77-103
        mol = gto.Mole()
        mol.atom = [[a, tuple(c)] for a, c in zip(atoms, coords)]
        mol.basis, mol.charge, mol.spin, mol.verbose = basis, charge, spin, 0
        mol.build()
        return mol

    def optimize_fragment(atoms, coords, spin):
        mol = create_pyscf_mol(atoms, coords, "6-31G(d)", 0, spin)
        mf = (gpu4pyscf.dft.RKS(mol) if spin == 0 else gpu4pyscf.dft.UKS(mol)).to_gpu()
        mf.xc = "B3LYP"
        mf.verbose = 0
        mf.conv_tol = 1e-6
        mf.max_cycle = 100
        mol_eq = optimize(mf, maxsteps=100)
        mf_opt = mf.__class__(mol_eq)
        if hasattr(mf, "xc"):
            mf_opt.xc = mf.xc
        mf_opt.verbose = 0
        return mol_eq.atom_coords(), mf_opt.kernel(), mf_opt

    t = 0.629
    ch4 = [(0, 0, 0), (t, t, t), (-t, -t, t), (-t, t, -t), (t, -t, -t)]
    _, e_ch4, mf4 = optimize_fragment(["C", "H", "H", "H", "H"], ch4, 0)
    xyz3, e_ch3, mf3 = optimize_fragment(["C", "H", "H", "H"], ch4[:4], 1)
    mol_h = create_pyscf_mol(["H"], [(0, 0, 0)], "6-31G(d)", 0, 1)
    mf_h = gpu4pyscf.dft.UKS(mol_h).to_gpu()
    mf_h.xc = "B3LYP"
    e_h = mf_h.kernel()
    assert mf4.converged and mf3.converged and mf_h.converged
    bde = e_ch3 + e_h - e_ch4
    assert 0.165 < bde < 0.190, bde            # electronic C-H dissociation energy of methane ~ 0.178 Ha (112 kcal/mol)
    # the methyl radical relaxes to a planar D3h structure: the carbon sits in the plane of the three hydrogens
    c, h1, h2, h3 = xyz3
    nrm = np.cross(h2 - h1, h3 - h1)
    assert abs(np.dot(c - h1, nrm / np.linalg.norm(nrm))) < 0.02
    assert abs(mf3.spin_square()[0] - 0.75) < 0.02
