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
