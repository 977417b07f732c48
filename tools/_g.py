import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computational-chemistry-ai_amd", "python"))
import torch
from pyscf import gto, scf, dft
from mi355scf import smiles_fixtures
sym, xyz = smiles_fixtures.TABLE["CC(C)Cc1ccc(cc1)C(C)C(=O)O"]()
mol = gto.Mole(); mol.atom = "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz)); mol.basis = "def2-TZVP"; mol.verbose = 0; mol.build()
mf = dft.RKS(mol); mf.xc = "B3LYP"; mf = mf.to_gpu(); mf.kernel()
g = mf.nuc_grad_method()
t0 = time.time(); g.kernel(); torch.cuda.synchronize(); print("grad s", time.time() - t0)
