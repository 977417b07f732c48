import sys
sys.path.insert(0, "computational-chemistry-ai_amd/python"); sys.path.insert(0, "tests")
import torch
from conftest import MOLECULES
from pyscf import gto, scf
from mi355scf import df as dfm
orig = dfm.pivoted_cholesky
def wrapped(dm, rank, rtol=1e-10):
    L = orig(dm, rank, rtol)
    L2 = orig(dm, rank, 1.0)
    r = float((dm - L2 @ L2.t()).abs().max()) if L2 is not None else -1
    ev = torch.linalg.eigvalsh(dm)
    print("chol rank", rank, "ok" if L is not None else "FAIL", "resid %.2e" % r, "eig lo %.2e hi %.2e n>1e-8: %d" % (float(ev[0]), float(ev[-1]), int((ev > 1e-8).sum())), flush=True)
    return L
dfm.pivoted_cholesky = wrapped
mol = gto.Mole(); mol.atom = MOLECULES["h2co"]; mol.basis = "6-31G(d)"; mol.verbose = 0; mol.build()
mf = scf.RHF(mol).density_fit()
print(mf.kernel(), mf.with_df.k_path)
