import sys, time
sys.path.insert(0, "computational-chemistry-ai_amd/python")
import torch
from mi355scf import smiles_fixtures, df as dfm
from pyscf import gto, scf, dft
sym, xyz = smiles_fixtures.TABLE["CC(C)Cc1ccc(cc1)C(C)C(=O)O"]()
mol = gto.Mole(); mol.atom = "; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz)); mol.basis = "def2-TZVP"; mol.verbose = 0; mol.build()
mf = dft.RKS(mol, xc="B3LYP").density_fit()
mf.kernel()
d = mf.with_df
B = d._B
n, na, _ = B.shape
dm = mf._dm
def t(label, fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); print(f"{label:40s} {(time.time()-t0)/reps*1e3:8.2f} ms", flush=True); return out
rho = t("rho bmm", lambda: torch.bmm(B, dm.unsqueeze(2)).sum(dim=0).squeeze(1))
t("rho einsum", lambda: torch.einsum("ipj,ij->p", B, dm))
t("rho matmul view", lambda: (B.transpose(0, 1).reshape(na, n * n) if False else torch.matmul(B.permute(1, 0, 2).reshape(na, -1), dm.reshape(-1))) if False else (B * dm.unsqueeze(1)).sum(dim=(0, 2)))
t("J matmul(rho,B)", lambda: torch.matmul(rho, B))
t("J einsum", lambda: torch.einsum("p,ipj->ij", rho, B))
L = t("pivoted cholesky", lambda: dfm.pivoted_cholesky(dm, d.rank_hint))
Y = t("Y = B L", lambda: torch.matmul(B.reshape(n * na, n), L).reshape(n, na * L.shape[1]))
K = torch.zeros(n, n, dtype=torch.float64, device=B.device)
t("K += Y Y^T (xc_vmat)", lambda: d._eng.xc_vmat(Y, Y, K))
t("K = Y Y^T (matmul)", lambda: Y @ Y.t())
t("whole get_jk", lambda: d.get_jk(dm))
