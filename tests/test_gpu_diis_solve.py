"""Device-side Pulay solve (`mi_diis_solve`, row a10) against numpy.linalg.solve -- what PySCF's lib.diis does on the host
(call site in the reference: every `mf.kernel()`, templates/calculate_energy.py:205)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_diis_solve_matches_numpy_and_updates_the_gram_matrix():
    import torch
    from mi355scf.mole import Mole
    from mi355scf.engine import Engine
    from mi355scf import fixtures
    eng = Engine(Mole(atom=fixtures.H2O, basis="sto-3g", verbose=0).build())
    rng = np.random.default_rng(1)
    space, worst = 8, 0.0
    for _ in range(200):
        m = int(rng.integers(1, space + 1))
        slot = int(rng.integers(0, m))
        V = rng.normal(size=(space, 12)) * 10.0 ** int(rng.integers(-8, 1))      # error vectors from 1e-8 to 1 in size
        Bfull = V @ V.T
        B = Bfull.copy()
        B[slot, :] = B[:, slot] = -7.0                                             # stale row / column: must be replaced
        part = np.zeros((space, 16))
        for i in range(m):
            part[i] = Bfull[i, slot] / 16.0
        Bd = torch.as_tensor(B, device="cuda").contiguous()
        cd = torch.zeros(space, dtype=torch.float64, device="cuda")
        eng.diis_solve(torch.as_tensor(part.ravel(), device="cuda"), m, slot, space, Bd, cd)
        A = np.zeros((m + 1, m + 1))
        A[0, 1:] = A[1:, 0] = 1.0
        A[1:, 1:] = Bfull[:m, :m]
        rhs = np.zeros(m + 1)
        rhs[0] = 1.0
        ref = np.linalg.solve(A, rhs)[1:]
        got = cd.cpu().numpy()[:m]
        assert abs(got.sum() - 1.0) < 1e-9
        worst = max(worst, np.abs(got - ref).max() / max(1.0, np.abs(ref).max()) / max(1.0, np.linalg.cond(A) * 1e-16 / 1e-9))
        Bn = Bd.cpu().numpy()
        assert np.allclose(Bn[:m, slot], Bfull[:m, slot], rtol=1e-13, atol=0) and np.allclose(Bn[slot, :m], Bfull[slot, :m], rtol=1e-13, atol=0)
    assert worst < 1e-9, worst
    # singular system (all error vectors zero): no extrapolation, the newest Fock matrix alone
    Bd = torch.zeros(space, space, dtype=torch.float64, device="cuda")
    cd = torch.zeros(space, dtype=torch.float64, device="cuda")
    eng.diis_solve(torch.zeros(space * 16, dtype=torch.float64, device="cuda"), 3, 1, space, Bd, cd)
    assert np.array_equal(cd.cpu().numpy()[:3], [0.0, 1.0, 0.0])
