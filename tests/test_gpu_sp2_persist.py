"""Planned purification as ONE resident launch with grid barriers (`sp2_plan_persist_kernel`) against the one-launch-per-pass
sequence (`sp2_plan_kernel`) and against the projector from `eigh` -- row a11, the density from the Fock matrix (the reference
reaches it inside `mf.kernel()`, templates/calculate_energy.py:205)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("atom,basis", [("H2O", "sto-3g"), ("BENZENE", "cc-pVDZ"), ("BENZENE", "cc-pVTZ")])
def test_resident_launch_is_bit_identical_to_the_pass_per_launch_sequence(atom, basis):
    import torch
    from mi355scf.mole import Mole
    from mi355scf.engine import Engine
    from mi355scf import fixtures, sp2plan
    eng = Engine(Mole(atom=getattr(fixtures, atom), basis=basis, verbose=0).build())
    n = eng.nao
    nocc = max(1, n // 5)
    rng = np.random.default_rng(n)
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    e = np.sort(np.concatenate([rng.uniform(-20.0, -0.4, nocc), rng.uniform(0.1, 30.0, n - nocc)]))
    F = (q * e) @ q.T
    F = 0.5 * (F + F.T)
    P = q[:, :nocc] @ q[:, :nocc].T
    dev = eng.device
    Fd = torch.as_tensor(F, device=dev)
    nbd = (n + 15) // 16
    for margin in (0.15, 0.02):                       # two plans of different length (even / odd pass counts are both exercised)
        for extra in (0, 1):
            coef = sp2plan.plan(*sp2plan.bounds_from_spectrum(e, nocc, inner_margin=margin))
            assert coef is not None
            if extra:                                 # one more (idempotent-preserving) pass: x -> x^2 folds nothing new at the end
                coef = np.vstack([coef, [[1.0, 0.0, 0.0]]])
            out = {}
            for persist in (0, 1, 2):                # 2: write-through stores / L2-bypassing loads instead of cache maintenance
                eng.set_option("sp2_persist", persist)
                A = torch.full((2, n, n), float("nan"), dtype=torch.float64, device=dev)
                B = torch.full((2, n, n), float("nan"), dtype=torch.float64, device=dev)
                tr = torch.zeros(64 * 80, dtype=torch.float64, device=dev)
                for _ in range(3):                    # repeated launches reuse the monotonic barrier counter
                    res, off = eng.sp2_iterate_planned(Fd, A, B, coef, tr, out_scale=2.0)
                torch.cuda.synchronize()
                out[persist] = (res[0].cpu().numpy().copy(), tr[:off + 64].cpu().numpy().copy(), off)
            eng.set_option("sp2_persist", 0)
            for v in (1, 2):
                assert out[0][2] == out[v][2]
                assert np.array_equal(out[0][0], out[v][0])
                assert np.array_equal(out[0][1], out[v][1])      # the partial traces of EVERY pass
            X = out[1][0]
            assert np.abs(X - 2.0 * P).max() < 1e-10
            t = out[1][1][out[1][2]:out[1][2] + 2 * nbd].reshape(nbd, 2).sum(axis=0)
            assert abs(t[0] - nocc) < 1e-9 and abs(t[0] - t[1]) < 1e-9
