"""meta-GGA path (`--method M06-2X`, templates/calculate_energy.py:263, calculate_bde.py:105): the HIP
functional kernels against the complex-step oracle point by point, Kohn-Sham energies against the oracle's numpy SCF, the
open-shell kernel against the closed-shell one, analytic gradients against finite differences.  The M06-2X parameter tables
are unverified-memory in BOTH implementations; their limits are pinned in tests/test_host_logic.py."""
import numpy as np
import pytest
import torch

from conftest import MOLECULES

pytestmark = pytest.mark.gpu

TERMS = {"TPSS": [(1.0, "tpss_x"), (1.0, "tpss_c")], "M06-2X": [(1.0, "m062x_x"), (1.0, "m062x_c")]}


def _points(n, seed):
    rng = np.random.default_rng(seed)
    rho = 10 ** rng.uniform(-4, 1.5, n)
    x = 10 ** rng.uniform(-2, 1, n)                           # reduced gradient scale
    g = x * rho ** (4.0 / 3)
    tw = g * g / (8 * rho)
    tau = tw * (1 + 10 ** rng.uniform(-3, 1.5, n))            # tau >= tau_W
    return rho, g, tau


@pytest.mark.parametrize("name", ["TPSS", "M06-2X"])
def test_closed_shell_kernel_matches_complex_step_oracle(name):
    from mi355scf.dft import XC_IDS
    from mi355scf.engine import Engine
    from mi355scf.mole import Mole
    from oracle import dft as od
    eng = Engine(Mole(atom="H 0 0 0; H 0 0 0.74", basis="sto-3g", verbose=0).build())
    rho, g, tau = _points(4000, 1)
    terms = [(c, XC_IDS[k]) for c, k in TERMS[name]]
    r4 = torch.zeros(4, len(rho), dtype=torch.float64, device=eng.device)
    r4[0] = torch.as_tensor(rho)
    r4[1] = torch.as_tensor(g)
    w = torch.ones(len(rho), dtype=torch.float64, device=eng.device)
    e, wv = eng.xc_eval_mgga(terms, r4, torch.as_tensor(tau, device=eng.device), w)
    eo, vr, vs, vt = od.eval_xc_mgga(TERMS[name], rho, g * g, tau)
    scale = np.abs(eo) + 1e-12
    assert np.abs(e.cpu().numpy() - eo).max() < 1e-12 * max(1.0, np.abs(eo).max())
    wv = wv.cpu().numpy()
    assert (np.abs(2 * wv[0] - vr) / (np.abs(vr) + 1e-8)).max() < 1e-8
    assert (np.abs(wv[1] / (2 * g) - vs) * rho ** (4.0 / 3) * g / scale).max() < 1e-8      # energy-weighted
    assert (np.abs(4 * wv[4] - vt) * tau / scale).max() < 1e-8


@pytest.mark.parametrize("name", ["TPSS", "M06-2X"])
def test_spin_kernel_matches_oracle_and_closed_shell_limit(name):
    from mi355scf.dft import XC_IDS
    from mi355scf.engine import Engine
    from mi355scf.mole import Mole
    from oracle import dft as od
    eng = Engine(Mole(atom="H 0 0 0; H 0 0 0.74", basis="sto-3g", verbose=0).build())
    n = 3000
    ra, ga, ta = _points(n, 2)
    rb, gb, tb = _points(n, 3)
    rb[:300] = 0.0; gb[:300] = 0.0; tb[:300] = 0.0             # fully polarised points
    rng = np.random.default_rng(4)
    cosab = rng.uniform(-1, 1, n)
    gbx, gby = gb * cosab, gb * np.sqrt(1 - cosab ** 2)
    terms = [(c, XC_IDS[k]) for c, k in TERMS[name]]
    dev = eng.device

    def pack(r, gx, gy):
        t = torch.zeros(4, n, dtype=torch.float64, device=dev)
        t[0], t[1], t[2] = torch.as_tensor(r), torch.as_tensor(gx), torch.as_tensor(gy)
        return t
    w = torch.ones(n, dtype=torch.float64, device=dev)
    e, wva, wvb = eng.xc_eval_mgga_spin(terms, pack(ra, ga, 0 * ga), pack(rb, gbx, gby), torch.as_tensor(ta, device=dev),
                                        torch.as_tensor(tb, device=dev), w)
    saa, sab, sbb = ga * ga, ga * gbx, gb * gb
    eo, d = od.eval_xc_mgga_spin(TERMS[name], ra, rb, saa, sab, sbb, ta, tb)
    scale = np.abs(eo) + 1e-12
    assert np.abs(e.cpu().numpy() - eo).max() < 1e-12 * max(1.0, np.abs(eo).max())
    wva, wvb = wva.cpu().numpy(), wvb.cpu().numpy()
    ok = rb > 0
    assert (np.abs(2 * wva[0] - d[0]) / (np.abs(d[0]) + 1e-8)).max() < 1e-7
    assert (np.abs(2 * wvb[0] - d[1])[ok] / (np.abs(d[1][ok]) + 1e-8)).max() < 1e-7
    # gradient parts: wva = 2 v_aa ga + v_ab gb ; wvb = 2 v_bb gb + v_ab ga  (ga along x)
    assert (np.abs(wva[1] - (2 * d[2] * ga + d[3] * gbx)) * ga / scale).max() < 1e-7
    assert (np.abs(wvb[2] - 2 * d[4] * gby)[ok] * gb[ok] / scale[ok]).max() < 1e-7
    assert (np.abs(4 * wva[4] - d[5]) * ta / scale).max() < 1e-7
    assert (np.abs(4 * wvb[4] - d[6])[ok] * tb[ok] / scale[ok]).max() < 1e-7
    # closed-shell limit: spin kernel at (rho/2, rho/2) == closed-shell kernel
    rho, g, tau = _points(n, 5)
    r4 = pack(rho, g, 0 * g)
    e1, wv1 = eng.xc_eval_mgga(terms, r4, torch.as_tensor(tau, device=dev), w)
    e2, wa2, wb2 = eng.xc_eval_mgga_spin(terms, 0.5 * r4, 0.5 * r4, torch.as_tensor(0.5 * tau, device=dev), torch.as_tensor(0.5 * tau, device=dev), w)
    assert float((e1 - e2).abs().max()) < 1e-12 * max(1.0, float(e1.abs().max()))
    assert float((wv1[0] - wa2[0]).abs().max() / wv1[0].abs().max()) < 1e-10


@pytest.mark.parametrize("xc,basis", [("TPSS", "6-31g(d)"), ("M06-2X", "6-31g(d)")])
def test_rks_meta_gga_energy_matches_oracle(xc, basis):
    from mi355scf.mole import Mole
    from oracle import dft as od
    from pyscf import dft
    mol = Mole(atom=MOLECULES["h2o"], basis=basis, verbose=0).build()
    mf = dft.RKS(mol, xc=xc)
    mf.small_rho_cutoff = 0.0
    mf.init_guess = "1e"
    e = mf.kernel()
    r = od.rks_mgga(mol, xc)
    assert mf.converged and r["converged"]
    assert abs(e - r["e_tot"]) < 1e-7, (e, r["e_tot"])
    assert abs(float(mf._nelec_grid) - r["nelec_grid"]) < 1e-8
    # water: TPSS ~ -76.4, M06-2X ~ -76.3 with this small basis (sanity window, not a pin)
    assert -76.6 < e < -76.1


def test_meta_gga_gradient_matches_finite_difference_and_uks_equals_rks():
    from pyscf import dft, gto
    mol = gto.Mole()
    mol.atom, mol.basis, mol.verbose = MOLECULES["h2o"], "6-31g(d)", 0
    mol.build()
    for xc in ("TPSS", "M06-2X"):
        mf = dft.RKS(mol, xc=xc)
        mf.conv_tol = 1e-11
        mf.small_rho_cutoff = 0.0
        e = mf.kernel()
        g = mf.nuc_grad_method().kernel()
        uk = dft.UKS(mol, xc=xc)
        uk.conv_tol = 1e-11
        eu = uk.kernel()
        assert abs(e - eu) < 1e-8
        gu = uk.nuc_grad_method().kernel()
        assert np.abs(g - gu).max() < 1e-6
        # d/dR of E_xc[D] at fixed density on a grid FROZEN in space == analytic XC gradient without weight response
        dm = mf._dm
        gx = mf.nuc_grad_method().grad_xc(dm)
        coords0, w0 = mf.grids.coords.clone(), mf.grids.weights.clone()
        R = mol.atom_coords()
        h = 1e-3
        for ia, x in ((0, 2), (1, 1), (2, 2)):
            vals = []
            for sgn in (+1, -1):
                Rn = R.copy()
                Rn[ia, x] += sgn * h
                mf2 = dft.RKS(mol.set_geom_(Rn, unit="Bohr", inplace=False), xc=xc)
                mf2.mol.verbose = 0
                mf2._setup_once()
                mf2.grids.coords, mf2.grids.weights = coords0, w0
                vals.append(float(mf2.nr_rks(dm)[1]))
            assert abs((vals[0] - vals[1]) / (2 * h) - gx[ia, x]) < 2e-6, (xc, ia, x)
    # an open shell: OH radical, M06-2X (the BDE template's default functional on radicals), converges to a doublet
    oh = gto.Mole()
    oh.atom, oh.basis, oh.spin, oh.verbose = "O 0 0 0; H 0 0 0.97", "6-31g(d)", 1, 0
    oh.build()
    uk = dft.UKS(oh, xc="M06-2X")
    e = uk.kernel()
    assert uk.converged and -75.8 < e < -75.6 and abs(uk.spin_square()[0] - 0.75) < 0.02
