"""Density fitting (`mf.density_fit()`): three-/two-index integrals from the HIP Rys kernels
against the oracle's brute-force restatement, fitted J/K against dense numpy algebra (1e-9), and the fitted SCF energy
against the exact-integral one (fitting error of the generated even-tempered auxiliary basis, l_aux <= 3)."""
import numpy as np
import pytest
import torch

from conftest import MOLECULES

pytestmark = pytest.mark.gpu


def _sym(n, seed):
    a = np.random.default_rng(seed).standard_normal((n, n))
    return 0.5 * (a + a.T)


@pytest.mark.parametrize("name,basis", [("h2o", "6-31g(d)"), ("h2co", "cc-pvdz")])
def test_df_integrals_and_jk_match_oracle(name, basis):
    from mi355scf import df
    from mi355scf.engine import Engine
    from mi355scf.mole import Mole
    from oracle import df as odf
    mol = Mole(atom=MOLECULES[name], basis=basis, verbose=0).build()
    eng = Engine(mol)
    d = df.DF(mol).build(eng)
    assert d.naux > 2 * mol.nao
    j3o, j2o = odf.integrals(mol, d.auxmol)
    assert np.abs(d.int2c.cpu().numpy() - j2o).max() < 1e-9
    D = _sym(mol.nao, 3)
    J, K = d.get_jk(torch.as_tensor(D, device=eng.device))
    Jo, Ko = odf.jk(j3o, j2o, D)
    assert np.abs(J.cpu().numpy() - Jo).max() < 1e-9 * max(1.0, np.abs(Jo).max())
    assert np.abs(K.cpu().numpy() - Ko).max() < 1e-9 * max(1.0, np.abs(Ko).max())
    # fitted J/K approximate the exact ones (the point of the fit), without being equal to them
    Je, Ke = eng.get_jk(D)
    # (a random density stresses the d x d products, which would need l_aux = 4: the engine's auxiliary functions stop at f)
    assert 1e-9 < float((J - Je).abs().max()) < 0.3 and float((K - Ke).abs().max()) < 0.3


def test_density_fitted_scf_energy_close_to_exact():
    from mi355scf import fixtures
    from pyscf import gto, scf, dft
    # generated even-tempered auxiliary basis with l_aux <= 4 (g): measured 3.8e-5 / 7.0e-5 Ha (tools/df_beta.py); 1e-4 is what
    # VERDICT r01 item 7 asked for (with l_aux <= 3 the error saturated at 2e-4 whatever the even-tempered ratio)
    for atom, basis, tol in ((MOLECULES["h2co"], "6-31G(d)", 1e-4), (fixtures.BENZENE, "cc-pVDZ", 1e-4)):
        mol = gto.Mole()
        mol.atom, mol.basis, mol.verbose = atom, basis, 0
        mol.build()
        e0 = scf.RHF(mol).kernel()
        mf = scf.RHF(mol).density_fit()
        e1 = mf.kernel()
        assert mf.converged and mf.with_df is not None
        assert 1e-9 < abs(e1 - e0) < tol, (e1, e0)
        # the SCF's exchange builds went through the pivoted-Cholesky factor of the density (4 N^2 N_aux n_occ flops); the dense
        # 4 N^3 N_aux route (any density: rank_hint unset) gives the same energy
        assert mf.with_df.k_path == "low rank"
        mf.with_df.rank_hint = None
        dm = mf.make_rdm1()
        vj, vk = mf.get_jk(mol, dm)
        assert mf.with_df.k_path == "dense"
        mf.with_df.rank_hint = mol.nelectron // 2
        vj2, vk2 = mf.get_jk(mol, dm)
        assert mf.with_df.k_path == "low rank" and np.abs(np.asarray(vk2) - np.asarray(vk)).max() < 1e-10
    ks = dft.RKS(mol, xc="B3LYP").density_fit()
    ek = ks.kernel()
    assert ks.converged and abs(ek - dft.RKS(mol, xc="B3LYP").kernel()) < 1e-3
    g = mf.nuc_grad_method().kernel()      # analytic gradient of the fitted energy (tests below)
    assert np.abs(g.sum(axis=0)).max() < 1e-7


def _displaced(mol, ia, x, h):
    R = mol.atom_coords().copy()
    R[ia, x] += h
    return mol.set_geom_(R, unit="Bohr", inplace=False)


@pytest.mark.parametrize("name,basis", [("h2o", "6-31g(d)"), ("h2co", "6-31g")])
def test_df_derivative_integrals_match_oracle(name, basis):
    """`mi_df_grad` (three- and two-index derivative integrals contracted with dense densities on the fly) against the
    ORACLE's fitted integrals at displaced geometries: F(R) = sum Z3 (ab|P)(R) + sum Z2 (P|Q)(R) with random symmetric Z3, Z2,
    fourth-order central differences (h, 2h)."""
    from mi355scf import df
    from mi355scf.engine import Engine
    from mi355scf.mole import Mole
    from oracle import df as odf
    mol = Mole(atom=MOLECULES[name], basis=basis, verbose=0).build()
    eng = Engine(mol)
    d = df.DF(mol).build(eng)
    n, na = mol.nao, d.naux
    rng = np.random.default_rng(11)
    z3 = rng.standard_normal((n, n, na))
    z3 = 0.5 * (z3 + z3.transpose(1, 0, 2))
    z2 = _sym(na, 12)
    aux_eng = Engine(d._aux_packed, device=eng.device)
    for a3, a2 in ((z3, None), (None, z2), (z3, z2)):
        g = torch.zeros(mol.natm, 3, dtype=torch.float64, device=eng.device)
        eng.df_grad(aux_eng, torch.as_tensor(a3, device=eng.device).contiguous() if a3 is not None else None,
                    torch.as_tensor(a2, device=eng.device).contiguous() if a2 is not None else None, g)
        g = g.cpu().numpy()
        assert np.abs(g.sum(axis=0)).max() < 1e-9 * max(1.0, np.abs(g).max())      # translational invariance

        def F(m):
            am = Mole(atom=[(s_, xyz) for s_, xyz in m._atom], basis=df.even_tempered_aux(m), unit="Bohr", verbose=0).build()
            j3, j2 = odf.integrals(m, am)
            return (0.0 if a3 is None else float((a3 * j3).sum())) + (0.0 if a2 is None else float((a2 * j2).sum()))
        h = 2e-3
        ref = np.zeros((mol.natm, 3))
        for ia in range(mol.natm if a3 is None or a2 is None else 1):      # the combined call: one atom is enough
            for x in range(3):
                d1 = (F(_displaced(mol, ia, x, h)) - F(_displaced(mol, ia, x, -h))) / (2 * h)
                d2 = (F(_displaced(mol, ia, x, 2 * h)) - F(_displaced(mol, ia, x, -2 * h))) / (4 * h)
                ref[ia, x] = (4.0 * d1 - d2) / 3.0
        rows = slice(0, mol.natm if a3 is None or a2 is None else 1)
        assert np.abs(g[rows] - ref[rows]).max() < 1e-7 * max(1.0, np.abs(ref).max()), (g, ref)
    aux_eng.close()


@pytest.mark.parametrize("method", ["RHF", "B3LYP", "UHF", "PBE", "UB3LYP"])
def test_density_fitted_gradient_matches_finite_difference_of_the_fitted_energy(method):
    """`mf.density_fit().nuc_grad_method().kernel()` against central differences of the fitted SCF energy: 1e-6 for HF; for the
    functionals the analytic gradient leaves out the grid-weight response (as PySCF does by default [MEM]; 2e-4, see
    test_gpu_grad), which the exact-integral gradient shares: there g_DF - g_exact is compared with FD(E_DF - E_exact)."""
    from pyscf import gto, scf, dft
    ks = method not in ("RHF", "UHF")
    open_shell = method.startswith("U")

    def make(atom, fit, unit="Angstrom"):
        mol = gto.Mole()
        mol.atom, mol.basis, mol.verbose, mol.unit = atom, "6-31G(d)", 0, unit
        if open_shell:
            mol.charge, mol.spin = 1, 1
        mol.build()
        if method == "UB3LYP":
            mf = dft.UKS(mol)
            mf.xc = "B3LYP"
        else:
            mf = scf.RHF(mol) if method == "RHF" else scf.UHF(mol) if method == "UHF" else dft.RKS(mol, xc=method)
        if fit:
            mf = mf.density_fit()
        mf.conv_tol = 1e-12
        return mol, mf
    mol, mf = make(MOLECULES["h2o"], True)
    mf.kernel()
    assert mf.converged
    g = mf.nuc_grad_method().kernel()
    # exchange densities through the low-rank factor of each spin density == the dense N^2 N_aux^2 route
    dms = mf._dm if not open_shell else [mf._dm[0], mf._dm[1]]
    ga, gb = (mf.with_df.grad_jk(dms, 1.0, factorize=f).cpu().numpy() for f in (True, False))
    assert np.abs(ga - gb).max() < 1e-10 and np.abs(ga).max() > 1e-2
    if ks:
        _m, mfe = make(MOLECULES["h2o"], False)
        mfe.kernel()
        g = g - mfe.nuc_grad_method().kernel()
    assert np.abs(g.sum(axis=0)).max() < 1e-7
    R = mol.atom_coords()
    h = 1e-3
    for ia, x in ((0, 2), (1, 1), (2, 2)):
        e = []
        for sgn in (+1, -1):
            Rd = R.copy()
            Rd[ia, x] += sgn * h
            atom = [(mol.atom_symbol(k), tuple(Rd[k])) for k in range(mol.natm)]
            _m, mfd = make(atom, True, unit="Bohr")
            ed = mfd.kernel()
            assert mfd.converged
            if ks:
                _m, mfx = make(atom, False, unit="Bohr")
                ed -= mfx.kernel()
            e.append(ed)
        fd = (e[0] - e[1]) / (2 * h)
        assert abs(g[ia, x] - fd) < 1e-6, (method, ia, x, g[ia, x], fd)
