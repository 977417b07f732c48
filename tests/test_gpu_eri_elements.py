"""Element-wise pins of rows a3/a4: the resident tiles against the oracle's McMurchie-Davidson integrals,
integral by integral -- not only through J/K contractions -- and the Schwarz factors against the oracle's.
  * H2O/cc-pVTZ: the whole tensor (`mi_eri_unpack` vs `orc_eri_full`, 58^4 elements, s..f shells);
  * ibuprofen/def2-TZVP (BASELINE config 5, 105 GB resident) and C60/6-31G* shard 0 of 8 (config 4, 63 GB): seeded random
    shell quartets read back with `mi_eri_read_quartet`, stratified over the angular classes incl. (ff|ff) / (dd|dd).
Tolerance 1e-10 absolute (FP64; integrals are O(1e-3 .. 1))."""
import numpy as np
import pytest

from conftest import MOLECULES

pytestmark = pytest.mark.gpu


def _fixture_mol(smiles, basis):
    from mi355scf import smiles_fixtures
    from mi355scf.mole import Mole
    sym, xyz = smiles_fixtures.TABLE[smiles]()
    return Mole(atom="; ".join(f"{s} {x:.6f} {y:.6f} {z:.6f}" for s, (x, y, z) in zip(sym, xyz)), basis=basis, verbose=0).build()


def test_full_tensor_and_schwarz_h2o_ccpvtz():
    from mi355scf.engine import Engine
    from mi355scf.mole import Mole
    from oracle import oracle as orc
    mol = Mole(atom=MOLECULES["h2o"], basis="cc-pvtz", verbose=0).build()
    eng = Engine(mol)
    eng.prepare_eri(1e-13)
    o = orc.Oracle(mol)
    ref = o.eri_full()
    got = eng.eri_dense().cpu().numpy()
    assert np.abs(got - ref).max() < 1e-10, np.abs(got - ref).max()
    q, qo = eng.schwarz(), o.schwarz()
    kept = q > 0
    assert kept.sum() > 0.9 * q.size
    assert np.abs(q - qo)[kept].max() < 1e-11
    assert (qo[~kept] * qo.max() < 1e-13).all()          # dropped pairs really are negligible
    # a shell block read back through the debug ABI equals the same block of the dense tensor
    loc = mol.ao_loc_nr()
    for sh in ((mol.nbas - 1, 3, 7, 0), (5, 5, 5, 5), (2, 9, 11, 11)):
        blk = eng.eri_read_quartet(*sh)
        sl = tuple(slice(loc[s_], loc[s_ + 1]) for s_ in sh)
        assert np.abs(blk - ref[sl]).max() < 1e-10


def _sample_quartets(mol, n, seed, classes):
    """Seeded shell quartets: `classes` (tuples of four l) first, then uniformly random ones."""
    rng = np.random.default_rng(seed)
    ls = mol._bas[:, 1]
    by_l = {l: np.where(ls == l)[0] for l in set(ls.tolist())}
    out = []
    for cls in classes:
        for _ in range(n // (4 * len(classes))):
            out.append(tuple(int(rng.choice(by_l[l])) for l in cls))
    while len(out) < n:
        out.append(tuple(int(x) for x in rng.integers(0, mol.nbas, 4)))
    return out


def _compare_sampled(eng, o, quartets, allow_missing):
    nchk = nmiss = nscreened = 0
    worst = 0.0
    for sh in quartets:
        got = eng.eri_read_quartet(*sh)
        if np.isnan(got).any():                 # some of its tiles live on another rank: compare what is here
            assert allow_missing
            nmiss += 1
        ref = o.eri_shell(*sh)
        ok = ~np.isnan(got)
        if not ok.any():
            continue
        zero = ok & (got == 0.0)
        if zero.any():                          # Schwarz-screened tiles: the integrals are below the threshold
            nscreened += 1
            assert np.abs(ref[zero]).max() < 1e-11
        worst = max(worst, np.abs(got - ref)[ok].max())
        nchk += 1
    assert worst < 1e-10, worst
    return nchk, nmiss, nscreened


def test_sampled_quartets_ibuprofen_def2tzvp():
    """BASELINE config 5's tensor (N = 573, 237 shells, 105 GB of resident tiles)."""
    from mi355scf.engine import Engine, release_cache
    from oracle import oracle as orc
    mol = _fixture_mol("CC(C)Cc1ccc(cc1)C(C)C(=O)O", "def2-TZVP")
    assert (mol.nao, mol.nbas) == (573, 237)
    eng = Engine(mol)
    st = eng.prepare_eri(1e-13)
    assert st["stored_bytes"] > 90e9
    classes = [(3, 3, 3, 3), (2, 2, 2, 2), (3, 2, 1, 0), (3, 3, 2, 2), (1, 1, 1, 1), (2, 1, 2, 0), (0, 0, 0, 0), (3, 0, 3, 0)]
    nchk, _nmiss, _nscr = _compare_sampled(eng, orc.Oracle(mol), _sample_quartets(mol, 2000, 11, classes), allow_missing=False)
    assert nchk == 2000
    eng.close()
    release_cache()


def test_sampled_quartets_c60_631gs_shard():
    """BASELINE config 4: rank 0 of 8 of the C60/6-31G* store (N = 840, 360 shells, ~63 GB per rank)."""
    from mi355scf.engine import Engine, release_cache
    from oracle import oracle as orc
    mol = _fixture_mol("C60", "6-31G*")
    assert (mol.nao, mol.nbas) == (840, 360)
    eng = Engine(mol)
    st = eng.prepare_eri(1e-13, rank=0, nranks=8)
    assert 55e9 < st["stored_bytes"] < 72e9
    classes = [(2, 2, 2, 2), (2, 1, 1, 0), (1, 1, 1, 1), (2, 2, 1, 1), (0, 0, 0, 0), (2, 0, 2, 0)]
    nchk, nmiss, _nscr = _compare_sampled(eng, orc.Oracle(mol), _sample_quartets(mol, 2000, 12, classes), allow_missing=True)
    assert nchk > 300 and nmiss > 1000           # about 1/8 of the quartets is resident on this rank
    eng.close()
    release_cache()
