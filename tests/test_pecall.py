"""PECaller per-(site, sample) genotype likelihoods (fill_sample_like, src/pecaller.c:2448-2507).

Golden vectors: tests/golden/pecall_like.npz, captured from an instrumented scratch build of the reference
(tests/golden/make_golden_pecall.py): for 4032 (site, pass) records of a 6-sample run, the inputs of the function
(counts, alpha means, norm) and its outputs (like[14], initial_call, initial_p)."""
import os
import numpy as np
import pytest
import fixtures
import oracle_py


def _gold():
    return np.load(os.path.join(fixtures.GOLD, "pecall_like.npz"))


def _by_norm(g):
    for nv in np.unique(g["norm"]):
        idx = np.nonzero(g["norm"] == nv)[0]
        yield float(nv), idx


def _check(like, best, margin, g, idx, tol):
    called = g["tot"][idx] > 2
    # the reference leaves like[] of skipped samples untouched (stale); only called samples are defined
    assert np.array_equal(best[called], g["best"][idx][called])
    assert np.all(best[~called] == 14)
    d = np.abs(like[called] - g["like"][idx][called])
    assert d.max() <= tol, d.max()
    dm = np.abs(margin[called] - g["margin"][idx][called])
    assert dm.max() <= tol
    return d.max()


def test_oracle_matches_reference_dump():
    """bit-exact on the host: same formula, same libm"""
    g = _gold()
    assert len(g["norm"]) == 4032 and set(np.unique(g["passes"])) == {1, 2, 3, 4, 5}
    for nv, idx in _by_norm(g):
        like, best, margin = oracle_py.site_like(g["reads"][idx], g["alpha"][idx], nv)
        assert _check(like, best, margin, g, idx, 0.0) == 0.0


def test_factln_table_formula():
    L = oracle_py.pecall_lib()
    import math
    assert L.ora_factln(0) == 0.0 and L.ora_factln(1) == 0.0
    for n in (2, 5, 40):
        assert abs(L.ora_factln(n) - math.lgamma(n + 1)) < 1e-12
    # above 40 the reference switches to the truncated 6-term Lanczos series (pecaller.c:3163-3181): ~1e-10 off lgamma
    # just above the switch, which is why the table must come from the reference's formula and not from lgamma
    assert 1e-11 < abs(L.ora_factln(41) - math.lgamma(42)) < 1e-9
    for n in (100, 5000, 20000):
        assert abs(L.ora_factln(n) - math.lgamma(n + 1)) < 1e-9


@pytest.mark.gpu
def test_gpu_matches_reference_dump_and_oracle():
    """north_star tolerance for the caller: 1e-6 on fp64 likelihoods (here they come out identical to ~1e-12), calls exact"""
    from pecaller_amd.pecall import PecallDev
    g = _gold()
    dev = PecallDev(0)
    worst = 0.0
    for nv, idx in _by_norm(g):
        like, best, margin = dev.site_like(g["reads"][idx], g["alpha"][idx], nv)
        worst = max(worst, _check(like, best, margin, g, idx, 1e-6))
        ol, ob, om = oracle_py.site_like(g["reads"][idx], g["alpha"][idx], nv)
        assert np.array_equal(best, ob)
        assert np.abs(like - ol).max() <= 1e-6
    assert worst < 1e-9
    dev.close()


@pytest.mark.gpu
def test_gpu_64_samples_30x_and_deep_counts():
    """config 5 shape: 64 samples, 30x; plus counts beyond the 10000-entry ln n! table (device gammln branch)"""
    from pecaller_amd.pecall import PecallDev
    rng = np.random.default_rng(5)
    n_sites, indiv = 3000, 64
    depth = rng.poisson(30, size=(n_sites, indiv))
    reads = np.zeros((n_sites, indiv, 6), np.uint16)
    ref = rng.integers(0, 4, n_sites)
    for a in range(4):
        reads[:, :, a] = rng.binomial(depth, np.where(ref[:, None] == a, 0.988, 0.004))
    reads[::97, ::5, 4] = 7
    reads[5::211, 3, :4] = [9000, 4000, 2500, 30]          # tot > 10000
    alpha = rng.dirichlet(np.ones(6) * 0.3, size=(n_sites, 14))
    dev = PecallDev(0)
    for norm in (1.0, 2.5, 6.25):
        like, best, margin = dev.site_like(reads, alpha, norm)
        ol, ob, om = oracle_py.site_like(reads, alpha, norm)
        assert np.array_equal(best, ob)
        assert np.abs(like - ol).max() <= 1e-6 and np.abs(margin - om).max() <= 1e-6
    dev.close()


def test_site_oracle_threads_do_not_change_results():
    """tests/oracle_py.call_sites hands blocks of columns to several callers side by side (columns are independent): the same arrays as
    one caller over all of them, with a pedigree as well (its de-novo counts travel per column)"""
    rng = np.random.default_rng(4)
    n, n_sites = 70, 300
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    dom[::41] = 14
    reads = np.zeros((n_sites, n, 6), np.int64)
    idx = np.arange(n_sites)
    is_var = rng.random(n_sites) < 0.2
    alt = rng.integers(0, 6, n_sites)
    for i in range(n):
        d = rng.poisson(25, n_sites)
        dose = np.where(is_var, rng.binomial(2, 0.3, n_sites), 0)
        e = rng.binomial(d, 0.004)
        ar = rng.binomial(d - e, dose / 2.0)
        reads[idx, i, np.where(dom < 4, dom, 0)] += d - e - ar
        reads[idx, i, alt] += ar
        reads[idx, i, rng.integers(0, 4, n_sites)] += e
    reads = reads.astype(np.uint16)
    dad = np.full(n, -1, np.int32)
    mom = np.full(n, -1, np.int32)
    sex = (1 + (np.arange(n) % 2)).astype(np.int32)
    for c in range(0, 18, 3):
        dad[c], mom[c] = c + 1, c + 2
        sex[c + 1], sex[c + 2] = 1, 2
    for ped in (None, dict(dad=dad, mom=mom, sex=sex, denovo_rate=1e-5)):
        one = oracle_py.call_sites(reads, dom, ped=ped, threads=1)
        den1 = oracle_py.call_sites.denovo.copy()
        many = oracle_py.call_sites(reads, dom, ped=ped, threads=5)
        assert all(np.array_equal(a, b) for a, b in zip(one, many))
        assert np.array_equal(den1, oracle_py.call_sites.denovo)
    assert (one[2] > 0).sum() > 20
