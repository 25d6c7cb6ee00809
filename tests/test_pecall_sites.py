"""PECaller per-site caller: the oracle (oracle/pecall_site_oracle.c) against the text the reference itself printed"""
import numpy as np
import pytest
import oracle_py
import pecall_sites_fixture as fx


@pytest.mark.parametrize("tag", ["pecall_sites", "pecall_ped", "pecall_wide", "pecall_wide300"])
def test_site_oracle_matches_reference_text(tag):
    f = fx.load(tag)
    call, p, typ, ac, npass = oracle_py.call_sites(f["reads"], f["dom"], ped=f.get("ped"))
    den = oracle_py.call_sites.denovo
    n_base = n_snp = 0
    bad = []
    for i, pos in enumerate(f["pos"]):
        pos1 = int(pos) + 1                      # every site of the fixture lies in the first contig
        exp = f["base_rows"].get(pos1)
        if exp is None:
            continue                              # no sample had a read there, or a non-ACGT reference base
        got = fx.base_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i])
        n_base += 1
        if got != exp:
            bad.append((pos1, exp, got))
        exp_s = f["snp_rows"].get(pos1)
        if typ[i] > 0:
            n_snp += 1
            got_s = fx.snp_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i], typ[i], ac[i], den[i])
            if got_s != exp_s:
                bad.append((pos1, exp_s, got_s))
        elif exp_s is not None:
            bad.append((pos1, exp_s, None))
    assert n_base == len(f["base_rows"]) and n_snp == len(f["snp_rows"]), (n_base, len(f["base_rows"]), n_snp, len(f["snp_rows"]))
    assert not bad, bad[:3]
    assert npass.max() >= 2                      # the fixture exercises the alpha re-estimation
    if tag == "pecall_wide300":
        assert f["reads"].shape[1] == 300       # beyond 256 samples (make_golden_pecall_wide.py --samples 300 --sites 400)
    if tag == "pecall_wide":
        assert f["reads"].shape[1] == 100       # more than 64 samples: the reference takes any INDIV (pecaller.c:251-257)
    if tag == "pecall_ped":
        assert (den[typ > 0] > 0).sum() >= 10     # DENOVO_ rows


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["pecall_sites", "pecall_wide", "pecall_wide300"])
def test_gpu_site_caller_matches_reference_text_and_oracle(tag):
    """the 8-sample fixture, 100 samples (two chunks of 64: a lane stands for a sample of each) and 300 samples (eight chunks: beyond
    the 256 of rounds 1-3) against the text the reference printed and against the oracle"""
    from pecaller_amd.pecall import PecallDev
    f = fx.load(tag)
    dev = PecallDev(0)
    call, p, typ, ac, npass = dev.call_sites(f["reads"], f["dom"])
    ocall, op, otyp, oac, onpass = oracle_py.call_sites(f["reads"], f["dom"])
    assert np.array_equal(call, ocall)
    assert np.array_equal(typ, otyp) and np.array_equal(ac, oac) and np.array_equal(npass, onpass)
    assert np.max(np.abs(p - op)) <= 1e-6          # north_star: genotype posteriors within 1e-6
    bad = []
    for i, pos in enumerate(f["pos"]):
        pos1 = int(pos) + 1
        exp = f["base_rows"].get(pos1)
        if exp is not None and fx.base_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i]) != exp:
            bad.append((pos1, exp))
    # every row as the reference printed it (%g, six significant digits): the device's posteriors have been bit-equal to the CPU's on
    # every column compared so far (bench.py: max_abs_dposterior 0.0 over 1.8 M columns)
    assert not bad, bad[:3]
    dev.close()


@pytest.mark.gpu
def test_gpu_site_caller_64_samples_and_haploid():
    """BASELINE config 5's width (64 samples), and the haploid mode, against the oracle on seeded columns"""
    from pecaller_amd.pecall import PecallDev
    rng = np.random.default_rng(5)
    n_sites, n = 600, 64
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    dom[::97] = 14                                   # reference N: skipped
    reads = np.zeros((n_sites, n, 6), np.uint16)
    for s in range(n_sites):
        var = rng.random() < 0.25
        q = rng.uniform(0.02, 0.5)
        alt = int(rng.integers(0, 6))
        for i in range(n):
            d = int(rng.poisson(30 if i % 9 else 4))
            r = int(dom[s]) if dom[s] < 4 else 0
            g = (alt if (var and rng.random() < q) else r, alt if (var and rng.random() < q) else r)
            for _ in range(d):
                al = g[int(rng.integers(0, 2))]
                if rng.random() < 0.004:
                    al = int(rng.integers(0, 4))
                if al == 5:
                    reads[s, i, r] += 1
                reads[s, i, al] += 1
    dev = PecallDev(0)
    for hap in (False, True):
        got = dev.call_sites(reads, dom, haploid=hap)
        exp = oracle_py.call_sites(reads, dom, haploid=hap)
        assert np.array_equal(got[0], exp[0]), ("calls", hap)
        assert np.max(np.abs(got[1] - exp[1])) <= 1e-6
        for a, b in zip(got[2:], exp[2:]):
            assert np.array_equal(a, b)
    dev.close()


def _wide_reference_columns(n, n_sites, seed, var_frac, deep_every):
    """mostly reference columns of n samples (30x, 0.4 % errors: a handful of samples per column miss the shortcut's margin by error reads
    alone), a few variant columns, a sample too deep for the head of the ln n! table in every deep_every-th column"""
    rng = np.random.default_rng(seed)
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    dom[::97] = 14
    depth = rng.integers(20, 40, n)
    depth[5] = 3
    is_var = rng.random(n_sites) < var_frac
    q = rng.uniform(0.01, 0.5, n_sites)
    alt = rng.integers(0, 6, n_sites)
    reads = np.zeros((n_sites, n, 6), np.int64)
    idx = np.arange(n_sites)
    r = np.where(dom < 4, dom, 0)
    for i in range(n):
        d = rng.poisson(depth[i], n_sites)
        dose = np.where(is_var, rng.binomial(2, q), 0)
        e = rng.binomial(d, 0.004)
        ar = rng.binomial(d - e, dose / 2.0)
        reads[idx, i, r] += d - e - ar
        reads[idx, i, alt] += ar
        reads[idx, i, r] += np.where(alt == 5, ar, 0)
        reads[idx, i, rng.integers(0, 4, n_sites)] += e
    reads[::deep_every, 7, r[::deep_every]] += 1600
    return reads.astype(np.uint16), dom


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_sites,var_frac", [(160, 800, 0.02), (300, 800, 0.02), (512, 160, 0.02)])
def test_gpu_site_caller_wide_small_beam(n, n_sites, var_frac):
    """beyond 128 samples the shortcut kernel works a chunk of 64 samples at a time and parks the unsettled samples' likelihoods in LDS for
    the small beam (round 4; pcs_fast_kernel<., 4|8>): columns whose unsettled samples are error reads only -- nearly all of a real run --
    against the oracle, with columns too deep for the table's head (listed for the beam search) and variant columns among them"""
    from pecaller_amd.pecall import PecallDev
    reads, dom = _wide_reference_columns(n, n_sites, 7000 + n, var_frac, 211)
    dev = PecallDev(0)
    got = dev.call_sites(reads, dom)
    exp = oracle_py.call_sites(reads, dom)
    assert np.array_equal(got[0], exp[0])
    assert np.max(np.abs(got[1] - exp[1])) <= 1e-6
    for a, b in zip(got[2:], exp[2:]):
        assert np.array_equal(a, b)
    assert (exp[4] == 1).sum() > 0.9 * n_sites and (exp[2] > 0).sum() >= 2
    # the same columns resident (sites_stage / sites_run: the heavy columns' beam search starts ahead of the shortcut kernels)
    dev.sites_stage(reads, dom)
    dev.sites_run()
    res = dev.sites_collect()
    for a, b in zip(res, got):
        assert np.array_equal(a, b)
    dev.close()


@pytest.mark.gpu
def test_gpu_site_caller_wide_small_beam_haploid_and_chromosome_types():
    """the wide shortcut kernel's small beam with HAPLOID set (6 genotypes, one allele per call) and with the column types that force it per
    column (chrY / chrMT in guide mode: bit 4 of the type) or change the site filter (chrY), 200 samples, against the oracle"""
    from pecaller_amd.pecall import PecallDev
    reads, dom = _wide_reference_columns(200, 600, 7777, 0.02, 211)
    rng = np.random.default_rng(12)
    chrom = rng.integers(0, 4, len(dom)).astype(np.uint8)
    chrom[chrom >= 2] |= 16
    dev = PecallDev(0)
    for kw in (dict(haploid=True), dict(chrom=chrom)):
        got = dev.call_sites(reads, dom, **kw)
        exp = oracle_py.call_sites(reads, dom, **kw)
        assert np.array_equal(got[0], exp[0]), kw.keys()
        assert np.max(np.abs(got[1] - exp[1])) <= 1e-6
        for a, b in zip(got[2:], exp[2:]):
            assert np.array_equal(a, b), kw.keys()
        assert (exp[4] == 1).sum() > 0.9 * len(dom)
    dev.close()


@pytest.mark.gpu
def test_gpu_site_caller_deep_columns():
    """columns with samples thousands of reads deep: the shortcut kernel then stages the whole ln n! table instead of its head
    (the head form leaves the deep columns on a list for a second launch with the whole table), and counts above 10,000 take the table's log formula; mixed with ordinary columns
    in the same call, a second call with only shallow columns (the head serves again)"""
    from pecaller_amd.pecall import PecallDev
    rng = np.random.default_rng(11)
    n_sites, n = 3000, 24
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    depth = np.where(rng.random((n_sites, n)) < 0.1, rng.integers(1500, 12000, (n_sites, n)), rng.poisson(30, (n_sites, n)))
    err = rng.binomial(depth, 0.004)
    reads = np.zeros((n_sites, n, 6), np.int64)
    idx = np.arange(n_sites)
    is_var = rng.random(n_sites) < 0.05
    alt = (dom + rng.integers(1, 4, n_sites)) % 4
    for i in range(n):
        dose = np.where(is_var, rng.binomial(2, 0.3, n_sites), 0)
        good = depth[:, i] - err[:, i]
        ar = rng.binomial(good, dose / 2.0)
        reads[idx, i, dom] += good - ar
        reads[idx, i, alt] += ar
        reads[idx, i, rng.integers(0, 4, n_sites)] += err[:, i]
    reads = np.minimum(reads, 65535).astype(np.uint16)
    dev = PecallDev(0)
    for sl in (slice(0, n_sites), slice(0, 1)):
        r = reads[sl] if sl.stop > 1 else np.minimum(reads[:500], 40)
        dm = dom[sl] if sl.stop > 1 else dom[:500]
        got = dev.call_sites(r, dm)
        exp = oracle_py.call_sites(r, dm)
        assert np.array_equal(got[0], exp[0])
        assert np.max(np.abs(got[1] - exp[1])) <= 1e-6
        for a, b in zip(got[2:], exp[2:]):
            assert np.array_equal(a, b)
    dev.close()


@pytest.mark.gpu
def test_gpu_site_caller_with_pedigree():
    """two trios, a second child and a child without a sampled father: calls, posteriors, types and DENOVO_ counts"""
    from pecaller_amd.pecall import PecallDev
    f = fx.load("pecall_ped")
    ped = f["ped"]
    off, lst = oracle_py.kid_lists(ped["dad"], ped["mom"], ped["order"])
    dev = PecallDev(0)
    dev.set_pedigree(ped["dad"], ped["mom"], ped["sex"], off, lst, ped["denovo_rate"])
    for chrom_code in (0, 1, 2, 3):                # autosome, X, Y, MT change which parent explains a child's allele
        chrom = np.full(len(f["dom"]), chrom_code, np.uint8)
        got = dev.call_sites(f["reads"], f["dom"], chrom=chrom)
        exp = oracle_py.call_sites(f["reads"], f["dom"], chrom=chrom, ped=ped)
        assert np.array_equal(got[0], exp[0]), chrom_code
        assert np.max(np.abs(got[1] - exp[1])) <= 1e-6
        for a, b in zip(got[2:], exp[2:]):
            assert np.array_equal(a, b)
        assert np.array_equal(dev.denovo, oracle_py.call_sites.denovo)
        if chrom_code == 0:
            assert (dev.denovo > 0).sum() >= 10
            bad = [pos1 for i, pos1 in enumerate(f["pos"].astype(int) + 1)
                   if got[2][i] > 0 and fx.snp_row("chr1", pos1, chr(f["ref"][i]), got[0][i], got[1][i], got[2][i], got[3][i], dev.denovo[i])
                   != f["snp_rows"].get(pos1)]
            assert len(bad) <= 1, bad[:3]          # the reference's own rows (a posterior may differ in the sixth digit)
    dev.set_pedigree(None, None, None, None, None, 0)
    got = dev.call_sites(f["reads"][:500], f["dom"][:500])
    exp = oracle_py.call_sites(f["reads"][:500], f["dom"][:500])
    assert np.array_equal(got[0], exp[0])
    dev.close()


def test_site_oracle_guide_mode_matches_reference_text():
    """BED guide mode: every position of the intervals, uncovered ones included; chrY / chrMT columns with HAPLOID forced"""
    f = fx.load_guide()
    call, p, typ, ac, _ = oracle_py.call_sites(f["reads"], f["dom"], chrom=f["chrom"])
    n = 0
    for i, (c, p1, refch) in enumerate(f["key"]):
        if f["dom"][i] > 3:
            assert (c, p1) not in f["base_rows"]
            continue
        n += 1
        assert fx.base_row(c, p1, refch, call[i], p[i]) == f["base_rows"][(c, p1)], (c, p1)
        if typ[i] > 0:
            assert fx.snp_row(c, p1, refch, call[i], p[i], typ[i], ac[i]) == f["snp_rows"][(c, p1)], (c, p1)
        else:
            assert (c, p1) not in f["snp_rows"]
    assert n == len(f["base_rows"]) and len(f["snp_rows"]) > 30


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 5, 13, 33])
def test_gpu_site_caller_odd_sample_counts(n):
    """sample counts that are not multiples of 8 (and below 4: the single-pass rule, pecaller.c:1470), against the oracle"""
    from pecaller_amd.pecall import PecallDev
    rng = np.random.default_rng(100 + n)
    n_sites = 1500
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    reads = np.zeros((n_sites, n, 6), np.uint16)
    for s in range(n_sites):
        var = rng.random() < 0.3
        q = rng.uniform(0.05, 0.6)
        alt = int(rng.integers(0, 6))
        r = int(dom[s])
        for i in range(n):
            d = int(rng.poisson(28))
            g = tuple(alt if (var and rng.random() < q) else r for _ in range(2))
            cnt = np.bincount(rng.integers(0, 2, d), minlength=2)
            for k in (0, 1):
                al = g[k]
                if al == 5:
                    reads[s, i, r] += cnt[k]
                reads[s, i, al] += cnt[k]
    dev = PecallDev(0)
    got = dev.call_sites(reads, dom)
    exp = oracle_py.call_sites(reads, dom)
    assert np.array_equal(got[0], exp[0])
    assert np.max(np.abs(got[1] - exp[1])) <= 1e-6
    for a, b in zip(got[2:], exp[2:]):
        assert np.array_equal(a, b)
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,ped", [(65, False), (128, False), (150, False), (200, True), (256, False), (257, False), (400, True), (512, False)])
def test_gpu_site_caller_more_than_64_samples(n, ped):
    """sample counts on both sides of the chunk boundaries (2 chunks up to 128, 4 up to 256, 8 up to 512), with and without a pedigree that
    spans the chunks, against the oracle; 513 is refused"""
    from pecaller_amd.pecall import PecallDev
    from pecaller_amd.pemap import PemapError
    rng = np.random.default_rng(300 + n)
    n_sites = 400 if n <= 128 else 160 if n <= 256 else 60          # (the CPU oracle's time grows faster than the sample count)
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    dom[::53] = 14
    depth = rng.integers(12, 40, n)
    depth[3] = 2
    depth[n - 2] = 5
    is_var = rng.random(n_sites) < 0.3
    q = rng.uniform(0.02, 0.5, n_sites)
    alt = rng.integers(0, 6, n_sites)
    reads = np.zeros((n_sites, n, 6), np.int64)
    idx = np.arange(n_sites)
    r = np.where(dom < 4, dom, 0)
    for i in range(n):
        d = rng.poisson(depth[i], n_sites)
        dose = np.where(is_var, rng.binomial(2, q), 0)
        e = rng.binomial(d, 0.004)
        ar = rng.binomial(d - e, dose / 2.0)
        reads[idx, i, r] += d - e - ar
        reads[idx, i, alt] += ar
        reads[idx, i, r] += np.where(alt == 5, ar, 0)      # an insertion is counted on top of the base it follows
        reads[idx, i, rng.integers(0, 4, n_sites)] += e
    reads = reads.astype(np.uint16)
    pd = None
    if ped:
        # trios whose members sit in different chunks of 64: child i, father i + 64, mother i + 128
        dad = np.full(n, -1, np.int32)
        mom = np.full(n, -1, np.int32)
        sex = (1 + (np.arange(n) % 2)).astype(np.int32)
        for c in range(0, 40, 3):
            dad[c], mom[c] = c + 64, c + 128
            sex[c + 64], sex[c + 128] = 1, 2
        pd = dict(dad=dad, mom=mom, sex=sex, denovo_rate=1e-5)
    dev = PecallDev(0)
    if pd:
        kids = [[k for k in range(n) if dad[k] == i or mom[k] == i] for i in range(n)]
        off = np.concatenate([[0], np.cumsum([len(k) for k in kids])]).astype(np.int32)
        dev.set_pedigree(dad, mom, sex, off, np.array([k for ks in kids for k in ks] + [0], np.int32), 1e-5)
    got = dev.call_sites(reads, dom)
    gden = dev.denovo.copy()
    exp = oracle_py.call_sites(reads, dom, ped=pd)
    assert np.array_equal(got[0], exp[0])
    assert np.max(np.abs(got[1] - exp[1])) <= 1e-6
    for a, b in zip(got[2:], exp[2:]):
        assert np.array_equal(a, b)
    if pd:
        assert np.array_equal(gden, oracle_py.call_sites.denovo)
    assert (got[2] > 0).sum() > n_sites // 8 and (got[4].max() >= 2 or n > 256)
    if n == 512:
        with pytest.raises(PemapError):
            dev.call_sites(np.zeros((4, 513, 6), np.uint16), dom[:4])
    dev.close()


@pytest.mark.gpu
def test_gpu_site_caller_sparse_posteriors():
    """pecall_dev_call_sites_sparse: the same calls, types and allele counts as the dense call, and its list holds exactly the columns
    in which the dense call has a posterior that is not 1, ascending, with the dense call's posteriors to the last bit -- on the
    reference's fixture, on columns with samples too deep for the table's head (the second pass appends), with more than 64 samples,
    through pinned and pageable buffers; a list that is too short fails and says how many rows are needed"""
    from pecaller_amd.pecall import PecallDev
    dev = PecallDev(0)

    def check(reads, dom, **kw):
        dense = dev.call_sites(reads, dom, **kw)
        post = dense[1].copy()
        call, (site, rows), typ, ac, npass = dev.call_sites_sparse(reads, dom, **kw)
        assert np.array_equal(call, dense[0]) and np.array_equal(typ, dense[2]) and np.array_equal(ac, dense[3]) and np.array_equal(npass, dense[4])
        want = np.nonzero((post != 1.0).any(axis=1))[0]
        assert np.array_equal(site, want.astype(np.uint32)), (len(site), len(want))
        assert np.array_equal(rows, post[want])
        return len(want)

    f = fx.load()
    n1 = check(f["reads"], f["dom"])
    assert n1 > 100
    # deep columns: the head form defers them, the form with the whole table calls them in a second pass
    rng = np.random.default_rng(12)
    n_sites, n = 2500, 20
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    depth = np.where(rng.random((n_sites, n)) < 0.03, rng.integers(1500, 9000, (n_sites, n)), rng.poisson(30, (n_sites, n)))
    reads = np.zeros((n_sites, n, 6), np.int64)
    idx = np.arange(n_sites)
    alt = (dom + rng.integers(1, 4, n_sites)) % 4
    is_var = rng.random(n_sites) < 0.05
    for i in range(n):
        dose = np.where(is_var, rng.binomial(2, 0.3, n_sites), 0)
        err = rng.binomial(depth[:, i], 0.004)
        ar = rng.binomial(depth[:, i] - err, dose / 2.0)
        reads[idx, i, dom] += depth[:, i] - err - ar
        reads[idx, i, alt] += ar
        reads[idx, i, rng.integers(0, 4, n_sites)] += err
    reads = np.minimum(reads, 65535).astype(np.uint16)
    assert check(reads, dom) > 0
    # a call right after it on shallow columns of the same shape: nothing of the deep call's posteriors is left in the list
    shallow = np.minimum(reads, 40)
    check(shallow, dom)
    # more than 64 samples (every column through the beam search's kernel)
    w = fx.load("pecall_wide")
    check(w["reads"][:600], w["dom"][:600])
    # a list that is too short
    with pytest.raises(Exception) as e:
        dev.call_sites_sparse(f["reads"], f["dom"], cap=8)
    assert "list holds 8" in str(e.value) and dev.sparse_needed == n1
    # and the object still works
    check(f["reads"][:500], f["dom"][:500])
    dev.close()
