"""Edge cases of the mapping path on the GPU against the oracle: ragged batches (every read length from the shortest to the
longest the reference can take, mixed in one batch), reads of N, low-complexity reads, bisulfite mode, length limits."""
import numpy as np
import pytest
import fixtures
import oracle_py
import refio

pytestmark = pytest.mark.gpu
COMP = np.zeros(256, np.uint8)
COMP[:] = ord("N")
for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
    COMP[a] = b


def ragged_reads(seed, n, lo, hi):
    """pairs from the fixture genome with independent random lengths per end, 1 % substitutions, a few indels, some junk"""
    rng = np.random.default_rng(seed)
    _, contigs = fixtures.genome()
    r1, r2 = [], []
    for k in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        la, lb = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
        fl = int(rng.integers(max(la, lb) + 20, max(la, lb) + 400))
        s = int(rng.integers(0, len(c) - fl - 1))
        a = c[s:s + la].copy()
        b = COMP[c[s + fl - lb:s + fl]][::-1].copy()
        for x in (a, b):
            m = rng.random(len(x)) < 0.01
            x[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(m.sum()))]
        if k % 50 == 0 and la > 40:
            a = np.concatenate([a[:20], a[23:]])                      # a 3-base deletion
        if k % 50 == 1 and lb > 40:
            b = np.concatenate([b[:25], np.frombuffer(b"AC", np.uint8), b[25:]])[:hi]
        if k % 97 == 0:
            a[:] = ord("N")                                          # filtered by the N rule
        if k % 97 == 1:
            a[:] = ord("A")                                          # every bucket over the too-many-spots limit
        if k % 97 == 2:
            a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, la)]   # junk
        if rng.random() < 0.5:
            a, b = b, a
        r1.append(a.tobytes())
        r2.append(b.tobytes())
    return refio.pack_reads(r1) + refio.pack_reads(r2)


# (the longest read of a batch picks the kernels' geometry: 278 -> 19 segments, 16 lanes x 19 columns; 60 -> 7, 8 x 13; 200 -> 13,
# 16 x 13; 160 -> 10, 16 x 10; the golden 2 x 245 set covers 16 segments and 16 x 16)
@pytest.mark.parametrize("lo,hi,bis", [(16, 278, False), (16, 60, False), (200, 278, False), (100, 200, False), (90, 160, True)])
def test_ragged_batches_match_oracle(lo, hi, bis):
    from pecaller_amd import PemapDev
    ix = fixtures.index() if not bis else None
    if bis:
        names, contigs = fixtures.genome()
        mers, ukmer, ustart, cs = refio.kmer_index(contigs, bisulfite=True)
        base = fixtures.index()
        ix = dict(base, mers=mers, ukmer=ukmer, ustart=ustart, contig_starts=cs)
    b1, l1, b2, l2 = ragged_reads(1000 + lo + hi, 3000, lo, hi)
    dev = PemapDev(0)
    dev.build_index(ix["genome"], ix["contig_len"], bisulfite=bis)
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85, bisulfite=bis)
    m1, m2, mt = dev.map_batch(b1, l1, b2, l2)
    counts, ins = dev.fetch_pileup()
    o = oracle_py.Oracle(ix, paired=True, min_dist=0, max_dist=500, min_align=0.85, bisulfite=bis)
    om1, om2, omt, _, _ = o.map_batch(b1, l1, b2, l2, threads=8)
    assert np.array_equal(m1, om1), np.nonzero(m1 != om1)[0][:10]
    assert np.array_equal(m2, om2), np.nonzero(m2 != om2)[0][:10]
    assert np.array_equal(mt, omt)
    assert np.array_equal(counts, o.counts())
    assert sorted(ins) == sorted(o.insertions())
    assert np.array_equal(np.array(dev.summary()), np.array(o.summary()))
    assert (m1 > 0).sum() > 1500
    dev.close()


def test_length_limits_are_errors():
    from pecaller_amd import PemapDev, PemapError
    ix = fixtures.index()
    dev = PemapDev(0)
    dev.build_index(ix["genome"], ix["contig_len"])
    dev.set_params(paired=False, min_dist=0, max_dist=500, min_align=0.85)
    with pytest.raises(PemapError):                 # an empty batch is an error, not a silent no-op
        dev.map_batch(np.zeros((0, 304), np.uint8), np.zeros(0, np.int32), None, None)
    for ln in (15, 279):
        buf = np.full((4, 304), ord("A"), np.uint8)
        lens = np.array([100, 100, ln, 100], np.int32)
        with pytest.raises(PemapError):
            dev.map_batch(buf, lens, None, None)
    dev.close()


def test_caller_argument_errors():
    from pecaller_amd.pecall import PecallDev
    from pecaller_amd import PemapError
    dev = PecallDev(0)
    with pytest.raises(PemapError):
        dev.call_sites(np.zeros((0, 8, 6), np.uint16), np.zeros(0, np.uint8))
    with pytest.raises(PemapError):
        dev.call_sites(np.zeros((4, 513, 6), np.uint16), np.zeros(4, np.uint8))       # more than 512 samples
    with pytest.raises(PemapError):
        dev.call_sites(np.zeros((4, 8, 6), np.uint16), np.zeros(4, np.uint8), theta=0.9)   # pecaller.c:305-309
    dev.close()


def _adversarial_genome(seed):
    """10 contigs of unique sequence with tandem repeats of periods 1..12 and near-duplicated blocks planted in them: the places
    where several diagonals of an SW window match equally well, or a deletion competes with mismatches"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    contigs = []
    for c in range(10):
        g = acgt[rng.integers(0, 4, 60000)].copy()
        pos = 2000
        for period in range(1, 13):
            unit = acgt[rng.integers(0, 4, period)]
            n = int(rng.integers(60, 400))
            rep = np.tile(unit, n // period + 1)[:n].copy()
            m = rng.random(n) < 0.02              # slightly impure repeats
            rep[m] = acgt[rng.integers(0, 4, int(m.sum()))]
            g[pos:pos + n] = rep
            pos += n + int(rng.integers(600, 3000))
        # a block copied 30..40 bases downstream of itself with a few changes (short-range near-duplicate)
        for _ in range(6):
            s = int(rng.integers(pos, len(g) - 2000))
            blk = g[s:s + 120].copy()
            d = int(rng.integers(1, 22))
            g[s + 120 + d:s + 240 + d] = blk
        contigs.append(g)
    return contigs


def _adversarial_reads(contigs, seed, n, L):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    r1, r2 = [], []
    for k in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        fl = int(rng.integers(L + 20, L + 380))
        # half of the fragments start inside or next to a planted repeat (the first 40 % of a contig holds them)
        s = int(rng.integers(1500, int(len(c) * 0.4))) if k % 2 == 0 else int(rng.integers(0, len(c) - fl - 30))
        ends = []
        for which in (0, 1):
            lo = s if which == 0 else s + fl - L
            kind = int(rng.integers(0, 8))
            if kind == 5:      # the read skips b reference bases at p: the one-deletion alignment of the rule's second case
                b = int(rng.integers(1, 22))
                p = int(rng.integers(20, L - 20))
                x = np.concatenate([c[lo:lo + p], c[lo + p + b:lo + L + b]]).copy()
            elif kind == 6:    # an inserted stretch
                a = int(rng.integers(1, 6))
                p = int(rng.integers(20, L - 20))
                x = np.concatenate([c[lo:lo + p], acgt[rng.integers(0, 4, a)], c[lo + p:lo + L - a]]).copy()
            else:
                x = c[lo:lo + L].copy()
            nsub = (0, 1, 2, 2, 3, 1, 0, 4)[kind]      # exact numbers of substitutions: the rule's case boundaries
            for q in rng.choice(L, size=nsub, replace=False):
                x[q] = acgt[(int(np.nonzero(acgt == x[q])[0][0]) + int(rng.integers(1, 4))) % 4]
            if kind == 7 and rng.random() < 0.3:
                x[int(rng.integers(0, L))] = ord("N")
            if which == 1:
                x = COMP[x][::-1].copy()
            ends.append(x)
        a, b = ends
        if rng.random() < 0.5:
            a, b = b, a
        r1.append(a.tobytes())
        r2.append(b.tobytes())
    return refio.pack_reads(r1) + refio.pack_reads(r2)


@pytest.mark.parametrize("L", [150, 100])
def test_gapless_rule_on_repeats_and_competing_gaps(L):
    """The gapless rule (pm_gapless_kernel) against the oracle's full DP where it is most exposed: tandem repeats (several
    diagonals perfect or tied), reads with exactly 0 / 1 / 2 / 3 substitutions, reads that skip 1..21 reference bases (the
    one-deletion alignment that beats two mismatches), insertions.  Coordinates, classes, pileup, insertions, and per hit
    the fp64 score bits and the start cell."""
    from pecaller_amd import PemapDev
    contigs = _adversarial_genome(77)
    mers, ukmer, ustart, cs = refio.kmer_index(contigs)
    ix = dict(mers=mers, ukmer=ukmer, ustart=ustart, genome=np.concatenate(contigs), contig_starts=cs,
              contig_len=np.array([len(c) for c in contigs], dtype=np.uint32))
    n = 6000
    b1, l1, b2, l2 = _adversarial_reads(contigs, 5 + L, n, L)
    dev = PemapDev(0)
    dev.build_index(ix["genome"], ix["contig_len"])
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
    m1, m2, mt = dev.map_batch(b1, l1, b2, l2)
    stats, _ = dev.run_stats()
    dbg = dev.debug_hits(2 * n)
    counts, ins = dev.fetch_pileup()
    o = oracle_py.Oracle(ix, paired=True, min_dist=0, max_dist=500, min_align=0.85)
    om1, om2, omt, d1, d2 = o.map_batch(b1, l1, b2, l2, debug=True, threads=8)
    assert np.array_equal(m1, om1), np.nonzero(m1 != om1)[0][:10]
    assert np.array_equal(m2, om2), np.nonzero(m2 != om2)[0][:10]
    assert np.array_equal(mt, omt)
    assert np.array_equal(counts, o.counts())
    assert sorted(ins) == sorted(o.insertions())
    multi = 0
    for which, od in ((0, d1), (1, d2)):
        nh = dbg["n_hits"][which::2]
        assert np.array_equal(nh, od["n_hits"])
        for i in np.nonzero(nh)[0]:
            k = nh[i]
            e = 2 * i + which
            assert np.array_equal(dbg["score"][e, :k].view(np.uint64), od["score"][i, :k].view(np.uint64)), (e, k)
            assert np.array_equal(dbg["start_k"][e, :k], od["start"][i, :k, 0]), e
            assert np.array_equal(dbg["start_i"][e, :k], od["start"][i, :k, 1]), e
            multi += k > 1
    # the rule decided a good share and left a good share to the DP; multi-hit ends (repeats) occurred
    assert stats["gapless"] > n // 2 and stats["sw_dirs"] > n // 4 and multi > 50, (stats, multi)
    dev.close()


def _band_reads(contigs, seed, n, L):
    """reads for the banded DP's corner: the best ungapped placement has 0..8 mismatches (the kernel takes <= 6) while the best
    alignment may hold a gap -- an indel a few bases from either end of the read (its tail then counts as a handful of
    mismatches), of any length up to the window's slop, with substitutions, Ns and tandem repeats around it"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    r1, r2 = [], []
    for k in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        fl = int(rng.integers(L + 20, L + 380))
        s = int(rng.integers(1500, int(len(c) * 0.4))) if k % 3 == 0 else int(rng.integers(0, len(c) - fl - 60))
        ends = []
        for which in (0, 1):
            lo = s if which == 0 else s + fl - L
            kind = int(rng.integers(0, 6))
            tail = int(rng.integers(1, 13))                    # bases between the indel and the nearer end
            p = tail if rng.random() < 0.5 else L - tail
            if kind in (0, 1):      # the read skips b reference bases near an end
                b = int(rng.integers(1, 22))
                x = np.concatenate([c[lo:lo + p], c[lo + p + b:lo + L + b]]).copy()
            elif kind in (2, 3):    # inserted bases near an end (up to beyond the band's K = 5)
                a = int(rng.integers(1, 9))
                x = np.concatenate([c[lo:lo + p], acgt[rng.integers(0, 4, a)], c[lo + p:lo + L - a]]).copy()
            else:
                x = c[lo:lo + L].copy()
            nsub = int(rng.integers(0, 9)) if kind >= 4 else int(rng.integers(0, 4))    # 6 | 7 is the kernel's boundary
            for q in rng.choice(L, size=nsub, replace=False):
                other = acgt[acgt != x[q]]            # (x[q] may be one of the planted reference Ns)
                x[q] = other[int(rng.integers(0, len(other)))]
            if rng.random() < 0.1:
                x[int(rng.integers(0, L))] = ord("N")
            if which == 1:
                x = COMP[x][::-1].copy()
            ends.append(x)
        a, b = ends
        if rng.random() < 0.5:
            a, b = b, a
        r1.append(a.tobytes())
        r2.append(b.tobytes())
    return refio.pack_reads(r1) + refio.pack_reads(r2)


@pytest.mark.parametrize("L", [150, 64, 250])
def test_banded_dp_on_indels_near_read_ends(L):
    """pm_band_kernel against the oracle's full DP where a band is most exposed: its problems are those with few mismatches on
    the best diagonal, which includes reads whose best alignment has a gap a few bases from an end (the ungapped tail costs
    less than 6 mismatches) -- deletions up to the window's slop, insertions beyond the band's half-width, 0..8 substitutions
    (6 | 7 is where problems stop going to the band), reference Ns in the window (contigs are N-free here; the read's N is
    the letter with its own rule).  Per hit the fp64 score bits and the traceback's start cell; coordinates, classes, pileup
    and insertions."""
    from pecaller_amd import PemapDev
    contigs = _adversarial_genome(78)
    for c in contigs[:3]:          # a few reference Ns: a window with one takes the rewritten-letters form of a column
        c[np.random.default_rng(len(c)).integers(0, len(c), 300)] = ord("N")
    mers, ukmer, ustart, cs = refio.kmer_index(contigs)
    ix = dict(mers=mers, ukmer=ukmer, ustart=ustart, genome=np.concatenate(contigs), contig_starts=cs,
              contig_len=np.array([len(c) for c in contigs], dtype=np.uint32))
    n = 5000
    b1, l1, b2, l2 = _band_reads(contigs, 9 + L, n, L)
    dev = PemapDev(0)
    dev.build_index(ix["genome"], ix["contig_len"])
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
    m1, m2, mt = dev.map_batch(b1, l1, b2, l2)
    stats, _ = dev.run_stats()
    dbg = dev.debug_hits(2 * n)
    counts, ins = dev.fetch_pileup()
    o = oracle_py.Oracle(ix, paired=True, min_dist=0, max_dist=500, min_align=0.85)
    om1, om2, omt, d1, d2 = o.map_batch(b1, l1, b2, l2, debug=True, threads=8)
    assert np.array_equal(m1, om1), np.nonzero(m1 != om1)[0][:10]
    assert np.array_equal(m2, om2), np.nonzero(m2 != om2)[0][:10]
    assert np.array_equal(mt, omt)
    assert np.array_equal(counts, o.counts())
    assert sorted(ins) == sorted(o.insertions())
    for which, od in ((0, d1), (1, d2)):
        nh = dbg["n_hits"][which::2]
        assert np.array_equal(nh, od["n_hits"])
        for i in np.nonzero(nh)[0]:
            k = nh[i]
            e = 2 * i + which
            assert np.array_equal(dbg["score"][e, :k].view(np.uint64), od["score"][i, :k].view(np.uint64)), (e, k)
            assert np.array_equal(dbg["start_k"][e, :k], od["start"][i, :k, 0]), e
            assert np.array_equal(dbg["start_i"][e, :k], od["start"][i, :k, 1]), e
    # the band took a good share, the full DP still had work (more than 6 mismatches), and gaps were found
    # ("banded" counts both passes of a multi-hit end; cells_dirs are the full DP's cells: > n // 50 problems of L x (L + 21))
    assert stats["banded"] > n // 4 and stats["cells_dirs"] > (n // 50) * L * (L + 21) and len(ins) > n // 50, stats
    dev.close()


def _repeat_family_genome(seed):
    """10 contigs of unique sequence with one repeat family planted in them: 95 exact copies of a 260-base tile and 95 copies of a
    variant that differs from it in every 16th base.  Every 16-mer of a read from the family then has a bucket of 95 positions
    (below too_many_spots, pemapper.c:163) plus a single-substitution neighbour bucket of 95: about 1,900 positions on the
    read's strand, over the 1,024 the look-up / vote kernels keep in LDS, so the end goes to the big-end list and through
    pm_seed_kernel's global spill area."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    contigs = [acgt[rng.integers(0, 4, 80000)].copy() for _ in range(10)]
    tile = acgt[rng.integers(0, 4, 260)].copy()
    var = tile.copy()
    for q in range(5, 260, 16):
        var[q] = acgt[(int(np.nonzero(acgt == var[q])[0][0]) + 1 + q % 3) % 4]
    spots = []
    for k in range(190):
        c = k % 10
        s = 1500 + (k // 10) * 3900 + int(rng.integers(0, 600))
        contigs[c][s:s + 260] = tile if k < 95 else var
        spots.append((c, s))
    return contigs, spots


def _family_reads(contigs, spots, seed, n, L):
    """pairs with one end inside a copy of the family and the mate in the unique flank (so that the pair resolves), and pairs
    with both ends inside copies; 1 % substitutions"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    r1, r2 = [], []
    for k in range(n):
        c, s = spots[int(rng.integers(0, len(spots)))]
        g = contigs[c]
        lo = s + int(rng.integers(0, 260 - L + 1)) if k % 4 else s - int(rng.integers(1, 40))
        fl = int(rng.integers(L + 150, 480))
        if k % 3 == 0:
            lo = lo - fl + L                  # the family end is the fragment's far end
        lo = max(0, min(lo, len(g) - fl - 1))
        a = g[lo:lo + L].copy()
        b = COMP[g[lo + fl - L:lo + fl]][::-1].copy()
        for x in (a, b):
            m = rng.random(len(x)) < 0.01
            x[m] = acgt[rng.integers(0, 4, int(m.sum()))]
        if rng.random() < 0.5:
            a, b = b, a
        r1.append(a.tobytes())
        r2.append(b.tobytes())
    return refio.pack_reads(r1) + refio.pack_reads(r2)


@pytest.mark.parametrize("env", [{}, {"PEMAP_SEED_BLOCKS_PER_CU": "1"}, {"PEMAP_SEED_BLOCKS_PER_CU": "1", "PEMAP_BIG_BLOCKS_PER_CU": "8"},
                                 {"PEMAP_PIPELINE": "0", "PEMAP_SEED_BLOCKS_PER_CU": "2"}])
def test_big_ends_take_the_spill_path(env, monkeypatch):
    """Read-ends with more than 1,024 positions on a strand leave the wave-per-end kernels and are seeded by pm_seed_kernel in
    list mode with its per-block spill area in HBM.  Asserted: such ends occur (stats["big_ends"]), and hits, scores,
    coordinates, classes, pileup and insertions equal the oracle's.  The grid knobs cover the geometry that faulted in round 1
    (list-mode grid larger than the grid the spill scratch was sized for: gpurun_out/ab_rep15.log) -- the scratch is now sized
    for the larger grid and the kernel is told its capacity -- and the monolithic one-stream form of the seed stage."""
    from pecaller_amd import PemapDev
    for k, v in env.items():
        monkeypatch.setenv(k, v)           # the knobs are read by pemap_dev_create, once per object
    contigs, spots = _repeat_family_genome(4242)
    mers, ukmer, ustart, cs = refio.kmer_index(contigs)
    ix = dict(mers=mers, ukmer=ukmer, ustart=ustart, genome=np.concatenate(contigs), contig_starts=cs,
              contig_len=np.array([len(c) for c in contigs], dtype=np.uint32))
    n, L = 700, 150
    b1, l1, b2, l2 = _family_reads(contigs, spots, 99, n, L)
    dev = PemapDev(0)
    dev.build_index(ix["genome"], ix["contig_len"])
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
    m1, m2, mt = dev.map_batch(b1, l1, b2, l2)
    stats, _ = dev.run_stats()
    dbg = dev.debug_hits(2 * n)
    counts, ins = dev.fetch_pileup()
    dev.close()
    o = oracle_py.Oracle(ix, paired=True, min_dist=0, max_dist=500, min_align=0.85)
    om1, om2, omt, d1, d2 = o.map_batch(b1, l1, b2, l2, debug=True, threads=8)
    if env.get("PEMAP_PIPELINE") != "0":
        assert stats["big_ends"] > n // 4, stats        # (the monolithic form has no big-end list: every end is its own)
    assert np.array_equal(m1, om1), np.nonzero(m1 != om1)[0][:10]
    assert np.array_equal(m2, om2), np.nonzero(m2 != om2)[0][:10]
    assert np.array_equal(mt, omt)
    assert np.array_equal(counts, o.counts())
    assert sorted(ins) == sorted(o.insertions())
    many = 0
    for which, od in ((0, d1), (1, d2)):
        nh = dbg["n_hits"][which::2]
        assert np.array_equal(nh, od["n_hits"])
        for i in np.nonzero(nh)[0]:
            k = nh[i]
            e = 2 * i + which
            assert np.array_equal(dbg["spot"][e, :k], od["spot"][i, :k]), e
            assert np.array_equal(dbg["orient"][e, :k], od["orient"][i, :k]), e
            assert np.array_equal(dbg["score"][e, :k].view(np.uint64), od["score"][i, :k].view(np.uint64)), (e, k)
            many += k >= 90
    assert many > n // 4, many              # the family's ends really had ~95 or ~190 candidate hits each
    assert (m1 > 0).sum() > n // 2


def _edge_genome(seed):
    """12 short contigs (so that a good share of the reads lies at a contig end and gets a clipped SW window), with N runs,
    IUPAC letters and single N's sprinkled inside the sequence (.seq holds upper-case letters: lower case cannot occur)"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    odd = np.frombuffer(b"NRYKMSW", np.uint8)
    contigs = []
    for c in range(12):
        n = int(rng.integers(2500, 6000))
        g = acgt[rng.integers(0, 4, n)].copy()
        for _ in range(n // 300):                       # an odd letter every ~300 bases
            g[int(rng.integers(0, n))] = odd[int(rng.integers(0, len(odd)))]
        s = int(rng.integers(300, n - 400))
        g[s:s + int(rng.integers(3, 30))] = ord("N")    # an N run
        contigs.append(g)
    return contigs


def _edge_reads(contigs, seed, n, L, bis):
    """half of the fragments touch a contig end: the read starts within 0..12 bases of the first base or ends within 0..12 of
    the last one, so that the window (spot - 10 .. spot + L + 10, clipped to the contig: pemapper.c:1047-1081) has 1..21
    diagonals; exact numbers of substitutions around the rule's case boundaries; C -> T conversion in bisulfite mode"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    r1, r2 = [], []
    for k in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        fl = int(rng.integers(L + 10, min(len(c), L + 380)))
        if k % 2 == 0:
            s = int(rng.integers(0, 13)) if k % 4 == 0 else len(c) - fl - int(rng.integers(0, 13))
        else:
            s = int(rng.integers(0, len(c) - fl))
        s = max(0, min(s, len(c) - fl))
        ends = []
        for which in (0, 1):
            lo = s if which == 0 else s + fl - L
            x = c[lo:lo + L].copy()
            for q in rng.choice(L, size=(0, 1, 2, 2, 3, 1, 0, 4)[int(rng.integers(0, 8))], replace=False):
                x[q] = acgt[int(rng.integers(0, 4))]
            if which == 1:
                x = COMP[x][::-1].copy()
            if bis:
                conv = (x == ord("C")) & (rng.random(L) < 0.95)
                x[conv] = ord("T")
            ends.append(x)
        a, b = ends
        if rng.random() < 0.5:
            a, b = b, a
        r1.append(a.tobytes())
        r2.append(b.tobytes())
    return refio.pack_reads(r1) + refio.pack_reads(r2)


@pytest.mark.parametrize("bis", [False, True])
def test_gapless_rule_at_contig_ends_and_on_odd_letters(bis):
    """The three places where the gapless rule's inputs leave the common case -- windows clipped at a contig end (fewer than
    22 diagonals), reference N / IUPAC letters inside the window (pm_match's asymmetric N row), bisulfite mode -- on the
    device against the oracle's full DP: coordinates, classes, pileup, insertions, and per hit the score BITS, the start
    cell and the window geometry."""
    from pecaller_amd import PemapDev
    contigs = _edge_genome(31)
    mers, ukmer, ustart, cs = refio.kmer_index(contigs, bisulfite=bis)
    ix = dict(mers=mers, ukmer=ukmer, ustart=ustart, genome=np.concatenate(contigs), contig_starts=cs,
              contig_len=np.array([len(c) for c in contigs], dtype=np.uint32))
    n, L = 5000, 120
    b1, l1, b2, l2 = _edge_reads(contigs, 9 + int(bis), n, L, bis)
    dev = PemapDev(0)
    dev.build_index(ix["genome"], ix["contig_len"], bisulfite=bis)
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85, bisulfite=bis)
    m1, m2, mt = dev.map_batch(b1, l1, b2, l2)
    stats, _ = dev.run_stats()
    dbg = dev.debug_hits(2 * n)
    counts, ins = dev.fetch_pileup()
    dev.close()
    o = oracle_py.Oracle(ix, paired=True, min_dist=0, max_dist=500, min_align=0.85, bisulfite=bis)
    om1, om2, omt, d1, d2 = o.map_batch(b1, l1, b2, l2, debug=True, threads=8)
    assert np.array_equal(m1, om1), np.nonzero(m1 != om1)[0][:10]
    assert np.array_equal(m2, om2), np.nonzero(m2 != om2)[0][:10]
    assert np.array_equal(mt, omt)
    assert np.array_equal(counts, o.counts())
    assert sorted(ins) == sorted(o.insertions())
    clipped = 0
    for which, od in ((0, d1), (1, d2)):
        nh = dbg["n_hits"][which::2]
        assert np.array_equal(nh, od["n_hits"])
        for i in np.nonzero(nh)[0]:
            k = nh[i]
            e = 2 * i + which
            assert np.array_equal(dbg["win_start"][e, :k].astype(np.int32), od["win_start"][i, :k])
            assert np.array_equal(dbg["win_len"][e, :k], od["win_len"][i, :k])
            assert np.array_equal(dbg["score"][e, :k].view(np.uint64), od["score"][i, :k].view(np.uint64)), (e, k)
            assert np.array_equal(dbg["start_k"][e, :k], od["start"][i, :k, 0]), e
            assert np.array_equal(dbg["start_i"][e, :k], od["start"][i, :k, 1]), e
            clipped += int((od["win_len"][i, :k] < L + 21).sum())
    # clipped windows occurred in numbers, the rule decided a good share, and mapping still works
    # (in bisulfite mode only the end that reads the converted strand forward maps: every C of both ends was turned into T)
    assert clipped > n // 6 and stats["gapless"] > n // 3 and (m1 > 0).sum() > 2 * n // 5, (clipped, stats, int((m1 > 0).sum()))
