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


@pytest.mark.parametrize("lo,hi,bis", [(16, 278, False), (16, 60, False), (200, 278, False), (90, 160, True)])
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
        dev.call_sites(np.zeros((4, 65, 6), np.uint16), np.zeros(4, np.uint8))        # more than 64 samples
    with pytest.raises(PemapError):
        dev.call_sites(np.zeros((4, 8, 6), np.uint16), np.zeros(4, np.uint8), theta=0.9)   # pecaller.c:305-309
    dev.close()
