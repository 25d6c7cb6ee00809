"""The oracle (oracle/pemap_oracle.c) against the outputs of the compiled reference (tests/golden/).  CPU only."""
import numpy as np
import pytest
import fixtures
import oracle_py
import refio


def test_numpy_index_matches_reference_files():
    ix = fixtures.index()
    m = fixtures.meta()["index"]
    assert len(ix["mers"]) == m["n_mers"]
    assert refio.md5(ix["mers"]) == m["mdx_md5"]                 # .mdx bytes
    assert refio.md5(ix["ukmer"]) == m["idx_ukmer_md5"]          # run boundaries of the inflated .idx
    assert refio.md5(ix["ustart"]) == m["idx_ustart_md5"]
    lens, names, idepth = refio.read_sdx(fixtures.GOLD + "/g1.sdx")
    assert idepth == 16 and names == ix["names"]
    assert np.array_equal(np.cumsum([0] + lens), ix["contig_starts"])
    assert len(ix["genome"]) == m["seq_len"]


@pytest.mark.parametrize("name", ["r150", "r100", "r250"])
def test_oracle_matches_reference(name):
    ix = fixtures.index()
    s = fixtures.SETS[name]
    r1, l1, r2, l2 = fixtures.reads(name)
    o = oracle_py.Oracle(ix, paired=s["paired"], min_dist=0, max_dist=500, min_align=0.85)
    m1, m2, mt, _, _ = o.map_batch(r1, l1, r2, l2, threads=8)
    assert np.array_equal(m1, fixtures.golden_m(name, 1))
    if s["paired"]:
        assert np.array_equal(m2, fixtures.golden_m(name, 2))
    fixtures.check_pileup_against_golden(name, o.counts())
    tot, head, rows = fixtures.golden_summary(name)
    sm = o.summary()
    assert sm[0] == tot
    cls = dict(zip(["Unique Mate-Paired", "Unique Mate-Paired with slip", "Unique Single End", "Unique Mis-size",
                    "Non-Unique Mate-Paired", "Non-Unique Mis-size", "Fragment Mismatch", "Non-unique with no map",
                    "Neither Map"], sm[4:])) if s["paired"] else {
        "Unique Mapping": sm[6], "Non-Unique Mapping, discarded": sm[11], "No mapping reaches threshold": sm[12]}
    for k, v in cls.items():
        assert rows[k] == v, k
    assert rows["All"] == len(l1)
    # "%g" of the averages (pemapper.c:888-890)
    assert head[3] == "%g" % (sm[1] / sm[0])
    assert head[7] == "%g" % ((sm[2] / sm[3]) if sm[3] else 0.0)
    names, contigs = fixtures.genome()
    gold_ins, _ = fixtures.golden_insertions(name)
    assert fixtures.ins_to_named(o.insertions(), names, contigs) == gold_ins


def test_oracle_threads_agree():
    ix = fixtures.index()
    r1, l1, r2, l2 = fixtures.reads("r150")
    n = 3000
    a = oracle_py.Oracle(ix, paired=True)
    b = oracle_py.Oracle(ix, paired=True)
    ra = a.map_batch(r1[:n], l1[:n], r2[:n], l2[:n], threads=1)
    rb = b.map_batch(r1[:n], l1[:n], r2[:n], l2[:n], threads=5)
    for x, y in zip(ra[:3], rb[:3]):
        assert np.array_equal(x, y)
    assert np.array_equal(a.counts(), b.counts())
    assert a.insertions() == b.insertions()
