"""BASELINE config 0 at its own scale: a chr21-sized single-contig genome and 100,000 single-end 100-base reads, regenerated
from seeds (tools/synth.py), against what the unmodified reference pemapper produced for them (tests/golden/chr21.json, made
by tests/golden/make_golden_chr21.py)."""
import functools
import hashlib
import json
import os
import sys
import numpy as np
import pytest
import refio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "chr21.json")))


@functools.lru_cache(maxsize=None)
def data():
    import synth
    p = GOLD["params"]
    contigs = synth.make_genome(p["genome_seed"], 1, p["contig_len"], features=True)
    r1, _, _ = synth.make_reads(p["reads_seed"], contigs, p["n_reads"], p["read_len"])
    g = contigs[0]
    # the generator must reproduce the inputs the reference saw (same numpy, same code)
    assert hashlib.md5(g.tobytes()).hexdigest() == GOLD["genome_md5"], "synthetic genome differs from the one the golden was made with"
    assert hashlib.md5(b"\n".join(r1)).hexdigest() == GOLD["reads_md5"], "synthetic reads differ from the ones the golden was made with"
    buf, lens = refio.pack_reads(r1)
    return g, buf, lens


def check(m1, counts, ins, summary):
    assert [int(x) for x in m1[:64]] == GOLD["m1_head"]
    assert int((m1 > 0).sum()) == GOLD["mapped"]
    assert refio.md5(m1.astype("<u4")) == GOLD["m1_md5"]
    nz = np.nonzero(counts.astype(np.int64).sum(axis=1))[0]
    rec = np.zeros(len(nz), refio.PILE_DT)
    rec["pos"] = nz
    rec["c"] = counts[nz]
    assert len(rec) == GOLD["pileup_records"]
    assert refio.md5(rec) == GOLD["pileup_md5"]
    named = sorted(("chr21", int(p) + 1, s.decode() if isinstance(s, bytes) else s) for p, s in ins)
    assert hashlib.md5(repr(named).encode()).hexdigest() == GOLD["ins_md5"]
    head = [l for l in GOLD["summary"].split("\n") if l.startswith("Total Number")][0].split("\t")
    assert int(head[1]) == summary[0]
    assert head[3] == "%g" % (summary[1] / summary[0])


def test_oracle_chr21():
    import oracle_py
    g, buf, lens = data()
    mers, ukmer, ustart, cs = refio.kmer_index([g])
    assert len(mers) == GOLD["n_mers"]
    ix = dict(mers=mers, ukmer=ukmer, ustart=ustart, genome=g, contig_starts=cs, contig_len=np.array([len(g)], np.uint32), names=["chr21"])
    o = oracle_py.Oracle(ix, paired=False, min_dist=0, max_dist=500, min_align=0.85)
    m1, _, _, _, _ = o.map_batch(buf, lens, None, None, threads=8)
    check(m1, o.counts(), o.insertions(), o.summary())


@pytest.mark.gpu
def test_gpu_chr21():
    from pecaller_amd import PemapDev
    g, buf, lens = data()
    dev = PemapDev(0)
    dev.build_index(g, np.array([len(g)], np.uint32))
    dev.set_params(paired=False, min_dist=0, max_dist=500, min_align=0.85)
    m1, _, _ = dev.map_batch(buf, lens, None, None)
    counts, ins = dev.fetch_pileup()
    check(m1, counts, ins, dev.summary())
    dev.close()
