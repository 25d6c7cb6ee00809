"""Shared loaders for the committed golden fixtures (tests/golden/, made by make_golden.py from the compiled reference)."""
import functools
import json
import os
import numpy as np
import refio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SETS = {
    "r150": dict(prefix="g1", paired=True, trim=(0, 0)),
    "r100": dict(prefix="g1s", paired=False, trim=(0, 0)),
    "r250": dict(prefix="g1l", paired=True, trim=(3, 2)),      # pemapper_tsw trim_from_start=3 trim_from_end=2
}


@functools.lru_cache(maxsize=None)
def meta():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


@functools.lru_cache(maxsize=None)
def genome():
    names, contigs = refio.read_fasta(os.path.join(GOLD, "g1.fa.gz"))
    return names, contigs


@functools.lru_cache(maxsize=None)
def index():
    names, contigs = genome()
    mers, ukmer, ustart, cs = refio.kmer_index(contigs)
    g = np.concatenate(contigs)
    return dict(mers=mers, ukmer=ukmer, ustart=ustart, genome=g, contig_starts=cs,
                contig_len=np.array([len(c) for c in contigs], dtype=np.uint32), names=names)


@functools.lru_cache(maxsize=None)
def reads(name):
    s = SETS[name]
    out = []
    for k in (1, 2):
        if k == 2 and not s["paired"]:
            out += [None, None]
            continue
        r = refio.read_fastq(os.path.join(GOLD, "%s_%d_.fastq.gz" % (s["prefix"], k)))
        a, b = s["trim"]
        if (a, b) != (0, 0):
            r = [x[a:len(x) - b] for x in r]     # pemapper_tsw.c:693-704
        buf, lens = refio.pack_reads(r)
        out += [buf, lens]
    return tuple(out)


def golden_m(name, k):
    return np.fromfile(os.path.join(GOLD, "%s.m%d" % (name, k)), dtype="<u4")


def golden_summary(name):
    """-> (total_reads, class counts dict) parsed from the reference's summary.txt"""
    txt = open(os.path.join(GOLD, name + ".summary.txt")).read().split("\n")
    head = [l for l in txt if l.startswith("Total Number")][0].split("\t")
    rows = {}
    for l in txt:
        p = l.split("\t")
        if len(p) == 3 and p[0] not in ("Mapping Type",):
            rows[p[0]] = int(p[1])
    return int(head[1]), head, rows


def golden_insertions(name):
    rows = refio.read_indel(os.path.join(GOLD, name + ".indel.txt.gz"))
    return sorted((r[0], r[1], s.encode()) for r in rows for s in r[7]), rows


def ins_to_named(ins, names, contigs):
    real = np.concatenate([[0], np.cumsum([len(c) for c in contigs])])
    out = []
    for p, s in ins:
        c = int(np.searchsorted(real, p, side="right") - 1)
        out.append((names[c], int(p - real[c] + 1), s))
    return sorted(out)


def plane0_column(positions):
    """column of the reference's six that the device keeps in plane 0 at these positions (PmPile, pemap_kernels.hip.h): the
    reference base's own column where the letter is A / C / G / T, column A elsewhere"""
    g = index()["genome"][positions]
    return np.select([g == ord("A"), g == ord("C"), g == ord("G"), g == ord("T")], [0, 1, 2, 3], 0)


def check_pileup_against_golden(name, counts):
    """counts: [gsize][6] u16.  Compares with the md5 / record count / column sums / sampled records of the reference run."""
    m = meta()[name]
    nz = np.nonzero(counts.astype(np.int64).sum(axis=1))[0]
    rec = np.zeros(len(nz), refio.PILE_DT)
    rec["pos"] = nz
    rec["c"] = counts[nz]
    assert len(rec) == m["pileup_records"]
    assert [int(x) for x in rec["c"].sum(axis=0)] == m["pileup_colsum"]
    samp = np.load(os.path.join(GOLD, name + ".pileup_sample.npy"))
    assert np.array_equal(rec[::101], samp)
    assert refio.md5(rec) == m["pileup_md5"]
    return rec
