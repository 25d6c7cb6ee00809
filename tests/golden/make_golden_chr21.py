#!/usr/bin/env python3
"""BASELINE config 0 at its own scale (development container only; needs oracle/_ref): a chr21-sized single-contig genome,
100,000 single-end 100-base reads, mapped by the UNMODIFIED reference pemapper.

Nothing large is stored: genome and reads are regenerated from seeds by tools/synth.py (their md5 is recorded as a guard),
the index files the reference needs are written here from tests/refio.kmer_index (itself proved equal to the reference
builder's files on the small fixture; the reference builder cannot index 46 Mbp in this container), and
tests/golden/chr21.json keeps the md5 of the reference's .mfile words, pileup records and insertion rows, the summary text
and a few leading words.

  python3 tests/golden/make_golden_chr21.py [--work /tmp/gold_chr21]
"""
import argparse
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import zlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
PARAMS = dict(genome_seed=2101, contig_len=46700000, reads_seed=2102, n_reads=100000, read_len=100)


def generate():
    import synth
    contigs = synth.make_genome(PARAMS["genome_seed"], 1, PARAMS["contig_len"], features=True)
    r1, _, _ = synth.make_reads(PARAMS["reads_seed"], contigs, PARAMS["n_reads"], PARAMS["read_len"])
    return contigs, r1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_chr21")
    a = ap.parse_args()
    W = a.work
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W)
    import refio
    import synth
    contigs, r1 = generate()
    g = contigs[0]
    print("genome", len(g), flush=True)
    mers, ukmer, ustart, cs = refio.kmer_index(contigs)
    print("k-mers", len(mers), "distinct", len(ukmer), flush=True)
    with open(os.path.join(W, "c21.sdx"), "w") as f:
        f.write("1\n%d\tchr21\n16\n" % (len(g) - 15))
    with gzip.open(os.path.join(W, "c21.seq"), "wb", compresslevel=1) as f:
        f.write(g.tobytes())
    mers.astype("<u4").tofile(os.path.join(W, "c21.mdx"))
    # .idx = gz of pos_index[0 .. 2^32]: number of indexed positions whose 16-mer is below k
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(os.path.join(W, "c21.idx"), "wb") as f:
        step = 1 << 26
        for lo in range(0, 1 << 32, step):
            k = np.arange(lo, lo + step, dtype=np.uint64).astype(np.uint32)
            f.write(co.compress(ustart[np.searchsorted(ukmer, k, side="left")].astype("<u4").tobytes()))
        f.write(co.compress(np.array([len(mers)], "<u4").tobytes()))
        f.write(co.flush())
    print("index files written", flush=True)
    fq = os.path.join(W, "c21_1_.fastq.gz")
    synth.write_fastq(fq, r1, 1)
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "pemapper"), "out", os.path.join(W, "c21.sdx"), "s", fq, "N", "0.85", "8", "200000000"],
                          cwd=W, stdout=subprocess.DEVNULL)
    m1 = np.fromfile(fq + ".mfile", dtype="<u4")[:PARAMS["n_reads"]]
    pile = refio.read_pileup(os.path.join(W, "out.pileup.gz"))
    ins = refio.read_indel(os.path.join(W, "out.indel.txt.gz"))
    rec = dict(params=PARAMS, genome_len=int(len(g)), genome_md5=hashlib.md5(g.tobytes()).hexdigest(),
               reads_md5=hashlib.md5(b"\n".join(r1)).hexdigest(), n_mers=int(len(mers)),
               m1_md5=refio.md5(m1), m1_head=[int(x) for x in m1[:64]], mapped=int((m1 > 0).sum()),
               pileup_records=int(len(pile)), pileup_md5=refio.md5(pile),
               indel_md5=hashlib.md5(repr(ins).encode()).hexdigest(), indel_sites=len(ins),
               ins_md5=hashlib.md5(repr(sorted((r[0], r[1], s) for r in ins for s in r[7])).encode()).hexdigest(),
               insertions=sum(len(r[7]) for r in ins),
               summary=open(os.path.join(W, "out.summary.txt")).read())
    json.dump(rec, open(os.path.join(HERE, "chr21.json"), "w"), indent=1)
    print({k: v for k, v in rec.items() if k not in ("summary", "m1_head")})


if __name__ == "__main__":
    main()
