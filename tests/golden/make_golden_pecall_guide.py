#!/usr/bin/env python3
"""Golden vectors for pecaller's BED guide mode (development container only; needs oracle/_ref).

The unmodified reference run with a guide file (pecaller.c:925-1068) on 8 samples: every position of three intervals is
called, covered or not; the second and third contigs are renamed chrY / chrMT in a copy of the .sdx so that the forced
HAPLOID of those columns (955-957) is exercised.  Stores under tests/golden/:

  pecall_guide.npz          reads[record][sample][6] and pos[record] (the pileup records), sample names, column order
  pecall_guide.sdx / .bed   the renamed contig table and the guide intervals
  pecall_guide.base.txt.gz / .snp.txt / .dist.txt   what the reference wrote (rows sorted)

  python3 tests/golden/make_golden_pecall_guide.py [--work /tmp/gold_guide]
"""
import argparse
import gzip
import os
import shutil
import struct
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "pecaller")
BED = [("chr1", 1001, 1500), ("chrY", 500, 800), ("chrMT", 200, 400)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_guide")
    a = ap.parse_args()
    W = a.work
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refio
    names_c, seqs = refio.read_fasta(os.path.join(HERE, "g1.fa.gz"))
    seq = b"".join(x.tobytes() for x in seqs)
    sdx = open(os.path.join(HERE, "g1.sdx")).read().split("\n")
    sdx[2] = sdx[2].split("\t")[0] + "\tchrY"
    sdx[3] = sdx[3].split("\t")[0] + "\tchrMT"
    open(os.path.join(HERE, "pecall_guide.sdx"), "w").write("\n".join(sdx))
    shutil.copy(os.path.join(HERE, "pecall_guide.sdx"), os.path.join(W, "g1.sdx"))
    with gzip.open(os.path.join(W, "g1.seq"), "wb") as f:
        f.write(seq)
    with open(os.path.join(HERE, "pecall_guide.bed"), "w") as f:
        for b in BED:
            f.write("%s\t%d\t%d\n" % b)
    lens = [int(x.split("\t")[0]) + 15 for x in sdx[1:11]]
    starts = np.concatenate([[0], np.cumsum(lens)])
    cname = ["chr1", "chrY", "chrMT"]
    rng = np.random.default_rng(31337)
    n_samp = 8
    names = ["s%d" % i for i in range(n_samp)]
    depth = [30, 28, 33, 25, 38, 14, 9, 45]
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    pos, reads = [], []
    for cn, lo, hi in BED:
        ci = cname.index(cn)
        for p1 in range(lo - 20, hi + 20):        # records also just outside the intervals (skipped by the guide)
            g = int(starts[ci]) + p1 - 1
            r = code.get(seq[g])
            if r is None or rng.random() < 0.05:  # 5 % of the positions have no record at all
                continue
            var = rng.random() < 0.06
            alt = int(rng.integers(0, 6))
            q = rng.uniform(0.1, 0.6)
            col = np.zeros((n_samp, 6), np.uint16)
            for s in range(n_samp):
                if rng.random() < 0.03:
                    continue                       # this sample has no record here
                d = int(rng.poisson(depth[s]))
                if ci == 0:
                    gt = tuple(alt if (var and rng.random() < q) else r for _ in range(2))
                else:
                    a1 = alt if (var and rng.random() < q) else r
                    gt = (a1, a1)
                for _ in range(d):
                    al = gt[int(rng.integers(0, 2))]
                    if rng.random() < 0.004:
                        al = int(rng.integers(0, 4))
                    if al == 5:
                        col[s, r] += 1
                    col[s, al] += 1
            pos.append(g)
            reads.append(col)
    pos = np.array(pos, np.uint32)
    reads = np.array(reads, np.uint16)
    rundir = os.path.join(W, "run")
    os.makedirs(rundir)
    tail = int(starts[3]) + 5000
    for s in range(n_samp):
        recs = [struct.pack("<I6H", int(pos[i]), *[int(x) for x in reads[i, s]]) for i in range(len(pos)) if reads[i, s].sum() > 0]
        recs += [struct.pack("<I6H", tail + k, 20, 0, 0, 0, 0, 0) for k in range(40)]   # keeps the files open past the last interval
        with gzip.open(os.path.join(rundir, "%s.pileup.gz" % names[s]), "wb") as f:
            f.write(b"".join(recs))
    subprocess.check_call([REFBIN, "pileup", os.path.join(W, "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "n",
                           os.path.join(HERE, "pecall_guide.bed")], cwd=rundir, stdout=subprocess.DEVNULL)
    base = gzip.open(os.path.join(rundir, "out.base.gz"), "rt").read().split("\n")
    snp = open(os.path.join(rundir, "out.snp")).read().split("\n")
    cols = [c for c in base[0].split("\t")[3:] if c]
    with gzip.open(os.path.join(HERE, "pecall_guide.base.txt.gz"), "wt") as f:
        f.write(base[0] + "\n" + "\n".join(sorted(x for x in base[1:] if x)) + "\n")
    with open(os.path.join(HERE, "pecall_guide.snp.txt"), "w") as f:
        f.write(snp[0] + "\n" + "\n".join(sorted(x for x in snp[1:] if x)) + "\n")
    shutil.copy(os.path.join(rundir, "out.dist"), os.path.join(HERE, "pecall_guide.dist.txt"))
    np.savez_compressed(os.path.join(HERE, "pecall_guide.npz"), reads=reads, pos=pos, names=np.array(names), columns=np.array(cols),
                        tail=np.array([tail]))
    print("records", len(pos), "base rows", len([x for x in base[1:] if x]), "snp rows", len([x for x in snp[1:] if x]), "columns", cols)


if __name__ == "__main__":
    main()
