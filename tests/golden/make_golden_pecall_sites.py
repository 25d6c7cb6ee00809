#!/usr/bin/env python3
"""Golden vectors for the PECaller per-site caller (development container only; needs /root/reference and oracle/_ref).

Runs the UNMODIFIED reference (oracle/_ref/pecaller = gcc -O1 of /root/reference/src/pecaller.c, `make -C oracle ref`) with
one worker thread on synthetic binary pileups of 8 samples and stores, under tests/golden/:

  pecall_sites.npz          inputs: reads[site][sample][6] (u16), pos[site] (0-based index into .seq), sample names
  pecall_sites.base.txt.gz  the reference's <out>.base.gz rows (one per site it processed), sorted
  pecall_sites.snp.txt      the reference's <out>.snp rows, sorted
  pecall_sites.dist.txt     the reference's <out>.dist (coverage statistics of the dispatcher; includes the 40 padding columns)

Only data is stored: inputs and the text the reference printed.

  python3 tests/golden/make_golden_pecall_sites.py [--work /tmp/gold_sites]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_sites")
    a = ap.parse_args()
    W = a.work
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refio
    contigs = refio.read_fasta(os.path.join(HERE, "g1.fa.gz"))
    seq = b"".join(x.tobytes() for x in contigs[1])
    shutil.copy(os.path.join(HERE, "g1.sdx"), os.path.join(W, "g1.sdx"))
    with gzip.open(os.path.join(W, "g1.seq"), "wb") as f:
        f.write(seq)
    rng = np.random.default_rng(4242)
    n_samp = 8
    names = ["s%d" % i for i in range(n_samp)]
    depth = [30, 28, 33, 25, 38, 14, 3, 45]
    first, n_sites = 100, 6000
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    # planted variants shared by the samples: allele frequency q, genotypes under Hardy-Weinberg
    is_var = rng.random(n_sites) < 0.03
    q = rng.uniform(0.05, 0.6, n_sites)
    alt = rng.integers(0, 6, n_sites)
    reads = np.zeros((n_sites, n_samp, 6), np.uint16)
    # special stretches: 3000-3300 a second alternative allele in some samples (MULTIALLELIC), 3300-3600 a fifth of the reads
    # replaced by random bases (MESS), 3600-3900 variants private to one sample sequenced at depth ~9 (LOW)
    alt2 = (alt + 1 + rng.integers(0, 3, n_sites)) % 4
    for i in range(n_sites):
        r = code.get(seq[first + i])
        if r is None:
            continue
        for s in range(n_samp):
            d = int(rng.poisson(depth[s]))
            if 2000 <= i < 2100:
                d = int(rng.poisson(2))       # a shallow stretch: average depth < 8 -> every call 'N'
            g = (r, r)
            if is_var[i]:
                g = tuple(int(alt[i]) if rng.random() < q[i] else r for _ in range(2))
            err = 0.004
            if 3000 <= i < 3300 and i % 5 == 0:
                g = tuple(int([alt[i] % 4, alt2[i], r][int(rng.integers(0, 3))]) for _ in range(2))
            if 3300 <= i < 3600 and i % 7 == 0:
                err = 0.2
                g = (r, int(alt[i]) % 4) if s % 2 else (r, r)
            if 3600 <= i < 3900 and i % 6 == 0:
                g = (int(alt[i]) % 4, int(alt[i]) % 4) if s == 6 else (r, r)
                if s == 6:
                    d = int(rng.poisson(9))
            cnt = np.zeros(6, np.int64)
            for _ in range(d):
                al = g[int(rng.integers(0, 2))]
                if rng.random() < err:
                    al = int(rng.integers(0, 4))
                if al == 5:                    # an insertion is counted on top of the base it follows
                    cnt[r] += 1
                    cnt[5] += 1
                else:
                    cnt[al] += 1
            reads[i, s] = cnt
    rundir = os.path.join(W, "run")
    os.makedirs(rundir)
    pos = first + np.arange(n_sites)
    pad = 40                                   # sites in flight when the reader finishes are lost (pecaller.c:1071, 1207)
    for s in range(n_samp):
        recs = []
        for i in range(n_sites):
            if reads[i, s].sum() > 0:
                recs.append(struct.pack("<I6H", int(pos[i]), *[int(x) for x in reads[i, s]]))
        for k in range(pad):
            recs.append(struct.pack("<I6H", int(first + n_sites + k), 20, 0, 0, 0, 0, 0))
        with gzip.open(os.path.join(rundir, "%s.pileup.gz" % names[s]), "wb") as f:
            f.write(b"".join(recs))
    subprocess.check_call([REFBIN, "pileup", os.path.join(W, "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "n"], cwd=rundir,
                          stdout=subprocess.DEVNULL)
    base = gzip.open(os.path.join(rundir, "out.base.gz"), "rt").read().split("\n")
    hdr, rows = base[0], sorted(x for x in base[1:] if x)
    # the column order is the directory order the reference saw
    cols = [c for c in hdr.split("\t")[3:] if c]
    snp = open(os.path.join(rundir, "out.snp")).read().split("\n")
    snp_rows = sorted(x for x in snp[1:] if x)
    keep_rows = [x for x in rows if int(x.split("\t")[1]) <= contig_pos(pos[-1], contigs)]
    keep_snp = [x for x in snp_rows if int(x.split("\t")[1]) <= contig_pos(pos[-1], contigs)]
    with gzip.open(os.path.join(HERE, "pecall_sites.base.txt.gz"), "wt") as f:
        f.write(hdr + "\n" + "\n".join(keep_rows) + "\n")
    with open(os.path.join(HERE, "pecall_sites.snp.txt"), "w") as f:
        f.write(snp[0] + "\n" + "\n".join(keep_snp) + "\n")
    shutil.copy(os.path.join(rundir, "out.dist"), os.path.join(HERE, "pecall_sites.dist.txt"))
    np.savez_compressed(os.path.join(HERE, "pecall_sites.npz"), reads=reads, pos=pos.astype(np.uint32), names=np.array(names),
                        columns=np.array(cols), pad=np.array([pad]))
    print("sites", n_sites, "base rows", len(keep_rows), "snp rows", len(keep_snp), "columns", cols)


def contig_pos(p, contigs):
    return int(p) + 1          # all sites lie in the first contig


if __name__ == "__main__":
    main()
