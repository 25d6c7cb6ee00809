#!/usr/bin/env python3
"""Golden vectors for the per-site caller with MORE THAN 64 samples (development container only; needs /root/reference and
oracle/_ref): the reference takes any INDIV (pecaller.c:251-257); the device caller maps a lane to a sample of each chunk of 64.

Runs the UNMODIFIED reference (oracle/_ref/pecaller, gcc -O1, one worker thread) on synthetic binary pileups of 100 samples
over 1,500 columns and stores, under tests/golden/:

  (python3 tests/golden/make_golden_pecall_wide.py --samples 300 --sites 400 --tag pecall_wide300: the same beyond 256 samples)
  pecall_wide.npz          inputs: reads[site][sample][6] (u16), pos[site], sample names, the reference's column order
  pecall_wide.base.txt.gz  the reference's <out>.base.gz rows, sorted
  pecall_wide.snp.txt      the reference's <out>.snp rows, sorted

Only data is stored: inputs and the text the reference printed.

  python3 tests/golden/make_golden_pecall_wide.py [--work /tmp/gold_wide]
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
    ap.add_argument("--work", default="/tmp/gold_wide")
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--sites", type=int, default=1500)
    ap.add_argument("--tag", default="pecall_wide", help="name of the stored files (pecall_wide300: --samples 300 --sites 400, beyond 256 samples)")
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
    rng = np.random.default_rng(10100)
    n_samp = a.samples
    names = ["w%03d" % i for i in range(n_samp)]
    first, n_sites = 5000, a.sites
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    ref = np.array([code.get(seq[first + i], -1) for i in range(n_sites)])
    depth = rng.integers(18, 45, n_samp)
    depth[7] = 3                # a sample under the depth floor most of the time
    if n_samp > 70:
        depth[70] = 6
    is_var = rng.random(n_sites) < 0.04
    q = rng.uniform(0.03, 0.6, n_sites)
    alt = rng.integers(0, 6, n_sites)
    reads = np.zeros((n_sites, n_samp, 6), np.int64)
    for i in range(n_sites):
        r = ref[i]
        if r < 0:
            continue
        d = rng.poisson(depth)
        if 400 <= i < 440:
            d = rng.poisson(2, n_samp)          # a shallow stretch: average depth < 8 -> every call 'N'
        a1 = np.where(is_var[i] & (rng.random(n_samp) < q[i]), alt[i], r)
        a2 = np.where(is_var[i] & (rng.random(n_samp) < q[i]), alt[i], r)
        for s in range(n_samp):
            pick = np.where(rng.random(d[s]) < 0.5, a1[s], a2[s])
            e = rng.random(d[s]) < 0.004
            pick = np.where(e, rng.integers(0, 4, d[s]), pick)
            for al in pick:
                if al == 5:                    # an insertion is counted on top of the base it follows
                    reads[i, s, r] += 1
                    reads[i, s, 5] += 1
                else:
                    reads[i, s, al] += 1
    reads = reads.astype(np.uint16)
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
    subprocess.check_call([REFBIN, "pileup", os.path.join(W, "g1.sdx"), str(n_samp + 5), "out", "0.95", "0.001", "n", "2", "n"], cwd=rundir,
                          stdout=subprocess.DEVNULL)
    base = gzip.open(os.path.join(rundir, "out.base.gz"), "rt").read().split("\n")
    hdr, rows = base[0], sorted(x for x in base[1:] if x)
    cols = [c for c in hdr.split("\t")[3:] if c]      # the column order is the directory order the reference saw
    snp = open(os.path.join(rundir, "out.snp")).read().split("\n")
    snp_rows = sorted(x for x in snp[1:] if x)
    last = int(pos[-1]) + 1                            # all sites lie in the first contig
    keep_rows = [x for x in rows if int(x.split("\t")[1]) <= last]
    keep_snp = [x for x in snp_rows if int(x.split("\t")[1]) <= last]
    with gzip.open(os.path.join(HERE, a.tag + ".base.txt.gz"), "wt") as f:
        f.write(hdr + "\n" + "\n".join(keep_rows) + "\n")
    with open(os.path.join(HERE, a.tag + ".snp.txt"), "w") as f:
        f.write(snp[0] + "\n" + "\n".join(keep_snp) + "\n")
    np.savez_compressed(os.path.join(HERE, a.tag + ".npz"), reads=reads, pos=pos.astype(np.uint32), names=np.array(names),
                        columns=np.array(cols), pad=np.array([pad]))
    print("sites", n_sites, "samples", n_samp, "base rows", len(keep_rows), "snp rows", len(keep_snp))


if __name__ == "__main__":
    main()
