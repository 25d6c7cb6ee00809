#!/usr/bin/env python3
"""Golden vectors for the PECaller per-site caller WITH a pedigree (development container only; needs oracle/_ref).

Like make_golden_pecall_sites.py, with `use_pedfile = y`: two trios, a second child, and a child whose father is not
sampled (the reference then reuses the previous kid's father genotype, pecaller.c:2590-2600).  Children inherit one
allele from each parent; 1 % of the columns carry a new allele in one child.  Stores under tests/golden/:

  pecall_ped.npz           reads[site][sample][6], pos, sample names, the reference's column order, the pedigree
  pecall_ped.base.txt.gz   the reference's <out>.base.gz rows, sorted
  pecall_ped.snp.txt       the reference's <out>.snp rows (with DENOVO_ types), sorted

  python3 tests/golden/make_golden_pecall_ped.py [--work /tmp/gold_ped]
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
PED = [("fam1", "s0", "0", "0", 1), ("fam1", "s1", "0", "0", 2), ("fam1", "s2", "s0", "s1", 1),
       ("fam2", "s3", "0", "0", 1), ("fam2", "s4", "0", "0", 2), ("fam2", "s5", "s3", "s4", 2), ("fam2", "s6", "s3", "s4", 1),
       ("fam3", "s7", "0", "s4", 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_ped")
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
    rng = np.random.default_rng(9091)
    names = [p[1] for p in PED]
    n_samp = len(names)
    dad = [names.index(p[2]) if p[2] != "0" else -1 for p in PED]
    mom = [names.index(p[3]) if p[3] != "0" else -1 for p in PED]
    sex = [p[4] for p in PED]
    depth = [32, 30, 34, 27, 36, 29, 22, 40]
    first, n_sites = 20000, 4000
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    is_var = rng.random(n_sites) < 0.04
    q = rng.uniform(0.1, 0.6, n_sites)
    alt = rng.integers(0, 6, n_sites)
    novo = rng.random(n_sites) < 0.01
    reads = np.zeros((n_sites, n_samp, 6), np.uint16)
    for i in range(n_sites):
        r = code.get(seq[first + i])
        if r is None:
            continue
        geno = [None] * n_samp
        for s in range(n_samp):                 # founders first: the table lists parents before children
            def pop():
                return int(alt[i]) if (is_var[i] and rng.random() < q[i]) else r
            if dad[s] < 0 and mom[s] < 0:
                geno[s] = (pop(), pop())
            else:
                a1 = geno[dad[s]][int(rng.integers(0, 2))] if dad[s] >= 0 else pop()
                a2 = geno[mom[s]][int(rng.integers(0, 2))] if mom[s] >= 0 else pop()
                geno[s] = (a1, a2)
        if novo[i]:
            kid = [2, 5, 6, 7][int(rng.integers(0, 4))]
            geno[kid] = (geno[kid][0], int((r + 1 + rng.integers(0, 3)) % 4))
        for s in range(n_samp):
            d = int(rng.poisson(depth[s]))
            cnt = np.zeros(6, np.int64)
            for _ in range(d):
                al = geno[s][int(rng.integers(0, 2))]
                if rng.random() < 0.004:
                    al = int(rng.integers(0, 4))
                if al == 5:
                    cnt[r] += 1
                    cnt[5] += 1
                else:
                    cnt[al] += 1
            reads[i, s] = cnt
    rundir = os.path.join(W, "run")
    os.makedirs(rundir)
    pos = first + np.arange(n_sites)
    pad = 40
    for s in range(n_samp):
        recs = [struct.pack("<I6H", int(pos[i]), *[int(x) for x in reads[i, s]]) for i in range(n_sites) if reads[i, s].sum() > 0]
        recs += [struct.pack("<I6H", int(first + n_sites + k), 20, 0, 0, 0, 0, 0) for k in range(pad)]
        with gzip.open(os.path.join(rundir, "%s.pileup.gz" % names[s]), "wb") as f:
            f.write(b"".join(recs))
    with open(os.path.join(W, "ped.txt"), "w") as f:
        for p in PED:
            f.write("%s\t%s\t%s\t%s\t%d\n" % p)
    rate = 1e-6
    subprocess.check_call([REFBIN, "pileup", os.path.join(W, "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "y", os.path.join(W, "ped.txt"),
                           "%g" % rate], cwd=rundir, stdout=subprocess.DEVNULL)
    base = gzip.open(os.path.join(rundir, "out.base.gz"), "rt").read().split("\n")
    hdr = base[0]
    cols = [c for c in hdr.split("\t")[3:] if c]
    last = int(pos[-1]) + 1
    rows = sorted(x for x in base[1:] if x and int(x.split("\t")[1]) <= last)
    snp = open(os.path.join(rundir, "out.snp")).read().split("\n")
    snp_rows = sorted(x for x in snp[1:] if x and int(x.split("\t")[1]) <= last)
    with gzip.open(os.path.join(HERE, "pecall_ped.base.txt.gz"), "wt") as f:
        f.write(hdr + "\n" + "\n".join(rows) + "\n")
    with open(os.path.join(HERE, "pecall_ped.snp.txt"), "w") as f:
        f.write(snp[0] + "\n" + "\n".join(snp_rows) + "\n")
    np.savez_compressed(os.path.join(HERE, "pecall_ped.npz"), reads=reads, pos=pos.astype(np.uint32), names=np.array(names),
                        columns=np.array(cols), dad=np.array(dad), mom=np.array(mom), sex=np.array(sex), denovo_rate=np.array([rate]))
    print("sites", n_sites, "base rows", len(rows), "snp rows", len(snp_rows), "columns", cols)
    import collections
    print(collections.Counter(x.split("\t")[5] for x in snp_rows))


if __name__ == "__main__":
    main()
