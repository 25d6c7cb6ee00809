#!/usr/bin/env python3
"""Golden vectors for the PECaller likelihood kernel (development container only; needs /root/reference).

The reference's pecaller prints calls and posteriors only, so the per-sample likelihoods of fill_sample_like
(src/pecaller.c:2448-2507) are captured from an INSTRUMENTED SCRATCH BUILD: the reference source is streamed through
a three-line insertion (a binary dump right after the fill_sample_like call, src/pecaller.c:1365) into a temporary
directory, compiled there at -O1 and run with one worker thread on synthetic pileups.  Nothing
of the reference is copied into the repository: only inputs (pileup counts, alpha means, norm) and outputs (like[14],
initial_call, initial_p) of that function are stored, in tests/golden/pecall_like.npz.

  python3 tests/golden/make_golden_pecall.py [--work /tmp/gold_pc]
"""
import argparse
import gzip
import os
import struct
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFSRC = "/root/reference/src/pecaller.c"
CALL = "fill_sample_like (samples, d_alpha_mean, max_gen, INDIV, ordered_samples, new_norm[pass], average_depth);"
DUMP = r'''
	{ FILE *df_ = fopen ("like_dump.bin", "ab"); int ii_, jj_; double nn_ = new_norm[pass]; int hdr_[4];
	  hdr_[0] = INDIV; hdr_[1] = pass; hdr_[2] = max_gen; hdr_[3] = min_depth_needed; fwrite (hdr_, sizeof (int), 4, df_); fwrite (&nn_, sizeof (double), 1, df_);
	  for (ii_ = 0; ii_ < MAX_GENOTYPES; ii_++) fwrite (d_alpha_mean[ii_], sizeof (double), NO_ALLELES, df_);
	  for (ii_ = 0; ii_ < INDIV; ii_++) { int ic_ = samples[ii_]->initial_call; fwrite (samples[ii_]->reads, sizeof (int), NO_ALLELES, df_);
	    fwrite (samples[ii_]->like, sizeof (double), MAX_GENOTYPES, df_); fwrite (&samples[ii_]->initial_p, sizeof (double), 1, df_); fwrite (&ic_, sizeof (int), 1, df_);
	    jj_ = samples[ii_]->tot; fwrite (&jj_, sizeof (int), 1, df_); }
	  fclose (df_); }
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_pc")
    ap.add_argument("--index", default="/tmp/gold/g1", help="basename of an existing .sdx/.seq pair (make_golden.py)")
    a = ap.parse_args()
    W = a.work
    os.makedirs(W, exist_ok=True)
    src = open(REFSRC).read()
    assert src.count(CALL) == 1
    with open(os.path.join(W, "pecaller_dump.c"), "w") as f:
        f.write(src.replace(CALL, CALL + DUMP))
    subprocess.check_call(["gcc", "-O1", "-w", "-o", os.path.join(W, "pecaller_dump"), os.path.join(W, "pecaller_dump.c"),
                           "-lm", "-lz", "-lpthread"])
    os.remove(os.path.join(W, "pecaller_dump.c"))
    # synthetic pileups: 6 samples, sites 1..4000 of the first contig, Poisson(30) depth (two shallow samples), 0.4 % error,
    # planted SNPs / deletions / insertions
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refio
    seq = gzip.open(a.index + ".seq", "rb").read()
    rng = np.random.default_rng(777)
    n_sites, n_samp = 4000, 6
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    rundir = os.path.join(W, "run")
    os.makedirs(rundir, exist_ok=True)
    for f in os.listdir(rundir):
        os.remove(os.path.join(rundir, f))
    var_site = rng.random(n_sites) < 0.02
    var_alt = rng.integers(0, 6, n_sites)
    for s in range(n_samp):
        recs = []
        mean_depth = [30, 30, 30, 12, 3, 45][s]
        for i in range(60, n_sites):
            r = code.get(seq[i])
            if r is None:
                continue
            d = int(rng.poisson(mean_depth))
            cnt = np.zeros(6, np.int64)
            gt = (r, r)
            if var_site[i] and rng.random() < 0.5:
                gt = (r, int(var_alt[i])) if rng.random() < 0.7 else (int(var_alt[i]), int(var_alt[i]))
            for _ in range(d):
                al = gt[int(rng.integers(0, 2))]
                if rng.random() < 0.004:
                    al = int(rng.integers(0, 4))
                cnt[al] += 1
            if cnt.sum() > 0:
                recs.append(struct.pack("<I6H", i, *[int(x) for x in cnt]))
        with gzip.open(os.path.join(rundir, "s%d.pileup.gz" % s), "wb") as f:
            f.write(b"".join(recs))
    subprocess.check_call([os.path.join(W, "pecaller_dump"), "pileup", a.index + ".sdx", "11", "out", "0.95", "0.001", "n", "2", "n"],
                          cwd=rundir, stdout=subprocess.DEVNULL)
    raw = open(os.path.join(rundir, "like_dump.bin"), "rb").read()
    per_samp = 6 * 4 + 14 * 8 + 8 + 4 + 4
    reads, alpha, like, margin, best, norm, passes, tots = [], [], [], [], [], [], [], []
    o = 0
    while o < len(raw):
        indiv, ps, max_gen, min_depth = struct.unpack_from("<4i", raw, o)
        o += 16
        nn, = struct.unpack_from("<d", raw, o)
        o += 8
        al = np.frombuffer(raw, "<f8", 84, o).reshape(14, 6)
        o += 84 * 8
        rr, ll, mm, bb, tt = [], [], [], [], []
        for _ in range(indiv):
            rr.append(np.frombuffer(raw, "<i4", 6, o)); o += 24
            ll.append(np.frombuffer(raw, "<f8", 14, o)); o += 112
            mm.append(struct.unpack_from("<d", raw, o)[0]); o += 8
            bb.append(struct.unpack_from("<i", raw, o)[0]); o += 4
            tt.append(struct.unpack_from("<i", raw, o)[0]); o += 4
        reads.append(rr); alpha.append(al); like.append(ll); margin.append(mm); best.append(bb); norm.append(nn); passes.append(ps); tots.append(tt)
    assert max_gen == 14 and min_depth == 2
    reads = np.array(reads, np.uint16)
    keep = np.arange(len(reads))
    np.savez_compressed(os.path.join(HERE, "pecall_like.npz"), reads=reads[keep], alpha=np.array(alpha)[keep],
                        like=np.array(like)[keep], margin=np.array(margin)[keep], best=np.array(best, np.int8)[keep],
                        norm=np.array(norm)[keep], passes=np.array(passes, np.int8)[keep], tot=np.array(tots, np.int32)[keep])
    print("records", len(reads), "passes", np.bincount(np.array(passes)))


if __name__ == "__main__":
    main()
