#!/usr/bin/env python3
"""Golden vector for the END of pecaller's BED guide mode (development container only; needs oracle/_ref).

The reference's guide loop runs `while (running_files > 0)` (pecaller.c:952): the column at which the last pileup stream ends is
the last one printed, however far the BED interval goes on.  Same records as pecall_guide.npz (make_golden_pecall_guide.py), but
every record behind chrMT:CUT is dropped and no stream has padding records, so that all eight streams end inside the third
interval [200, 400].  Stores under tests/golden/:

  pecall_guide_end.json                  {"cut": CUT}
  pecall_guide_end.base.txt.gz / .dist.txt   what the unmodified reference wrote (rows sorted; the last rows in flight when the
                                             dispatcher stops are lost by the reference, pecaller.c:1071, 1207 -- the test allows for them)

  python3 tests/golden/make_golden_pecall_guide_end.py [--work /tmp/gold_guide_end]
"""
import argparse
import gzip
import json
import os
import shutil
import struct
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "pecaller")
CUT = 300


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_guide_end")
    a = ap.parse_args()
    W = a.work
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refio
    _, seqs = refio.read_fasta(os.path.join(HERE, "g1.fa.gz"))
    shutil.copy(os.path.join(HERE, "pecall_guide.sdx"), os.path.join(W, "g1.sdx"))
    with gzip.open(os.path.join(W, "g1.seq"), "wb") as f:
        f.write(b"".join(x.tobytes() for x in seqs))
    lens, cn, _ = refio.read_sdx(os.path.join(HERE, "pecall_guide.sdx"))
    starts = np.concatenate([[0], np.cumsum(np.array(lens) + 15)])
    z = np.load(os.path.join(HERE, "pecall_guide.npz"))
    limit = int(starts[cn.index("chrMT")]) + CUT - 1          # 0-based index into .seq of chrMT:CUT
    rundir = os.path.join(W, "run")
    os.makedirs(rundir)
    for s, nm in enumerate([str(x) for x in z["names"]]):
        recs = [struct.pack("<I6H", int(z["pos"][i]), *[int(x) for x in z["reads"][i, s]]) for i in range(len(z["pos"]))
                if z["reads"][i, s].sum() > 0 and int(z["pos"][i]) <= limit]
        with gzip.open(os.path.join(rundir, "%s.pileup.gz" % nm), "wb") as f:
            f.write(b"".join(recs))
    subprocess.check_call([REFBIN, "pileup", os.path.join(W, "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "n",
                           os.path.join(HERE, "pecall_guide.bed")], cwd=rundir, stdout=subprocess.DEVNULL)
    base = gzip.open(os.path.join(rundir, "out.base.gz"), "rt").read().split("\n")
    with gzip.open(os.path.join(HERE, "pecall_guide_end.base.txt.gz"), "wt") as f:
        f.write(base[0] + "\n" + "\n".join(sorted(x for x in base[1:] if x)) + "\n")
    shutil.copy(os.path.join(rundir, "out.dist"), os.path.join(HERE, "pecall_guide_end.dist.txt"))
    json.dump({"cut": CUT}, open(os.path.join(HERE, "pecall_guide_end.json"), "w"))
    rows = [x for x in base[1:] if x]
    print("base rows", len(rows), "last chrMT row", max(int(x.split("\t")[1]) for x in rows if x.startswith("chrMT")))


if __name__ == "__main__":
    main()
