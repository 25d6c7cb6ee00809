#!/usr/bin/env python3
"""Golden vector for pileup streams that are NOT in ascending order (development container only; needs oracle/_ref).

The reference's dispatcher takes the lowest pending position of all streams, column after column (find_lowest, pecaller.c:865-923,
1820-1833), whatever order a stream's records come in: a record behind its successor becomes a column of its own as soon as it is the
lowest pending one, a position that occurs twice in a stream gives two columns.  pemapper never writes such a file; a drop-in caller
has to take one all the same.  Same columns as pecall_sites.npz (make_golden_pecall_sites.py), with three disturbances:
  sample 2: the records of columns 1000 and 1001 swapped; the record of column 2500 moved behind that of column 2510
  sample 5: the record of column 1500 twice (the second time with every count halved)
Stores under tests/golden/: pecall_unordered.json (the disturbances), pecall_unordered.base.txt.gz / .snp.txt / .dist.txt (what the
unmodified reference wrote, rows sorted).

  python3 tests/golden/make_golden_pecall_unordered.py [--work /tmp/gold_unordered]
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
SPEC = {"swap": [2, 1000, 1001], "late": [2, 2500, 2510], "twice": [5, 1500]}


def stream_records(z, s, spec=SPEC):
    """the 16-byte records of sample s, disturbed as `spec` says (column numbers are indices into z["pos"])"""
    reads, pos, pad = z["reads"], z["pos"], int(z["pad"][0])
    order = [i for i in range(len(pos)) if reads[i, s].sum() > 0]
    extra = {}
    if s == spec["swap"][0]:
        a, b = order.index(spec["swap"][1]), order.index(spec["swap"][2])
        order[a], order[b] = order[b], order[a]
    if s == spec["late"][0]:
        a = order.index(spec["late"][1])
        x = order.pop(a)
        order.insert(order.index(spec["late"][2]) + 1, x)
    recs = []
    for i in order:
        recs.append(struct.pack("<I6H", int(pos[i]), *[int(x) for x in reads[i, s]]))
        if s == spec["twice"][0] and i == spec["twice"][1]:
            recs.append(struct.pack("<I6H", int(pos[i]), *[int(x) // 2 for x in reads[i, s]]))
    recs += [struct.pack("<I6H", int(pos[-1]) + 1 + k, 20, 0, 0, 0, 0, 0) for k in range(pad)]
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_unordered")
    a = ap.parse_args()
    W = a.work
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refio
    _, seqs = refio.read_fasta(os.path.join(HERE, "g1.fa.gz"))
    shutil.copy(os.path.join(HERE, "g1.sdx"), os.path.join(W, "g1.sdx"))
    with gzip.open(os.path.join(W, "g1.seq"), "wb") as f:
        f.write(b"".join(x.tobytes() for x in seqs))
    z = np.load(os.path.join(HERE, "pecall_sites.npz"))
    names = [str(x) for x in z["names"]]
    rundir = os.path.join(W, "run")
    os.makedirs(rundir)
    for s, nm in enumerate(names):
        with gzip.open(os.path.join(rundir, "%s.pileup.gz" % nm), "wb") as f:
            f.write(b"".join(stream_records(z, s)))
    subprocess.check_call([REFBIN, "pileup", os.path.join(W, "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "n"], cwd=rundir, stdout=subprocess.DEVNULL)
    base = gzip.open(os.path.join(rundir, "out.base.gz"), "rt").read().split("\n")
    snp = open(os.path.join(rundir, "out.snp")).read().split("\n")
    last = int(z["pos"][-1]) + 1
    keep = lambda rows: sorted(x for x in rows if x and int(x.split("\t")[1]) <= last)      # (not the padding columns)
    with gzip.open(os.path.join(HERE, "pecall_unordered.base.txt.gz"), "wt") as f:
        f.write(base[0] + "\n" + "\n".join(keep(base[1:])) + "\n")
    with open(os.path.join(HERE, "pecall_unordered.snp.txt"), "w") as f:
        f.write(snp[0] + "\n" + "\n".join(keep(snp[1:])) + "\n")
    shutil.copy(os.path.join(rundir, "out.dist"), os.path.join(HERE, "pecall_unordered.dist.txt"))
    json.dump(SPEC, open(os.path.join(HERE, "pecall_unordered.json"), "w"))
    rows = keep(base[1:])
    from collections import Counter
    c = Counter(int(x.split("\t")[1]) for x in rows)
    print("base rows", len(rows), "positions printed more than once:", sorted(p for p, n in c.items() if n > 1))


if __name__ == "__main__":
    main()
