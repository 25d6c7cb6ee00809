#!/usr/bin/env python3
"""wall clock of the host program pecaller_hip on N pileup columns x S samples of bench.py's config-4 generator (30x Poisson, 0.4 %
error, 1 variant per kb), on a genome of its own (the generator's reference bases), the generated columns laid down `repeats` times:  python3 tools/pecaller_cli_throughput.py [columns=1000000] [samples=64] [repeats=1]
(the program prints its own split: merge + device + text)"""
import gzip, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, refio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # the generated columns laid down REP times, one after the other
OFF = 1000              # (position 0 means 'stream ended' to the merge, pecaller.c:891-907)
reads, dom = bench.pecall_columns(n, S)              # [n][S][6] u16; dom = reference base code of the generator
# a genome of its own: the letter at every column is the generator's reference base
genome = np.full(OFF + n * REP + 1000, ord("A"), np.uint8)
genome[OFF:OFF + n * REP] = np.tile(np.frombuffer(b"ACGT", np.uint8)[dom], REP)
W = tempfile.mkdtemp()
open(os.path.join(W, "g1.sdx"), "w").write("1\n%d\tchr1\n" % len(genome))
gzip.open(os.path.join(W, "g1.seq"), "wb", compresslevel=1).write(genome.tobytes() + b"N" * 15)
run = os.path.join(W, "run")
os.mkdir(run)
rec = np.dtype([("pos", "<u4"), ("c", "<u2", 6)])


def write_sample(s):
    a = np.zeros(n, rec)
    a["c"] = reads[:, s, :]
    keep = a["c"].sum(axis=1) > 0
    with open(os.path.join(run, "s%03d.pileup.gz" % s), "wb") as f:
        for k in range(REP):
            a["pos"] = np.arange(n, dtype=np.uint32) + OFF + k * n
            f.write(gzip.compress(a[keep].tobytes(), compresslevel=1))      # (gz members concatenate)


import multiprocessing
with multiprocessing.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool:
    pool.map(write_sample, range(S))
n = n * REP
t0 = time.time()
out = subprocess.run([os.path.join(ROOT, "pecaller_amd", "pecaller_hip"), "pileup", os.path.join(W, "g1.sdx"), str(S), "out", "0.95", "0.001", "n", "24", "n"],
                     cwd=run, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
dt = time.time() - t0
txt = out.stdout.decode(errors="replace")
for l in txt.splitlines():
    if "pecaller_hip:" in l:
        print(l.strip())
assert out.returncode == 0, txt[-2000:]
print("%d columns x %d samples: wall %.2f s = %.3f M columns/s end to end; out.base.gz %d MB, out.snp %d rows" %
      (n, S, dt, n / dt / 1e6, os.path.getsize(os.path.join(run, "out.base.gz")) >> 20, sum(1 for _ in open(os.path.join(run, "out.snp")))))
shutil.rmtree(W)
