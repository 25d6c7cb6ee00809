#!/usr/bin/env python3
"""wall clock of the host program pecaller_hip on N pileup columns x S samples of bench.py's config-4 generator (30x Poisson, 0.4 %
error, 1 variant per kb), laid over the golden fixture's genome:  python3 tools/pecaller_cli_throughput.py [columns=1000000] [samples=64]
(the program prints its own split: merge + device + text)"""
import gzip, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, refio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
_, seqs = refio.read_fasta(os.path.join(ROOT, "tests", "golden", "g1.fa.gz"))
genome = np.concatenate(seqs)
n = min(n, len(genome) - 2000)
OFF = 1000              # (position 0 means 'stream ended' to the merge, pecaller.c:891-907)
reads, dom = bench.pecall_columns(n, S)              # [n][S][6] u16; dom = reference base code of the generator
# the generator's reference base is not the fixture genome's letter: rotate the four base columns so that it is
code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
g = np.array([code.get(int(c), 0) for c in genome[OFF:OFF + n]], np.int64)
rot = (g - dom.astype(np.int64)) % 4
idx = (np.arange(4)[None, :] - rot[:, None]) % 4
r2 = reads.copy()
r2[:, :, :4] = np.take_along_axis(reads[:, :, :4], np.broadcast_to(idx[:, None, :], (n, S, 4)), axis=2)
W = tempfile.mkdtemp()
shutil.copy(os.path.join(ROOT, "tests", "golden", "g1.sdx"), os.path.join(W, "g1.sdx"))
gzip.open(os.path.join(W, "g1.seq"), "wb", compresslevel=1).write(genome.tobytes())
run = os.path.join(W, "run")
os.mkdir(run)
rec = np.dtype([("pos", "<u4"), ("c", "<u2", 6)])
for s in range(S):
    a = np.zeros(n, rec)
    a["pos"] = np.arange(n, dtype=np.uint32) + OFF
    a["c"] = r2[:, s, :]
    a = a[a["c"].sum(axis=1) > 0]
    with gzip.open(os.path.join(run, "s%03d.pileup.gz" % s), "wb", compresslevel=1) as f:
        f.write(a.tobytes())
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
