#!/usr/bin/env python3
"""pemapper_hip, the host program, on an hg38-SIZED index (tools/cli_throughput.sh uses the 5 Mbp fixture genome): the bench's synthetic
3.1 Gbp genome written as <base>.sdx / <base>.seq, its synthetic 2 x 150 reads written as fastq -- one plain pair, and 8 gz pairs whose
quality lines have the entropy of real ones -- and the program run on each as the reference's would be.  Prints the program's own
"read and mapped" rate (input parsing included) and the wall time end to end (genome load, index build on the device, output files).
    python3 tools/cli_hg38.py [pairs=1000000] [repeat_frac=0.02] [threads=24] [copies=8: the files hold the pairs that many times]
Host-side files go to a scratch directory under $TMPDIR (~4 GB + the outputs); removed at the end."""
import gzip, os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def fastq_bytes(rows, lens, first, qual_rng=None):
    """fixed-width records: '@r%09d\\n' + 150 letters + '\\n+\\n' + 150 qualities + '\\n' (all reads of the bench are 150 long)"""
    n, L = len(lens), int(lens[0])
    assert (lens == L).all()
    hdr = np.frombuffer(("".join("@r%09d\n" % (first + i) for i in range(n))).encode(), np.uint8).reshape(n, 12)
    H = 12
    out = np.empty((n, H + L + 3 + L + 1), np.uint8)
    out[:, :H] = hdr
    out[:, H:H + L] = rows[:, :L]
    out[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", np.uint8)
    if qual_rng is None:
        out[:, H + L + 3:H + 2 * L + 3] = ord("I")
    else:
        q = np.clip(qual_rng.normal(36, 4, (n, L)).astype(np.int32), 2, 40)
        low = qual_rng.random((n, L)) < 0.05
        q[low] = qual_rng.integers(2, 20, int(low.sum()))
        out[:, H + L + 3:H + 2 * L + 3] = (q + 33).astype(np.uint8)
    out[:, -1] = ord("\n")
    return out.tobytes()


def gz_write(args):
    path, data = args
    with open(path, "wb") as f:
        f.write(gzip.compress(data, 6))
    return os.path.getsize(path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
    threads = sys.argv[3] if len(sys.argv) > 3 else "24"
    copies = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    from pecaller_amd.pemap import PemapDev
    W = tempfile.mkdtemp(prefix="cli_hg38_")
    try:
        t0 = time.time()
        dev = PemapDev(0)
        gsize, ncont = 3100000000, 25
        d_g, cl = dev.synth_genome(20240601, gsize, ncont, rf)
        dev.build_index_resident(d_g, gsize, cl)
        idepth = dev.index_info()[3]
        genome = dev.read_buffer(2, np.uint8)
        assert len(genome) == gsize and int(cl.sum()) == gsize
        genome.tofile(os.path.join(W, "big.seq"))
        del genome
        with open(os.path.join(W, "big.sdx"), "w") as f:
            f.write("%d\n" % ncont)
            for i, c in enumerate(cl):
                f.write("%d\tchr%d\n" % (int(c) - 15, i + 1))
            f.write("%d\n" % idepth)
        dev.synth_reads(20240602, n, 150, paired=True)
        r1, l1, r2, l2 = dev.staged_reads()
        dev.close()
        del dev
        print("genome (%.2f Gbp, %d contigs, %.0f %% repeat tiles), index and %d read pairs made on the device in %.1f s" % (gsize / 1e9, ncont, 100 * rf, n, time.time() - t0))
        t0 = time.time()
        for k, (r, l) in ((1, (r1, l1)), (2, (r2, l2))):
            with open(os.path.join(W, "plain_%d_.fastq" % k), "wb") as f:
                for a in range(0, n, 250000):
                    f.write(fastq_bytes(r[a:a + 250000], l[a:a + 250000], a))
        # 8 gz pairs, qualities of real entropy
        per = n // 8
        jobs = []
        rng = np.random.default_rng(5)
        for fno in range(8):
            sl = slice(fno * per, (fno + 1) * per)
            for k, (r, l) in ((1, (r1, l1)), (2, (r2, l2))):
                jobs.append((os.path.join(W, "q%d_%d_.fastq.gz" % (fno, k)), fastq_bytes(r[sl], l[sl], fno * per, rng)))
        with ProcessPoolExecutor(8) as ex:
            sizes = list(ex.map(gz_write, jobs))
        with open(os.path.join(W, "a1.txt"), "w") as f1, open(os.path.join(W, "a2.txt"), "w") as f2:
            for fno in range(8):
                f1.write(os.path.join(W, "q%d_1_.fastq.gz\n" % fno))
                f2.write(os.path.join(W, "q%d_2_.fastq.gz\n" % fno))
        print("fastq written in %.1f s: plain %.2f GB per mate file, gz with qualities of real entropy %.1f MB per mate file (8 pairs of %d)" %
              (time.time() - t0, os.path.getsize(os.path.join(W, "plain_1_.fastq")) / 1e9, sizes[0] / 1e6, per))
        del r1, r2
        # the files `copies` times over (text and gzip members concatenate; a batch's first call pays for allocations and page-locking:
        # a million pairs alone measure that)
        def repeat(path):
            data = open(path, "rb").read()
            with open(path, "wb") as f:
                for _ in range(copies):
                    f.write(data)
        if copies > 1:
            for k in (1, 2):
                repeat(os.path.join(W, "plain_%d_.fastq" % k))
                for fno in range(8):
                    repeat(os.path.join(W, "q%d_%d_.fastq.gz" % (fno, k)))
        exe = os.path.join(ROOT, "pecaller_amd", "pemapper_hip")
        runs = (("one plain pair", ["p", os.path.join(W, "plain_1_.fastq"), os.path.join(W, "plain_2_.fastq")], n * copies),
                ("8 gz pairs, array mode", ["pa", os.path.join(W, "a1.txt"), os.path.join(W, "a2.txt")], 8 * per * copies))
        for name, files, pairs in runs:
            out = os.path.join(W, "out_" + files[0])
            t0 = time.time()
            p = subprocess.run([exe, out, os.path.join(W, "big.sdx")] + files + ["500", "0", "N", "0.85", threads, "2000000000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            wall = time.time() - t0
            log = p.stdout.decode(errors="replace")
            if p.returncode:
                print(log[-2000:])
                raise SystemExit("pemapper_hip failed on " + name)
            lines = [ln.strip() for ln in log.split("\n") if "read and mapped" in ln]
            print("%s (%d pairs): %s" % (name, pairs, lines[-1] if lines else "?"))
            for ln in log.split("\n"):
                if "pemapper_hip:" in ln and "read and mapped" not in ln:
                    print("   " + ln.strip()[:200])
            summ = open(out + ".summary.txt").read() if os.path.exists(out + ".summary.txt") else ""
            tot = [ln for ln in summ.split("\n") if ln.startswith("Total Number of Mapping reads")]
            print("   wall %.1f s end to end = %.2f M reads/s with the 3.1 GB genome file read, the index and its 8 look-up copies built on the device and the output files written (%s, pileup %.2f GB)" %
                  (wall, 2 * pairs / wall / 1e6, tot[0][:70].replace("\t", " ") if tot else "no summary", os.path.getsize(out + ".pileup.gz") / 1e9))
    finally:
        shutil.rmtree(W, ignore_errors=True)


if __name__ == "__main__":
    main()
