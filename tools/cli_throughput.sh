#!/bin/bash
# wall-clock rate of the host program pemapper_hip on N x 20 k pairs (the golden fixture concatenated: gz members concatenate), from
# gz input and from the same reads as plain text.   tools/cli_throughput.sh [copies=100] [copies for the array-mode run=copies]
set -e
N=${1:-100}
ROOT=$(pwd); W=$(mktemp -d); cd $W
cp $ROOT/tests/golden/g1.sdx .
python3 - <<PY
import gzip, sys
sys.path.insert(0, "$ROOT/tests")
import refio, numpy as np
_, seqs = refio.read_fasta("$ROOT/tests/golden/g1.fa.gz")
gzip.open("g1.seq", "wb", compresslevel=1).write(np.concatenate(seqs).tobytes())
PY
for k in 1 2; do for i in $(seq $N); do cat $ROOT/tests/golden/g1_${k}_.fastq.gz; done > big_${k}_.fastq.gz; gzip -dc big_${k}_.fastq.gz > plain_${k}_.fastq; done
ls -la big_1_.fastq.gz plain_1_.fastq | awk '{print "bytes per mate file:", $5, $9}'
cat plain_1_.fastq plain_2_.fastq > /dev/null     # (page cache: the plain files were written a moment ago)
for kind in big plain plain; do
  sfx=fastq.gz; [ $kind = plain ] && sfx=fastq
  T0=$(date +%s.%N)
  $ROOT/pecaller_amd/pemapper_hip out_$kind g1.sdx p ${kind}_1_.$sfx ${kind}_2_.$sfx 500 0 N 0.85 24 2000000000 > log_$kind.txt 2>&1 || { tail -3 log_$kind.txt; exit 1; }
  T1=$(date +%s.%N)
  grep "read and mapped" log_$kind.txt
  python3 -c "
import os; n=os.path.getsize('${kind}_1_.$sfx.mfile')//4; w=$T1-$T0; print('$kind input: pairs', n, 'wall %.2f s' % w, '= %.2f M reads/s end to end (start-up, index build of the 5 Mbp fixture genome and output included)' % (2*n/w/1e6))"
done
# ---- array mode: the same reads as 8 gz file pairs into one output set (the reference's usual run, pemapper.c:307-348), the files
#      read and mapped side by side by the program's file workers
M=$(((${2:-$N}) / 8)); [ $M -lt 1 ] && M=1     # (second argument: copies for the array-mode run, all 8 files together)
for f in $(seq 8); do for k in 1 2; do for i in $(seq $M); do cat $ROOT/tests/golden/g1_${k}_.fastq.gz; done > part${f}_${k}_.fastq.gz; done; echo $W/part${f}_1_.fastq.gz >> a1.txt; echo $W/part${f}_2_.fastq.gz >> a2.txt; done
T0=$(date +%s.%N)
$ROOT/pecaller_amd/pemapper_hip out_arr g1.sdx pa a1.txt a2.txt 500 0 N 0.85 24 2000000000 > log_arr.txt 2>&1 || { tail -3 log_arr.txt; exit 1; }
T1=$(date +%s.%N)
grep "files by" log_arr.txt
python3 -c "
import os; n=sum(os.path.getsize('part%d_1_.fastq.gz.mfile' % f)//4 for f in range(1,9)); w=$T1-$T0; print('array mode, 8 gz file pairs: pairs', n, 'wall %.2f s' % w, '= %.2f M reads/s end to end' % (2*n/w/1e6))"
# ---- the same array-mode run on reads whose quality lines have the entropy of real ones (the fixture's are one letter repeated, which
#      inflates three times as fast per byte): same reads, same mapping, 2.3:1 instead of 6:1 compression
python3 - <<PY
import gzip, numpy as np
rng = np.random.default_rng(5)
for k in (1, 2):
    lines = gzip.open("$ROOT/tests/golden/g1_%d_.fastq.gz" % k, "rb").read().split(b"\n")
    for i in range(3, len(lines), 4):
        n = len(lines[i])
        q = np.clip(rng.normal(36, 4, n).astype(int), 2, 40)
        q[rng.random(n) < 0.05] = rng.integers(2, 20)
        lines[i] = (q + 33).astype(np.uint8).tobytes()
    open("q_%d_.fastq.gz" % k, "wb").write(gzip.compress(b"\n".join(lines), 6))
PY
rm -f a1.txt a2.txt
for f in $(seq 8); do for k in 1 2; do for i in $(seq $M); do cat q_${k}_.fastq.gz; done > qpart${f}_${k}_.fastq.gz; done; echo $W/qpart${f}_1_.fastq.gz >> a1.txt; echo $W/qpart${f}_2_.fastq.gz >> a2.txt; done
ls -la qpart1_1_.fastq.gz | awk '{print "bytes per mate file, qualities of high entropy:", $5}'
$ROOT/pecaller_amd/pemapper_hip out_q g1.sdx pa a1.txt a2.txt 500 0 N 0.85 24 2000000000 > log_q.txt 2>&1 || { tail -3 log_q.txt; exit 1; }
grep "files by" log_q.txt
cmp out_q.pileup.gz out_arr.pileup.gz > /dev/null 2>&1 && echo "pileup of the run with the other quality lines: identical bytes"
cmp out_big.pileup.gz out_plain.pileup.gz > /dev/null 2>&1 && echo "pileups of the two runs: identical bytes" || { gzip -dc out_big.pileup.gz | md5sum; gzip -dc out_plain.pileup.gz | md5sum; }
cd /; rm -rf $W
