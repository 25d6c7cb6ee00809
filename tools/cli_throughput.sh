#!/bin/bash
# wall-clock rate of the host program on 1 M pairs (the 20 k-pair fixture concatenated 50 times: gz members concatenate)
set -e
ROOT=$(pwd); W=$(mktemp -d); cd $W
cp $ROOT/tests/golden/g1.sdx .
python3 - <<PY
import gzip, sys
sys.path.insert(0, "$ROOT/tests")
import refio, numpy as np
_, seqs = refio.read_fasta("$ROOT/tests/golden/g1.fa.gz")
gzip.open("g1.seq", "wb", compresslevel=1).write(np.concatenate(seqs).tobytes())
PY
for k in 1 2; do for i in $(seq 50); do cat $ROOT/tests/golden/g1_${k}_.fastq.gz; done > big_${k}_.fastq.gz; done
ls -la big_1_.fastq.gz | awk '{print "fastq.gz bytes per mate file:", $5}'
T0=$(date +%s.%N)
$ROOT/pecaller_amd/pemapper_hip out g1.sdx p big_1_.fastq.gz big_2_.fastq.gz 500 0 N 0.85 24 200000000 > log.txt 2>&1 || { tail -3 log.txt; exit 1; }
T1=$(date +%s.%N)
python3 -c "print('wall %.2f s' % ($T1 - $T0))"
python3 -c "
import os; n=os.path.getsize('big_1_.fastq.gz.mfile')//4; print('pairs', n)"
cd /; rm -rf $W
