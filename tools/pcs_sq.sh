#!/bin/bash
# SQ counters of the caller's kernels on BASELINE config 4's columns (resident): instructions per column, waiting share
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/pcs_sq_$i -- python3 $ROOT/tools/pecall_kernel_time.py ${1:-1000000} > $ROOT/gpurun_out/pcs_sq.log 2>&1 || { tail -3 $ROOT/gpurun_out/pcs_sq.log; exit 1; }
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob('gpurun_out/pcs_sq_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        n = re.sub(r'^void ', '', r['Kernel_Name']); n = re.sub(r'\(.*$', '', n)
        if n.startswith('pcs_'):
            acc[n][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in sorted(acc.items()):
    print(k, {c: round(v / 5 / 1e6, 2) for c, v in sorted(d.items())}, '(millions per run of the columns; 5 runs profiled)')
PY
