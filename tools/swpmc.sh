#!/bin/bash
# SQ counters of the SW kernel with 8 and 12 resident waves per CU (serial pipeline); gpurun_out/swpmc_<w>_<pass>/
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in 8 12; do
  i=0
  for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    PEMAP_SW_WAVES_PER_CU=$w PEMAP_PIPELINE=2 rocprofv3 --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/swpmc_${w}_$i -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $ROOT/gpurun_out/swpmc.log 2>&1 || { tail -3 $ROOT/gpurun_out/swpmc.log; exit 1; }
  done
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for w in (8, 12):
    acc = collections.defaultdict(list)
    for fn in glob.glob('gpurun_out/swpmc_%d_*/**/*counter_collection.csv' % w, recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'pm_sw_kernel<19, true>' in r['Kernel_Name'] and int(r['Grid_Size']) > 0:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('sw waves/CU', w, {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())}, '(millions, mean per launch)')
PY
