#!/bin/bash
# SQ counters of the per-site caller kernel on the config-5 shaped columns
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/pcspmc_$i -- python3 $ROOT/tools/pecall_bench.py --sites 200000 --cpu-sites 100 > $ROOT/gpurun_out/pcspmc.log 2>&1 || { tail -3 $ROOT/gpurun_out/pcspmc.log; exit 1; }
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob('gpurun_out/pcspmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'pcs_call_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) > 64 * 2000:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(max(v) / 1e6, 2) for k, v in sorted(acc.items())}, '(millions, the 200 k-site launch)')
PY
