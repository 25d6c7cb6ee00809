#!/bin/bash
# instruction mix of every kernel of the final pipeline: SQ counters per launch, two --pmc passes; gpurun_out/sqmix_<pass>/
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/sqmix_$i -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > $ROOT/gpurun_out/sqmix.log 2>&1 || { tail -3 $ROOT/gpurun_out/sqmix.log; exit 1; }
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, re, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('gpurun_out/sqmix_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        n = re.sub(r'^void ', '', r['Kernel_Name']); n = re.sub(r'\(.*$', '', n)
        if n.startswith('pm_'):
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in acc.items():
    out[k] = {c: {"launches": len(v), "sum_per_step_M": round(sum(v) / 4 / 1e6, 2)} for c, v in sorted(d.items())}   # 4 steps profiled (warm-up + timed, at the seam and resident)
json.dump(out, open('gpurun_out/sqmix.json', 'w'), indent=1)
for k, d in sorted(out.items()):
    print(k, {c: v["sum_per_step_M"] for c, v in d.items()})
PY
