#!/bin/bash
# the step against the seed waves per CU: tools/ab_lookup_waves.sh [config=hg38_150] [steps=12] [waves...=6 7 8]
CFG=${1:-hg38_150}; ST=${2:-12}; shift 2 2>/dev/null
for w in ${@:-6 7 8}; do
  PEMAP_LOOKUP_WAVES=$w timeout -k 10 300 python bench.py --config $CFG --steps $ST --warmup 2 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > gpurun_out/r3_abw_$w.log 2> gpurun_out/r3_abw_$w.err
  python3 -c "
import json;d=json.loads(open('gpurun_out/r3_abw_$w.log').read().strip().splitlines()[-1]);print('$CFG waves $w', {k:d[k] for k in ['value','ms_per_step','resident_value','resident_ms_per_step']}, d['roofline']['avg_launch_ms'], d['roofline']['kernel_ms'])"
done
