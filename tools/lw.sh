#!/bin/bash
# experiment: persistent wave-per-end look-up kernel; CFGS = list of "waves_per_cu:batch"
for cfg in ${CFGS:-4:8 4:16 6:4 6:8 8:8}; do
  lw=${cfg%%:*}; lb=${cfg##*:}
  PEMAP_LOOKUP_WAVES=$lw PEMAP_LOOKUP_BATCH=$lb PEMAP_PIPELINE=${PIPE:-1} timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-seconds ${CPUS:-0} > gpurun_out/lw.log 2>&1 || { tail -5 gpurun_out/lw.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/lw.log').read().strip().splitlines()[-1]);print('lw $lw batch $lb',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['cpu_baseline'].get('gpu_vs_cpu_mismatches'))"
done
