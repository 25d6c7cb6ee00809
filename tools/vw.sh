#!/bin/bash
# experiment: wave-per-end vote kernel, waves per CU (0 = workgroup kernel); PIPE=2 gives stand-alone kernel times (x steps)
for vw in ${VWS:-0 8 12 14 16}; do
  PEMAP_VOTE_WAVES=$vw PEMAP_PIPELINE=${PIPE:-1} timeout -k 10 200 python bench.py --steps ${STEPS:-6} --warmup 2 --cpu-seconds ${CPUS:-0} > gpurun_out/vw.log 2>&1 || { tail -5 gpurun_out/vw.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/vw.log').read().strip().splitlines()[-1]);print('vote waves $vw',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['cpu_baseline'].get('gpu_vs_cpu_mismatches') if d['cpu_baseline'] else None, d['counters_per_step'].get('big_ends'))"
done
