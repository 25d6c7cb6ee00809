#!/bin/bash
# experiment: vote placement (0 ALU stream, 1 memory stream, 2 own stream) x look-up waves per CU
for cfg in ${CFGS}; do
  IFS=: read vm lw <<< "$cfg"
  PEMAP_VOTE_ON_MEM=$vm PEMAP_LOOKUP_WAVES=$lw timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-seconds ${CPUS:-0} > gpurun_out/vs.log 2>&1 || { tail -5 gpurun_out/vs.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/vs.log').read().strip().splitlines()[-1]);print('vote stream $vm lw $lw',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'], d['cpu_baseline'].get('gpu_vs_cpu_mismatches') if d['cpu_baseline'] else '')"
done
