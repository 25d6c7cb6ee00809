#!/bin/bash
# experiment: CFGS = list of "lookup_waves_per_cu:batch:sw_waves_per_cu:vote_on_mem:vote_waves_per_cu"
for cfg in ${CFGS}; do
  IFS=: read lw lb sw vm vw <<< "$cfg"
  PEMAP_VOTE_WAVES=${vw:-12} PEMAP_VOTE_ON_MEM=${vm:-0} PEMAP_SW_WAVES_PER_CU=$sw PEMAP_LOOKUP_WAVES=$lw PEMAP_LOOKUP_BATCH=$lb PEMAP_PIPELINE=${PIPE:-1} timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-seconds ${CPUS:-0} > gpurun_out/g.log 2>&1 || { tail -5 gpurun_out/g.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/g.log').read().strip().splitlines()[-1]);print('lw $lw batch $lb sw $sw votemem $vm vw $vw',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
done
