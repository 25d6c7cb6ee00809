#!/bin/bash
# experiment: CFGS = list of "lookup_waves:sw_waves:vote_waves:walk_blocks" per CU
for cfg in ${CFGS}; do
  IFS=: read lw sw vw wb <<< "$cfg"
  PEMAP_WALK_BLOCKS_PER_CU=$wb PEMAP_VOTE_WAVES=$vw PEMAP_SW_WAVES_PER_CU=$sw PEMAP_LOOKUP_WAVES=$lw timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/g2.log 2>&1 || { tail -5 gpurun_out/g2.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/g2.log').read().strip().splitlines()[-1]);print('lw $lw sw $sw vw $vw wb $wb',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
done
