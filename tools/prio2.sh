#!/bin/bash
# experiment: wave issue priorities, CFGS = "sw:lookup:vote:lookup_waves"
for cfg in ${CFGS}; do
  IFS=: read a b c lw <<< "$cfg"
  PEMAP_SW_PRIO=$a PEMAP_LOOKUP_PRIO=$b PEMAP_VOTE_PRIO=$c PEMAP_LOOKUP_WAVES=${lw:-7} timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/p2.log 2>&1 || { tail -5 gpurun_out/p2.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/p2.log').read().strip().splitlines()[-1]);print('prio sw $a lookup $b vote $c lw $lw',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
done
