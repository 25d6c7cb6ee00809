#!/bin/bash
# timing experiment: vote kernel cut short after each phase (results are wrong for probe != 0)
for p in 1 2 3 4 0; do
  PEMAP_VOTE_PROBE=$p PEMAP_PIPELINE=2 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/vp$p.log 2>&1 || exit 1
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/vp$p.log').read().strip().splitlines()[-1]);print('probe',$p,d['ms_per_step'],d['roofline']['kernel_ms'])"
done
