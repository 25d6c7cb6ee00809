#!/bin/bash
# experiment: resident walk blocks per CU (fewer walkers keep their direction lines in L2)
for w in ${WBS:-1 2 4 8 16}; do
  PEMAP_WALK_BLOCKS_PER_CU=$w PEMAP_PIPELINE=${PIPE:-2} timeout -k 10 200 python bench.py --steps ${STEPS:-2} --warmup 1 --cpu-seconds 0 > gpurun_out/wk.log 2>&1 || { tail -5 gpurun_out/wk.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/wk.log').read().strip().splitlines()[-1]);k=d['roofline']['kernel_ms'];print('walk blocks/CU $w', d['ms_per_step'], 'walk', k['walk'], 'steps', d['steps'])"
done
