#!/bin/bash
# experiment: CFGS = "walk_blocks_per_cu:pile_blocks_per_cu"
for cfg in ${CFGS}; do
  IFS=: read wb pb <<< "$cfg"
  PEMAP_WALK_BLOCKS_PER_CU=$wb PEMAP_PILE_BLOCKS_PER_CU=$pb PEMAP_PIPELINE=${PIPE:-2} timeout -k 10 200 python bench.py --steps ${STEPS:-2} --warmup 1 --cpu-seconds 0 > gpurun_out/pl.log 2>&1 || { tail -5 gpurun_out/pl.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/pl.log').read().strip().splitlines()[-1]);k=d['roofline']['kernel_ms'];print('walk $wb pile $pb', d['value'], d['ms_per_step'], 'walk+pile', k['walk'], 'steps', d['steps'])"
done
