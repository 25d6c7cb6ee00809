#!/bin/bash
# stand-alone kernel times (PEMAP_PIPELINE=2: the same kernels on one stream) of the replica look-up kernels
# CFGS = list of "version:waves_per_cu"
for cfg in ${CFGS:-1:6 1:12 2:3 2:6 2:12}; do
  v=${cfg%%:*}; lw=${cfg##*:}
  PEMAP_LOOKUP_V=$v PEMAP_LOOKUP_WAVES=$lw PEMAP_PIPELINE=${PIPE:-2} timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu > gpurun_out/rs.log 2>&1 || { tail -5 gpurun_out/rs.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/rs.log').read().strip().splitlines()[-1]);print('v $v lw $lw', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
