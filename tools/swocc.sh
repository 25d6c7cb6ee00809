#!/bin/bash
# experiment: SW kernel time against its resident waves per CU (serial pipeline, stand-alone kernel times)
for w in ${SWW:-4 6 8 10 12 16}; do
  PEMAP_SW_WAVES_PER_CU=$w PEMAP_PIPELINE=2 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/sw.log 2>&1 || { tail -5 gpurun_out/sw.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);k=d['roofline']['kernel_ms'];print('sw waves/CU $w', 'sw_single x2 =', 2*k['sw_single'], 'walk x2 =', 2*k['walk'])"
done
