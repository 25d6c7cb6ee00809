#!/bin/bash
# experiment: CU-masked look-up stream, with and without the LDS pad that limits its occupancy
for cfg in "0 20" "8 0" "12 0" "16 0" "20 0" "12 20" "16 20"; do
  set -- $cfg
  PEMAP_MEM_CUS=$1 PEMAP_LOOKUP_LDS_PAD_KB=$2 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 > gpurun_out/cm.log 2>&1 || { tail -5 gpurun_out/cm.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/cm.log').read().strip().splitlines()[-1]);print('cus $1 pad $2',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
done
