#!/bin/bash
for p in 0 1 -1 0 1 -1; do
  PEMAP_MEM_PRIO=$p timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/pr.log 2>&1 || { tail -5 gpurun_out/pr.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/pr.log').read().strip().splitlines()[-1]);print('mem prio $p',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
done
