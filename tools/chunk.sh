#!/bin/bash
# experiment: pairs per chunk of the two-stream pipeline
for cp in ${CPS:-65536 131072 196608 262144 524288}; do
  PEMAP_CHUNK_PAIRS=$cp PEMAP_DIR_BUDGET_GB=120 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 > gpurun_out/c.log 2>&1 || { tail -5 gpurun_out/c.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/c.log').read().strip().splitlines()[-1]);print('chunk pairs $cp',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['counters_per_step']['chunks'])"
done
