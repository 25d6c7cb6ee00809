#!/bin/bash
# experiment: CFGS = list of "chunk_pairs:lookup_waves:lookup_batch"
for cfg in ${CFGS}; do
  IFS=: read cp lw lb <<< "$cfg"
  PEMAP_CHUNK_PAIRS=$cp PEMAP_LOOKUP_WAVES=$lw PEMAP_LOOKUP_BATCH=$lb timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/g3.log 2>&1 || { tail -5 gpurun_out/g3.log; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/g3.log').read().strip().splitlines()[-1]);print('chunk $cp lw $lw batch $lb',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
done
