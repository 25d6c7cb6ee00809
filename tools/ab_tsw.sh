#!/bin/bash
# like abn.sh for BASELINE config 3's shape: tools/ab_tsw.sh repeats "ENV_A" "ENV_B" ...
reps=$1; shift
for i in $(seq $reps); do
  for e in "$@"; do
    env $e timeout -k 10 300 python bench.py --config tsw250 --steps 4 --warmup 1 --no-cpu --no-pecaller > gpurun_out/ab_tsw.log 2>&1 || { tail -5 gpurun_out/ab_tsw.log; exit 1; }
    python3 -c "
import json,sys;d=json.loads([l for l in open('gpurun_out/ab_tsw.log') if l.startswith('{')][-1]);k=d['roofline']['kernel_ms'];print('tsw $e', '| seam', d['value'], d['ms_per_step'], 'resident', d['resident_value'], d['resident_ms_per_step'], 'lookup', k['lookup'], 'sw', k['sw_single'], 'walk', k['walk'])"
  done
done
