#!/bin/bash
# several environments on the same box, round robin: tools/abn.sh repeats "ENV_A" "ENV_B" ...   (X=1 style; several assignments in one string are fine)
reps=$1; shift
for i in $(seq $reps); do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu --no-secondary --no-pecaller > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1]);k=d['roofline']['kernel_ms'];print('$e', '| seam', d['value'], d['ms_per_step'], 'resident', d['resident_value'], d['resident_ms_per_step'], 'lookup', k['lookup'], 'vote', k['vote'], 'sw', k['sw_single'], 'walk', k['walk'])"
  done
done
