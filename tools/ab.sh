#!/bin/bash
# A/B of two environments on the same box, alternating: tools/ab.sh "ENV_A" "ENV_B" [repeats]
# (X=1 style assignments; use "X=" for the default).  Prints the seam rate, the resident rate and the per-step kernel times.
for i in $(seq ${3:-3}); do
  for e in "$1" "$2"; do
    env $e timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu --no-secondary --no-pecaller > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    python3 -c "
import json,sys;d=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1]);k=d['roofline']['kernel_ms'];print('$e', 'seam', d['value'], d['ms_per_step'], 'resident', d['resident_value'], d['resident_ms_per_step'], 'lookup', k['lookup'], 'vote', k['vote'], 'sw', k['sw_single'], 'walk', k['walk'])"
  done
done
