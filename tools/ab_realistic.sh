#!/bin/bash
# A/B of environment settings on the `realistic` genome (2 % repeat tiles) as the main workload: tools/ab_realistic.sh <steps> "<ENV=V ...>" ...   ("-" = defaults)
ST=$1; shift
i=0
for setting in "$@"; do
  i=$((i+1))
  ( [ "$setting" != "-" ] && export $setting
    timeout -k 10 300 python bench.py --repeat-frac 0.02 --seed 20240702 --steps $ST --warmup 2 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > gpurun_out/r3_abr_$i.log 2> gpurun_out/r3_abr_$i.err )
  python3 -c "
import json;d=json.loads(open('gpurun_out/r3_abr_$i.log').read().strip().splitlines()[-1]);print('realistic [$setting]', {k:d[k] for k in ['value','ms_per_step','resident_value','resident_ms_per_step','mapped_frac']}, d['roofline']['avg_launch_ms'], d['roofline']['kernel_ms'])"
done
