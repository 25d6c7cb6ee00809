#!/bin/bash
# A/B of environment settings on a mapper-only bench run: tools/ab_env.sh <config> <steps> "<ENV=V ENV=V>" "<...>" ...   ("-" = defaults)
CFG=$1; ST=$2; shift 2
i=0
for setting in "$@"; do
  i=$((i+1))
  ( [ "$setting" != "-" ] && export $setting
    timeout -k 10 300 python bench.py --config $CFG --steps $ST --warmup 2 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > gpurun_out/r3_abe_$i.log 2> gpurun_out/r3_abe_$i.err )
  python3 -c "
import json;d=json.loads(open('gpurun_out/r3_abe_$i.log').read().strip().splitlines()[-1]);print('$CFG [$setting]', {k:d[k] for k in ['value','ms_per_step','resident_value','resident_ms_per_step']}, d['roofline']['avg_launch_ms'])"
done
