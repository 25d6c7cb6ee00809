#!/bin/bash
# A/B of compile-time variants (tools/variants.sh) on a mapper-only bench run: tools/ab_lib.sh <config> <steps> "<lib name or -> [ENV=V ...]" ...
CFG=$1; ST=$2; shift 2
i=0
for setting in "$@"; do
  i=$((i+1))
  set -- $setting
  lib=$1; shift
  ( [ "$lib" != "-" ] && export PEMAP_LIB=$PWD/pecaller_amd/libpemap_hip.$lib.so
    for e in "$@"; do export $e; done
    timeout -k 10 300 python bench.py --config $CFG --steps $ST --warmup 2 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > gpurun_out/r3_abl_$i.log 2> gpurun_out/r3_abl_$i.err )
  python3 -c "
import json;d=json.loads(open('gpurun_out/r3_abl_$i.log').read().strip().splitlines()[-1]);print('$CFG [$setting]', {k:d[k] for k in ['value','ms_per_step','resident_value','resident_ms_per_step']}, d['roofline']['avg_launch_ms'], d['counters_per_step']['big_ends'])"
done
