#!/bin/bash
# several environments on the same box, one bench run each, in the order given (repeat an entry to see the run-to-run spread):
#   tools/abm.sh "X=1 Y=2" "X=" "PEMAP_LIB=$PWD/pecaller_amd/libpemap_hip.variant.so" ...
# Prints the seam rate, the resident rate, the per-step kernel times and the big-end count; stderr of every run goes to gpurun_out/abm.err
: > gpurun_out/abm.err
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --steps ${ABM_STEPS:-8} --warmup 2 --no-cpu --no-secondary --no-pecaller > gpurun_out/abm.log 2>> gpurun_out/abm.err || { tail -5 gpurun_out/abm.log gpurun_out/abm.err; exit 1; }
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/abm.log').read().strip().splitlines()[-1]);r=d['roofline'];k=r['kernel_ms'];c=d['counters_per_step'];print('$e', '| seam', d['value'], d['ms_per_step'], '| resident', d['resident_value'], d['resident_ms_per_step'], '| seed/launch', r['avg_launch_ms'], 'frac', r['frac'], '| seed', k['seed'], 'lookup', k['lookup'], 'sw', k['sw_single'], 'walk', k['walk'], '| big', c['big_ends'])"
done
