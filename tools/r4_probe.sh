#!/bin/bash
# phase probes of pm_seed4_kernel in the pipeline and alone (PEMAP_PIPELINE=2), then the SQ instruction mix (tools/sqmix.sh)
for mode in 1 2; do
PEMAP_PIPELINE=$mode PEMAP_LIB=$PWD/pecaller_amd/libpemap_hip.probes.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > gpurun_out/probe_m$mode.log 2> gpurun_out/probe_m$mode.err || exit 1
echo "pipeline $mode: $(grep pm_s4_probe gpurun_out/probe_m$mode.err | tail -2 | cut -c1-600)"
grep '^{' gpurun_out/probe_m$mode.log | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('   kernel ms/step', d['roofline']['kernel_ms'], 'step', d['resident_ms_per_step'])"
done
tools/sqmix.sh
