#!/bin/bash
# the GPU path against the CPU oracle at other seeds, error rates and a 125-base read length (coordinates and classes of ~800 K pairs
# each):  tools/other_rates.sh  ->  gpurun_out/other_rates.txt
out=gpurun_out/other_rates.txt; : > $out
run () {
  timeout -k 10 500 python bench.py --steps 3 --warmup 1 --cpu-seconds 25 --no-secondary --no-pecaller "$@" > gpurun_out/other_rates.log 2>&1 || { tail -5 gpurun_out/other_rates.log; exit 1; }
  python3 - "$*" >> $out <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/other_rates.log') if l.startswith('{')][-1])
c = d['cpu_baseline']; s = d['counters_per_step']
print(sys.argv[1], '|', d['resident_value'], 'M reads/s resident |', 'mismatches', c['gpu_vs_cpu_mismatches'], 'of', c['compared_pairs'], 'pairs | per step: gapless', s['gapless'], 'banded', s['banded'], 'of', s['sw_score'], 'problems, full DP cells', s['cells_dirs'], 'big ends', s['big_ends'])
PY
  tail -1 $out
}
run --seed 1 --sub-rate 0.02 --indel-rate 0.002
run --seed 2 --sub-rate 0.005 --indel-rate 0.001
run --seed 3 --read-len 125 --sub-rate 0.015 --indel-rate 0.003
run --seed 4 --read-len 100 --sub-rate 0.03 --indel-rate 0.0005
