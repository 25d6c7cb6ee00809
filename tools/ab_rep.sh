#!/bin/bash
# A/B of environment settings on one box: each line of CFGS is a list of VAR=value pairs (use _ for none)
while read -r cfg; do
  [ -z "$cfg" ] && continue
  [ "$cfg" = "_" ] && envs="" || envs="$cfg"
  env $envs timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/ab_rep.tmp 2>&1 || { tail -5 gpurun_out/ab_rep.tmp; exit 1; }
  python3 -c "
import json,sys
l=[x for x in open('gpurun_out/ab_rep.tmp').read().splitlines() if x.startswith('{\"metric\"')][-1]
d=json.loads(l);print('$cfg |', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done <<< "$CFGS"
