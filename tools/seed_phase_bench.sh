#!/bin/bash
# timing probe: seed kernel time when it stops after phase k (1 look-ups, 2 gather, 3 sort, 4 vote count, 0 full)
for k in 1 2 3 4 0; do
  PEMAP_SEED_PHASE=$k timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('phase', $k, r['roofline']['kernel_ms'])"
done
