#!/bin/bash
# the caller's resident rate against the heavy-column threshold and the waves of the early beam search: tools/r4_pc_sweep.sh "t..." "w..."
for t in ${1:-4 6 8 12}; do for w in ${2:-2 3}; do
  echo "PECALL_HEAVY_MIN=$t PECALL_HEAVY_WAVES=$w: $(PECALL_LIST_STATS=1 PECALL_HEAVY_MIN=$t PECALL_HEAVY_WAVES=$w python tools/pecall_kernel_time.py 2000000 2>&1 | tail -3 | cut -c1-150 | tr '\n' ' ')"
done; done
