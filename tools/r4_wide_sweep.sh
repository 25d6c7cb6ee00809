#!/bin/bash
# the caller beyond 128 samples against PECALL_HEAVY_MIN_WIDE (samples with variant reads from which a column's beam search starts first)
for n in "256 1000000" "512 200000"; do
for h in 0 2 3 5 8 13; do
echo -n "HEAVY_MIN_WIDE=$h: "
PECALL_HEAVY_MIN_WIDE=$h PECALL_LIST_STATS=1 timeout -k 10 200 python tools/pecall_wide_time.py $n 2>&1 | tail -3 | cut -c1-170 | tr '\n' ' '
echo
done
done
