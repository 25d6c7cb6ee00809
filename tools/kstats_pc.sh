#!/bin/bash
# per-kernel times of the per-site caller on resident columns: tools/kstats_pc.sh <sites>
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kstats_pc; mkdir -p $OUT
export PECALL_LIST_STATS=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/pecall_kernel_time.py ${1:-2000000} > $OUT/log.txt 2>&1
cd $ROOT
grep "pecall\]\|kernel ms" $OUT/log.txt | tail -3
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-50s calls %5s total_ms %10.3f avg_ms %10.3f" % (r['Name'][:50], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
PY
find $OUT -name "*.csv" -size +2M -delete
