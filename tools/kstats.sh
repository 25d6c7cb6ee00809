#!/bin/bash
# per-kernel time of a short mapper-only bench run:  tools/kstats.sh <tag> [ENV=VAL ...]   ->  gpurun_out/kstats_<tag>.txt
# (env assignments are exported here, before the profiler starts: no env hop after `--`)
TAG=$1; shift
for e in "$@"; do export "$e"; done
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu --no-secondary --no-pecaller ${KSTATS_ARGS} > $OUT/log.txt 2>&1
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/kstats_$TAG.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s total_ms %10.3f avg_us %10.1f pct %s" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
cat gpurun_out/kstats_$TAG.txt
tail -1 $OUT/log.txt | cut -c1-300
find $OUT -name "*.csv" -size +2M -delete
