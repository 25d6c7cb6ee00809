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
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $OUT/timeline.txt <<'PY'
# the kernels of the last run, in start order: offset from the run's first kernel, duration, queue
import csv,sys
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:34],r.get('Queue_Id','?')) for r in csv.DictReader(open(sys.argv[1])) if 'pcs_' in r['Kernel_Name']]
rows.sort()
# runs are separated by gaps of more than 5 ms
runs=[[rows[0]]]
for a,b in zip(rows,rows[1:]):
    if b[0]-max(x[1] for x in runs[-1])>5e6: runs.append([])
    runs[-1].append(b)
last=runs[-1]; t0=last[0][0]
print("last run: %d kernels, %.2f ms from first start to last end" % (len(last),(max(x[1] for x in last)-t0)/1e6))
for s,e,n,q in last:
    print("%8.3f  +%7.3f ms  q%-3s %s" % ((s-t0)/1e6,(e-s)/1e6,q,n))
PY
head -40 $OUT/timeline.txt
find $OUT -name "*.csv" -size +2M -delete
