#!/bin/bash
# memory-side counters of the fused seed kernel (alone on the GPU: PEMAP_PIPELINE=2) and of the random-line microbenchmark, for
# comparison: address translation, L1 -> L2 request latency, L2 -> fabric requests.   tools/pmc_mem.sh <tag>
# (one --pmc pass per counter group; the program itself follows `--`)
TAG=${1:-mem}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export PEMAP_PIPELINE=2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_MISS_sum" \
           "TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/bench$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --no-pecaller > $OUT/bench$i.log 2>&1 || { tail -3 $OUT/bench$i.log; exit 1; }
  rocprofv3 --pmc $set --output-format csv -d $OUT/micro$i -- $ROOT/tools/micro/line_gather > $OUT/micro$i.log 2>&1 || { tail -3 $OUT/micro$i.log; exit 1; }
  echo "pass $i done"
done
cd $ROOT
python3 - $OUT > gpurun_out/pmc_$TAG.txt <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
for kind in ("bench", "micro"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for fn in glob.glob(out + "/%s*/**/*counter_collection.csv" % kind, recursive=True):
        seen = set()
        for r in csv.DictReader(open(fn)):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (fn, k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
        for (f, k, d) in seen:
            cnt[(f, k)] += 1
    for k in sorted(agg, key=lambda k: -agg[k].get("TCP_TCC_READ_REQ_sum", 0)):
        if not (k.startswith("void pm_seed4") or k.startswith("pm_") or "gather" in k):
            continue
        n = max(v for (f, kk), v in cnt.items() if kk == k)
        print(kind, k[:50], "dispatches/pass", n)
        for c, v in sorted(agg[k].items()):
            print("    %-34s %16.0f   per dispatch %14.0f" % (c, v, v / n))
PY
cat gpurun_out/pmc_$TAG.txt
find $OUT -name "*.csv" -size +4M -delete
