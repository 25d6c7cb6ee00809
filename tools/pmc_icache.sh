#!/bin/bash
# instruction-cache counters per kernel of a short mapper-only bench run:  tools/pmc_icache.sh [ENV=VAL ...]
for e in "$@"; do export "$e"; done
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_icache; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --no-pecaller > $OUT/a.log 2>&1 || { tail -3 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --no-pecaller > $OUT/b.log 2>&1 || { tail -3 $OUT/b.log; exit 1; }
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        if k.startswith("pm_"):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items(), key=lambda x: -x[1].get("SQC_ICACHE_REQ", 0)):
    req, miss = d.get("SQC_ICACHE_REQ", 0), d.get("SQC_ICACHE_MISSES", 0)
    print("%-28s icache req %12.0f misses %12.0f (%.1f %%)  ifetch %12.0f  wave-cycles %14.0f wait_inst_any %14.0f" % (k[:28], req, miss, 100 * miss / max(req, 1), d.get("SQ_IFETCH", 0), d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_WAIT_INST_ANY", 0)))
PY
