#!/bin/bash
# rocprofv3 passes behind profiles/: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc runs.
# Run on the GPU box from the repo root:  tools/profile.sh <tag> [config]   (config: hg38_150 (default) or tsw250)
#   -> gpurun_out/prof_<tag>/{stats.csv, pmc.json, ...}; copy pmc.json to profiles/r04_bench_pmc_<config>.json
# (the program itself follows `--`: no env / bash -c hop between the profiler and python3)
set -e
TAG=${1:-r03}
CFG=${2:-hg38_150}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --no-cpu --no-secondary --no-pecaller --realistic-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 4 --warmup 1 $ARGS > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/fetch.log 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/write.log 2>&1
echo "WRITE_SIZE pass done"
cd $ROOT
python3 tools/pmc_summary.py $OUT 6    # (2 timed + 1 warm-up) steps at the seam + the same resident
tail -1 $OUT/stats.log | cut -c1-400
# keep the merge small: the raw per-dispatch CSVs stay on the box
find $OUT -name "*.csv" -size +8M -delete
