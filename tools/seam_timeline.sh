#!/bin/bash
# kernel trace of a short bench run -> idle gaps at the seam and resident: tools/seam_timeline.sh [waves]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/seam_tl; rm -rf $OUT; mkdir -p $OUT
[ -n "$1" ] && export PEMAP_LOOKUP_WAVES=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu --no-secondary --no-pecaller --realistic-steps 0 > $OUT/log.txt 2>&1
cd $ROOT
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_gaps.py $t
tail -1 $OUT/log.txt | cut -c1-200
find $OUT -name "*.csv" -size +8M -delete
