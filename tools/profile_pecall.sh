#!/bin/bash
# rocprofv3 passes of the per-site caller on BASELINE config 4's columns (resident runs of tools/pecall_kernel_time.py): kernel stats,
# then FETCH_SIZE and WRITE_SIZE in separate --pmc runs.  tools/profile_pecall.sh <tag> [columns=2000000]
#   -> gpurun_out/prof_pecall_<tag>/{stats.csv, pmc.json, timeline.txt}; copy pmc.json to profiles/r04_pecall_pmc.json
set -e
TAG=${1:-r03}
N=${2:-2000000}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_pecall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/pecall_kernel_time.py $N > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/pecall_kernel_time.py $N > $OUT/fetch.log 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/tools/pecall_kernel_time.py $N > $OUT/write.log 2>&1
echo "WRITE_SIZE pass done"
cd $ROOT
python3 tools/pmc_summary.py $OUT 5    # 5 resident runs of the columns (after a 20000-column warm-up call)
python3 - $OUT $N <<'PY'
import json, sys
d, n = sys.argv[1], int(sys.argv[2])
pm = json.load(open(d + "/pmc.json"))
pm["columns"] = n
json.dump(pm, open(d + "/pmc.json", "w"), indent=1)
tot = 0.0
for cn, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    for k, v in sorted(pm[cn].items()):
        if k.startswith("pcs_"):
            b = v["mean_KB_per_launch"] * v["launches"] * 1024.0 / 5
            tot += factor * b
            print("%-12s %-28s %4d launches  %10.1f MB per run of %d columns (as counted)" % (cn, k[:28], v["launches"], b / 1e6, n))
print("HBM bytes per column: %.0f with FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of a coalesced stream at 64 bytes: MI355X_MICROARCH.md; the columns' rows alone are 770 bytes); SURVEY's algorithmic figure: 7936" % (tot / n))
PY
grep "kernel ms" $OUT/stats.log
find $OUT -name "*.csv" -size +8M -delete
