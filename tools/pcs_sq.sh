#!/bin/bash
# SQ counters of the caller's kernels on BASELINE config 4's columns (resident): instructions per column, busy quad-cycles of the VALUs
# and of the LDS, waiting share.  tools/pcs_sq.sh [columns]  ->  gpurun_out/pcs_sq.json (copy to profiles/r04_pecall_sq.json: bench.py's
# pecaller.roofline reads valu_issue_frac / lds_frac from it, tied to the sha of the device sources)
ROOT=$(pwd)
N=${1:-2000000}
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/pcs_sq_$i -- python3 $ROOT/tools/pecall_kernel_time.py $N > $ROOT/gpurun_out/pcs_sq.log 2>&1 || { tail -3 $ROOT/gpurun_out/pcs_sq.log; exit 1; }
done
cd $ROOT
python3 - $N <<'PY'
import csv, glob, collections, re, json, sys
sys.path.insert(0, '.')
import bench
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob('gpurun_out/pcs_sq_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        n = re.sub(r'^void ', '', r['Kernel_Name']); n = re.sub(r'\(.*$', '', n)
        if n.startswith('pcs_'):
            acc[n][r['Counter_Name']] += float(r['Counter_Value'])
RUNS = 5    # pecall_kernel_time.py runs the columns five times (its warm-up call of 20,000 columns is 1 % of one run)
out = {"columns": int(sys.argv[1]), "runs": RUNS, "sha_pecall": bench.kernel_sources_sha("pecall_"), "unit": "counter value per run of the columns",
       "kernels": {k: {c: round(v / RUNS, 1) for c, v in sorted(d.items())} for k, d in sorted(acc.items())}}
json.dump(out, open('gpurun_out/pcs_sq.json', 'w'), indent=1)
for k, d in out["kernels"].items():
    print(k, {c: round(v / 1e6, 2) for c, v in d.items()}, '(millions per run)')
PY
