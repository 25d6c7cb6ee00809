#!/bin/bash
# The scaling curve as the driver launches it: N = 1, 2, 4, 8 ranks of bench.py on ONE node, one rank per GPU over RCCL
# (backend nccl), back to back.  Run on an 8-GPU MI355X node from the repo root:
#     tools/scale.sh [steps] [warmup] [port]        ->  gpurun_out/scale_N<k>.json (rank 0's JSON line), gpurun_out/scale.txt
# Nothing here is edited per node: bench.py reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.
# Per-N JSON fields to read: value (whole-job M reads/s), ms_per_step (max over ranks), n_gpus, scaling
# "weak", ranks[] (per rank: ms_per_step, lookup_replicas, hbm_free_after_setup_GiB, setup_s), timings.index_bcast_s (RCCL broadcast of the 32 GB the
# reference's arrays take), timings.pileup_reduce_s and timings.pileup_reduce_checked_total (the grand total of the summed counters, asserted equal to the sum of the ranks' totals).
# A rank without room for the look-up replicas ends the run (exit 3) unless --allow-fallback is given.
set -e
STEPS=${1:-20}; WARM=${2:-5}; PORT=${3:-29517}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
: > gpurun_out/scale.txt
for N in 1 2 4 8; do
  if [ $N -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-secondary --no-pecaller --realistic-steps 0 > gpurun_out/scale_N$N.log 2> gpurun_out/scale_N$N.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps $STEPS --warmup $WARM > gpurun_out/scale_N$N.log 2> gpurun_out/scale_N$N.err
  fi
  grep '^{' gpurun_out/scale_N$N.log | tail -1 > gpurun_out/scale_N$N.json
  python3 - $N <<'PY' | tee -a gpurun_out/scale.txt
import json, sys
n = sys.argv[1]
d = json.loads(open("gpurun_out/scale_N%s.json" % n).read())
print("N=%s value %.2f M reads/s  ms_per_step %.3f  timings %s" % (n, d["value"], d["ms_per_step"], {k: v for k, v in d.get("timings", {}).items() if k in ("index_bcast_s", "pileup_reduce_s")}))
PY
done
