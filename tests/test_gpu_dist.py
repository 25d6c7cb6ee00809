"""The multi-GPU path on real device objects, rehearsed with two ranks on the one GPU of the box (gloo; the look-up replicas
off so that two index replicas fit): what a node runs with RCCL, minus the transport."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
import fixtures

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(args, tmp_path, port, timeout=900):
    env = dict(os.environ, PEMAP_REPLICAS="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    # a child process (never an exec of this one, which has touched the GPU)
    return subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)


def test_two_device_ranks_equal_one_process(tmp_path):
    """index_alloc -> broadcast into device buffers -> index_commit -> map disjoint shards on the device -> reduce_pileup on the
    device counters: coordinates in read order, classes, pileup (with the u16 wrap), insertions and summary must be the
    reference's for the whole read set."""
    r = _launch([os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path)], tmp_path, 29700 + os.getpid() % 200)
    assert r.returncode == 0 and b"rank 0 ok" in r.stdout and b"rank 1 ok" in r.stdout, r.stdout[-3000:].decode(errors="replace")
    z = np.load(tmp_path / "dist_result.npz")
    assert np.array_equal(z["m1"], fixtures.golden_m("r150", 1)) and np.array_equal(z["m2"], fixtures.golden_m("r150", 2))
    counts = z["counts"].copy()
    # take the two ranks' 40,000 out again, modulo 2^16: the worker bumped the low halves of the first 600 words of plane 0,
    # i.e. the counters of positions 0, 2, .. 1198 in the column of the reference base there (PmPile's rotation)
    rows = np.arange(0, 1200, 2)
    col = fixtures.plane0_column(rows)
    bumped = counts[rows, col].copy()
    counts[rows, col] = (bumped.astype(np.uint32) - 2 * 40000).astype(np.uint16)
    fixtures.check_pileup_against_golden("r150", counts)
    names, contigs = fixtures.genome()
    ins = sorted((int(p), bytes(s)) for p, s in zip(z["ins_pos"], z["ins_seq"]))
    assert fixtures.ins_to_named(ins, names, contigs) == fixtures.golden_insertions("r150")[0]
    tot, head, rows = fixtures.golden_summary("r150")
    sm = z["summary"]
    assert sm[0] == tot and rows["Unique Mate-Paired"] == sm[4] and rows["Neither Map"] == sm[12]
    assert head[3] == "%g" % (sm[1] / sm[0]) and head[7] == "%g" % (sm[2] / sm[3])


def test_bench_two_ranks_runs_the_broadcast_and_the_reduction(tmp_path):
    """bench.py --gpus 2 as the driver launches it, on a small genome: the N > 1 branch (broadcast into the device buffers,
    sharded reads, max-over-ranks timing, end-of-run pileup sum with its check) runs and prints one JSON line"""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--genome-size", "2.5e8", "--batch-pairs", "100000",
                 "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary", "--no-pecaller"], tmp_path, 29500 + os.getpid() % 200)
    assert r.returncode == 0, r.stdout[-3000:].decode(errors="replace")
    rec = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["resident_value"] > 0
    assert rec["timings"]["pileup_reduce_s"] > 0 and rec["timings"]["pileup_reduce_checked_total"] > 0
    assert rec["counters_per_step"]["ends"] == 200000
    # every rank reports its own layout, memory and step time (a rank that fell back to the reference's table, or ran out of HBM, shows)
    assert [r["rank"] for r in rec["ranks"]] == [0, 1]
    for r in rec["ranks"]:
        assert r["lookup_replicas"] == 0 and r["hbm_free_after_setup_GiB"] > 1 and r["ms_per_step"] > 0 and r["resident_ms_per_step"] > 0
        assert 0.3 < r["mapped_frac"] <= 1.0
    assert rec["ms_per_step"] >= max(r["ms_per_step"] for r in rec["ranks"]) - 1e-6
    assert rec["config"]["lookup_replicas"] == 0 and rec["config"]["host_batches"] == 3
