#!/usr/bin/env python3
"""bench.py -- throughput of the PEMapper hot path on MI355X.

Metric (BASELINE.json): million reads mapped per second, whole job, synthetic 2 x 150 bp paired-end reads against an
hg38-sized index resident in HBM.  A "step" is one pass of the hot path (seed gather + diagonal vote, SW scoring,
pair selection, traceback + pileup) over one batch of `--batch-pairs` read pairs that are already resident in HBM.
Both ends of a pair count as reads; every read of the batch counts (mapped or not), as it does for the reference.

  python bench.py --gpus 1 --steps 8 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU.  Rank 0 generates the genome and builds the index on its GPU; the other ranks receive
pos_index / mers / genome / contig table through an RCCL broadcast at start-up (nothing is exchanged while mapping);
every rank maps its own reads (weak scaling) into its own pileup counters.
The last line printed by rank 0 is the JSON record.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md), GB/s


class DevArray:
    """exposes a device pointer handed out by the C-ABI to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n_items, typestr):
        self.__cuda_array_interface__ = {"shape": (n_items,), "typestr": typestr, "data": (ptr, False), "version": 2}


def dev_tensor(torch, dev, which, typestr, itemsize):
    ptr, nbytes = dev.buffer(which)
    return torch.as_tensor(DevArray(ptr, nbytes // itemsize, typestr), device="cuda")


def bcast_chunks(dist, t, src=0, chunk=1 << 28):
    n = t.numel()
    for o in range(0, n, chunk):
        dist.broadcast(t[o:min(n, o + chunk)], src=src)


def algorithmic_bytes_per_end(L, P_per_end, H_per_end):
    """B(L) = S*49*2*8 [pos_index pairs] + 4P [bucket payload] + H(L+21) [reference windows]
    + L [read] + 4L [pileup read-modify-write] + 4 [mfile]"""
    S = L // 16 + (0 if L % 16 == 0 else 1)
    return S * 49 * 2 * 8 + 4.0 * P_per_end + H_per_end * (L + 21) + L + 4 * L + 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-size", type=float, default=3.1e9)
    ap.add_argument("--contigs", type=int, default=25)
    ap.add_argument("--repeat-frac", type=float, default=0.5)
    ap.add_argument("--batch-pairs", type=int, default=1000000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-rate", type=float, default=0.01)
    ap.add_argument("--indel-rate", type=float, default=0.0002)
    ap.add_argument("--seed", type=int, default=20240601)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--index-mode", default="bcast", choices=["bcast", "build"],
                    help="bcast: rank 0 builds, RCCL broadcast; build: every rank builds its own replica (tests)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pecaller_amd import PemapDev

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dev_id = local_rank % ndev
    torch.cuda.set_device(dev_id)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=a.backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    gsize = int(a.genome_size)
    dev = PemapDev(dev_id)
    t_setup = time.time()
    timings = {}
    # ---- index: build on rank 0, broadcast to the replicas
    if rank == 0 or a.index_mode == "build":
        t0 = time.time()
        d_g, contig_len = dev.synth_genome(a.seed, gsize, a.contigs, a.repeat_frac)
        dev.build_index_resident(d_g, gsize, contig_len)
        dev.free(d_g)
        timings["index_build_s"] = time.time() - t0
    if world > 1 and a.index_mode == "bcast":
        t0 = time.time()
        info = [dev.index_info() if rank == 0 else None]
        dist.broadcast_object_list(info, src=0)
        n_mers, gs, n_contigs, idepth = info[0]
        if rank != 0:
            dev.index_alloc(n_mers, gs, n_contigs, idepth)
        cuda_dev = torch.device("cuda", dev_id)
        for which, ts, isz in ((0, "<i4", 4), (1, "<i4", 4), (2, "|u1", 1), (3, "<i4", 4)):
            t = dev_tensor(torch, dev, which, ts, isz)
            if a.backend == "nccl":
                bcast_chunks(dist, t, 0)
            else:       # gloo rehearsal on one box: through host memory
                h = t.cpu()
                bcast_chunks(dist, h, 0)
                if rank != 0:
                    t.copy_(h.to(cuda_dev))
        torch.cuda.synchronize()
        if rank != 0:
            dev.index_commit()
        timings["index_bcast_s"] = time.time() - t0
    n_mers, gs, n_contigs, idepth = dev.index_info()

    # ---- reads: (steps + warmup) batches resident in HBM, different reads per rank
    B = a.batch_pairs
    n_batches = a.steps + a.warmup
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
    dev.synth_reads(a.seed + 7, B * n_batches, a.read_len, paired=True, sub_rate=a.sub_rate, indel_rate=a.indel_rate,
                    first_read=rank * B * n_batches)
    timings["setup_s"] = time.time() - t_setup

    for w in range(a.warmup):
        dev.run_slice(w * B, B, sync=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kt = {"seed": [], "sw_single": [], "sw_multi": [], "select": [], "sw_redo": [], "walk": [], "lookup": [], "vote": []}
    agg = None
    # the K steps are queued back to back (each step = one slice of the resident reads) and synchronised once: the
    # library pipelines the look-ups of a step's first chunk under the previous step's last chunk
    for s in range(a.steps):
        dev.run_slice((a.warmup + s) * B, B, sync=False)
    dev.sync()
    st, tm = dev.run_stats()        # totals over the K queued steps
    for k in kt:
        kt[k].append(tm[k] / a.steps)
    agg = st
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    reads_per_step = 2 * B
    value = world * a.steps * reads_per_step / dt / 1e6

    if rank == 0:
        # ---- roofline of the dominant kernel, live HIP-event durations on the kernels' own stream
        avg_ms = {k: float(np.mean(v)) for k, v in kt.items()}
        split = avg_ms["lookup"] > 0
        cand = {k: v for k, v in avg_ms.items() if k != ("seed" if split else "lookup") and (split or k != "vote")}
        dom = max(cand, key=cand.get)
        ends = agg["ends"]
        n_rep, rec_bytes = dev.lookup_replicas()
        P_e = agg["positions"] / ends
        H_e = agg["sw_score"] / ends
        L = a.read_len
        S = L // 16 + (0 if L % 16 == 0 else 1)
        n_single = (agg["sw_dirs"] - agg["redo"]) / ends      # single-hit problems the DP scored, with nibbles (the rest: gapless rule)
        n_multi = (agg["sw_score"] - (agg["sw_dirs"] - agg["redo"]) - agg.get("gapless", 0)) / ends
        # SW geometry as pick_geom (pemap_capi.hip): lanes per alignment, columns per lane
        lanes, W = (8, 13) if L <= 104 else (8, 19) if L <= 152 else (8, 26) if L <= 208 else (8, 32) if L <= 256 else (8, 38)
        if L > 104 and os.environ.get("PEMAP_GAPLESS", "2") != "0" and "PEMAP_SW_LANES" not in os.environ:
            lanes, W = (16, 10) if L <= 160 else (16, 13) if L <= 208 else (16, 16) if L <= 256 else (16, 19)   # the default beside the gapless rule
        slab = lanes * ((L + 21 + lanes + 15) // 16 * 16) * ((W * 4 + 31) // 32) * 4
        per_end = {
            "seed": S * 49 * 2 * 8 + 4.0 * P_e + L,                     # pos_index pairs + bucket payload + the read
            # ... + the (key, segment) lists written for the vote; with the look-up replicas an entry is 4 bytes, not a pair
            "lookup": S * 49 * 2 * (4 if n_rep else 8) + 4.0 * P_e + L + 5.0 * P_e,
            "vote": 5.0 * P_e + H_e * 16,
            "sw_single": n_single * (L + 21 + L + slab),                # window + read in, direction nibbles out
            "sw_multi": n_multi * (L + 21 + L),
            "select": H_e * 16 + 12,
            "sw_redo": (agg["redo"] / ends) * (L + 21 + L + slab),
            "walk": (agg["walks"] / ends) * ((L + 21) * 0.5 + 4 * L) + 4,   # nibbles along the path + pileup RMW + mfile
        }
        # one "launch" of the dominant kernel = one chunk of the step (the run is cut into chunks that pipeline on two
        # streams); algorithmic bytes per launch / average launch duration (HIP events on the kernel's own stream)
        # `achieved` prices the launch with SURVEY.md 8(d)'s per-unit figure B(L) (the whole path's algorithmic bytes per read-end:
        # one launch of the dominant kernel carries one chunk of ends through the path's bound); the kernel's own algorithmic
        # bytes (for the look-up kernel: B's index and payload terms plus the lists it writes for the vote) are reported beside it
        launches = max(1, agg["chunks"] // a.steps)
        launch_ms = avg_ms[dom] / launches
        total_b = algorithmic_bytes_per_end(L, P_e, H_e)
        alg_bytes = total_b * (ends / a.steps) / launches
        kernel_bytes = per_end[dom] * (ends / a.steps) / launches
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        kname = "pm_%s_kernel" % dom
        if dom == "lookup" and os.environ.get("PEMAP_LOOKUP_WAVES", "6") != "0":
            # the persistent wave-per-end forms of the look-up kernel: against the 8 replicas of the table (default on an
            # MI355X), or against the reference's table
            kname = ("pm_lookup_rep2_kernel" if os.environ.get("PEMAP_LOOKUP_V", "1") == "2" else "pm_lookup_rep_kernel") if n_rep \
                else "pm_lookup_wave_kernel"
        elif dom == "vote" and os.environ.get("PEMAP_VOTE_WAVES", "1024") != "0":
            kname = "pm_vote_wave_kernel"
        elif dom.startswith("sw_"):
            kname = "pm_sw_kernel"
        traffic, tsrc = pmc_traffic(kname, gs, B, L)
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                "launches_per_step": launches, "avg_launch_ms": round(launch_ms, 3), "algorithmic_bytes_per_launch": round(alg_bytes),
                "kernel_algorithmic_bytes_per_launch": round(kernel_bytes),
                "kernel_achieved": round(kernel_bytes / (launch_ms * 1e-3) / 1e9, 2),
                "kernel_ms": {k: round(v, 3) for k, v in avg_ms.items()},
                "bytes_per_end_path": round(total_b, 1), "P_per_end": round(P_e, 2), "H_per_end": round(H_e, 3),
                "path_GBs": round(total_b * ends / a.steps / (dt / a.steps) / 1e9, 2),
                "cells_per_s": round((agg["cells_score"] + agg["cells_dirs"]) / dt, 0)}
        if dom != "lookup" and split:
            # the path's HBM-heavy kernel beside the dominant one: same accounting
            lk_name = ("pm_lookup_rep2_kernel" if os.environ.get("PEMAP_LOOKUP_V", "1") == "2" else "pm_lookup_rep_kernel") if n_rep else "pm_lookup_wave_kernel"
            lk_ms = avg_ms["lookup"] / launches
            lk_traffic, lk_src = pmc_traffic(lk_name, gs, B, L)
            roof["lookup_kernel"] = {"kernel": lk_name, "avg_launch_ms": round(lk_ms, 3), "achieved": round(alg_bytes / (lk_ms * 1e-3) / 1e9, 2),
                                     "frac": round(alg_bytes / (lk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": lk_traffic, "traffic_source": lk_src,
                                     "kernel_algorithmic_bytes_per_launch": round(per_end["lookup"] * (ends / a.steps) / launches)}
        cpu = None
        if world == 1 and not a.no_cpu:
            cpu = cpu_baseline(dev, a, B)
        rec = {
            "metric": "M reads mapped/sec (whole node), 2x150bp PE synthetic hg38",
            "value": round(value, 4), "unit": "M reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "hg38-sized synthetic index resident in HBM (%.2f Gbp, %d contigs, %d%% repeat tiles), "
                                   "2x%dbp paired-end reads, %d pairs per step per GPU" % (gs / 1e9, n_contigs, int(a.repeat_frac * 100), L, B),
                       "genome_size": gs, "n_mers": n_mers, "batch_pairs": B, "read_len": L, "sub_rate": a.sub_rate,
                       "indel_rate": a.indel_rate, "sharding": "reads split by rank, index replica per GPU",
                       "lookup_replicas": n_rep, "lookup_record_bytes": rec_bytes},
            "roofline": roof, "cpu_baseline": cpu, "timings": {k: round(v, 2) for k, v in timings.items()},
            "counters_per_step": {k: int(v / a.steps) for k, v in agg.items()},
        }
        print(json.dumps(rec), flush=True)
    barrier()
    dev.close()
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel, gsize, B, L):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs
    of this same command, profiles/r01_bench_pmc_*.json, newest pipeline first), valid for the default workload only.
    FETCH_SIZE is taken at face value: on this path's random 8- to 64-byte reads it equals TCC_EA0_RDREQ x 64 B, one 64-byte
    request each (calibrated with tools/micro/gather_calib.hip; the 1/2 factor of MI355X_MICROARCH.md applies to coalesced
    streams)."""
    if not (gsize == 3100000000 and B == 1000000 and L == 150):
        return None, None
    for name in ("r01_bench_pmc_final3.json", "r01_bench_pmc_v12.json", "r01_bench_pmc_replicas.json", "r01_bench_pmc_final.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            pm = json.load(open(path))
            f = [v for k, v in pm["FETCH_SIZE"].items() if k.startswith(kernel)][0]["mean_KB_per_launch"]
            w = [v for k, v in pm["WRITE_SIZE"].items() if k.startswith(kernel)][0]["mean_KB_per_launch"]
            return round((f + w) * 1024.0), "profiles/" + name
        except Exception:
            continue
    return None, None


def cpu_baseline(dev, a, B):
    """the oracle (CPU restatement of the reference loop, pthreads) on a bounded sample of the same workload, same index;
    its coordinates and classes are also compared with what the GPU produced for the same reads (checker role)"""
    import oracle_py
    t0 = time.time()
    pos_index = dev.read_buffer(0, np.uint32)
    mers = dev.read_buffer(1, np.uint32)
    genome = dev.read_buffer(2, np.uint8)
    cs = dev.read_buffer(3, np.uint32)
    t_copy = time.time() - t0
    r1, l1, r2, l2 = dev.staged_reads()
    first = a.warmup * B
    ix = dict(pos_index=pos_index, mers=mers, genome=genome, contig_starts=cs)
    o = oracle_py.Oracle(ix, paired=True, min_dist=0, max_dist=500, min_align=0.85)
    # GPU results of the first timed batch, for the comparison
    dev.run_slice(first, B, sync=True)
    g1, g2, gt = dev.collect(B)
    n = 4000
    done = 0
    spent = 0.0
    mism = 0
    while True:
        lo, hi = first + done, first + done + n
        if hi > first + B:
            break
        t1 = time.time()
        m1, m2, mt, _, _ = o.map_batch(r1[lo:hi], l1[lo:hi], r2[lo:hi], l2[lo:hi], threads=a.cpu_threads)
        spent += time.time() - t1
        mism += int((m1 != g1[done:done + n]).sum() + (m2 != g2[done:done + n]).sum() + (mt != gt[done:done + n]).sum())
        done += n
        if spent >= a.cpu_seconds:
            break
        rate = done / spent
        n = int(max(4000, min(rate * (a.cpu_seconds - spent), 200000)))
    return {"value": round(2 * done / spent / 1e6, 5), "unit": "M reads/s", "cores": a.cpu_threads, "kind": "port",
            "gpu_vs_cpu_mismatches": mism, "compared_pairs": done,
            "sample": "%d pairs of the first timed batch, same index (copied back from HBM in %.1f s), %.1f s of CPU time"
                      % (done, t_copy, spent)}


if __name__ == "__main__":
    main()
