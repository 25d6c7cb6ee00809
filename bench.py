#!/usr/bin/env python3
"""bench.py -- throughput of the PEMapper hot path on MI355X.

Metric (BASELINE.json): million reads mapped per second, whole job, synthetic 2 x 150 bp paired-end reads against an
hg38-sized index resident in HBM.  A "step" is one pass of the hot path (seed gather + diagonal vote, SW scoring,
pair selection, traceback + pileup) over one batch of `--batch-pairs` read pairs.  Both ends of a pair count as reads;
every read of the batch counts (mapped or not), as it does for the reference; the mapped share is printed beside it.

`value` is timed at the reference's seam (first batch submitted to last result returned): every step hands
HOST buffers of reads to pemap_dev_submit_batch -- the call that stands where pthread_create(map_everything) stands,
pemapper.c:684 -- and takes m1 / m2 / mapping_type back in host memory, two batches in flight as the reference's reader
thread keeps its workers fed.  `resident_value` is the same K steps with the reads already in HBM and the results left
there (what round 1 printed as `value`).

  python bench.py --gpus 1 --steps 8 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU.  Rank 0 generates the genome and builds the index on its GPU; the other ranks receive
pos_index / mers / genome / contig table through an RCCL broadcast at start-up (nothing is exchanged while mapping);
every rank maps its own reads (weak scaling) into its own pileup counters, which are summed once at the end
(`timings.pileup_reduce_s`, outside the timed region as the reference's final genome walk is).
The same JSON line carries two more measurements: `secondary` = BASELINE config 3's shape (pemapper_tsw reads: 2 x 250
bases trimmed 3 / 2, 5 % of the read-ends with one 1..10-base indel), and `pecaller` = BASELINE config 4 (the per-site
caller on 64-sample 30x pileup columns).  The last line printed by rank 0 is the JSON record.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md), GB/s
PECALL_BYTES_PER_SITE = 64 * 12 + 64 * 14 * 8      # 64 x 12 B in + 64 x 14 x 8 B likelihoods = 7,936 B


def algorithmic_bytes_per_end(L, P_per_end, H_per_end):
    """B(L) = S*49*2*8 [pos_index pairs] + 4P [bucket payload] + H(L+21) [reference windows]
    + L [read] + 4L [pileup read-modify-write] + 4 [mfile]"""
    S = L // 16 + (0 if L % 16 == 0 else 1)
    return S * 49 * 2 * 8 + 4.0 * P_per_end + H_per_end * (L + 21) + L + 4 * L + 4


def sw_geometry(L):
    """lanes per alignment and columns per lane as pick_geom (pemap_capi.hip) chooses them"""
    if L <= 104:
        return 8, 13
    if os.environ.get("PEMAP_GAPLESS", "2") == "0" and L <= 152:
        return 8, 19
    return (16, 10) if L <= 160 else (16, 13) if L <= 208 else (16, 16) if L <= 256 else (16, 19)


def lookup_kernel_name(n_rep):
    """the kernel that serves the look-ups: with the 8 table replicas the fused look-up + vote kernel"""
    return "pm_seed4_kernel" if n_rep else "pm_lookup_wave_kernel"


def mapper_leg(dev, a, L, B, steps, warmup, rank, world, barrier, allmax, indel_read_frac=0.0, seed_off=7, seam=True):
    """K steps of the mapping path on batches of B pairs of L-base reads: at the seam (host buffers in, results out) and on
    resident reads.  -> dict of rates, per-step kernel times and counters of the seam run (or of the resident run if seam=False)"""
    n_batches = steps + warmup
    # at most --host-batches distinct batches exist (generated on the device, staged on the host, pinned); step k maps batch k modulo
    # that number.  (Every batch of a run was distinct until round 3: (steps + warm-up) x 1 M x 2 x 160 B = 8 GB of pinned host memory
    # per rank with the driver's 20 + 5 steps, 64 GB on an 8-GPU node.)
    NB = max(1, min(n_batches, a.host_batches))
    first_read = rank * B * NB

    def synth():
        if indel_read_frac > 0:
            dev.synth_reads_indel(a.seed + seed_off, B * NB, L, paired=True, sub_rate=a.sub_rate, indel_read_frac=indel_read_frac,
                                  first_read=first_read)
        else:
            dev.synth_reads(a.seed + seed_off, B * NB, L, paired=True, sub_rate=a.sub_rate, indel_rate=a.indel_rate, first_read=first_read)

    out = {}
    synth()
    host = None
    if seam:
        # ---- at the seam: the reads start in host memory (the reference's batch buffers), the results end there
        t0 = time.time()
        r1, l1, r2, l2 = host = dev.staged_reads()
        dev.pin_host(r1)            # the reference allocates its batch buffers once (pd_node_alloc) and reuses them: pinned once, here
        dev.pin_host(r2)
        out["host_staging_s"] = time.time() - t0
        res = {}

        trace = [] if os.environ.get("BENCH_TRACE") else None

        def run(k_from, k_to, depth=2):
            tick = []
            for k in range(k_from, k_to):
                s = slice((k % NB) * B, (k % NB + 1) * B)
                ta = time.perf_counter()
                tick.append((k, dev.submit_batch(r1[s], l1[s], r2[s], l2[s])))
                tb = time.perf_counter()
                if len(tick) > depth:
                    kk, t = tick.pop(0)
                    res[kk] = dev.wait_batch(t)
                if trace is not None:
                    trace.append(("submit %d" % k, round((tb - ta) * 1e3, 2), "then wait", round((time.perf_counter() - tb) * 1e3, 2)))
            for kk, t in tick:
                ta = time.perf_counter()
                res[kk] = dev.wait_batch(t)
                if trace is not None:
                    trace.append(("wait %d" % kk, round((time.perf_counter() - ta) * 1e3, 2)))
        run(0, warmup)
        dev.sync()                  # the warm-up's kernels are accounted and out of the counters
        barrier()
        t0 = time.perf_counter()
        run(warmup, n_batches)
        out["seam_dt_local"] = time.perf_counter() - t0
        dt = allmax(out["seam_dt_local"])
        st, tm = dev.run_stats()        # totals over the K timed steps
        if trace:
            print("seam trace (ms):", trace, file=sys.stderr)
        out.update(seam_dt=dt, stats=st, times=tm, results=res)
        mapped = sum(int((res[k][0] > 0).sum() + (res[k][1] > 0).sum()) for k in range(warmup, n_batches))
        out["mapped_frac"] = mapped / float(2 * B * steps)
        dev.unpin_host(r1)
        dev.unpin_host(r2)
        synth()                     # back to one resident read set (same reads)
    # ---- resident: reads already in HBM, results left there; the K steps queued back to back and synchronised once (the library
    #      pipelines the look-ups of a step's first chunk under the previous step's last chunk)
    for w in range(warmup):
        dev.run_slice((w % NB) * B, B, sync=True)
    barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        dev.run_slice(((warmup + s) % NB) * B, B, sync=False)
    dev.sync()
    out["resident_dt_local"] = time.perf_counter() - t0
    out["resident_dt"] = allmax(out["resident_dt_local"])
    out.setdefault("seam_dt_local", out["resident_dt_local"])
    out.setdefault("mapped_frac", 0.0)
    if not seam:
        st, tm = dev.run_stats()
        out.update(stats=st, times=tm)
    out["host"] = host
    out["host_batches"] = NB
    return out


def roofline_of(leg, dt, steps, L, n_rep, gs, B, config="hg38_150"):
    """roofline of the dominant kernel of a mapper leg: SURVEY.md 8(d)'s algorithmic bytes per read-end x the read-ends one
    launch carries / that kernel's average launch duration (HIP events on the kernel's own stream, taken over the timed region)"""
    agg, tm = leg["stats"], leg["times"]
    avg_ms = {k: v / steps for k, v in tm.items()}
    split = avg_ms["lookup"] > 0
    cand = {k: v for k, v in avg_ms.items() if k != ("seed" if split else "lookup") and (split or k != "vote")}
    dom = max(cand, key=cand.get)
    ends = agg["ends"]
    P_e = agg["positions"] / ends
    H_e = agg["sw_score"] / ends
    S = L // 16 + (0 if L % 16 == 0 else 1)
    n_single = (agg["sw_dirs"] - agg["redo"]) / ends      # single-hit problems the DP scored, with nibbles (the rest: gapless rule)
    n_multi = (agg["sw_score"] - (agg["sw_dirs"] - agg["redo"]) - agg.get("gapless", 0)) / ends
    n_band = agg.get("banded", 0) / ends
    lanes, W = sw_geometry(L)
    slab = lanes * ((L + 21 + lanes + 15) // 16 * 16) * ((W * 4 + 31) // 32) * 4
    per_end = {
        "seed": S * 49 * 2 * 8 + 4.0 * P_e + L,                     # pos_index pairs + bucket payload + the read
        # ... + the (key, segment) lists written for the vote; with the look-up replicas an entry is 4 bytes, not a pair
        "lookup": S * 49 * 2 * (4 if n_rep else 8) + 4.0 * P_e + L + 5.0 * P_e,
        "vote": 5.0 * P_e + H_e * 16,
        # window + read in, direction nibbles out (a banded problem writes 16 bytes per read column, not the whole slab)
        "sw_single": (n_single - n_band) * (L + 21 + L + slab) + n_band * (L + 21 + L + 16 * (L + 1)),
        "sw_multi": n_multi * (L + 21 + L),
        "select": H_e * 16 + 12,
        "sw_redo": (agg["redo"] / ends) * (L + 21 + L + slab),
        "walk": (agg["walks"] / ends) * ((L + 21) * 0.5 + 4 * L) + 4,   # nibbles along the path + pileup RMW + mfile
    }
    # one "launch" of the dominant kernel = one chunk of the step (the run is cut into chunks that pipeline on several
    # streams).  `achieved` prices the launch with SURVEY.md 8(d)'s per-unit figure B(L) (the whole path's algorithmic bytes per
    # read-end); the kernel's own algorithmic bytes are reported beside it
    launches = max(1, agg["chunks"] // steps)
    launch_ms = avg_ms[dom] / launches
    total_b = algorithmic_bytes_per_end(L, P_e, H_e)
    alg_bytes = total_b * (ends / steps) / launches
    kernel_bytes = per_end[dom] * (ends / steps) / launches
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    kname = "pm_%s_kernel" % dom
    if dom == "lookup":
        kname = lookup_kernel_name(n_rep)
    elif dom == "vote" and os.environ.get("PEMAP_VOTE_WAVES", "1024") != "0":
        kname = "pm_vote_wave_kernel"
    elif dom.startswith("sw_"):
        kname = "pm_sw_kernel"
    traffic, tsrc = pmc_traffic(kname, gs, B, L, config)
    step_traffic, _ = pmc_traffic(None, gs, B, L, config)
    roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
            "launches_per_step": launches, "avg_launch_ms": round(launch_ms, 3), "algorithmic_bytes_per_launch": round(alg_bytes),
            "kernel_algorithmic_bytes_per_launch": round(kernel_bytes),
            "kernel_achieved": round(kernel_bytes / (launch_ms * 1e-3) / 1e9, 2),
            "kernel_ms": {k: round(v, 3) for k, v in avg_ms.items()},
            "bytes_per_end_path": round(total_b, 1), "P_per_end": round(P_e, 2), "H_per_end": round(H_e, 3),
            "path_GBs": round(total_b * ends / steps / (dt / steps) / 1e9, 2),
            "path_frac": round(total_b * ends / steps / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5),
            "step_traffic_bytes": step_traffic,
            "step_traffic_over_algorithmic": (round(step_traffic / (total_b * ends / steps), 2) if step_traffic else None),
            "cells_per_s": round((agg["cells_score"] + agg["cells_dirs"] + agg.get("cells_band", 0)) / dt, 0)}
    if step_traffic:
        # what the two streams share: the step's counter traffic as 64-byte lines per second -- nearly all of them
        # single lines at random addresses -- against what the chip delivers of those at best (tools/micro/line_gather: 48.8 G/s)
        lines_s = step_traffic / 64.0 / (dt / steps)
        roof["random_lines"] = {"step_G_lines_per_s": round(lines_s / 1e9, 2), "ceiling_G_lines_per_s": 48.8, "frac": round(lines_s / 48.8e9, 3),
                                "source": "profiles/r01_line_gather.txt"}
    if dom != "lookup" and split:
        # the path's HBM-heavy kernel beside the dominant one: same accounting
        lk_name = lookup_kernel_name(n_rep)
        lk_ms = avg_ms["lookup"] / launches
        lk_traffic, lk_src = pmc_traffic(lk_name, gs, B, L, config)
        roof["lookup_kernel"] = {"kernel": lk_name, "avg_launch_ms": round(lk_ms, 3), "achieved": round(alg_bytes / (lk_ms * 1e-3) / 1e9, 2),
                                 "frac": round(alg_bytes / (lk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": lk_traffic, "traffic_source": lk_src,
                                 "kernel_algorithmic_bytes_per_launch": round(per_end["lookup"] * (ends / steps) / launches)}
    return roof


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
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline's primary sample; 0 = the physical cores of the host this process may run on "
                         "(SURVEY.md 8(d)); samples with 24 (the authors' default) and 16 threads are reported beside it")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the pemapper_tsw 2x250 leg")
    ap.add_argument("--no-pecaller", action="store_true", help="skip the PECaller leg")
    ap.add_argument("--config", default="hg38_150", choices=["hg38_150", "tsw250"],
                    help="tsw250: only BASELINE config 3's leg is run and printed as the record's headline fields (profiling)")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--realistic-steps", type=int, default=3,
                    help="steps of the `realistic` data point: the same reads on a genome with few repeat tiles, where (as on real hg38) "
                         "over 95 %% of the reads map and nearly every end has an alignment to score; 0 = skip")
    ap.add_argument("--realistic-repeat-frac", type=float, default=0.02)
    ap.add_argument("--pecall-sites", type=int, default=2000000,
                    help="columns per launch of the PECaller leg (a launch ends with its slowest column: the few hundred-configuration "
                         "variant columns take ~50-90 ms each on one wave, so short launches measure that tail, not the rate)")
    ap.add_argument("--pecall-wide-sites", type=int, default=1000000, help="columns of the 128-sample data point of the PECaller leg (0 = skip)")
    ap.add_argument("--pecall-wide256-sites", type=int, default=400000, help="columns of the 256-sample data point of the PECaller leg (0 = skip)")
    ap.add_argument("--pecall-cpu-seconds", type=float, default=10.0)
    ap.add_argument("--host-batches", type=int, default=4, help="distinct batches staged (and pinned) on the host; the steps cycle through them")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: do not fail when a rank had no room for the look-up replicas and maps from the reference's table (slower, same results)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--index-mode", default="bcast", choices=["bcast", "build"],
                    help="bcast: rank 0 builds, RCCL broadcast; build: every rank builds its own replica (tests)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pecaller_amd import PemapDev
    from pecaller_amd import dist as pd

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
    on_gpu = a.backend == "nccl"
    if world > 1 and on_gpu:
        # one small device collective NOW: RCCL's communicator, its buffers and torch's context are allocated before the library takes
        # ~290 of the device's 309 GB (index + 128 GiB of look-up replicas + work arrays); afterwards nothing of size is allocated
        warm = torch.ones(1 << 20, dtype=torch.int32, device="cuda")
        dist.all_reduce(warm)
        dist.broadcast(warm, src=0)
        torch.cuda.synchronize()
        assert int(warm[0].item()) == world
        del warm

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        """the bracketing barrier + max over ranks of a rank's wall time"""
        barrier()
        if world > 1:
            tt = torch.tensor([x], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return x

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
        # (four numbers as a tensor: no pickled-object collective on the path)
        it = torch.tensor(list(dev.index_info()) if rank == 0 else [0, 0, 0, 0], dtype=torch.int64, device="cuda" if on_gpu else "cpu")
        dist.broadcast(it, src=0)
        n_mers, gs, n_contigs, idepth = (int(x) for x in it.tolist())
        if rank != 0:
            dev.index_alloc(n_mers, gs, n_contigs, idepth)
        # torch views of the library's device buffers: RCCL writes straight into the index the kernels read (a gloo rehearsal on
        # one box goes through host copies, pecaller_amd/dist.py)
        pd.broadcast_tensors(dist, [pd.device_tensor(torch, dev, which) for which in (0, 1, 2, 3)], src=0)
        torch.cuda.synchronize()
        if rank != 0:
            dev.index_commit()
        timings["index_bcast_s"] = time.time() - t0
    n_mers, gs, n_contigs, idepth = dev.index_info()
    n_rep, rec_bytes = dev.lookup_replicas()
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
    timings["setup_s"] = time.time() - t_setup

    def gather_ranks(vals):
        """one row of numbers per rank -> list of rows on every rank (tensor all_gather: works under RCCL and gloo alike)"""
        t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        if world == 1:
            return [t.tolist()]
        rows = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(rows, t)
        return [r.tolist() for r in rows]
    # every rank's layout and what is left of its HBM after set-up: a rank that had no room for the replicas maps from the reference's
    # table at ~0.6x the rate and would set the max-over-ranks time -- the run fails rather than print that as the node's number
    free_b, total_b = torch.cuda.mem_get_info()
    setup_rows = gather_ranks([n_rep, free_b / 2.0 ** 30, total_b / 2.0 ** 30, timings["setup_s"]])
    fell_back = [r for r, row in enumerate(setup_rows) if int(row[0]) != 8]
    if world > 1 and fell_back and len(fell_back) < world and not a.allow_fallback and os.environ.get("PEMAP_REPLICAS") != "0":
        if rank == 0:
            print("bench.py: rank(s) %s map without the look-up replicas (free HBM per rank after set-up, GiB: %s); --allow-fallback accepts that"
                  % (fell_back, [round(row[1], 1) for row in setup_rows]), file=sys.stderr)
        raise SystemExit(3)

    B = a.batch_pairs
    tsw_only = a.config == "tsw250"
    TSW_L = 245                     # 250-base reads after pemapper_tsw's trims of 3 / 2 (pemapper_tsw.c:693-704; the golden's values)
    main_leg = None
    if not tsw_only:
        main_leg = mapper_leg(dev, a, a.read_len, B, a.steps, a.warmup, rank, world, barrier, allmax)
    sec_leg = None
    if tsw_only or not a.no_secondary:
        k, w = (a.steps, a.warmup) if tsw_only else (a.secondary_steps, 1)
        sec_leg = mapper_leg(dev, a, TSW_L, B, k, w, rank, world, barrier, allmax, indel_read_frac=0.05, seed_off=11, seam=True)
        sec_leg["steps"] = k

    leg0 = sec_leg if tsw_only else main_leg
    step_rows = gather_ranks([leg0["seam_dt_local"] / a.steps * 1e3, leg0["resident_dt_local"] / a.steps * 1e3, leg0["mapped_frac"],
                              leg0["stats"]["big_ends"] / a.steps])
    ranks = [{"rank": r, "lookup_replicas": int(setup_rows[r][0]), "hbm_free_after_setup_GiB": round(setup_rows[r][1], 1),
              "hbm_total_GiB": round(setup_rows[r][2], 1), "setup_s": round(setup_rows[r][3], 2), "ms_per_step": round(step_rows[r][0], 3),
              "resident_ms_per_step": round(step_rows[r][1], 3), "mapped_frac": round(step_rows[r][2], 4), "big_ends_per_step": int(step_rows[r][3])}
             for r in range(world)]
    # ---- end of the run: the per-GPU pileup partials are summed (the reference's one shared all_base_list, pemapper.c:156)
    if world > 1:
        cnt = pd.device_tensor(torch, dev, 4)
        def grand_total(t):       # of the 16-bit counters, two to a word (in pieces: the masks are temporaries of the piece's size)
            tot = torch.zeros((), dtype=torch.int64, device=t.device)
            flat = t.view(-1)
            for o in range(0, flat.numel(), 1 << 27):
                c = flat[o:o + (1 << 27)]
                tot += (c & 0xFFFF).sum(dtype=torch.int64) + ((c >> 16) & 0xFFFF).sum(dtype=torch.int64)
            return tot
        local = grand_total(cnt).reshape(1)
        tot = local.clone() if on_gpu else local.cpu()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        barrier()
        t0 = time.perf_counter()
        pd.reduce_pileup(dist, cnt)
        timings["pileup_reduce_s"] = allmax(time.perf_counter() - t0)
        after = int(grand_total(cnt).item())
        # every rank now holds the sum: the grand total of the counters equals the sum of the ranks' totals before
        assert after == int(tot.item()), (after, int(tot.item()))
        timings["pileup_reduce_checked_total"] = after

    if rank == 0:
        L = TSW_L if tsw_only else a.read_len
        leg = sec_leg if tsw_only else main_leg
        steps = a.steps
        reads_per_step = 2 * B
        dt = leg["seam_dt"]
        value = world * steps * reads_per_step / dt / 1e6
        resident = world * steps * reads_per_step / leg["resident_dt"] / 1e6
        roof = roofline_of(leg, dt, steps, L, n_rep, gs, B)
        cpu = None
        if world == 1 and not a.no_cpu:
            cpu = cpu_baseline(dev, a, B, leg, a.warmup)
        rec = {
            "metric": "M reads mapped/sec (whole node), 2x150bp PE synthetic hg38",
            "value": round(value, 4), "unit": "M reads/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "timed_region": "host buffers of reads in -> m1/m2/mapping_type in host memory (pemap_dev_submit_batch / pemap_dev_wait_batch, "
                            "2 batches in flight, PCIe both ways included)",
            "resident_value": round(resident, 4), "resident_ms_per_step": round(leg["resident_dt"] / steps * 1e3, 3),
            "mapped_frac": round(leg["mapped_frac"], 4), "mapped_reads_per_s_M": round(value * leg["mapped_frac"], 4),
            "config": {"workload": "hg38-sized synthetic index resident in HBM (%.2f Gbp, %d contigs, %d%% repeat tiles), "
                                   "2x%dbp paired-end reads, %d pairs per step per GPU" % (gs / 1e9, n_contigs, int(a.repeat_frac * 100), L, B),
                       "genome_size": gs, "n_mers": n_mers, "batch_pairs": B, "read_len": L, "sub_rate": a.sub_rate,
                       "indel_rate": a.indel_rate, "sharding": "reads split by rank, index replica per GPU",
                       "lookup_replicas": min(r["lookup_replicas"] for r in ranks), "lookup_record_bytes": rec_bytes,
                       "host_batches": leg["host_batches"]},
            "ranks": ranks,
            "roofline": roof, "cpu_baseline": cpu, "timings": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in timings.items()},
            "counters_per_step": {k: int(v / steps) for k, v in leg["stats"].items()},
        }
        if tsw_only:
            rec["metric"] = "M reads mapped/sec, pemapper_tsw path, 2x250bp (trimmed 3/2) 5% indel-enriched reads"
        elif sec_leg is not None:
            ks = sec_leg["steps"]
            sdt = sec_leg["seam_dt"]
            rdt = sec_leg["resident_dt"]
            rec["secondary"] = {
                "config": "pemapper_tsw path: 2x250bp reads trimmed 3/2 (245-base rows at the seam), 5%% of the read-ends with one 1..10-base "
                          "indel, %d pairs per step, same index" % B,
                "value": round(world * ks * reads_per_step / sdt / 1e6, 4), "unit": "M reads/s", "steps": ks, "ms_per_step": round(sdt / ks * 1e3, 3),
                "timed_region": "host buffers of reads in -> results in host memory, as the headline's",
                "resident_value": round(world * ks * reads_per_step / rdt / 1e6, 4), "resident_ms_per_step": round(rdt / ks * 1e3, 3),
                "mapped_frac": round(sec_leg["mapped_frac"], 4),
                "roofline": roofline_of(sec_leg, sdt, ks, TSW_L, n_rep, gs, B, config="tsw250"),
                "counters_per_step": {k: int(v / ks) for k, v in sec_leg["stats"].items()}}
        if world == 1 and not tsw_only and a.realistic_steps > 0:
            # ---- the workload's own weakness: a third of its reads never seed (high-copy repeat tiles, N blocks), so H = 0.66
            #      alignments per end where real hg38 reads map at > 95 % with H >= 1.  The same path on a genome of the same size
            #      with 2 % repeat tiles: every number of the headline again, with its own check against the CPU oracle
            t0 = time.time()
            dev.close()                 # (the index build needs the room the replicas and the work arrays hold: a fresh object)
            dev = PemapDev(dev_id)
            dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
            d_g, contig_len = dev.synth_genome(a.seed + 101, gsize, a.contigs, a.realistic_repeat_frac)
            dev.build_index_resident(d_g, gsize, contig_len)
            dev.free(d_g)
            t_build = time.time() - t0
            rl = mapper_leg(dev, a, a.read_len, B, a.realistic_steps, 1, rank, world, barrier, allmax, seed_off=23)
            ks = a.realistic_steps
            rcpu = None
            if not a.no_cpu:
                import copy
                a2 = copy.copy(a)
                a2.cpu_seconds = min(a.cpu_seconds, 5.0)
                rcpu = cpu_baseline(dev, a2, B, rl, 1)
            st = rl["stats"]
            rec["realistic"] = {
                "config": "the headline's reads on a %.2f Gbp genome with %.0f%% repeat tiles (seed + 101), index rebuilt in %.1f s"
                          % (gs / 1e9, a.realistic_repeat_frac * 100, t_build),
                "value": round(ks * reads_per_step / rl["seam_dt"] / 1e6, 4), "unit": "M reads/s", "steps": ks,
                "ms_per_step": round(rl["seam_dt"] / ks * 1e3, 3), "resident_value": round(ks * reads_per_step / rl["resident_dt"] / 1e6, 4),
                "resident_ms_per_step": round(rl["resident_dt"] / ks * 1e3, 3), "mapped_frac": round(rl["mapped_frac"], 4),
                "H_per_end": round(st["sw_score"] / st["ends"], 3), "P_per_end": round(st["positions"] / st["ends"], 2),
                "kernel_ms": {k: round(v / ks, 3) for k, v in rl["times"].items()},
                "cpu_check": None if rcpu is None else {k: rcpu[k] for k in ("value", "cores", "gpu_vs_cpu_mismatches", "compared_pairs")},
                "counters_per_step": {k: int(v / ks) for k, v in st.items()}}
            # (beside `value`: what a genome with hg38's share of mappable sequence gives)
            rec["realistic_value"] = rec["realistic"]["value"]
            rec["realistic_ms_per_step"] = rec["realistic"]["ms_per_step"]
        if world == 1 and not a.no_pecaller and not tsw_only:
            dev.close()
            dev = None
            rec["pecaller"] = pecaller_leg(a)
        print(json.dumps(rec), flush=True)
    barrier()
    if dev is not None:
        dev.close()
    if world > 1:
        dist.destroy_process_group()


def kernel_sources_sha(prefix=""):
    """sha1 over the device sources (prefix "pemap_": the mapper's, "pecall_": the caller's): a counter profile belongs to the
    kernels it was taken on"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for fn in sorted(glob.glob(os.path.join(ROOT, "pecaller_amd", "csrc", prefix + "*.hip*"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, gsize, B, L, config="hg38_150"):
    """HBM bytes per launch of `kernel` (or, kernel=None, per step over all kernels) from the committed rocprofv3 PMC passes
    (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, profiles/r04_bench_pmc_<config>.json, written by
    tools/profile.sh with the sha1 of the device sources it ran), valid for the default workload only -- and only while the kernels
    are the ones profiled: a different sha gives None and says so.  FETCH_SIZE is taken at face value: on this path's random 8- to
    64-byte reads it equals TCC_EA0_RDREQ x 64 B, one 64-byte request each (calibrated with tools/micro/gather_calib.hip; the 1/2
    factor of MI355X_MICROARCH.md applies to coalesced streams)."""
    if not (gsize == 3100000000 and B == 1000000 and L == (150 if config == "hg38_150" else 245)):
        return None, None
    name = "r04_bench_pmc_%s.json" % config
    path = os.path.join(ROOT, "profiles", name)
    try:
        pm = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    sha = kernel_sources_sha("pemap_")
    if pm.get("sha_pemap") != sha:
        return None, "profiles/%s is of other kernels (pemap_* sources %s, now %s): re-run tools/profile.sh" % (name, pm.get("sha_pemap"), sha)
    src = "profiles/%s (device sources pemap_* %s)" % (name, sha)
    if kernel is None:
        tot = 0.0
        for cn in ("FETCH_SIZE", "WRITE_SIZE"):
            for k, v in pm[cn].items():
                if k.startswith("pm_"):       # the mapping path's kernels (not the index build or the generators)
                    tot += v["mean_KB_per_launch"] * v["launches"] / pm.get("steps", 3)
        return round(tot * 1024.0), src
    tot = 0.0
    hit = False
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in pm[cn].items():
            if k.startswith(kernel):
                tot += v["mean_KB_per_launch"]
                hit = True
    return (round(tot * 1024.0), src) if hit else (None, src)


def pecall_sq_fracs(n, kernel_ms, clock_ghz=2.4, simds=1024):
    """share of a resident run of the caller in which the VALUs (SQ_ACTIVE_INST_VALU) and the LDS pipes (SQ_ACTIVE_INST_LDS) of the chip's
    SIMDs were busy: the counters are in units of four cycles, summed over the SIMDs; per run of the n columns from
    profiles/r04_pecall_sq.json (tools/pcs_sq.sh), valid for these columns and these device sources; the clock is taken at its 2.4 GHz
    peak (the shares are lower bounds: under load the chip runs at 1.95-2.1)"""
    path = os.path.join(ROOT, "profiles", "r04_pecall_sq.json")
    try:
        pm = json.load(open(path))
    except (OSError, ValueError):
        return None
    sha = kernel_sources_sha("pecall_")
    if pm.get("sha_pecall") != sha or pm.get("columns") != n:
        return None
    v = sum(k.get("SQ_ACTIVE_INST_VALU", 0.0) for k in pm["kernels"].values())
    l = sum(k.get("SQ_ACTIVE_INST_LDS", 0.0) for k in pm["kernels"].values())
    cyc = simds * kernel_ms * 1e-3 * clock_ghz * 1e9
    out = {"valu_issue_frac": round(4.0 * v / cyc, 4), "lds_frac": round(4.0 * l / cyc, 4),
           "source": "profiles/r04_pecall_sq.json (device sources pecall_* %s), %d SIMDs at %.1f GHz" % (sha, simds, clock_ghz)}
    tr, _ = pecall_pmc_traffic(n)
    hb = (tr / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else 0.0
    out["bound"] = max((("valu", out["valu_issue_frac"]), ("lds", out["lds_frac"]), ("hbm", hb)), key=lambda x: x[1])[0]
    return out


def pecall_pmc_traffic(n):
    """HBM bytes of one resident run of the caller's kernels over the n columns, from profiles/r04_pecall_pmc.json (tools/profile_pecall.sh:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pecall_kernel_time.py, 5 runs each) -- valid while the columns and the device
    sources are the profiled ones.  The kernels stream their columns (coalesced 768-byte rows in, 600 bytes out): WRITE_SIZE is taken as
    rocprofv3 reports it (KB); FETCH_SIZE is DOUBLED, as MI355X_MICROARCH.md prescribes for coalesced streaming reads on gfx950 (128-byte
    requests tallied at 64) -- the counter shows 395 bytes per column for rows of 770 bytes that the shortcut kernel reads whole."""
    path = os.path.join(ROOT, "profiles", "r04_pecall_pmc.json")
    try:
        pm = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    sha = kernel_sources_sha("pecall_")
    if pm.get("sha_pecall") != sha:
        return None, "profiles/r04_pecall_pmc.json is of other kernels (pecall_* sources %s, now %s): re-run tools/profile_pecall.sh" % (pm.get("sha_pecall"), sha)
    if pm.get("columns") != n:
        return None, "profiles/r04_pecall_pmc.json was taken on %s columns" % pm.get("columns")
    tot = 0.0
    for cn, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for k, v in pm[cn].items():
            if k.startswith("pcs_"):
                tot += factor * v["mean_KB_per_launch"] * v["launches"] / pm.get("steps", 5)
    return round(tot * 1024.0), "profiles/r04_pecall_pmc.json (device sources pecall_* %s; FETCH_SIZE x 2, the guide's correction for coalesced streams)" % sha


def physical_cores():
    """physical cores among the CPUs this process may run on: distinct sets of hardware-thread siblings (sysfs), the CPU count if
    sysfs does not say"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            seen.add(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip())
        except OSError:
            seen.add("cpu%d" % c)
    return max(1, len(seen))


def cpu_baseline(dev, a, B, leg, warmup):
    """the oracle (CPU restatement of the reference loop, pthreads) on a bounded sample of the same workload, same index;
    its coordinates and classes are also compared with what the GPU produced for the same reads (checker role)"""
    import oracle_py
    t0 = time.time()
    pos_index = dev.read_buffer(0, np.uint32)
    mers = dev.read_buffer(1, np.uint32)
    genome = dev.read_buffer(2, np.uint8)
    cs = dev.read_buffer(3, np.uint32)
    t_copy = time.time() - t0
    r1, l1, r2, l2 = leg["host"]
    ix = dict(pos_index=pos_index, mers=mers, genome=genome, contig_starts=cs)
    o = oracle_py.Oracle(ix, paired=True, min_dist=0, max_dist=500, min_align=0.85)

    def sample(threads, seconds, batch):
        """the oracle on the first pairs of timed batch `batch`, compared with what the GPU returned for them through the seam"""
        first = ((warmup + batch) % leg["host_batches"]) * B
        g1, g2, gt = leg["results"][warmup + batch]
        n = 4000
        done = 0
        spent = 0.0
        mism = 0
        while done + n <= B:
            lo, hi = first + done, first + done + n
            t1 = time.time()
            m1, m2, mt, _, _ = o.map_batch(r1[lo:hi], l1[lo:hi], r2[lo:hi], l2[lo:hi], threads=threads)
            spent += time.time() - t1
            s = slice(done, done + n)
            mism += int((m1 != g1[s]).sum() + (m2 != g2[s]).sum() + (mt != gt[s]).sum())
            done += n
            if spent >= seconds:
                break
            rate = done / spent
            n = int(max(4000, min(rate * (seconds - spent), 200000)))
        return done, spent, mism
    phys = physical_cores()
    nt = a.cpu_threads if a.cpu_threads > 0 else phys
    done, spent, mism = sample(nt, a.cpu_seconds, 0)
    out = {"value": round(2 * done / spent / 1e6, 5), "unit": "M reads/s", "cores": nt, "kind": "port",
           "gpu_vs_cpu_mismatches": mism, "compared_pairs": done, "host_cpus": os.cpu_count(), "host_physical_cores": phys,
           "sample": "%d pairs of the first timed batch on %d threads (= the host's physical cores unless --cpu-threads says otherwise), "
                     "same index (copied back from HBM in %.1f s), %.1f s of wall time" % (done, nt, t_copy, spent)}
    # the authors' default thread count (map_directory_array.pl:100) and the one-GPU box's CPU share (16): short samples on later batches
    for j, t in enumerate((24, 16)):
        if len(leg["results"]) > warmup + 1 + j and t != nt:
            dj, sj, mj = sample(t, min(5.0, a.cpu_seconds), 1 + j)
            out["threads_%d" % t] = {"value": round(2 * dj / sj / 1e6, 5), "compared_pairs": dj, "gpu_vs_cpu_mismatches": mj}
            out["gpu_vs_cpu_mismatches"] += mj
            out["compared_pairs"] += dj
    return out


def pecall_columns(n_sites, n, seed=777, var_rate=0.001, depth=30, err=0.004):
    """BASELINE config 4 / SURVEY.md 8(d): 30x Poisson depth, 0.4 % per-base error, one variant per kb with genotypes drawn
    under Hardy-Weinberg; the six counters of the pileup record per (column, sample)"""
    rng = np.random.default_rng(seed)
    dom = rng.integers(0, 4, n_sites).astype(np.uint8)
    d = rng.poisson(depth, (n_sites, n))
    e = rng.binomial(d, err)                   # reads replaced by a uniformly random base
    good = d - e
    is_var = rng.random(n_sites) < var_rate
    q = rng.uniform(0.02, 0.5, n_sites)
    alt = (dom + rng.integers(1, 4, n_sites)) % 4
    dose = np.where(is_var[:, None], rng.binomial(2, q[:, None], (n_sites, n)), 0)      # copies of the alternative allele
    alt_reads = rng.binomial(good, dose / 2.0)
    reads = np.zeros((n_sites, n, 6), np.int32)
    idx = np.arange(n_sites)
    for s in range(n):
        reads[idx, s, dom] += good[:, s] - alt_reads[:, s]
        reads[idx, s, alt] += alt_reads[:, s]
        reads[idx, s, rng.integers(0, 4, n_sites)] += e[:, s]
    return reads.astype(np.uint16), dom


def pecaller_leg(a):
    """BASELINE config 4: the per-site caller (call_single_base's body, pecaller.c:1207-1691, fill_sample_like 2448-2507 inside it)
    on synthetic 30x pileup columns of 64 samples; device-resident and seam-inclusive rates, the kernel's roofline by SURVEY.md
    8(d)'s 7.9 KB per column, the CPU oracle on a sample of the same columns with calls and posteriors compared"""
    from pecaller_amd.pecall import PecallDev
    import oracle_py
    n, S = a.pecall_sites, 64
    t0 = time.time()
    reads, dom = pecall_columns(n, S)
    t_gen = time.time() - t0
    pc = PecallDev(0)
    pc.call_sites(reads[:20000], dom[:20000])        # warm-up: tables, allocations
    # the seam: the columns and the result arrays are the host program's tile buffers -- allocated once and page-locked once, as
    # pecaller_hip does (pecall_dev_pin_host; the reference allocates its per-slot arrays once too, pecaller.c:1149-1206) -- and
    # every call moves host columns in and calls + posteriors out
    out = pc.out_arrays(n, S)
    for arr in (reads, dom) + out:
        pc.pin_host(arr)
    pc.call_sites(reads, dom, out=out)               # (device arrays of the full size, staging: allocated here)
    t0 = time.perf_counter()
    call, post, typ, ac, npass = pc.call_sites(reads, dom, out=out)
    seam_dt = time.perf_counter() - t0
    # ... and with the posteriors as a list of the columns in which one is not 1 (pecall_dev_call_sites_sparse): 94 bytes per column
    # come back instead of 606
    sp_site, sp_rows = np.zeros(max(1024, n // 8), np.uint32), np.zeros((max(1024, n // 8), S))
    pc.pin_host(sp_site)
    pc.pin_host(sp_rows)
    pc.call_sites_sparse(reads, dom, out=out, sparse_out=(sp_site, sp_rows))
    t0 = time.perf_counter()
    s_call, (s_site, s_rows), s_typ, s_ac, s_np = pc.call_sites_sparse(reads, dom, out=out, sparse_out=(sp_site, sp_rows))
    seam_sparse_dt = time.perf_counter() - t0
    sparse_cols = len(s_site)
    s_call = s_call.copy()
    for arr in (reads, dom, sp_site, sp_rows) + out:
        pc.unpin_host(arr)
    # the same call from pageable memory (the library stages through its own pinned buffers)
    out2 = pc.out_arrays(n, S)
    t0 = time.perf_counter()
    c3, p3 = pc.call_sites(reads, dom, out=out2)[:2]
    seam_pageable_dt = time.perf_counter() - t0
    assert np.array_equal(c3, call) and np.array_equal(p3, post)
    want = np.nonzero((p3 != 1.0).any(axis=1))[0]
    assert np.array_equal(s_call, c3) and np.array_equal(s_site, want.astype(np.uint32)) and np.array_equal(s_rows, p3[want])
    del out2, c3, p3
    pc.sites_stage(reads, dom)
    kms = [pc.sites_run() for _ in range(3)]
    kernel_ms = float(np.mean(kms))
    c2, p2 = pc.sites_collect()[:2]
    assert np.array_equal(c2, call) and np.array_equal(p2, post)
    # beyond a lane per sample (round 4): 128 samples -- the shortcut kernel with two samples per lane -- and 256 -- a chunk of 64 samples
    # at a time, the unsettled samples' likelihoods parked in LDS for the small beam; resident columns, a sample of them against the oracle
    def wide_point(nw, SW, seed, chk, form):
        wr, wd = pecall_columns(nw, SW, seed=seed)
        pc.sites_stage(wr, wd)
        wms = [pc.sites_run() for _ in range(2)]
        wc, wp = pc.sites_collect()[:2]
        chk = min(nw, chk)
        oc, op = oracle_py.call_sites(wr[:chk], wd[:chk])[:2]
        return {"samples": SW, "columns": nw, "value": round(nw / (min(wms) * 1e-3) / 1e6, 4), "unit": "M columns/s", "kernel_ms": round(min(wms), 2), "form": form,
                "calls_equal_oracle": bool(np.array_equal(wc[:chk], oc)), "max_abs_dposterior": float(np.max(np.abs(wp[:chk] - op))), "compared_columns": chk}
    wide = wide256 = None
    if a.pecall_wide_sites > 0:
        wide = wide_point(a.pecall_wide_sites, 128, 778, 2000,
                          "pcs_fast_kernel<2048, 2> (two samples per lane) + pcs_call_kernel<2> of the columns it and pcs_heavy_kernel list")
    if a.pecall_wide256_sites > 0:
        wide256 = wide_point(a.pecall_wide256_sites, 256, 779, 600,
                             "pcs_fast_kernel<2048, 4> (a chunk of 64 samples at a time) + pcs_call_kernel<4> of the columns it and pcs_heavy_kernel list")
    pc.close()
    # CPU: the oracle, one caller per thread on disjoint slices of the same columns (columns are independent)
    nt = a.cpu_threads if a.cpu_threads > 0 else physical_cores()
    per = 2000
    stop_at = time.time() + a.pecall_cpu_seconds
    done = []

    def work(i):
        k = i
        got = []
        while time.time() < stop_at and (k + 1) * per <= n:
            s = slice(k * per, (k + 1) * per)
            got.append((s, oracle_py.call_sites(reads[s], dom[s], threads=1)))
            k += nt
        return got
    t0 = time.time()
    with ThreadPoolExecutor(nt) as ex:
        for got in ex.map(work, range(nt)):
            done += got
    cpu_dt = time.time() - t0
    m = sum(s.stop - s.start for s, _ in done)
    dmax = 0.0
    calls_eq = types_eq = True
    for s, (oc, op, otyp, oac, onp) in done:
        dmax = max(dmax, float(np.max(np.abs(post[s] - op))))
        calls_eq = calls_eq and bool(np.array_equal(call[s], oc)) and bool(np.array_equal(ac[s], oac))
        types_eq = types_eq and bool(np.array_equal(typ[s], otyp))
    achieved = PECALL_BYTES_PER_SITE * n / (kernel_ms * 1e-3) / 1e9
    traffic, tsrc = pecall_pmc_traffic(n)
    sq = pecall_sq_fracs(n, kernel_ms)
    return {"metric": "M pileup columns called/sec, 64 samples, 30x", "value": round(n / (kernel_ms * 1e-3) / 1e6, 4), "unit": "M columns/s",
            "timed_region": "pcs_fast_kernel + pcs_call_kernel on columns resident in HBM, in chunks of 2^18 columns, the beam searches of the chunks on four streams beside the next chunks' shortcut kernels (HIP events around all of them, mean of 3 runs)",
            "seam_value": round(n / seam_dt / 1e6, 4), "seam_timed_region": "pecall_dev_call_sites: host columns in, calls + posteriors out (PCIe included; the caller's buffers page-locked once, copies and kernels of neighbouring chunks side by side)",
            "seam_pageable_value": round(n / seam_pageable_dt / 1e6, 4),
            "seam_sparse_value": round(n / seam_sparse_dt / 1e6, 4), "seam_sparse_region": "pecall_dev_call_sites_sparse: as seam_value, the posteriors as the list of the %d columns (of %d) in which one differs from 1; checked against the dense call" % (sparse_cols, n),
            "dtype": "f64", "data": "synthetic", "n_gpus": 1,
            "config": {"workload": "%d pileup columns x %d samples, 30x Poisson depth, 0.4%% error, 1 variant/kb under HWE, seed 777, "
                                   "prob_to_call 0.95, theta 0.001, diploid, no pedigree" % (n, S), "generated_in_s": round(t_gen, 1)},
            "variant_rows": int((typ > 0).sum()), "passes_histogram": np.bincount(npass).tolist(), "samples_128": wide, "samples_256": wide256,
            "kernel_ms_runs": [round(x, 2) for x in kms],
            "roofline": {"bound": sq["bound"] if sq else "hbm", "kernel": "pcs_fast_kernel+pcs_call_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": round(kernel_ms, 3),
                         # what the kernels are bound by is not HBM (their traffic is a fifth of the algorithmic figure: the likelihood table
                         # look-ups stay in LDS): the share of the run in which the VALUs / the LDS pipes of the 1,024 SIMDs were busy, from
                         # the SQ counters of profiles/r04_pecall_sq.json, beside the HBM share of the counted traffic; `bound` = the largest
                         "valu_issue_frac": sq["valu_issue_frac"] if sq else None, "lds_frac": sq["lds_frac"] if sq else None,
                         "hbm_traffic_frac": (round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None),
                         "sq_source": sq["source"] if sq else None,
                         "algorithmic_bytes_per_launch": PECALL_BYTES_PER_SITE * n, "bytes_per_site": PECALL_BYTES_PER_SITE},
            "cpu_baseline": {"value": round(m / cpu_dt / 1e6, 5), "unit": "M columns/s", "cores": nt, "kind": "port",
                             "sample": "%d of the same columns, %d oracle callers on %d threads, %.1f s" % (m, nt, nt, cpu_dt),
                             "calls_and_allele_counts_equal": calls_eq, "site_types_equal": types_eq, "max_abs_dposterior": dmax,
                             "tolerance": 1e-6}}


if __name__ == "__main__":
    main()
