"""One rank of the multi-GPU path on real device objects (run by tests/test_gpu_dist.py under torch.distributed.run, two ranks
on the one GPU of the box, gloo): the start-up of bench.py -- rank 0 builds the index, the others index_alloc, the broadcast
lands in torch views of the library's device buffers, index_commit -- then every rank maps its shard of the golden reads ON THE
DEVICE, the u32 counters are summed with pecaller_amd.dist.reduce_pileup on the device buffers, and rank 0 writes what a
single process would have to produce."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fixtures
    from pecaller_amd import PemapDev, dist as pd
    dev = PemapDev(0)
    if rank == 0:
        ix = fixtures.index()
        dev.build_index(ix["genome"], ix["contig_len"])
    info = [dev.index_info() if rank == 0 else None]
    dist.broadcast_object_list(info, src=0)
    if rank != 0:
        dev.index_alloc(*info[0])
    pd.broadcast_tensors(dist, [pd.device_tensor(torch, dev, w) for w in (0, 1, 2, 3)], src=0, chunk=1 << 26)
    torch.cuda.synchronize()
    if rank != 0:
        dev.index_commit()
    assert dev.index_info() == info[0]
    r1, l1, r2, l2 = fixtures.reads("r150")
    n = len(l1)
    lo, hi = pd.shard_range(n, rank, world)
    dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
    cnt = pd.device_tensor(torch, dev, 4)
    # the reference's counters are unsigned short and wrap (pemapper.c:53-58): every rank starts 600 of them at 40,000, so that
    # the sum of two ranks passes 65,535 and the writer's truncation is what a single u16 counter would have done
    # (words 0 .. 599 of the first plane: the low halves are the counters of positions 0, 2, .. 1198 in the reference base's column)
    cnt[:600] += 40000
    torch.cuda.synchronize()
    m1, m2, mt = dev.map_batch(r1[lo:hi], l1[lo:hi], r2[lo:hi], l2[lo:hi])
    pd.reduce_pileup(dist, cnt, chunk=1 << 22)
    torch.cuda.synchronize()
    summ = pd.merge_summaries(dist, torch, dev.summary())
    gathered = [None] * world
    dist.all_gather_object(gathered, (m1, m2, mt))
    counts, ins = dev.fetch_pileup()          # every rank holds the sum; insertions stay per rank (concatenated below)
    all_ins = [None] * world
    dist.all_gather_object(all_ins, ins)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dist_result.npz"), counts=counts, summary=summ,
                 m1=np.concatenate([g[0] for g in gathered]), m2=np.concatenate([g[1] for g in gathered]),
                 mt=np.concatenate([g[2] for g in gathered]),
                 ins_pos=np.array([p for part in all_ins for p, _ in part], np.int64),
                 ins_seq=np.array([s for part in all_ins for _, s in part], dtype="S300"))
    dist.barrier()
    dev.close()
    dist.destroy_process_group()
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
