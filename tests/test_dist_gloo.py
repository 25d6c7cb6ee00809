"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): reads sharded by rank, every rank maps its slice with
the oracle standing in for the device (same results by the parity tests), then the two collectives of
pecaller_amd.dist -- index broadcast and pileup sum -- must reproduce the single-process result bit for bit."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
N = 1200


def _worker(rank, world, port, q):
    import fixtures
    import oracle_py
    from pecaller_amd import dist as pd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ix = fixtures.index()
    # index broadcast: rank 0 owns the arrays, the others start from zeros and must end up identical
    names = ["mers", "ukmer", "ustart", "genome", "contig_starts"]
    ts = []
    for k in names:
        a = ix[k]
        t = torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a.copy())
        if rank != 0:
            t = torch.zeros_like(t)
        ts.append(t)
    pd.broadcast_tensors(dist, ts, src=0, chunk=1 << 18)
    mine = {}
    for k, t in zip(names, ts):
        a = t.numpy()
        mine[k] = a.view(np.uint32) if ix[k].dtype == np.uint32 else a
        assert np.array_equal(mine[k], ix[k]), k
    # map this rank's slice
    r1, l1, r2, l2 = fixtures.reads("r150")
    lo, hi = pd.shard_range(N, rank, world)
    o = oracle_py.Oracle(mine, paired=True)
    m1, m2, mt, _, _ = o.map_batch(r1[lo:hi], l1[lo:hi], r2[lo:hi], l2[lo:hi], threads=2)
    counts = torch.from_numpy(pd.pack_counts(o.counts()).copy())
    pd.reduce_pileup(dist, counts, chunk=1 << 20)
    summ = pd.merge_summaries(dist, torch, o.summary())
    gathered = [None] * world
    dist.all_gather_object(gathered, (m1, m2, mt, o.insertions()))
    if rank == 0:
        q.put((pd.unpack_counts(counts.numpy(), len(mine["genome"])), summ, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one():
    import fixtures
    import oracle_py
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    counts, summ, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ix = fixtures.index()
    r1, l1, r2, l2 = fixtures.reads("r150")
    o = oracle_py.Oracle(ix, paired=True)
    m1, m2, mt, _, _ = o.map_batch(r1[:N], l1[:N], r2[:N], l2[:N], threads=4)
    assert np.array_equal(counts.reshape(-1, 6), o.counts())
    assert np.array_equal(summ, o.summary())
    assert np.array_equal(np.concatenate([g[0] for g in gathered]), m1)
    assert np.array_equal(np.concatenate([g[1] for g in gathered]), m2)
    assert np.array_equal(np.concatenate([g[2] for g in gathered]), mt)
    assert sorted(sum((g[3] for g in gathered), [])) == o.insertions()
    assert np.array_equal(m1, fixtures.golden_m("r150", 1)[:N])


class _OneRank:
    """a stand-in for torch.distributed with one rank whose all_reduce adds a second rank's tensor (the halves arrive in the
    order reduce_pileup sends them: low, high)"""

    class ReduceOp:
        SUM = 0

    def __init__(self, other_lo, other_hi):
        self.parts = [other_lo, other_hi]

    def get_backend(self):
        return "gloo"

    def all_reduce(self, t, op=None):
        t += self.parts.pop(0)


def test_u16_wrap_survives_the_reduction():
    """the reference's counters are unsigned short: the device packs two to a word, the reduction sums the halves apart and
    truncates once -- every counter wraps on its own, a low half's overflow does not reach its neighbour"""
    import torch
    from pecaller_amd import dist as pd
    a = np.array([[65535, 40000, 1, 0, 65535, 7], [1, 2, 3, 65535, 65535, 65535]], dtype=np.uint16)
    b = np.array([[1, 40000, 65535, 0, 65535, 9], [65535, 0, 0, 1, 65535, 1]], dtype=np.uint16)
    wa, wb = pd.pack_counts(a), pd.pack_counts(b)
    t = torch.from_numpy(wa.copy())
    tb = torch.from_numpy(wb.copy())
    pd.reduce_pileup(_OneRank(tb & 0xFFFF, (tb >> 16) & 0xFFFF), t)
    expect = ((a.astype(np.uint32) + b.astype(np.uint32)) & 0xFFFF).astype(np.uint16)
    assert np.array_equal(pd.unpack_counts(t.numpy(), 2), expect)
    assert np.array_equal(pd.unpack_counts(pd.pack_counts(a), 2), a)
