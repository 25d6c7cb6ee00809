"""Multi-GPU plumbing of the mapping path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by reads: every rank holds a full index replica and its own pileup counters, nothing is exchanged
while mapping.  Two collectives exist, both outside the timed loop:
  * broadcast_index: rank `src` loaded or built the index; the others receive pos_index / mers / genome / contig table;
  * reduce_pileup:   sum of the counters over ranks (the reference's shared all_base_list, pemapper.c:156).  The
    reference's counters are unsigned short and wrap; the device keeps them the same way, two 16-bit counters to a 32-bit
    word (PmPile, pemap_kernels.hip.h).  The two halves of every word are summed separately in 32 bits and truncated to 16
    when they are packed again: the arithmetic of one u16 counter that saw all ranks' increments.
shard_range gives the contiguous slice of reads a rank maps; .mfile entries are concatenated in rank order.
The tensors may live on the GPU (wrapping the device pointers of pemap_dev_buffer) or on the CPU (gloo tests).
"""
import numpy as np


def shard_range(n_reads, rank, world):
    """contiguous, balanced: the first n_reads % world ranks get one extra read"""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DevArray:
    """exposes a device pointer handed out by the C-ABI to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n_items, typestr):
        self.__cuda_array_interface__ = {"shape": (n_items,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_tensor(torch, dev, which):
    """torch view (no copy) of one of the object's resident buffers; u32 arrays are viewed as int32"""
    ptr, nbytes = dev.buffer(which)
    if which == 2:
        return torch.as_tensor(DevArray(ptr, nbytes, "|u1"), device="cuda")
    return torch.as_tensor(DevArray(ptr, nbytes // 4, "<i4"), device="cuda")


def _chunks(t, chunk):
    n = t.numel()
    for o in range(0, n, chunk):
        yield t[o:min(n, o + chunk)]


def _host_bounce(dist, t):
    """gloo moves host memory only: a device tensor goes through a host copy (rehearsals of several ranks on one GPU);
    RCCL ("nccl") works on the device buffers directly"""
    return t.is_cuda and dist.get_backend() != "nccl"


def broadcast_tensors(dist, tensors, src=0, chunk=1 << 28):
    for t in tensors:
        for c in _chunks(t, chunk):
            if _host_bounce(dist, c):
                h = c.cpu()
                dist.broadcast(h, src=src)
                if dist.get_rank() != src:
                    c.copy_(h)
            else:
                dist.broadcast(c, src=src)


def reduce_pileup(dist, counts, dst=None, chunk=1 << 27):
    """in-place sum over all ranks (all_reduce, or reduce to `dst`: only rank `dst` then holds the sum, the others keep their own
    counters) of the counters, an int32 view of words that hold two 16-bit counters each: the halves are summed apart and packed
    again modulo 2^16"""
    for c in _chunks(counts.view(-1), chunk):
        h = c.cpu() if _host_bounce(dist, c) else c
        lo = h & 0xFFFF
        hi = (h >> 16) & 0xFFFF
        for part in (lo, hi):
            if dst is None:
                dist.all_reduce(part, op=dist.ReduceOp.SUM)
            else:
                dist.reduce(part, dst=dst, op=dist.ReduceOp.SUM)
        if dst is not None and dist.get_rank() != dst:
            continue            # (a reduce leaves the other ranks' buffers unspecified: their own counters stay as they were)
        lo &= 0xFFFF
        hi &= 0xFFFF
        # (hi << 16 in 32 bits: values of 0x8000 and more land in the sign bit, which is the bit pattern wanted)
        packed = lo | (hi << 16)
        c.copy_(packed)
    return counts


def counts_to_u16(counts_i32):
    """the 16-bit counters of the packed words, in memory order (numpy or torch int32 array)"""
    a = counts_i32.cpu().numpy() if hasattr(counts_i32, "cpu") else np.asarray(counts_i32)
    return np.ascontiguousarray(a).view(np.uint16)


def pack_counts(counts_u16):
    """[positions][6] u16 columns -> six planes, two positions to a word, each plane padded to a whole number of 256-byte blocks
    (pemap_capi.hip: pile_plane_words).  The device's planes additionally rotate the four base columns by the reference letter
    (PmPile, pemap_kernels.hip.h) -- a bijection per position that the element-wise sum over ranks does not see; host-side
    rehearsals (tests/test_dist_gloo.py) use this unrotated form"""
    n = counts_u16.shape[0]
    plane_words = ((n + 2) // 2 + 63) & ~63
    out = np.zeros((6, 2 * plane_words), np.uint16)
    out[:, :n] = np.asarray(counts_u16, np.uint16).T
    return out.reshape(-1).view(np.int32)


def unpack_counts(words_i32, n_positions):
    """inverse of pack_counts"""
    a = np.ascontiguousarray(words_i32).view(np.uint16).reshape(6, -1)
    return np.ascontiguousarray(a[:, :n_positions].T)


def merge_summaries(dist, torch, summary13):
    """scalar totals and class histogram are plain sums over ranks (pemapper.c:1238-1265).  RCCL has no host tensors: under
    "nccl" the 13 numbers travel through the device."""
    t = torch.as_tensor(np.asarray(summary13, dtype=np.int64))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
