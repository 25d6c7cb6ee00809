"""ctypes mirror of include/pemap_hip.h (same names, same argument meaning, errors raised as PemapError)."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (PEMAP_LIB: another build of the same library, for A/B runs of compile-time variants -- tools/variants.sh)
LIB_PATH = os.environ.get("PEMAP_LIB") or os.path.join(HERE, "libpemap_hip.so")
MAX_HITS = 200
MIN_READ = 16
MAX_READ = 278
CLASS_NAMES = ["UNIQUE_MATE", "UNIQUE_SLIP", "UNIQUE_SINGLE", "UNIQUE_MIS", "NON_MATE", "NON_MIS", "FRAG_MIS", "NON_NO",
               "NEITHER_MAP"]

# every symbol include/pemap_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "pemap_dev_last_error", "pemap_dev_create", "pemap_dev_destroy", "pemap_dev_load_index", "pemap_dev_build_index",
    "pemap_dev_build_index_resident", "pemap_dev_index_alloc", "pemap_dev_index_commit", "pemap_dev_set_lookup_replicas",
    "pemap_dev_lookup_replicas", "pemap_dev_buffer",
    "pemap_dev_index_info", "pemap_dev_read_buffer", "pemap_dev_set_params", "pemap_dev_map_batch",
    "pemap_dev_submit_batch", "pemap_dev_wait_batch", "pemap_dev_pin_host", "pemap_dev_unpin_host",
    "pemap_dev_stage_reads", "pemap_dev_run", "pemap_dev_run_slice", "pemap_dev_collect", "pemap_dev_sync",
    "pemap_dev_synth_genome", "pemap_dev_synth_reads", "pemap_dev_synth_reads_indel", "pemap_dev_staged_reads", "pemap_dev_staged_info",
    "pemap_dev_free", "pemap_dev_fetch_pileup", "pemap_dev_fetch_records", "pemap_dev_reset_pileup", "pemap_dev_summary",
    "pemap_dev_run_stats", "pemap_dev_debug_hits",
    "pecall_dev_create", "pecall_dev_destroy", "pecall_dev_last_error", "pecall_dev_site_like", "pecall_dev_stage",
    "pecall_dev_run", "pecall_dev_collect", "pecall_dev_call_sites", "pecall_dev_call_sites_sparse", "pecall_dev_set_pedigree",
    "pecall_dev_sites_stage", "pecall_dev_sites_run", "pecall_dev_sites_collect", "pecall_dev_pin_host", "pecall_dev_unpin_host",
]

PILE_DT = np.dtype([("pos", "<u4"), ("c", "<u2", (6,))])


class PemapError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen the in-tree library.  There is deliberately no fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PemapError("%s is missing: run `python -m pecaller_amd.build` (hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i, u64, dbl = C.c_void_p, C.c_int, C.c_uint64, C.c_double
        L.pemap_dev_last_error.restype = C.c_char_p
        L.pemap_dev_last_error.argtypes = [vp]
        L.pemap_dev_create.argtypes = [C.POINTER(vp), i]
        L.pemap_dev_destroy.argtypes = [vp]
        L.pemap_dev_destroy.restype = None
        L.pemap_dev_load_index.argtypes = [vp, vp, vp, u64, vp, u64, vp, i, i]
        L.pemap_dev_build_index.argtypes = [vp, vp, u64, vp, i, i]
        L.pemap_dev_build_index_resident.argtypes = [vp, vp, u64, vp, i, i]
        L.pemap_dev_index_alloc.argtypes = [vp, u64, u64, i, i]
        L.pemap_dev_index_commit.argtypes = [vp]
        L.pemap_dev_set_lookup_replicas.argtypes = [vp, i]
        L.pemap_dev_lookup_replicas.argtypes = [vp, C.POINTER(i), C.POINTER(u64)]
        L.pemap_dev_buffer.argtypes = [vp, i, C.POINTER(vp), C.POINTER(u64)]
        L.pemap_dev_index_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(i), C.POINTER(i)]
        L.pemap_dev_read_buffer.argtypes = [vp, i, u64, vp, u64]
        L.pemap_dev_set_params.argtypes = [vp, i, i, i, dbl, i]
        L.pemap_dev_map_batch.argtypes = [vp, vp, vp, vp, vp, i, i, vp, vp, vp]
        L.pemap_dev_submit_batch.argtypes = [vp, vp, vp, vp, vp, i, i, vp, vp, vp, C.POINTER(u64)]
        L.pemap_dev_wait_batch.argtypes = [vp, u64]
        L.pemap_dev_pin_host.argtypes = [vp, vp, u64]
        L.pemap_dev_unpin_host.argtypes = [vp, vp]
        L.pemap_dev_stage_reads.argtypes = [vp, vp, vp, vp, vp, i, i]
        L.pemap_dev_run.argtypes = [vp, i]
        L.pemap_dev_run_slice.argtypes = [vp, i, i, i]
        L.pemap_dev_collect.argtypes = [vp, vp, vp, vp]
        L.pemap_dev_sync.argtypes = [vp]
        L.pemap_dev_synth_genome.argtypes = [vp, u64, u64, i, dbl, C.POINTER(vp), vp]
        L.pemap_dev_synth_reads.argtypes = [vp, u64, i, i, i, dbl, dbl, u64]
        L.pemap_dev_synth_reads_indel.argtypes = [vp, u64, i, i, i, dbl, dbl, u64]
        L.pemap_dev_staged_reads.argtypes = [vp, vp, vp, vp, vp, i]
        L.pemap_dev_staged_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
        L.pemap_dev_free.argtypes = [vp, vp]
        L.pemap_dev_fetch_pileup.argtypes = [vp, vp, vp, vp]
        L.pemap_dev_fetch_records.argtypes = [vp, u64, u64, vp, u64, C.POINTER(u64)]
        L.pemap_dev_reset_pileup.argtypes = [vp]
        L.pemap_dev_summary.argtypes = [vp, vp]
        L.pemap_dev_run_stats.argtypes = [vp, vp, vp]
        L.pemap_dev_debug_hits.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


INS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_char_p, C.c_int)


def _p(a):
    return None if a is None else a.ctypes.data


class PemapDev:
    """One device object (= one GPU), mirroring the pemap_dev_* entry points one to one."""

    def __init__(self, device_id=0):
        self.L = load_library()
        h = C.c_void_p()
        if self.L.pemap_dev_create(C.byref(h), device_id):
            raise PemapError(self.L.pemap_dev_last_error(None).decode())
        self.h = h
        self.paired = True
        self.gsize = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.pemap_dev_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ck(self, rc):
        if rc:
            raise PemapError(self.L.pemap_dev_last_error(self.h).decode())

    # ---- index
    def load_index(self, pos_index, mers, genome, contig_starts, idepth=16):
        assert pos_index.dtype == np.uint32 and len(pos_index) == (1 << 32) + 1
        self._ck(self.L.pemap_dev_load_index(self.h, _p(pos_index), _p(mers), len(mers), _p(genome), len(genome),
                                             _p(contig_starts), len(contig_starts) - 1, idepth))
        self.gsize = len(genome)

    def build_index(self, genome, contig_len, bisulfite=False):
        genome = np.ascontiguousarray(genome, dtype=np.uint8)
        cl = np.ascontiguousarray(contig_len, dtype=np.uint32)
        self._ck(self.L.pemap_dev_build_index(self.h, _p(genome), len(genome), _p(cl), len(cl), int(bisulfite)))
        self.gsize = len(genome)

    def build_index_resident(self, d_genome, genome_size, contig_len, bisulfite=False):
        cl = np.ascontiguousarray(contig_len, dtype=np.uint32)
        self._ck(self.L.pemap_dev_build_index_resident(self.h, d_genome, genome_size, _p(cl), len(cl), int(bisulfite)))
        self.gsize = genome_size

    def index_alloc(self, n_mers, genome_size, n_contigs, idepth=16):
        self._ck(self.L.pemap_dev_index_alloc(self.h, n_mers, genome_size, n_contigs, idepth))
        self.gsize = genome_size

    def index_commit(self):
        self._ck(self.L.pemap_dev_index_commit(self.h))

    def set_lookup_replicas(self, n):
        """-1 = when the memory is there (default), 0 = the reference's table only, 8 = required."""
        self._ck(self.L.pemap_dev_set_lookup_replicas(self.h, int(n)))

    def lookup_replicas(self):
        n = C.c_int(0)
        b = C.c_uint64(0)
        self._ck(self.L.pemap_dev_lookup_replicas(self.h, C.byref(n), C.byref(b)))
        return n.value, b.value

    def index_info(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_int(), C.c_int()
        self._ck(self.L.pemap_dev_index_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def buffer(self, which):
        p, n = C.c_void_p(), C.c_uint64()
        self._ck(self.L.pemap_dev_buffer(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_buffer(self, which, dtype, offset_bytes=0, n_bytes=None):
        _, nb = self.buffer(which)
        if n_bytes is None:
            n_bytes = nb - offset_bytes
        out = np.empty(n_bytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._ck(self.L.pemap_dev_read_buffer(self.h, which, offset_bytes, _p(out), n_bytes))
        return out

    # ---- mapping
    def set_params(self, paired=True, min_dist=0, max_dist=500, min_align=0.85, bisulfite=False):
        self.paired = bool(paired)
        self._ck(self.L.pemap_dev_set_params(self.h, int(paired), min_dist, max_dist, min_align, int(bisulfite)))

    def map_batch(self, r1, l1, r2=None, l2=None):
        n = len(l1)
        m1 = np.zeros(n, np.uint32)
        m2 = np.zeros(n, np.uint32) if self.paired else None
        mt = np.zeros(n, np.int32)
        self._ck(self.L.pemap_dev_map_batch(self.h, _p(r1), _p(l1), _p(r2), _p(l2), n, r1.shape[1], _p(m1), _p(m2), _p(mt)))
        self._n = n
        return m1, m2, mt

    def submit_batch(self, r1, l1, r2=None, l2=None):
        """-> ticket; the result arrays are filled when wait_batch(ticket) returns (they, and the inputs, are kept alive here)"""
        n = len(l1)
        m1 = np.zeros(n, np.uint32)
        m2 = np.zeros(n, np.uint32) if self.paired else None
        mt = np.zeros(n, np.int32)
        t = C.c_uint64()
        self._ck(self.L.pemap_dev_submit_batch(self.h, _p(r1), _p(l1), _p(r2), _p(l2), n, r1.shape[1], _p(m1), _p(m2), _p(mt),
                                               C.byref(t)))
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[t.value] = (m1, m2, mt, r1, l1, r2, l2)
        return t.value

    def wait_batch(self, ticket):
        self._ck(self.L.pemap_dev_wait_batch(self.h, ticket))
        m1, m2, mt = self._inflight.pop(ticket)[:3]
        return m1, m2, mt

    def pin_host(self, a):
        self._ck(self.L.pemap_dev_pin_host(self.h, a.ctypes.data, a.nbytes))

    def unpin_host(self, a):
        self._ck(self.L.pemap_dev_unpin_host(self.h, a.ctypes.data))

    def stage_reads(self, r1, l1, r2=None, l2=None):
        self._ck(self.L.pemap_dev_stage_reads(self.h, _p(r1), _p(l1), _p(r2), _p(l2), len(l1), r1.shape[1]))
        self._n = len(l1)

    def run(self, sync=True):
        self._ck(self.L.pemap_dev_run(self.h, int(sync)))

    def run_slice(self, first, n, sync=True):
        self._ck(self.L.pemap_dev_run_slice(self.h, first, n, int(sync)))
        self._n = n

    def sync(self):
        self._ck(self.L.pemap_dev_sync(self.h))

    def collect(self, n=None):
        n = self._n if n is None else n
        m1 = np.zeros(n, np.uint32)
        m2 = np.zeros(n, np.uint32) if self.paired else None
        mt = np.zeros(n, np.int32)
        self._ck(self.L.pemap_dev_collect(self.h, _p(m1), _p(m2), _p(mt)))
        return m1, m2, mt

    def synth_genome(self, seed, genome_size, n_contigs, repeat_frac=0.5):
        p = C.c_void_p()
        cl = np.zeros(n_contigs, np.uint32)
        self._ck(self.L.pemap_dev_synth_genome(self.h, seed, genome_size, n_contigs, repeat_frac, C.byref(p), _p(cl)))
        return p.value, cl

    def free(self, d_ptr):
        self._ck(self.L.pemap_dev_free(self.h, d_ptr))

    def synth_reads(self, seed, n, read_len, paired=True, sub_rate=0.01, indel_rate=0.0002, first_read=0):
        self.paired = bool(paired)
        self._ck(self.L.pemap_dev_synth_reads(self.h, seed, n, read_len, int(paired), sub_rate, indel_rate, first_read))
        self._n = n

    def synth_reads_indel(self, seed, n, read_len, paired=True, sub_rate=0.01, indel_read_frac=0.05, first_read=0):
        self.paired = bool(paired)
        self._ck(self.L.pemap_dev_synth_reads_indel(self.h, seed, n, read_len, int(paired), sub_rate, indel_read_frac, first_read))
        self._n = n

    def staged_info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.L.pemap_dev_staged_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, bool(c.value)

    def staged_reads(self):
        n, stride, paired = self.staged_info()
        r1 = np.zeros((n, stride), np.uint8)
        l1 = np.zeros(n, np.int32)
        r2 = np.zeros((n, stride), np.uint8) if paired else None
        l2 = np.zeros(n, np.int32) if paired else None
        self._ck(self.L.pemap_dev_staged_reads(self.h, _p(r1), _p(l1), _p(r2), _p(l2), stride))
        return r1, l1, r2, l2

    # ---- results
    def fetch_pileup(self, want_counts=True):
        counts = np.zeros((self.gsize, 6), np.uint16) if want_counts else None
        ins = []

        def cb(user, pos, seq, ln):
            ins.append((int(pos), seq[:ln]))
        cbf = INS_CB(cb)
        self._ck(self.L.pemap_dev_fetch_pileup(self.h, _p(counts), C.cast(cbf, C.c_void_p), None))
        return counts, sorted(ins)

    def fetch_records(self, first=0, count=None):
        count = self.gsize - first if count is None else count
        n = C.c_uint64()
        self._ck(self.L.pemap_dev_fetch_records(self.h, first, count, None, 0, C.byref(n)))
        out = np.zeros(n.value, PILE_DT)
        if n.value:
            self._ck(self.L.pemap_dev_fetch_records(self.h, first, count, _p(out), n.value, C.byref(n)))
        return out

    def reset_pileup(self):
        self._ck(self.L.pemap_dev_reset_pileup(self.h))

    def summary(self):
        out = np.zeros(13, np.int64)
        self._ck(self.L.pemap_dev_summary(self.h, _p(out)))
        return out

    def run_stats(self):
        s = np.zeros(16, np.uint64)
        t = np.zeros(8, np.float32)
        self._ck(self.L.pemap_dev_run_stats(self.h, _p(s), _p(t)))
        keys = ["ends", "positions", "sw_score", "sw_dirs", "cells_score", "cells_dirs", "pile_incs", "n_ins", "walks", "redo", "big_ends", "chunks", "gapless", "banded", "cells_band", "mono_ends"]
        tk = ["seed", "sw_single", "sw_multi", "select", "sw_redo", "walk", "lookup", "vote"]
        return dict(zip(keys, (int(x) for x in s))), dict(zip(tk, (float(x) for x in t)))

    def debug_hits(self, n_ends):
        nh = np.zeros(n_ends, np.int32)
        spot = np.zeros((n_ends, MAX_HITS), np.uint32)
        orient = np.zeros((n_ends, MAX_HITS), np.uint8)
        ws = np.zeros((n_ends, MAX_HITS), np.uint32)
        wl = np.zeros((n_ends, MAX_HITS), np.int32)
        sc = np.zeros((n_ends, MAX_HITS), np.float64)
        sk = np.zeros((n_ends, MAX_HITS), np.int32)
        si = np.zeros((n_ends, MAX_HITS), np.int32)
        self._ck(self.L.pemap_dev_debug_hits(self.h, _p(nh), _p(spot), _p(orient), _p(ws), _p(wl), _p(sc), _p(sk), _p(si)))
        return dict(n_hits=nh, spot=spot, orient=orient, win_start=ws, win_len=wl, score=sc, start_k=sk, start_i=si)
