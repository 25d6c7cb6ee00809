"""ctypes front-end of oracle/liboracle_pemap.so (the CPU restatement).  Checker side only: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by pecaller_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_HITS = 200


class OraIndex(C.Structure):
    _fields_ = [("pos_index", C.c_void_p), ("mers", C.c_void_p), ("ukmer", C.c_void_p), ("ustart", C.c_void_p),
                ("n_ukmer", C.c_uint64), ("n_mers", C.c_uint64), ("genome", C.c_void_p), ("genome_size", C.c_uint64),
                ("contig_starts", C.c_void_p), ("n_contigs", C.c_int), ("idepth", C.c_int)]


class OraParams(C.Structure):
    _fields_ = [("paired", C.c_int), ("min_dist", C.c_int), ("max_dist", C.c_int), ("min_align", C.c_double),
                ("bisulfite", C.c_int)]


END_DBG = np.dtype([("n_hits", "<i4"), ("spot", "<u4", (MAX_HITS,)), ("orient", "u1", (MAX_HITS,)),
                    ("win_start", "<i4", (MAX_HITS,)), ("win_len", "<i4", (MAX_HITS,)), ("score", "<f8", (MAX_HITS,)),
                    ("start", "<i4", (MAX_HITS, 3))], align=True)
INS_DT = np.dtype([("pos", "<u4"), ("len", "<u2"), ("seq", "S300")], align=True)

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", "liboracle_pemap.so")
        src = os.path.join(ROOT, "oracle", "pemap_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), so])
        L = C.CDLL(so)
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.POINTER(OraIndex), C.POINTER(OraParams)]
        L.ora_destroy.argtypes = [C.c_void_p]
        L.ora_map_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_counts.restype = C.c_void_p
        L.ora_counts.argtypes = [C.c_void_p]
        L.ora_n_ins.restype = C.c_long
        L.ora_n_ins.argtypes = [C.c_void_p]
        L.ora_ins_log.restype = C.c_void_p
        L.ora_ins_log.argtypes = [C.c_void_p]
        L.ora_summary.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_sw.restype = C.c_double
        L.ora_sw.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ora_find_chrom.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.ora_neighbours.argtypes = [C.c_uint32, C.c_void_p]
        L.ora_kmer.restype = C.c_uint32
        L.ora_kmer.argtypes = [C.c_char_p]
        assert C.sizeof(OraIndex) == 80
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


class Oracle:
    """index = dict(mers, ukmer, ustart, genome (uint8 array), contig_starts[, pos_index])"""

    def __init__(self, index, paired=True, min_dist=0, max_dist=500, min_align=0.85, bisulfite=False, idepth=16):
        self.L = lib()
        self.keep = index
        ix = OraIndex()
        ix.pos_index = _p(index.get("pos_index"))
        ix.mers = _p(index["mers"])
        ix.ukmer = _p(index.get("ukmer"))
        ix.ustart = _p(index.get("ustart"))
        ix.n_ukmer = 0 if index.get("ukmer") is None else len(index["ukmer"])
        ix.n_mers = len(index["mers"])
        ix.genome = _p(index["genome"])
        ix.genome_size = len(index["genome"])
        ix.contig_starts = _p(index["contig_starts"])
        ix.n_contigs = len(index["contig_starts"]) - 1
        ix.idepth = idepth
        pr = OraParams(int(paired), min_dist, max_dist, min_align, int(bisulfite))
        self.paired = paired
        self.gsize = len(index["genome"])
        self.h = self.L.ora_create(C.byref(ix), C.byref(pr))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_destroy(self.h)
            self.h = None

    def map_batch(self, r1, l1, r2=None, l2=None, debug=False, threads=1):
        n = len(l1)
        stride = r1.shape[1]
        m1 = np.zeros(n, np.uint32)
        m2 = np.zeros(n, np.uint32) if self.paired else None
        mt = np.zeros(n, np.int32)
        d1 = np.zeros(n, END_DBG) if debug else None
        d2 = np.zeros(n, END_DBG) if (debug and self.paired) else None
        rc = self.L.ora_map_batch(self.h, _p(r1), _p(l1), _p(r2), _p(l2), n, stride, _p(m1), _p(m2), _p(mt), _p(d1),
                                  _p(d2), threads)
        assert rc == 0
        return m1, m2, mt, d1, d2

    def counts(self):
        p = self.L.ora_counts(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(self.gsize, 6))

    def insertions(self):
        n = self.L.ora_n_ins(self.h)
        if n == 0:
            return []
        p = self.L.ora_ins_log(self.h)
        buf = (C.c_char * (n * INS_DT.itemsize)).from_address(p)
        a = np.frombuffer(buf, dtype=INS_DT, count=n)
        return sorted((int(x["pos"]), bytes(x["seq"])[:int(x["len"])]) for x in a)

    def summary(self):
        out = np.zeros(13, np.int64)
        self.L.ora_summary(self.h, _p(out))
        return out


def sw(ref, seq, bisulfite=False, planes=False):
    L = lib()
    st = np.zeros(3, np.int32)
    pl = np.zeros(3 * (len(ref) + 1) * (len(seq) + 1)) if planes else None
    s = L.ora_sw(bytes(ref), len(ref), bytes(seq), len(seq), int(bisulfite), _p(st), _p(pl))
    return s, st, (pl.reshape(3, len(ref) + 1, len(seq) + 1) if planes else None)


# ---- PECaller likelihood oracle (oracle/pecall_oracle.c)
_plib = None


def pecall_lib():
    global _plib
    if _plib is None:
        so = os.path.join(ROOT, "oracle", "liboracle_pecall.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in ("pecall_oracle.c", "pecall_site_oracle.c")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), so])
        L = C.CDLL(so)
        L.ora_site_like.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
        L.ora_factln.restype = C.c_double
        L.ora_factln.argtypes = [C.c_int]
        L.ora_caller_create.restype = C.c_void_p
        L.ora_caller_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.ora_caller_destroy.argtypes = [C.c_void_p]
        L.ora_caller_max_list.argtypes = [C.c_void_p]
        L.ora_call_sites.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_caller_set_ped.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        _plib = L
    return _plib


def site_like(reads, alpha_mean, norm, max_gen=14, min_depth=2):
    L = pecall_lib()
    reads = np.ascontiguousarray(reads, np.uint16)
    alpha_mean = np.ascontiguousarray(alpha_mean, np.float64)
    n_sites, indiv = reads.shape[:2]
    like = np.zeros((n_sites, indiv, 14))
    best = np.zeros((n_sites, indiv), np.int8)
    margin = np.zeros((n_sites, indiv))
    L.ora_site_like(_p(reads), _p(alpha_mean), n_sites, indiv, max_gen, min_depth, norm, _p(like), _p(best), _p(margin))
    return like, best, margin


def kid_lists(dad, mom, order=None):
    """kids of every sample in ped-file order (pecaller.c:574-590): CSR offsets and list"""
    n = len(dad)
    kids = [[] for _ in range(n)]
    for i in (order if order is not None else range(n)):
        if dad[i] >= 0:
            kids[dad[i]].append(i)
        if mom[i] >= 0:
            kids[mom[i]].append(i)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(k) for k in kids])
    return off, np.array([x for k in kids for x in k] + [0], np.int32)


def call_sites(reads, dom, threshold=0.95, theta=0.001, haploid=False, chrom=None, ped=None, threads=None):
    """the per-site caller oracle (oracle/pecall_site_oracle.c): reads[n_sites][indiv][6], dom[n_sites] in 0..3 (else skipped),
    chrom[n_sites] 0 autosome / 1 X / 2 Y / 3 MT, ped = dict(dad, mom, sex, order, denovo_rate) or None
    -> call[n_sites][indiv] (0..13, 14 = N), p, site type, allele counts, passes; call_sites.denovo = d_count per site.
    Columns are independent: blocks of them go to `threads` callers side by side (default: up to 8, one per CPU; a variant column of a
    few hundred samples costs the oracle seconds), each with its own handle, writing its rows of the shared result arrays."""
    import threading
    L = pecall_lib()
    reads = np.ascontiguousarray(reads, np.uint16)
    dom = np.ascontiguousarray(dom, np.uint8)
    n_sites, indiv = reads.shape[:2]
    cy = np.ascontiguousarray(chrom, np.uint8) if chrom is not None else np.zeros(n_sites, np.uint8)
    call = np.zeros((n_sites, indiv), np.int8)
    p = np.zeros((n_sites, indiv))
    typ = np.zeros(n_sites, np.int8)
    ac = np.zeros((n_sites, 6), np.int32)
    npass = np.zeros(n_sites, np.int8)
    den = np.zeros(n_sites, np.int32)
    if ped is not None:
        dad = np.ascontiguousarray(ped["dad"], np.int32)
        mom = np.ascontiguousarray(ped["mom"], np.int32)
        sex = np.ascontiguousarray(ped["sex"], np.int32)
        off, lst = kid_lists(dad, mom, ped.get("order"))
    block = 8 if indiv > 64 else 64
    blocks = iter(range(0, n_sites, block))
    lock = threading.Lock()
    max_list = [0]
    errors = []

    def worker():
        h = L.ora_caller_create(indiv, int(haploid), float(threshold), float(theta))
        try:
            if ped is not None:
                L.ora_caller_set_ped(h, _p(dad), _p(mom), _p(sex), _p(off), _p(lst), float(ped["denovo_rate"]))
            while True:
                with lock:
                    a = next(blocks, None)
                if a is None:
                    break
                b = min(a + block, n_sites)
                L.ora_call_sites(h, _p(reads[a:b]), _p(dom[a:b]), _p(cy[a:b]), b - a, _p(call[a:b]), _p(p[a:b]), _p(typ[a:b]), _p(ac[a:b]), _p(npass[a:b]),
                                 _p(den[a:b]))
            with lock:
                max_list[0] = max(max_list[0], L.ora_caller_max_list(h))      # longest configuration list seen (coverage statistic)
        except Exception as e:          # (surfaced by the caller: a thread's exception would otherwise be lost)
            errors.append(e)
        finally:
            L.ora_caller_destroy(h)

    nt = threads if threads else max(1, min(8, os.cpu_count() or 1, (n_sites + block - 1) // block))
    if nt <= 1:
        worker()
    else:
        ts = [threading.Thread(target=worker) for _ in range(nt)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    if errors:
        raise errors[0]
    call_sites.max_list = max_list[0]
    call_sites.denovo = den
    return call, p, typ, ac, npass
