"""ctypes mirror of the pecall_dev_* entry points of include/pemap_hip.h (PECaller per-site genotype likelihoods)."""
import ctypes as C
import numpy as np
from .pemap import load_library, PemapError

MAX_GEN = 14
ALLELES = 6


def _p(a):
    return None if a is None else a.ctypes.data


class PecallDev:
    def __init__(self, device_id=0):
        L = load_library()
        vp, i, dbl = C.c_void_p, C.c_int, C.c_double
        L.pecall_dev_create.argtypes = [C.POINTER(vp), i]
        L.pecall_dev_destroy.argtypes = [vp]
        L.pecall_dev_destroy.restype = None
        L.pecall_dev_last_error.argtypes = [vp]
        L.pecall_dev_last_error.restype = C.c_char_p
        L.pecall_dev_site_like.argtypes = [vp, vp, vp, i, i, i, i, dbl, vp, vp, vp]
        L.pecall_dev_stage.argtypes = [vp, vp, vp, i, i]
        L.pecall_dev_run.argtypes = [vp, i, i, i, i, dbl, i]
        L.pecall_dev_collect.argtypes = [vp, i, i, vp, vp, vp]
        L.pecall_dev_call_sites.argtypes = [vp, vp, vp, vp, C.c_long, i, i, dbl, dbl, vp, vp, vp, vp, vp, vp]
        L.pecall_dev_call_sites_sparse.argtypes = [vp, vp, vp, vp, C.c_long, i, i, dbl, dbl, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp]
        L.pecall_dev_set_pedigree.argtypes = [vp, i, vp, vp, vp, vp, vp, dbl]
        L.pecall_dev_sites_stage.argtypes = [vp, vp, vp, vp, C.c_long, i]
        L.pecall_dev_sites_run.argtypes = [vp, i, dbl, dbl, C.POINTER(C.c_float)]
        L.pecall_dev_sites_collect.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.pecall_dev_pin_host.argtypes = [vp, vp, C.c_uint64]
        L.pecall_dev_unpin_host.argtypes = [vp, vp]
        self.L = L
        h = vp()
        if L.pecall_dev_create(C.byref(h), device_id):
            raise PemapError(L.pecall_dev_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.pecall_dev_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ck(self, rc):
        if rc:
            raise PemapError(self.L.pecall_dev_last_error(self.h).decode())

    def site_like(self, reads, alpha_mean, norm, max_gen=14, min_depth=2):
        """reads [n_sites][indiv][6] u16, alpha_mean [n_sites][14][6] f64 -> like [n_sites][indiv][14], best, margin"""
        reads = np.ascontiguousarray(reads, np.uint16)
        alpha_mean = np.ascontiguousarray(alpha_mean, np.float64)
        n_sites, indiv = reads.shape[:2]
        like = np.zeros((n_sites, indiv, MAX_GEN))
        best = np.zeros((n_sites, indiv), np.int8)
        margin = np.zeros((n_sites, indiv))
        self._ck(self.L.pecall_dev_site_like(self.h, _p(reads), _p(alpha_mean), n_sites, indiv, max_gen, min_depth, norm, _p(like),
                                             _p(best), _p(margin)))
        return like, best, margin

    def stage(self, reads, alpha_mean):
        reads = np.ascontiguousarray(reads, np.uint16)
        alpha_mean = np.ascontiguousarray(alpha_mean, np.float64)
        self._shape = reads.shape[:2]
        self._ck(self.L.pecall_dev_stage(self.h, _p(reads), _p(alpha_mean), reads.shape[0], reads.shape[1]))

    def run(self, norm, max_gen=14, min_depth=2, sync=True):
        self._ck(self.L.pecall_dev_run(self.h, self._shape[0], self._shape[1], max_gen, min_depth, norm, int(sync)))

    def collect(self):
        n_sites, indiv = self._shape
        like = np.zeros((n_sites, indiv, MAX_GEN))
        best = np.zeros((n_sites, indiv), np.int8)
        margin = np.zeros((n_sites, indiv))
        self._ck(self.L.pecall_dev_collect(self.h, n_sites, indiv, _p(like), _p(best), _p(margin)))
        return like, best, margin

    def set_pedigree(self, dad, mom, sex, kid_off, kid_list, denovo_rate):
        """parents as sample indices (-1 = none), sex, kids of i = kid_list[kid_off[i]:kid_off[i+1]] in ped-file order; None clears"""
        if dad is None:
            self._ck(self.L.pecall_dev_set_pedigree(self.h, 0, None, None, None, None, None, 0.0))
            return
        a = [np.ascontiguousarray(x, np.int32) for x in (dad, mom, sex, kid_off, kid_list)]
        self._ck(self.L.pecall_dev_set_pedigree(self.h, len(a[0]), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), float(denovo_rate)))

    def sites_stage(self, reads, ref_base, chrom=None):
        reads = np.ascontiguousarray(reads, np.uint16)
        ref_base = np.ascontiguousarray(ref_base, np.uint8)
        cy = None if chrom is None else np.ascontiguousarray(chrom, np.uint8)
        self._sshape = reads.shape[:2]
        self._ck(self.L.pecall_dev_sites_stage(self.h, _p(reads), _p(ref_base), _p(cy), reads.shape[0], reads.shape[1]))

    def sites_run(self, threshold=0.95, theta=0.001, haploid=False):
        """the kernel on the staged columns -> its duration in ms (HIP events on the object's stream)"""
        ms = C.c_float(0)
        self._ck(self.L.pecall_dev_sites_run(self.h, int(haploid), float(threshold), float(theta), C.byref(ms)))
        return ms.value

    def sites_collect(self):
        n_sites, indiv = self._sshape
        call = np.zeros((n_sites, indiv), np.int8)
        post = np.zeros((n_sites, indiv))
        typ = np.zeros(n_sites, np.int8)
        ac = np.zeros((n_sites, ALLELES), np.int32)
        npass = np.zeros(n_sites, np.int8)
        self.denovo = np.zeros(n_sites, np.int32)
        self._ck(self.L.pecall_dev_sites_collect(self.h, _p(call), _p(post), _p(typ), _p(ac), _p(npass), _p(self.denovo)))
        return call, post, typ, ac, npass

    def pin_host(self, a):
        self._ck(self.L.pecall_dev_pin_host(self.h, a.ctypes.data, a.nbytes))

    def unpin_host(self, a):
        self._ck(self.L.pecall_dev_unpin_host(self.h, a.ctypes.data))

    @staticmethod
    def out_arrays(n_sites, indiv, posterior=True):
        """the six result arrays of call_sites, touched (np.zeros leaves the pages to the first write); posterior=False: a token
        array in its place (call_sites_sparse does not fill it)"""
        out = (np.zeros((n_sites, indiv), np.int8), np.zeros((n_sites, indiv) if posterior else (1, 1)), np.zeros(n_sites, np.int8), np.zeros((n_sites, ALLELES), np.int32),
               np.zeros(n_sites, np.int8), np.zeros(n_sites, np.int32))
        for a in out:
            a.fill(0)
        return out

    def call_sites_sparse(self, reads, ref_base, threshold=0.95, theta=0.001, haploid=False, chrom=None, cap=None, out=None, sparse_out=None):
        """call_sites with the posteriors as a list (pecall_dev_call_sites_sparse): -> call, (post_site [k] ascending, post_rows [k][indiv]),
        site_type, allele_count, n_pass; columns that are not listed have posterior 1 for every sample.  cap = rows the list may take
        (default: one per 8 columns, at least 1024); out / sparse_out: arrays to reuse (out as for call_sites, its posterior unused)"""
        reads = np.ascontiguousarray(reads, np.uint16)
        ref_base = np.ascontiguousarray(ref_base, np.uint8)
        n_sites, indiv = reads.shape[:2]
        cy = None if chrom is None else np.ascontiguousarray(chrom, np.uint8)
        call, _, typ, ac, npass, den = out if out is not None else self.out_arrays(n_sites, indiv, posterior=False)
        if sparse_out is not None:
            site, rows = sparse_out
            cap = len(site)
        else:
            cap = int(cap) if cap is not None else max(1024, n_sites // 8)
            site, rows = np.empty(cap, np.uint32), np.empty((cap, indiv), np.float64)
        n = C.c_uint64(0)
        self.denovo = den
        self.sparse_needed = 0
        rc = self.L.pecall_dev_call_sites_sparse(self.h, _p(reads), _p(ref_base), _p(cy), n_sites, indiv, int(haploid), float(threshold),
                                                 float(theta), _p(call), _p(site), _p(rows), cap, C.byref(n), _p(typ), _p(ac), _p(npass), _p(den))
        self.sparse_needed = int(n.value)
        self._ck(rc)
        return call, (site[:n.value], rows[:n.value]), typ, ac, npass

    def call_sites(self, reads, ref_base, threshold=0.95, theta=0.001, haploid=False, chrom=None, out=None):
        """the whole per-site caller (pecaller.c:1207-1691): reads [n_sites][indiv][6] u16, ref_base [n_sites] (0..3 = ACGT, else
        skipped), chrom [n_sites] 0 autosome / 1 X / 2 Y / 3 MT -> call [n_sites][indiv] (0..13, 14 = 'N'), posterior, site_type,
        allele_count, n_pass; self.denovo = d_count per site (with a pedigree).  out: the arrays of out_arrays(), reused by a
        caller that keeps (and may have pinned) its buffers"""
        reads = np.ascontiguousarray(reads, np.uint16)
        ref_base = np.ascontiguousarray(ref_base, np.uint8)
        n_sites, indiv = reads.shape[:2]
        cy = None if chrom is None else np.ascontiguousarray(chrom, np.uint8)
        call, post, typ, ac, npass, den = out if out is not None else self.out_arrays(n_sites, indiv)
        self.denovo = den
        self._ck(self.L.pecall_dev_call_sites(self.h, _p(reads), _p(ref_base), _p(cy), n_sites, indiv, int(haploid), float(threshold),
                                              float(theta), _p(call), _p(post), _p(typ), _p(ac), _p(npass), _p(den)))
        return call, post, typ, ac, npass
