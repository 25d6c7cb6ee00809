#!/usr/bin/env python3
"""BASELINE config 5 as a timing (not the bench.py metric): the per-site caller on synthetic 30x pileup columns of 64 samples,
one variant per kb with Hardy-Weinberg genotypes, 0.4 % error; GPU sites/s (host buffers in and out) and the CPU oracle on a
sample of the same columns, with the calls compared.   python3 tools/pecall_bench.py [--sites 400000] [--cpu-sites 20000]"""
import argparse
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def columns(n_sites, n, seed=777, var_rate=0.001, depth=30, err=0.004):
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
    reads = np.zeros((n_sites, n, 6), np.int64)
    idx = np.arange(n_sites)
    for s in range(n):
        reads[idx, s, dom] += good[:, s] - alt_reads[:, s]
        reads[idx, s, alt] += alt_reads[:, s]
        reads[idx, s, rng.integers(0, 4, n_sites)] += e[:, s]
    return reads.astype(np.uint16), dom


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=400000)
    ap.add_argument("--cpu-sites", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=64)
    a = ap.parse_args()
    from pecaller_amd.pecall import PecallDev
    import oracle_py
    reads, dom = columns(a.sites, a.samples)
    dev = PecallDev(0)
    dev.call_sites(reads[:1000], dom[:1000])
    t0 = time.time()
    call, p, typ, ac, npass = dev.call_sites(reads, dom)
    dt = time.time() - t0
    m = min(a.cpu_sites, a.sites)
    t0 = time.time()
    oc, op, otyp, oac, onp = oracle_py.call_sites(reads[:m], dom[:m])
    dtc = time.time() - t0
    print({"sites": a.sites, "samples": a.samples, "gpu_sites_per_s": round(a.sites / dt), "gpu_s": round(dt, 3),
           "cpu_oracle_sites_per_s_1_thread": round(m / dtc), "variant_rows": int((typ > 0).sum()), "passes": np.bincount(npass).tolist(),
           "calls_equal": bool(np.array_equal(call[:m], oc)), "max_abs_posterior_diff": float(np.max(np.abs(p[:m] - op))),
           "types_equal": bool(np.array_equal(typ[:m], otyp))})


if __name__ == "__main__":
    main()
