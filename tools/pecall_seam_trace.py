"""Host-side trace of pecall_dev_call_sites (PECALL_SEAM_TRACE=1) on the bench's columns, and the PCIe rates of the box next to
it: the seam moves 768 bytes in and ~600 out per 64-sample column, so these rates bound it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from bench import pecall_columns
from pecaller_amd.pecall import PecallDev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
reads, dom = pecall_columns(n, 64)
# PCIe: pinned torch tensors, 256 MB pieces
h = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
g = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for name, (a, b) in (("h2d", (g, h)), ("d2h", (h, g))):
    a.copy_(b, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        a.copy_(b, non_blocking=True)
    torch.cuda.synchronize()
    print("%s pinned: %.1f GB/s" % (name, 8 * 0.268435456 / (time.perf_counter() - t0)))
s2 = torch.cuda.Stream()
t0 = time.perf_counter()
h2 = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
g2 = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    g.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(g2, non_blocking=True)
torch.cuda.synchronize()
print("both directions at once: %.1f GB/s each" % (8 * 0.268435456 / (time.perf_counter() - t0)))
del h, g, h2, g2
pc = PecallDev(0)
pc.call_sites(reads[:20000], dom[:20000])
out = pc.out_arrays(n, 64)
for arr in (reads, dom) + out:
    pc.pin_host(arr)
pc.call_sites(reads, dom, out=out)
os.environ["PECALL_SEAM_TRACE"] = "1"
t0 = time.perf_counter()
pc.call_sites(reads, dom, out=out)
print("seam, pinned: %.2f ms, %.1f M columns/s" % ((time.perf_counter() - t0) * 1e3, n / (time.perf_counter() - t0) / 1e6))
pc.sites_stage(reads, dom)
print("resident: %.2f ms" % pc.sites_run())
