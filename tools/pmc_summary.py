#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh: <dir>/stats.csv (kernel stats of the pm_* / ix_* kernels) and
<dir>/pmc.json (mean FETCH_SIZE / WRITE_SIZE per launch and kernel, in KB as rocprofv3 reports them)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

d = sys.argv[1]


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


st = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(d, "stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
out = {}
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = defaultdict(list)
    for fn in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == ctr:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    out[ctr] = {k: {"launches": len(v), "mean_KB_per_launch": sum(v) / len(v), "max_KB": max(v)}
                for k, v in acc.items() if k.startswith(("pm_", "pile_", "ix_", "pcs_", "pc_"))}
if len(sys.argv) > 2:
    out["steps"] = int(sys.argv[2])     # steps of the main configuration the profiled command ran (for per-step totals)
# the device sources the profile was taken on (bench.py drops a profile of other kernels)
import hashlib
h = hashlib.sha1()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for fn in sorted(glob.glob(os.path.join(root, "pecaller_amd", "csrc", "*.hip*"))):
    h.update(os.path.basename(fn).encode())
    h.update(open(fn, "rb").read())
out["kernel_sources_sha"] = h.hexdigest()[:16]
for pre in ("pemap_", "pecall_"):           # the mapper's and the caller's sources apart: a change in one leaves the other's profile valid
    h = hashlib.sha1()
    for fn in sorted(glob.glob(os.path.join(root, "pecaller_amd", "csrc", pre + "*.hip*"))):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    out["sha_" + pre.rstrip("_")] = h.hexdigest()[:16]
json.dump(out, open(os.path.join(d, "pmc.json"), "w"), indent=1)
print("wrote", os.path.join(d, "pmc.json"), {k: len(v) for k, v in out.items() if isinstance(v, dict)})
