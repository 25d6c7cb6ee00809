#!/usr/bin/env python3
"""Regenerate tests/golden/ from the UNMODIFIED reference programs (development container only).

Needs /root/reference (sources) -> `make -C oracle ref` builds oracle/_ref/{index_genome_whole,pemapper,pemapper_tsw}.
Steps: synthetic genome + three read sets (tools/synth.py, fixed seeds) -> reference index builder (about 2.5 min,
about 20 GB RSS: it allocates three 2^32-entry tables, index_genome_whole.c:185-187) -> reference mappers ->
collect small fixtures.  The 16 GiB .idx stream is reduced to the run boundaries of its prefix table (md5 only).

  python3 tests/golden/make_golden.py [--work /tmp/gold] [--skip-run]
"""
import argparse
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refio  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
SETS = {
    # name: (prefix, paired, program, extra args, synth args)
    "r150": ("g1", True, "pemapper", [], ["--pairs", "20000", "--read-len", "150"]),
    "r100": ("g1s", False, "pemapper", [], ["--pairs", "20000", "--read-len", "100", "--reads-only"]),
    "r250": ("g1l", True, "pemapper_tsw", ["3", "2"],
             ["--pairs", "5000", "--read-len", "250", "--indel-read-frac", "0.05", "--reads-only"]),
}


def run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold")
    ap.add_argument("--skip-run", action="store_true", help="only collect from an existing work directory")
    a = ap.parse_args()
    W = a.work
    os.makedirs(W, exist_ok=True)
    if not a.skip_run:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        for name, (pre, paired, prog, extra, sargs) in SETS.items():
            run([sys.executable, os.path.join(ROOT, "tools", "synth.py"), "--out", os.path.join(W, pre), "--seed", "12345",
                 "--contigs", "10", "--contig-len", "200000"] + sargs)
        # answers: Screen / max contigs / fasta / basename / not bisulfite   (index_genome_whole.c:117-167)
        p = subprocess.Popen([os.path.join(REF, "index_genome_whole")], stdin=subprocess.PIPE, cwd=W)
        p.communicate(("s\n100\n%s/g1.fa\n%s/g1\nn\n" % (W, W)).encode())
        assert p.returncode == 0
        for name, (pre, paired, prog, extra, sargs) in SETS.items():
            os.makedirs(os.path.join(W, name), exist_ok=True)
            f1 = os.path.join(W, pre + "_1_.fastq.gz")
            f2 = os.path.join(W, pre + "_2_.fastq.gz")
            if paired:
                cmd = [os.path.join(REF, prog), os.path.join(W, name, "out"), os.path.join(W, "g1.sdx"), "p", f1, f2,
                       "500", "0", "N", "0.85", "8", "200000000"] + extra
            else:
                cmd = [os.path.join(REF, prog), os.path.join(W, name, "out"), os.path.join(W, "g1.sdx"), "s", f1, "N",
                       "0.85", "8", "200000000"] + extra
            run(cmd, stdout=subprocess.DEVNULL)

    meta = {}
    # inputs
    with open(os.path.join(W, "g1.fa"), "rb") as f, gzip.GzipFile(os.path.join(HERE, "g1.fa.gz"), "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    shutil.copy(os.path.join(W, "g1.sdx"), os.path.join(HERE, "g1.sdx"))
    mdx = np.fromfile(os.path.join(W, "g1.mdx"), dtype="<u4")
    uk, us = refio.idx_to_compact(os.path.join(W, "g1.idx"))
    seq = gzip.open(os.path.join(W, "g1.seq"), "rb").read()
    meta["index"] = {"mdx_md5": refio.md5(mdx), "n_mers": int(len(mdx)), "idx_ukmer_md5": refio.md5(uk),
                     "idx_ustart_md5": refio.md5(us), "n_ukmer": int(len(uk)), "seq_md5": hashlib.md5(seq).hexdigest(),
                     "seq_len": len(seq)}
    for name, (pre, paired, prog, extra, sargs) in SETS.items():
        d = os.path.join(W, name)
        for k in (1, 2):
            fq = os.path.join(W, "%s_%d_.fastq.gz" % (pre, k))
            if k == 2 and not paired:
                continue
            shutil.copy(fq, os.path.join(HERE, os.path.basename(fq)))
            shutil.copy(fq + ".mfile", os.path.join(HERE, "%s.m%d" % (name, k)))
        shutil.copy(os.path.join(d, "out.summary.txt"), os.path.join(HERE, name + ".summary.txt"))
        shutil.copy(os.path.join(d, "out.indel.txt.gz"), os.path.join(HERE, name + ".indel.txt.gz"))
        pile = refio.read_pileup(os.path.join(d, "out.pileup.gz"))
        np.save(os.path.join(HERE, name + ".pileup_sample.npy"), pile[::101])
        meta[name] = {"program": prog, "extra_args": extra, "paired": paired, "pileup_md5": refio.md5(pile),
                      "pileup_records": int(len(pile)), "pileup_colsum": [int(x) for x in pile["c"].sum(axis=0)]}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
