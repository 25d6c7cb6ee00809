#!/usr/bin/env python3
"""Golden vectors for the two text tools behind the caller (development container only; needs /root/reference, perl with IO::Zlib
and oracle/_ref/snp_to_vcf = gcc -O3 of /root/reference/src/snp_to_vcf.c, `make -C oracle ref`).

Inputs are made here from data already under tests/golden/ (the reference caller's .snp rows of pecall_sites / pecall_ped, the
g1 genome) and stored with what the UNMODIFIED reference tools print for them:

  (cases: sites, ped = the reference caller's rows; multi = rows with three and more alleles, D and I among them, made here)
  downstream/<case>.snp.txt            the snp file given to merge_indel_snp.pl: rows shuffled, some moved to other contigs, a run of
                                       deletions at consecutive positions, rows on a contig the .sdx does not name
  downstream/<case>_indel/<s>.indel.txt.gz   one insertion file per sample (pemapper's format; the counts have one clear winner per
                                       position -- the script's pick among equal counts is not reproducible)
  downstream/<case>.merged.txt         what merge_indel_snp.pl wrote
  downstream/<case>.vcf.txt            what snp_to_vcf printed for the merged file at min_prob 0.9 (the ##fileDate line removed)

Only data is stored.   python3 tests/golden/make_golden_downstream.py [--work /tmp/gold_down]
"""
import argparse
import gzip
import os
import shutil
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_PL = "/root/reference/src/merge_indel_snp.pl"
REF_VCF = os.path.join(ROOT, "oracle", "_ref", "snp_to_vcf")
OUT = os.path.join(HERE, "downstream")


def write_seq(work):
    """<work>/g1.sdx + g1.seq as index_genome_whole lays them out: contigs with 15 filler bytes after each"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refio
    names, contigs = refio.read_fasta(os.path.join(HERE, "g1.fa.gz"))
    shutil.copy(os.path.join(HERE, "g1.sdx"), os.path.join(work, "g1.sdx"))
    with gzip.open(os.path.join(work, "g1.seq"), "wb") as f:
        for c in contigs:
            f.write(c.tobytes() + b"N" * 15)


def multi_rows(rng, n_samples):
    """rows as the caller prints them for columns with three or more alleles, insertions and deletions among them: the forms
    snp_to_vcf.c:324-433 takes apart (D after one or two letters, I before and after a D, the reference absent from the alleles)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refio
    names, contigs = refio.read_fasta(os.path.join(HERE, "g1.fa.gz"))
    het = {}
    for a, b, c in ("ACM", "AGR", "ATW", "CGS", "CTY", "GTK"):
        het[a + b] = het[b + a] = c
    rows = []
    for k in range(90):
        ci = int(rng.integers(0, 4))
        pos = int(rng.integers(50, 60000))
        ref = chr(contigs[ci][pos - 1])
        pool = [x for x in "ACGT" if x != ref]
        rng.shuffle(pool)
        al = [ref] if rng.random() < 0.85 else []
        al += pool[: int(rng.integers(1, 3))]
        if k % 3 != 2:
            al.append("D")
        if k % 3 != 1:
            al.append("I")
        al = sorted(al, key="ACGTDI".index)
        if len(al) < 3:
            al = sorted(set(al + [pool[-1]]), key="ACGTDI".index)
        calls = []
        for s in range(n_samples):
            a, b = rng.choice(al, 2)
            if a == b:
                g = a
            elif "D" in (a, b):
                g = "E"
            elif "I" in (a, b):
                g = "H"
            else:
                g = het[a + b]
            if rng.random() < 0.05:
                g = "N"
            calls += [g, str(rng.choice(["1", "0.99", "0.95123", "0.5", "0.899999"]))]
        ty = "MULTIALLELIC" if k % 7 else "DENOVO_MULTIALLELIC"
        rows.append("\t".join([names[ci], str(pos), ref, ",".join(al), ",".join(str(int(x)) for x in rng.integers(1, 9, len(al))), ty] + calls))
    return rows


def make_case(case, src_snp, rng, work):
    lines = open(os.path.join(HERE, src_snp)).read().split("\n")
    header, rows = lines[0], [l for l in lines[1:] if l]
    if case == "multi":
        rows = multi_rows(rng, len(header.split("\t")[6::2])) + rows[:40]
    samples = header.split("\t")[6::2]
    out_rows = []
    for i, l in enumerate(rows):
        f = l.split("\t")
        u = rng.random()
        if u < 0.25:
            f[0] = "chr3"
        elif u < 0.45:
            f[0] = "chr2"
        elif u < 0.48 and f[5] == "SNP":
            f[0] = "chrUn"                      # not in the .sdx: sorts as contig number 0
        out_rows.append(f)
    # a run of deletions at consecutive positions (absorbed into one row), one of them after a deletion on another contig at pos-1
    dels = [f for f in out_rows if f[5] in ("DEL", "DENOVO_DEL")]
    for f in dels[:3]:
        for extra in (1, 2, 3)[: 1 + int(rng.integers(0, 3))]:
            g = list(f)
            g[1] = str(int(f[1]) + extra)
            out_rows.append(g)
    if len(dels) > 4:
        g = list(dels[4])
        g[0] = "chr5"
        out_rows.append(g)
        g = list(dels[4])
        g[0] = "chr6"
        g[1] = str(int(g[1]) + 1)
        out_rows.append(g)
    # duplicates of one position (the stable sort keeps their file order)
    g = list(out_rows[5])
    g[4] = "0,0"
    out_rows.append(g)
    order = rng.permutation(len(out_rows))
    out_rows = [out_rows[i] for i in order]
    snp_path = os.path.join(OUT, case + ".snp.txt")
    with open(snp_path, "w") as f:
        f.write(header + "\n")
        for r in out_rows:
            f.write("\t".join(r) + "\n")
    # insertion files: every position whose row carries an I, except one (the script reports it and leaves the I)
    need = [(r[0], int(r[1])) for r in out_rows if r[5] in ("INS", "DENOVO_INS") or ("MULTIALLELIC" in r[5] and "I" in r[3].split(","))]
    need = sorted(set(need))
    skip = need[len(need) // 2] if len(need) > 3 else None
    d = os.path.join(OUT, case + "_indel")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    per_sample = {s: [] for s in samples}
    for (c, p) in need:
        if (c, p) == skip:
            continue
        n = int(rng.integers(1, 9))
        win = "".join(rng.choice(list("ACGT"), n))
        others = ["".join(rng.choice(list("ACGT"), int(rng.integers(1, 6)))) for _ in range(3)]
        others = [o for o in others if o != win]
        # the winner: 2 reads in every sample but the last (one row each); the others once or twice in single samples
        for s in samples[:-1]:
            per_sample[s].append((c, p, [win, win]))
        for k, o in enumerate(others):
            s = samples[(k + p) % len(samples)]
            per_sample[s].append((c, p, [o] + ([win] if k == 0 else [])))
    for s in samples:
        rows_s = per_sample[s]
        # rows at positions nobody needs, and a row with no sequence column
        for _ in range(20):
            rows_s.append(("chr%d" % rng.integers(1, 9), int(rng.integers(1, 100000)), ["ACGTT"]))
        rows_s.sort(key=lambda r: (r[0], r[1]))
        with gzip.open(os.path.join(d, s + ".indel.txt.gz"), "wt") as f:
            f.write("Fragment\tPositions\tReference Base\tTotal Coverage\tReference Reads\tNo Deletions\tNo Insertions\tInsertion Sequence\n")
            for (c, p, seqs) in rows_s:
                f.write("%s\t%d\tA\t%d\t%d\t0\t%d\t%s\n" % (c, p, 20 + len(seqs), 20, len(seqs), "\t".join(seqs)))
            f.write("chr1\t77\tA\t9\t9\t0\t0\n")
    # ---- the reference tools
    merged = os.path.join(OUT, case + ".merged.txt")
    subprocess.check_call(["perl", REF_PL, os.path.join(work, "g1.sdx"), snp_path, d, merged], stdout=subprocess.DEVNULL)
    vcf = subprocess.run([REF_VCF, os.path.join(work, "g1.sdx"), merged, "0.9"], stdout=subprocess.PIPE, check=True).stdout
    vcf = b"\n".join(l for l in vcf.split(b"\n") if not l.startswith(b"##fileDate=")) .replace(os.path.join(work, "g1.sdx").encode(), b"g1.sdx")
    open(os.path.join(OUT, case + ".vcf.txt"), "wb").write(vcf)
    print(case, len(out_rows), "rows in,", len(open(merged).read().split("\n")) - 2, "rows merged,", vcf.count(b"\n") - 6, "vcf rows,", len(need), "insertions")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gold_down")
    a = ap.parse_args()
    shutil.rmtree(a.work, ignore_errors=True)
    os.makedirs(a.work)
    os.makedirs(OUT, exist_ok=True)
    write_seq(a.work)
    rng = np.random.default_rng(20260)
    make_case("sites", "pecall_sites.snp.txt", rng, a.work)
    make_case("ped", "pecall_ped.snp.txt", rng, a.work)
    make_case("multi", "pecall_sites.snp.txt", rng, a.work)


if __name__ == "__main__":
    main()
