#!/usr/bin/env python3
"""Deterministic synthetic genomes and reads for parity tests and the CPU-side fixtures.

Not part of the product path: the product reads ordinary FASTA / fastq(.gz).  The generator follows
SURVEY.md section 8(d): uniform fragment starts, fragment length uniform in [300, 500], read 1/2
strand swapped with p = 0.5, per-base substitutions and indels, constant qualities.

Genome features that exercise the reference's edge rules (src/pemapper.c):
  * exact and diverged duplicated segments  -> multi-hit ends, NON_MATE / NON_MIS / UNIQUE_SLIP
  * a low-complexity run (poly-A, short tandem repeat) -> buckets >= too_many_spots (1602-1606)
  * N runs inside contigs and at contig ends -> index window resets, N-as-wildcard scoring (2020-2023)
"""
import argparse
import gzip
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
COMP[:] = ord("N")
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def make_genome(seed, n_contigs, contig_len, features=True):
    rng = np.random.default_rng(seed)
    contigs = []
    for c in range(n_contigs):
        ln = int(contig_len * (0.75 + 0.5 * rng.random())) if features else contig_len
        seq = BASES[rng.integers(0, 4, size=ln)]
        contigs.append(seq.copy())
    if features and contig_len >= 20000:
        # exact duplicates (within and across contigs)
        for _ in range(max(2, n_contigs)):
            src = contigs[rng.integers(0, n_contigs)]
            dst = contigs[rng.integers(0, n_contigs)]
            ln = int(rng.integers(300, 2500))
            s = int(rng.integers(0, len(src) - ln))
            d = int(rng.integers(0, len(dst) - ln))
            dst[d:d + ln] = src[s:s + ln]
        # diverged duplicates (1-3 % substitutions)
        for _ in range(max(2, n_contigs)):
            src = contigs[rng.integers(0, n_contigs)]
            dst = contigs[rng.integers(0, n_contigs)]
            ln = int(rng.integers(400, 3000))
            s = int(rng.integers(0, len(src) - ln))
            d = int(rng.integers(0, len(dst) - ln))
            seg = src[s:s + ln].copy()
            m = rng.random(ln) < rng.choice([0.01, 0.02, 0.03])
            seg[m] = BASES[rng.integers(0, 4, size=int(m.sum()))]
            dst[d:d + ln] = seg
        # reverse-complement duplicate
        src = contigs[0]
        dst = contigs[-1]
        ln = 1500
        s = int(rng.integers(0, len(src) - ln))
        d = int(rng.integers(0, len(dst) - ln))
        dst[d:d + ln] = COMP[src[s:s + ln]][::-1]
        # low complexity
        c0 = contigs[1 % n_contigs]
        p = int(rng.integers(1000, len(c0) - 2000))
        c0[p:p + 400] = ord("A")
        unit = np.frombuffer(b"ACACACGT", dtype=np.uint8)
        c0[p + 600:p + 600 + 800] = np.tile(unit, 100)
        # dispersed repeat family: 150 copies of a 300-mer with 5 % divergence
        fam = BASES[rng.integers(0, 4, size=300)]
        for _ in range(150):
            dst = contigs[rng.integers(0, n_contigs)]
            d = int(rng.integers(0, len(dst) - 300))
            seg = fam.copy()
            m = rng.random(300) < 0.05
            seg[m] = BASES[rng.integers(0, 4, size=int(m.sum()))]
            dst[d:d + 300] = seg
        # N runs: inside, and at the ends of some contigs
        for c in range(0, n_contigs, 3):
            cc = contigs[c]
            p = int(rng.integers(2000, len(cc) - 3000))
            cc[p:p + int(rng.integers(1, 200))] = ord("N")
            p = int(rng.integers(2000, len(cc) - 3000))
            cc[p] = ord("N")
        contigs[0][:50] = ord("N")
        contigs[-1][-37:] = ord("N")
    return contigs


def write_fasta(path, contigs, names=None):
    with open(path, "wb") as f:
        for i, c in enumerate(contigs):
            nm = names[i] if names else "chr%d" % (i + 1)
            f.write(b">" + nm.encode() + b"\n")
            b = c.tobytes()
            for k in range(0, len(b), 60):
                f.write(b[k:k + 60] + b"\n")


def mutate(rng, seq, sub, indel):
    """substitutions, then single-event indels (1-10 bp) with probability `indel` per base."""
    seq = seq.copy()
    m = rng.random(len(seq)) < sub
    if m.any():
        seq[m] = BASES[(np.searchsorted(BASES, seq[m]) + rng.integers(1, 4, size=int(m.sum()))) % 4]
    if indel > 0:
        out = []
        i = 0
        ev = np.nonzero(rng.random(len(seq)) < indel)[0]
        for p in ev:
            if p < i:
                continue
            out.append(seq[i:p])
            ln = int(rng.integers(1, 11)) if rng.random() < 0.3 else 1
            if rng.random() < 0.5:
                out.append(BASES[rng.integers(0, 4, size=ln)])  # insertion
                i = p
            else:
                i = p + ln  # deletion
        out.append(seq[i:])
        seq = np.concatenate(out)
    return seq


def make_reads(seed, contigs, n_pairs, read_len, sub=0.01, indel=0.0004, frag=(300, 500),
               junk_frac=0.01, n_frac=0.01, indel_read_frac=0.0, paired=True):
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    pc = lens / lens.sum()
    r1, r2, truth = [], [], []
    for n in range(n_pairs):
        if rng.random() < junk_frac:
            a = BASES[rng.integers(0, 4, size=read_len)]
            b = BASES[rng.integers(0, 4, size=read_len)]
            truth.append((-1, 0, 0))
        else:
            c = int(rng.choice(len(contigs), p=pc))
            g = contigs[c]
            fl = int(rng.integers(frag[0], frag[1] + 1))
            # allow fragments that touch the contig ends
            s = int(rng.integers(0, max(1, len(g) - fl)))
            fr = g[s:s + fl + 40]
            ind = indel
            if indel_read_frac > 0:
                ind = 0.0
            a = mutate(rng, fr, sub, ind)[:read_len]
            fr2 = COMP[g[max(0, s + fl - read_len - 40):s + fl]][::-1]
            b = mutate(rng, fr2, sub, ind)[:read_len]
            if indel_read_frac > 0 and rng.random() < indel_read_frac:
                # exactly one 1-10 bp indel in the middle of each end (config 4)
                for which in (0, 1):
                    x = a if which == 0 else b
                    p = int(rng.integers(30, len(x) - 30))
                    ln = int(rng.integers(1, 11))
                    if rng.random() < 0.5:
                        x = np.concatenate([x[:p], BASES[rng.integers(0, 4, size=ln)], x[p:]])[:read_len]
                    else:
                        x = np.concatenate([x[:p], x[p + ln:]])
                    if which == 0:
                        a = x
                    else:
                        b = x
            truth.append((c, s, fl))
            if rng.random() < 0.5:
                a, b = b, a
        if rng.random() < n_frac:
            a = a.copy()
            k = int(rng.integers(1, 25))
            a[rng.integers(0, len(a), size=k)] = ord("N")
        if rng.random() < n_frac:
            b = b.copy()
            b[int(rng.integers(0, len(b)))] = ord("N")
        r1.append(a.tobytes())
        r2.append(b.tobytes())
    return r1, r2, truth


def write_fastq(path, reads, tag):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d/%d\n" % (i, tag))
            f.write(r)
            f.write(b"\n+\n")
            f.write(b"I" * len(r))
            f.write(b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="output prefix")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--contigs", type=int, default=10)
    ap.add_argument("--contig-len", type=int, default=200000)
    ap.add_argument("--pairs", type=int, default=20000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub", type=float, default=0.01)
    ap.add_argument("--indel", type=float, default=0.0004)
    ap.add_argument("--indel-read-frac", type=float, default=0.0)
    ap.add_argument("--plain", action="store_true", help="no repeat/N features")
    ap.add_argument("--reads-only", action="store_true")
    a = ap.parse_args()
    contigs = make_genome(a.seed, a.contigs, a.contig_len, features=not a.plain)
    if not a.reads_only:
        write_fasta(a.out + ".fa", contigs)
    r1, r2, _ = make_reads(a.seed + 1, contigs, a.pairs, a.read_len, a.sub, a.indel,
                           indel_read_frac=a.indel_read_frac)
    write_fastq(a.out + "_1_.fastq.gz", r1, 1)
    write_fastq(a.out + "_2_.fastq.gz", r2, 2)


if __name__ == "__main__":
    main()
