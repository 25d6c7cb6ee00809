"""the per-site caller fixture (tests/golden/pecall_sites.*, made by tests/golden/make_golden_pecall_sites.py) and the
reference's text format for calls (pecaller.c:1564, 1578, 1600, 1675-1680)"""
import gzip
import os
import numpy as np
import refio

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
GEN = "ACGTDIMRWSYKEHN"           # int_to_gen, pecaller.c:2910-2943
TYPES = ["", "SNP", "DEL", "INS", "LOW", "MULTIALLELIC", "MESS"]


def load(tag="pecall_sites"):
    """tag = "pecall_sites" (no pedigree) or "pecall_ped" (two trios, made by make_golden_pecall_ped.py; adds "ped")"""
    z = np.load(os.path.join(GOLD, tag + ".npz"))
    names = [str(x) for x in z["names"]]
    cols = [str(x) for x in z["columns"]]
    perm = [names.index(c) for c in cols]          # the reference's sample order = its directory order
    reads = z["reads"][:, perm, :]
    pos = z["pos"]
    _, seqs = refio.read_fasta(os.path.join(GOLD, "g1.fa.gz"))
    seq = np.concatenate(seqs)
    ref = seq[pos]
    dom = np.full(len(pos), 14, np.uint8)
    for k, ch in enumerate(b"ACGT"):
        dom[ref == ch] = k
    base = gzip.open(os.path.join(GOLD, tag + ".base.txt.gz"), "rt").read().split("\n")
    snp = open(os.path.join(GOLD, tag + ".snp.txt")).read().split("\n")
    base_rows = {int(r.split("\t")[1]): r for r in base[1:] if r}
    snp_rows = {int(r.split("\t")[1]): r for r in snp[1:] if r}
    out = dict(reads=reads, pos=pos, ref=ref, dom=dom, base_rows=base_rows, snp_rows=snp_rows)
    if "dad" in z:
        # the pedigree in the reference's sample numbering (its column order); kids are listed in ped-file order
        inv = {old: new for new, old in enumerate(perm)}
        remap = lambda a: np.array([inv[int(a[o])] if a[o] >= 0 else -1 for o in perm], np.int32)
        out["ped"] = dict(dad=remap(z["dad"]), mom=remap(z["mom"]), sex=np.array([int(z["sex"][o]) for o in perm], np.int32),
                          order=[inv[o] for o in range(len(names))], denovo_rate=float(z["denovo_rate"][0]))
    return out


def base_row(contig, pos1, refch, call, p):
    return "%s\t%d\t%s" % (contig, pos1, refch) + "".join("\t%s\t%g" % (GEN[c], x) for c, x in zip(call, p))


def snp_row(contig, pos1, refch, call, p, typ, ac, denovo=0):
    alle = ",".join("ACGTDI"[a] for a in range(6) if ac[a] > 0)
    cnts = ",".join("%d" % ac[a] for a in range(6) if ac[a] > 0)
    return "%s\t%d\t%s\t%s\t%s\t%s" % (contig, pos1, refch, alle, cnts, ("DENOVO_" if denovo > 0 else "") + TYPES[typ]) + "".join("\t%s\t%g" % (GEN[c], x) for c, x in zip(call, p))


def load_guide():
    """the BED guide-mode fixture (make_golden_pecall_guide.py): the columns the guide visits, in the reference's sample
    order, with the chromosome byte the caller gets (type + 16 where HAPLOID is forced)"""
    z = np.load(os.path.join(GOLD, "pecall_guide.npz"))
    names = [str(x) for x in z["names"]]
    cols = [str(x) for x in z["columns"]]
    perm = [names.index(c) for c in cols]
    lens, cn, _ = refio.read_sdx(os.path.join(GOLD, "pecall_guide.sdx"))
    starts = np.concatenate([[0], np.cumsum(np.array(lens) + 15)])
    _, seqs = refio.read_fasta(os.path.join(GOLD, "g1.fa.gz"))
    seq = np.concatenate(seqs)
    rec = {int(p): z["reads"][i][perm] for i, p in enumerate(z["pos"])}
    ctype = {"chrx": 1, "chry": 2, "chrmt": 3}
    out_reads, dom, chrom, key = [], [], [], []
    for line in open(os.path.join(GOLD, "pecall_guide.bed")):
        c, lo, hi = line.split()
        ci = cn.index(c)
        t = ctype.get(c.lower(), 0)
        for p1 in range(int(lo), int(hi) + 1):
            g = int(starts[ci]) + p1 - 1
            out_reads.append(rec.get(g, np.zeros((len(names), 6), np.uint16)))
            ch = seq[g]
            dom.append(b"ACGT".index(bytes([ch])) if bytes([ch]) in (b"A", b"C", b"G", b"T") else 14)
            chrom.append(t | (16 if t in (2, 3) else 0))
            key.append((c, p1, chr(ch)))
    base = gzip.open(os.path.join(GOLD, "pecall_guide.base.txt.gz"), "rt").read().split("\n")
    snp = open(os.path.join(GOLD, "pecall_guide.snp.txt")).read().split("\n")
    rows = {(r.split("\t")[0], int(r.split("\t")[1])): r for r in base[1:] if r}
    srows = {(r.split("\t")[0], int(r.split("\t")[1])): r for r in snp[1:] if r}
    return dict(reads=np.array(out_reads, np.uint16), dom=np.array(dom, np.uint8), chrom=np.array(chrom, np.uint8), key=key,
                base_rows=rows, snp_rows=srows, names=names, columns=cols, z=z)
