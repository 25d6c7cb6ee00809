"""Test-side readers for the reference's on-disk formats and a numpy index builder (checker side only).

Formats: .sdx text, .seq gz letters, .mdx raw LE u32, .idx gz of 2^32+1 LE u32,
pileup = gz of {u32 pos; u16 A,C,G,T,Del,Ins}, .mfile raw u32 per read, indel text rows.
"""
import gzip
import hashlib
import zlib
import numpy as np

PILE_DT = np.dtype([("pos", "<u4"), ("c", "<u2", (6,))])


def read_fasta(path):
    names, seqs, cur = [], [], []
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if names:
                    seqs.append(b"".join(cur))
                names.append(line[1:].strip().decode())
                cur = []
            else:
                cur.append(line.strip())
    seqs.append(b"".join(cur))
    return names, [np.frombuffer(s.upper(), dtype=np.uint8) for s in seqs]


def read_fastq(path):
    """the reference's record rule (pemapper.c:649-748): sequence = 2nd line; then skip to a line starting with '@'."""
    op = gzip.open if str(path).endswith(".gz") else open
    reads = []
    with op(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i + 1 < len(lines):
        if lines[i].startswith(b"@"):
            reads.append(lines[i + 1])
            i += 4
        else:
            i += 1
    return reads


def pack_reads(reads, stride=304):
    n = len(reads)
    buf = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for i, r in enumerate(reads):
        buf[i, :len(r)] = np.frombuffer(r, dtype=np.uint8)
        lens[i] = len(r)
    return buf, lens


def kmer_index(contigs, bisulfite=False):
    """(mers, ukmer, ustart, contig_starts) exactly as index_genome_whole.c:169-177, 206-299, 334-342 define them:
    A=0 C=1 G=2 T=3, any other letter except N = 0, N resets the window, positions in len-15 compressed coordinates,
    grouped by k-mer ascending and inside a k-mer in genome order."""
    code = np.zeros(256, dtype=np.uint64)
    code[ord("C")] = 3 if bisulfite else 1
    code[ord("G")] = 2
    code[ord("T")] = 3
    keys, poss, starts = [], [], [0]
    gpos = 0
    for c in contigs:
        n = len(c)
        if n >= 16:
            cd = code[c]
            isn = (c == ord("N")).astype(np.int64)
            cs = np.concatenate([[0], np.cumsum(isn)])
            nwin = n - 15
            valid = (cs[16:16 + nwin] - cs[0:nwin]) == 0
            k = np.zeros(nwin, dtype=np.uint64)
            for i in range(16):
                k = (k << np.uint64(2)) | cd[i:i + nwin]
            p = np.nonzero(valid)[0]
            keys.append(k[p].astype(np.uint32))
            poss.append((p + gpos).astype(np.uint32))
        gpos += max(n - 15, 0) if n >= 15 else n - 15
        starts.append(gpos)
    keys = np.concatenate(keys) if keys else np.zeros(0, np.uint32)
    poss = np.concatenate(poss) if poss else np.zeros(0, np.uint32)
    order = np.argsort(keys, kind="stable")
    mers = poss[order]
    sk = keys[order]
    ukmer, first = np.unique(sk, return_index=True)
    ustart = np.concatenate([first, [len(sk)]]).astype(np.uint32)
    return mers, ukmer.astype(np.uint32), ustart, np.array(starts, dtype=np.uint32)


def idx_to_compact(idx_gz_path, chunk=1 << 26):
    """stream-inflate a reference .idx and return (ukmer, ustart) -- the run boundaries of the 2^32+1 prefix table."""
    d = zlib.decompressobj(16 + zlib.MAX_WBITS)
    uk, us = [], []
    base = 0
    prev = None
    carry = b""
    with open(idx_gz_path, "rb") as f:
        while True:
            raw = f.read(1 << 22)
            if not raw:
                break
            out = carry + d.decompress(raw)
            nfull = len(out) // 4
            a = np.frombuffer(out[:nfull * 4], dtype="<u4")
            carry = out[nfull * 4:]
            if len(a) == 0:
                continue
            # pos_index[k+1] > pos_index[k]  <=> k-mer k occurs
            ext = a if prev is None else np.concatenate([[prev], a])
            off = base if prev is None else base - 1
            ch = np.nonzero(ext[1:] != ext[:-1])[0]
            uk.append((ch + off).astype(np.uint64))
            us.append(ext[ch])
            prev = a[-1]
            base += len(a)
    assert base == (1 << 32) + 1, base
    ukmer = np.concatenate(uk).astype(np.uint32)
    ustart = np.concatenate(us + [np.array([prev], dtype=np.uint32)]).astype(np.uint32)
    return ukmer, ustart


def read_sdx(path):
    with open(path) as f:
        n = int(f.readline())
        lens, names = [], []
        for _ in range(n):
            a, b = f.readline().split()[:2]
            lens.append(int(a))
            names.append(b)
        idepth = int(f.readline())
    return lens, names, idepth


def read_pileup(path):
    with gzip.open(path, "rb") as f:
        raw = f.read()
    return np.frombuffer(raw, dtype=PILE_DT)


def read_indel(path):
    """rows -> sorted list of (contig, pos, ref, tot, ref_reads, dels, n_ins, tuple(sorted(ins strings)))"""
    rows = []
    with gzip.open(path, "rt") as f:
        f.readline()
        for line in f:
            p = line.rstrip("\n").split("\t")
            if len(p) < 7:
                continue
            rows.append((p[0], int(p[1]), p[2], int(p[3]), int(p[4]), int(p[5]), int(p[6]), tuple(sorted(p[7:]))))
    return sorted(rows)


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()
