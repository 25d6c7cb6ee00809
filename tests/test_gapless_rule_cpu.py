"""The gapless rule of pm_gapless_kernel (pecaller_amd/csrc/pemap_sw.hip.h, K3a in DESIGN.md), restated in Python and checked
against the oracle's full DP (oracle/pemap_oracle.c: smith_waterman_align, pemapper.c:1694-1748) on windows built to stress it:
short reads over a two-letter alphabet (several diagonals perfect or tied, repeats), exact numbers of substitutions around the
rule's case boundaries, reads that skip reference bases (the one-deletion alignment that must defer the decision to the DP),
insertions, N on either side.  Whenever the rule decides, the DP must give the same double, the same start cell, and planes
in which the traceback never leaves plane 0 along the winning diagonal.  CPU only: this pins the RULE; the kernel that
implements it is pinned against goldens and the oracle in the GPU tests."""
import numpy as np
import oracle_py

MISS = np.frombuffer(np.array([0xBFD5555555555555], dtype=np.uint64).tobytes(), dtype=np.float64)[0]   # the reference's -1/3


def match(r, q, bis=False):
    """init_bonus_matrices as a predicate (pemapper.c:2006-2035), the form pm_match takes; bisulfite mode (2024-2035) lets a
    reference C (either case) pair with a read T (either case)"""
    m = (r == q) | (q == ord("N")) | (q == ord("n"))
    m |= (r == ord("N")) & (q >= ord("N"))
    m |= (r == ord("n")) & (q >= ord("n"))
    if bis:
        m |= ((r == ord("C")) | (r == ord("c"))) & ((q == ord("T")) | (q == ord("t")))
    return m


def gapless_rule(ref, read, max_x=2, bis=False):
    """-> None (left to the DP) or (score, row of the start cell)"""
    nn, mm = len(ref), len(read)
    nd = nn - mm + 1
    if nd <= 0:
        return None
    idx = np.arange(nd)[:, None] + np.arange(mm)[None, :]
    eq = match(ref[idx], read[None, :], bis)
    x = (~eq).sum(axis=1)
    if (x <= 1).any():
        cand = np.nonzero(x <= 1)[0]
    elif max_x >= 2 and (x == 2).any():
        first = np.where(eq.all(axis=1), mm, np.argmin(eq, axis=1))            # perfect prefix length
        last = np.where(eq.all(axis=1), -1, mm - 1 - np.argmin(eq[:, ::-1], axis=1))
        suf = mm - 1 - last
        before = np.concatenate([[0], np.maximum.accumulate(first)[:-1]])
        if ((before + suf) >= mm).any():
            return None                                                          # a one-deletion alignment without mismatch
        cand = np.nonzero(x == 2)[0]
    else:
        return None
    best, bd = None, -1
    for d in cand:                      # ascending rows, strict '>'
        s = np.float64(0.0)
        for j in range(mm):             # the DP's left fold along the diagonal
            s = s + (np.float64(1.0) if eq[d, j] else MISS)
        if best is None or s > best:
            best, bd = s, int(d)
    return best, bd + mm


def windows(seed, n):
    rng = np.random.default_rng(seed)
    for k in range(n):
        mm = int(rng.integers(16, 44))
        slack = int(rng.integers(0, 22))
        alpha = np.frombuffer((b"AC", b"ACGT", b"ACG")[k % 3], np.uint8)
        ref = alpha[rng.integers(0, len(alpha), mm + slack)].copy()
        if k % 5 == 0:                                     # a tandem repeat window
            per = int(rng.integers(1, 7))
            ref = np.tile(ref[:per], len(ref) // per + 1)[:len(ref)].copy()
        d = int(rng.integers(0, slack + 1))
        kind = k % 8
        if kind == 6 and slack - d >= 1:                   # the read skips b reference bases
            b = int(rng.integers(1, slack - d + 1))
            p = int(rng.integers(1, mm))
            read = np.concatenate([ref[d:d + p], ref[d + p + b:d + mm + b]]).copy()
        elif kind == 7:                                    # an inserted base
            p = int(rng.integers(1, mm - 1))
            read = np.concatenate([ref[d:d + p], alpha[rng.integers(0, len(alpha), 1)], ref[d + p:d + mm - 1]]).copy()
        else:
            read = ref[d:d + mm].copy()
        for q in rng.choice(mm, size=(0, 1, 2, 2, 3, 1, 0, 1)[kind], replace=False):
            read[q] = alpha[(int(np.nonzero(alpha == read[q])[0][0]) + 1) % len(alpha)] if len(alpha) > 1 else read[q]
        if k % 11 == 0:
            read[int(rng.integers(0, mm))] = ord("N")
        if k % 13 == 0:
            ref[int(rng.integers(0, len(ref)))] = ord("N")
        yield ref, read


def test_rule_agrees_with_the_full_dp():
    decided = {1: 0, 2: 0}
    deferred = 0
    for ref, read in windows(20241004, 40000):
        r1 = gapless_rule(ref, read, max_x=1)
        r2 = gapless_rule(ref, read, max_x=2)
        if r2 is None:
            deferred += 1
            continue
        score, st, pl = oracle_py.sw(ref, read, planes=True)
        mm = len(read)
        assert np.float64(score).view(np.uint64) == np.float64(r2[0]).view(np.uint64), (ref.tobytes(), read.tobytes(), score, r2)
        assert (int(st[0]), int(st[1]), int(st[2])) == (0, r2[1], mm), (ref.tobytes(), read.tobytes(), st, r2)
        # the traceback from plane 0 looks at the three planes of the previous cell of the diagonal (pemapper.c:1799-1813):
        # plane 0 must win there at every step that is followed by another one
        i = r2[1]
        for j in range(mm, 1, -1):
            a, b, c = pl[0, i - 1, j - 1], pl[1, i - 1, j - 1], pl[2, i - 1, j - 1]
            assert not (b > a) and not (c > max(a, b)), (ref.tobytes(), read.tobytes(), i, j)
            i -= 1
        if r1 is not None:
            assert r1 == r2
            decided[1] += 1
        else:
            decided[2] += 1
    # both cases and the deferral all occurred in numbers
    assert decided[1] > 10000 and decided[2] > 1500 and deferred > 4000, (decided, deferred)


def edge_windows(seed, n, bis):
    """the geometry and the letters where pm_match and the number of diagonals leave the common case: windows clipped at a contig
    end (1 .. 21 diagonals instead of 22), reference N / IUPAC / lower-case letters inside the window, lower-case and N in the
    read, and -- in bisulfite mode -- reads whose C's were converted to T"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    odd_ref = np.frombuffer(b"NNRYKMSWacgtn", np.uint8)
    for k in range(n):
        mm = int(rng.integers(16, 48))
        slack = int(rng.integers(0, 21)) if k % 4 else int(rng.integers(0, 4))      # nn - mm = 0 .. 20: a clipped window
        alpha = acgt if k % 3 else acgt[:2]
        ref = alpha[rng.integers(0, len(alpha), mm + slack)].copy()
        if k % 7 == 0:
            per = int(rng.integers(1, 5))
            ref = np.tile(ref[:per], len(ref) // per + 1)[:len(ref)].copy()
        d = int(rng.integers(0, slack + 1))
        kind = k % 8
        if kind == 6 and slack - d >= 1:
            b = int(rng.integers(1, slack - d + 1))
            p = int(rng.integers(1, mm))
            read = np.concatenate([ref[d:d + p], ref[d + p + b:d + mm + b]]).copy()
        elif kind == 7:
            p = int(rng.integers(1, mm - 1))
            read = np.concatenate([ref[d:d + p], acgt[rng.integers(0, 4, 1)], ref[d + p:d + mm - 1]]).copy()
        else:
            read = ref[d:d + mm].copy()
        for q in rng.choice(mm, size=(0, 1, 2, 2, 3, 1, 0, 1)[kind], replace=False):
            read[q] = acgt[(int(rng.integers(1, 4)) + int(np.searchsorted(acgt, read[q]))) % 4]
        if bis:
            conv = (read == ord("C")) & (rng.random(mm) < 0.9)           # bisulfite conversion of the read's C's
            read[conv] = ord("T")
        # odd letters: in the reference window (after the read was drawn from it) and in the read
        for _ in range(int(rng.integers(0, 4))):
            ref[int(rng.integers(0, len(ref)))] = odd_ref[int(rng.integers(0, len(odd_ref)))]
        if k % 5 == 0:
            q = int(rng.integers(0, mm))
            read[q] = (ord("N"), ord("n"), read[q] | 0x20)[k % 3]
        yield ref, read


def _check(windows_iter, bis):
    decided = deferred = narrow = odd = 0
    for ref, read in windows_iter:
        r2 = gapless_rule(ref, read, max_x=2, bis=bis)
        if r2 is None:
            deferred += 1
            continue
        score, st, pl = oracle_py.sw(ref, read, bisulfite=bis, planes=True)
        mm = len(read)
        assert np.float64(score).view(np.uint64) == np.float64(r2[0]).view(np.uint64), (ref.tobytes(), read.tobytes(), score, r2)
        assert (int(st[0]), int(st[1]), int(st[2])) == (0, r2[1], mm), (ref.tobytes(), read.tobytes(), st, r2)
        i = r2[1]
        for j in range(mm, 1, -1):
            a, b, c = pl[0, i - 1, j - 1], pl[1, i - 1, j - 1], pl[2, i - 1, j - 1]
            assert not (b > a) and not (c > max(a, b)), (ref.tobytes(), read.tobytes(), i, j)
            i -= 1
        decided += 1
        narrow += len(ref) - mm < 21
        odd += bool(np.isin(ref, np.frombuffer(b"NRYKMSWacgtn", np.uint8)).any())
    return decided, deferred, narrow, odd


def test_rule_on_clipped_windows_and_odd_letters():
    decided, deferred, narrow, odd = _check(edge_windows(77, 20000, False), False)
    assert decided > 8000 and deferred > 2000 and narrow > 7000 and odd > 4000, (decided, deferred, narrow, odd)


def test_rule_in_bisulfite_mode():
    decided, deferred, narrow, odd = _check(edge_windows(78, 20000, True), True)
    assert decided > 8000 and deferred > 2000 and narrow > 7000 and odd > 4000, (decided, deferred, narrow, odd)


def _add_ones(y, n):
    """The fold by binades of pm_add_ones (pemap_sw.hip.h), restated with Python's doubles."""
    import math
    while n > 0 and y < 1.0:
        y = y + 1.0
        n -= 1
    while n > 0:
        e = math.frexp(y)[1] - 1                # 2^e <= y < 2^(e+1)
        top = (2 << e) - 1
        k = min(n, top - int(y))
        if k > 0:
            y = y + float(k)
            n -= k
        if n > 0:
            y = y + 1.0
            n -= 1
    return y


def test_fold_by_binades_is_the_step_by_step_fold():
    """pemapper.c:2062-2081 adds the bonus one read base at a time; the kernel takes the steps of a binade at once."""
    miss = -1.0 / 3.0
    for m1 in range(0, 300):
        start = float(m1) + miss
        y = start
        for n in range(0, 520 - m1):
            assert _add_ones(start, n) == y, (m1, n)
            y = y + 1.0
    # two mismatches: fold, one more -1/3, fold again
    rng = np.random.default_rng(5)
    for _ in range(3000):
        mm = int(rng.integers(20, 512))
        m1, m2 = sorted(rng.choice(mm, 2, replace=False).tolist())
        y = 0.0
        for c in range(mm):
            y = y + (miss if c in (m1, m2) else 1.0)
        z = float(m1) + miss
        z = _add_ones(z, m2 - m1 - 1)
        z = z + miss
        z = _add_ones(z, mm - m2 - 1)
        assert z == y, (mm, m1, m2)
