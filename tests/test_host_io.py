"""The host programs' gz plumbing (pecaller_amd/csrc/host_io.h): the parallel multi-member writer must inflate -- with Python's
gzip, i.e. zlib, what the reference's readers use -- to exactly the bytes written, and the threaded reader must return them in
pecaller's 4 + 12 byte pattern.  CPU only."""
import gzip
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parallel_gz_writer_and_threaded_reader(tmp_path):
    exe = str(tmp_path / "host_io_check")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "csrc", "host_io_check.c"), "-lz", "-lpthread"])
    for total, threads in ((0, 4), (5, 1), (150000000, 4)):
        out = str(tmp_path / ("t%d.gz" % total))
        r = subprocess.run([exe, out, str(total), str(threads)], stdout=subprocess.PIPE)
        assert r.returncode == 0 and b"ok %d" % total in r.stdout, (r.returncode, r.stdout)
        data = np.frombuffer(gzip.open(out, "rb").read(), np.uint8)
        i = np.arange(total, dtype=np.uint64)
        exp = ((((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(13)) ^ (i >> np.uint64(7))).astype(np.uint8)
        assert len(data) == total and np.array_equal(data, exp)
        if total > (64 << 20):
            # several members were written (the file holds more than one gzip header)
            raw = open(out, "rb").read()
            assert raw.count(b"\x1f\x8b\x08\x00") >= 4


def test_threaded_reader_dies_on_a_broken_stream(tmp_path):
    """gzread < 0 or a stream cut short is not an end of input: pecaller_hip must not call on the part that inflated."""
    exe = str(tmp_path / "host_io_check")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "csrc", "host_io_check.c"), "-lz", "-lpthread"])
    rng = np.random.default_rng(3)
    payload = rng.integers(0, 256, 5 << 20, dtype=np.uint8).tobytes()
    good = str(tmp_path / "good.gz")
    with gzip.open(good, "wb") as f:
        f.write(payload)
    r = subprocess.run([exe, good, "read"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"read %d" % len(payload) in r.stdout
    raw = open(good, "rb").read()
    cut = str(tmp_path / "cut.gz")
    open(cut, "wb").write(raw[:len(raw) // 2])
    r = subprocess.run([exe, cut, "read"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"truncated or not a valid gzip" in r.stderr, (r.returncode, r.stdout, r.stderr)
    bad = bytearray(raw)
    for k in range(len(raw) // 3, len(raw) // 3 + 64):
        bad[k] ^= 0x5A
    garbled = str(tmp_path / "garbled.gz")
    open(garbled, "wb").write(bytes(bad))
    r = subprocess.run([exe, garbled, "read"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"truncated or not a valid gzip" in r.stderr, (r.returncode, r.stdout, r.stderr)


# ---------------------------------------------------------------------------------------------------------------------------------
# fast_inflate.h + gzsrc: the bytes zlib gives, for every kind of deflate block and gzip header; damaged streams are refused
# ---------------------------------------------------------------------------------------------------------------------------------
import zlib
import pytest


@pytest.fixture(scope="module")
def fic(tmp_path_factory):
    d = tmp_path_factory.mktemp("fic")
    exe = str(d / "fast_inflate_check")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "csrc", "fast_inflate_check.c"), "-lz", "-lpthread"])
    return exe


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, name=None, extra=None, comment=None, hcrc=False, memlevel=8):
    """one gzip member, header fields by hand (RFC 1952)"""
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    body = c.compress(data) + c.flush()
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    h = bytes([0x1f, 0x8b, 8, flg, 1, 2, 3, 4, 0, 3])
    if extra is not None:
        h += len(extra).to_bytes(2, "little") + extra
    if name is not None:
        h += name + b"\0"
    if comment is not None:
        h += comment + b"\0"
    if hcrc:
        h += (zlib.crc32(h) & 0xFFFF).to_bytes(2, "little")
    return h + body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")


def _cases():
    rng = np.random.default_rng(11)
    fq = gzip.open(os.path.join(ROOT, "tests", "golden", "g1_1_.fastq.gz"), "rb").read()[:3000000]
    rnd = rng.integers(0, 256, 700000, dtype=np.uint8).tobytes()
    rec = np.zeros(200000, np.dtype([("pos", "<u4"), ("c", "<u2", 6)]))
    rec["pos"] = np.arange(200000) + 1000
    rec["c"] = rng.poisson(5, (200000, 6))
    pile = rec.tobytes()
    runs = b"".join(bytes([int(v)]) * int(n) for v, n in zip(rng.integers(65, 70, 3000), rng.integers(1, 600, 3000)))
    short = b"".join((b"ACGTAC"[: int(k)] * 50) for k in rng.integers(2, 7, 400))        # distances 2..6
    yield "fastq level 6", _gz(fq), fq
    yield "fastq level 1", _gz(fq, 1), fq
    yield "fastq level 9", _gz(fq, 9), fq
    yield "fastq fixed codes", _gz(fq[:400000], 6, zlib.Z_FIXED), fq[:400000]
    yield "fastq huffman only", _gz(fq[:400000], 6, zlib.Z_HUFFMAN_ONLY), fq[:400000]
    yield "fastq small hash (many blocks)", _gz(fq[:900000], 6, memlevel=1), fq[:900000]
    yield "random bytes (stored blocks)", _gz(rnd), rnd
    yield "level 0", _gz(fq[:300000], 0), fq[:300000]
    yield "pileup records", _gz(pile), pile
    yield "long runs (distance 1)", _gz(runs), runs
    yield "short periods", _gz(short), short
    yield "one byte", _gz(b"x"), b"x"
    yield "empty member", _gz(b""), b""
    yield "header fields", _gz(fq[:5000], name=b"reads_1_.fastq", extra=b"\x01\x02abc", comment=b"a comment", hcrc=True), fq[:5000]
    many = [fq[i:i + 70001] for i in range(0, 700010, 70001)]
    yield "ten members, an empty one among them", b"".join(_gz(m) for m in many[:5]) + _gz(b"") + b"".join(_gz(m) for m in many[5:]), b"".join(many)
    yield "bytes behind the last member", _gz(fq[:9000]) + b"\0" * 512, fq[:9000]
    yield "not compressed", fq[:1234567], fq[:1234567]
    yield "empty file", b"", b""
    big = fq * 12                                            # 36 MB: the buffer wraps many times at small blocks
    yield "36 MB", _gz(big, 1), big


@pytest.mark.parametrize("block", [4096, 65536, 1 << 20])
def test_fast_inflate_gives_zlibs_bytes(fic, tmp_path, block):
    for name, comp, plain in _cases():
        if block == 4096 and len(plain) > 4000000:
            continue
        src, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
        open(src, "wb").write(comp)
        r = subprocess.run([fic, "cat", src, out, str(block)], stdout=subprocess.PIPE)
        assert r.returncode == 0, (name, r.stdout)
        got = open(out, "rb").read()
        assert got == plain, (name, len(got), len(plain))
        want_mode = b"mode 1" if comp[:2] == b"\x1f\x8b" else b"mode 2"
        assert want_mode in r.stdout, (name, r.stdout)


def test_crc_by_carry_less_multiplies_is_zlibs(fic):
    r = subprocess.run([fic, "crc"], stdout=subprocess.PIPE)
    assert r.returncode == 0 and b"crc ok" in r.stdout, r.stdout


def test_fast_inflate_refuses_damaged_streams(fic, tmp_path):
    fq = gzip.open(os.path.join(ROOT, "tests", "golden", "g1_1_.fastq.gz"), "rb").read()[:600000]
    good = _gz(fq)
    src, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")

    def run(data):
        open(src, "wb").write(data)
        r = subprocess.run([fic, "cat", src, out, "65536"], stdout=subprocess.PIPE)
        return r.returncode, r.stdout, open(out, "rb").read()

    # cut anywhere: an error, and what was produced before it is a prefix of the data
    for cut in (11, 200, len(good) // 2, len(good) - 9, len(good) - 8, len(good) - 1):
        rc, msg, got = run(good[:cut])
        assert rc == 3 and fq.startswith(got), (cut, rc, msg)
    # the trailer's CRC, its length
    bad = bytearray(good); bad[-6] ^= 1
    assert run(bytes(bad))[0] == 3 and b"CRC" in run(bytes(bad))[1]
    bad = bytearray(good); bad[-2] ^= 1
    assert run(bytes(bad))[0] == 3 and b"length" in run(bytes(bad))[1]
    # a flipped bit in the data: refused somewhere (a code, a distance, or at the latest the CRC)
    rng = np.random.default_rng(5)
    for k in rng.integers(12, len(good) - 8, 60):
        bad = bytearray(good); bad[int(k)] ^= 1 << int(rng.integers(0, 8))
        rc, msg, got = run(bytes(bad))
        assert rc == 3, (int(k), msg)
    # a second member that is damaged: the first one's bytes came through
    rc, msg, got = run(good + good[:300])
    assert rc == 3 and got[:len(fq)] == fq


def test_fast_inflate_under_the_address_sanitizer(tmp_path):
    """damaged and random input must not touch memory outside the buffers: the decoder built with -fsanitize=address,undefined and fed
    streams with bytes overwritten, cut, or made of noise behind a valid header"""
    exe = str(tmp_path / "fic_asan")
    r = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(ROOT, "tests", "csrc", "fast_inflate_check.c"), "-lz", "-lpthread"], stderr=subprocess.PIPE)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime here: " + r.stderr.decode()[-200:])
    rng = np.random.default_rng(17)
    fq = gzip.open(os.path.join(ROOT, "tests", "golden", "g1_1_.fastq.gz"), "rb").read()[:200000]
    good = _gz(fq)
    src, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    seen = set()
    for trial in range(150):
        kind = trial % 3
        if kind == 0:
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 6))):
                k = int(rng.integers(10, len(bad)))
                bad[k] = int(rng.integers(0, 256))
        elif kind == 1:
            bad = bytearray(good[:int(rng.integers(10, len(good)))])
        else:
            bad = bytearray(good[:10]) + bytearray(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes())
        open(src, "wb").write(bytes(bad))
        r = subprocess.run([exe, "cat", src, out, "4096"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode in (0, 3), (trial, r.returncode, r.stderr[-600:])
        seen.add(r.returncode)
    assert 3 in seen
