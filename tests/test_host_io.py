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
