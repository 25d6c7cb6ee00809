"""The C host program (pecaller_amd/pemapper_hip) run as a drop-in for the reference's pemapper / pemapper_tsw command
lines on the golden fixture: output FILES compared with the reference's."""
import gzip
import os
import shutil
import subprocess
import numpy as np
import pytest
import fixtures
import refio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pecaller_amd", "pemapper_hip")


def _prep(tmp_path):
    """index files next to each other as the reference expects: <base>.sdx and <base>.seq (gz of the letters)"""
    ix = fixtures.index()
    shutil.copy(os.path.join(fixtures.GOLD, "g1.sdx"), tmp_path / "g1.sdx")
    with gzip.open(tmp_path / "g1.seq", "wb", compresslevel=1) as f:
        f.write(ix["genome"].tobytes())
    return str(tmp_path / "g1.sdx")


def _compare(name, out, f1, f2):
    m = fixtures.meta()[name]
    assert np.array_equal(np.fromfile(f1 + ".mfile", dtype="<u4"), fixtures.golden_m(name, 1))
    if f2:
        assert np.array_equal(np.fromfile(f2 + ".mfile", dtype="<u4"), fixtures.golden_m(name, 2))
    pile = refio.read_pileup(out + ".pileup.gz")
    assert len(pile) == m["pileup_records"] and refio.md5(pile) == m["pileup_md5"]
    assert refio.read_indel(out + ".indel.txt.gz") == refio.read_indel(os.path.join(fixtures.GOLD, name + ".indel.txt.gz"))
    assert open(out + ".summary.txt").read() == open(os.path.join(fixtures.GOLD, name + ".summary.txt")).read()


@pytest.mark.parametrize("name", ["r150", "r100", "r250"])
def test_cli_outputs_equal_reference(tmp_path, name):
    assert os.path.exists(EXE), "build with make -C pecaller_amd/csrc"
    sdx = _prep(tmp_path)
    s = fixtures.SETS[name]
    f1 = str(tmp_path / ("%s_1_.fastq.gz" % s["prefix"]))
    shutil.copy(os.path.join(fixtures.GOLD, os.path.basename(f1)), f1)
    f2 = None
    out = str(tmp_path / "out")
    extra = [str(x) for x in fixtures.meta()[name]["extra_args"]]
    if s["paired"]:
        f2 = str(tmp_path / ("%s_2_.fastq.gz" % s["prefix"]))
        shutil.copy(os.path.join(fixtures.GOLD, os.path.basename(f2)), f2)
        cmd = [EXE, out, sdx, "p", f1, f2, "500", "0", "N", "0.85", "8", "200000000"] + extra
    else:
        cmd = [EXE, out, sdx, "s", f1, "N", "0.85", "8", "200000000"] + extra
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    _compare(name, out, f1, f2)


def test_cli_array_mode_and_max_reads(tmp_path):
    """'pa' array files (pemapper.c:307-348) and the max_reads cut (709-710)"""
    sdx = _prep(tmp_path)
    f1 = str(tmp_path / "g1_1_.fastq.gz")
    f2 = str(tmp_path / "g1_2_.fastq.gz")
    shutil.copy(os.path.join(fixtures.GOLD, "g1_1_.fastq.gz"), f1)
    shutil.copy(os.path.join(fixtures.GOLD, "g1_2_.fastq.gz"), f2)
    (tmp_path / "a1.txt").write_text(f1 + "\n")
    (tmp_path / "a2.txt").write_text(f2 + "\n")
    out = str(tmp_path / "outa")
    subprocess.check_call([EXE, out, sdx, "pa", str(tmp_path / "a1.txt"), str(tmp_path / "a2.txt"), "500", "0", "N", "0.85", "8",
                           "5000"], stdout=subprocess.DEVNULL)
    m1 = np.fromfile(f1 + ".mfile", dtype="<u4")
    assert len(m1) == 5000 and np.array_equal(m1, fixtures.golden_m("r150", 1)[:5000])
    # bad arguments fail like the reference: usage text, exit status 1
    r = subprocess.run([EXE, out, sdx, "p", f1], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"Usage" in r.stdout
    r = subprocess.run([EXE, out, sdx, "s", f1, "N", "0.85", "1", "100"], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"Max_threads" in r.stdout


@pytest.mark.parametrize("workers", ["1", "4"])
def test_cli_array_mode_several_files_in_flight(tmp_path, workers, monkeypatch):
    """the reference's usual run: an array of file pairs into one output set (pemapper.c:307-348, map_directory_array.pl:92-100).
    The golden reads cut into five uneven file pairs (one of them plain text), read and mapped by four workers side by side (and by
    one, the old way): every pair's .mfile holds its reads' coordinates, and the pileup, the insertions and the summary of the set
    are the reference's for the whole read set (counters are sums, the order of the files is free)."""
    monkeypatch.setenv("PEMAPPER_FILE_WORKERS", workers)
    sdx = _prep(tmp_path)
    g1, g2 = (gzip.open(os.path.join(fixtures.GOLD, "g1_%d_.fastq.gz" % k)).read().split(b"\n") for k in (1, 2))
    cuts = [0, 3000, 3001, 9000, 14500, 20000]
    n1, n2 = [], []
    for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        fa, fb = str(tmp_path / ("part%d_1_.fastq" % k)), str(tmp_path / ("part%d_2_.fastq" % k))
        if k != 2:
            fa, fb = fa + ".gz", fb + ".gz"
        for fn, lines in ((fa, g1), (fb, g2)):
            data = b"\n".join(lines[4 * a:4 * b]) + b"\n"
            (gzip.open(fn, "wb", compresslevel=1) if fn.endswith(".gz") else open(fn, "wb")).write(data)
        n1.append(fa)
        n2.append(fb)
    (tmp_path / "a1.txt").write_text("\n".join(n1) + "\n")
    (tmp_path / "a2.txt").write_text("\n".join(n2) + "\n")
    out = str(tmp_path / "outm")
    subprocess.check_call([EXE, out, sdx, "pa", str(tmp_path / "a1.txt"), str(tmp_path / "a2.txt"), "500", "0", "N", "0.85", "16", "200000000"],
                          stdout=subprocess.DEVNULL)
    m = fixtures.meta()["r150"]
    for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        assert np.array_equal(np.fromfile(n1[k] + ".mfile", dtype="<u4"), fixtures.golden_m("r150", 1)[a:b]), k
        assert np.array_equal(np.fromfile(n2[k] + ".mfile", dtype="<u4"), fixtures.golden_m("r150", 2)[a:b]), k
    pile = refio.read_pileup(out + ".pileup.gz")
    assert len(pile) == m["pileup_records"] and refio.md5(pile) == m["pileup_md5"]
    assert refio.read_indel(out + ".indel.txt.gz") == refio.read_indel(os.path.join(fixtures.GOLD, "r150.indel.txt.gz"))
    assert open(out + ".summary.txt").read() == open(os.path.join(fixtures.GOLD, "r150.summary.txt")).read()


def test_cli_plain_text_input_and_uneven_mate_files(tmp_path):
    """the two mate files are scanned by two threads (fill_rows): plain-text fastq gives the files the gz input gives (gzopen reads
    both, pemapper.c:626); a second mate file that ends early ends the run at its last record, as the reference's loop does
    (pemapper.c:739-748); a first-mate read of 12 bases or fewer ends it there (663); text between records that does not start
    with '@' is skipped by the scan for the next header (722-733)"""
    sdx = _prep(tmp_path)
    g1, g2 = (gzip.open(os.path.join(fixtures.GOLD, "g1_%d_.fastq.gz" % k)).read() for k in (1, 2))
    f1, f2 = str(tmp_path / "p_1_.fastq"), str(tmp_path / "p_2_.fastq")
    open(f1, "wb").write(g1)
    open(f2, "wb").write(g2)
    out = str(tmp_path / "outp")
    log = subprocess.check_output([EXE, out, sdx, "p", f1, f2, "500", "0", "N", "0.85", "8", "200000000"])
    assert b"read and mapped in" in log
    _compare("r150", out, f1, f2)
    # second mate file cut after 7,000 records, junk lines between two records of the first, a short first-mate read at record 9,000
    l1, l2 = g1.split(b"\n"), g2.split(b"\n")
    l1[4 * 100:4 * 100] = [b"junk line", b"+another"]
    cut = l1[:]
    cut[4 * 9000 + 1 + 2] = b"ACGTACGTACGT"            # (+ 2: the junk lines shifted everything behind record 100)
    for tag, a, b, n in (("e", l1, l2[:4 * 7000], 7000), ("s", cut, l2, 9000)):
        fa, fb = str(tmp_path / (tag + "_1_.fastq")), str(tmp_path / (tag + "_2_.fastq"))
        open(fa, "wb").write(b"\n".join(a))
        open(fb, "wb").write(b"\n".join(b) + (b"\n" if tag == "e" else b""))
        subprocess.check_call([EXE, str(tmp_path / ("out" + tag)), sdx, "p", fa, fb, "500", "0", "N", "0.85", "8", "200000000"],
                              stdout=subprocess.DEVNULL)
        m1, m2 = np.fromfile(fa + ".mfile", dtype="<u4"), np.fromfile(fb + ".mfile", dtype="<u4")
        assert len(m1) == n and len(m2) == n, (tag, len(m1), len(m2))
        # every pair is mapped on its own (no state between pairs): the first n coordinates are the full run's
        assert np.array_equal(m1, fixtures.golden_m("r150", 1)[:n]) and np.array_equal(m2, fixtures.golden_m("r150", 2)[:n])


def test_index_builder_cli_writes_reference_files(tmp_path):
    """index_genome_hip: the reference builder's dialogue on stdin, its four files out (.sdx text, .seq letters, .mdx, .idx)"""
    exe = os.path.join(ROOT, "pecaller_amd", "index_genome_hip")
    assert os.path.exists(exe)
    names, contigs = fixtures.genome()
    fa = tmp_path / "g1.fa"
    with gzip.open(os.path.join(fixtures.GOLD, "g1.fa.gz"), "rb") as f, open(fa, "wb") as o:
        shutil.copyfileobj(f, o)
    ans = "S\n%d\n%s\n%s\nN\n" % (len(names) + 2, fa, tmp_path / "out")
    subprocess.run([exe], input=ans.encode(), stdout=subprocess.DEVNULL, check=True)
    assert open(tmp_path / "out.sdx").read() == open(os.path.join(fixtures.GOLD, "g1.sdx")).read()
    ix = fixtures.index()
    assert gzip.open(tmp_path / "out.seq", "rb").read() == ix["genome"].tobytes()
    assert np.array_equal(np.fromfile(tmp_path / "out.mdx", dtype="<u4"), ix["mers"])
    assert refio.md5(np.fromfile(tmp_path / "out.mdx", dtype="<u4")) == fixtures.meta()["index"]["mdx_md5"]
    uk, us = refio.idx_to_compact(str(tmp_path / "out.idx"))
    assert np.array_equal(uk, ix["ukmer"]) and np.array_equal(us, ix["ustart"])


def test_cli_maps_from_the_index_files_of_the_builder(tmp_path, monkeypatch):
    """index_genome_hip's four files chained into pemapper_hip with PEMAP_INDEX_FROM_FILES=1: the .idx is inflated and the .mdx
    read as the reference does (init_index_buffer, pemapper.c:2129-2155) and handed to pemap_dev_load_index; the outputs must be
    the reference's r150 files."""
    exe = os.path.join(ROOT, "pecaller_amd", "index_genome_hip")
    names, contigs = fixtures.genome()
    fa = tmp_path / "g1.fa"
    with gzip.open(os.path.join(fixtures.GOLD, "g1.fa.gz"), "rb") as f, open(fa, "wb") as o:
        shutil.copyfileobj(f, o)
    ans = "S\n%d\n%s\n%s\nN\n" % (len(names) + 2, fa, tmp_path / "gx")
    subprocess.run([exe], input=ans.encode(), stdout=subprocess.DEVNULL, check=True)
    f1 = str(tmp_path / "g1_1_.fastq.gz")
    f2 = str(tmp_path / "g1_2_.fastq.gz")
    shutil.copy(os.path.join(fixtures.GOLD, "g1_1_.fastq.gz"), f1)
    shutil.copy(os.path.join(fixtures.GOLD, "g1_2_.fastq.gz"), f2)
    out = str(tmp_path / "outf")
    monkeypatch.setenv("PEMAP_INDEX_FROM_FILES", "1")
    r = subprocess.run([EXE, out, str(tmp_path / "gx.sdx"), "p", f1, f2, "500", "0", "N", "0.85", "8", "200000000"], stdout=subprocess.PIPE)
    assert r.returncode == 0, r.stdout[-2000:]
    _compare("r150", out, f1, f2)
