"""The C host program pecaller_amd/pecaller_hip run with the reference's pecaller command line on the site-caller fixture:
its FILES against the oracle (for the column order the directory gave) and, where the order is the fixture's, the reference's text."""
import gzip
import os
import shutil
import struct
import subprocess
import numpy as np
import pytest
import oracle_py
import pecall_sites_fixture as fx
import refio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pecaller_amd", "pecaller_hip")


@pytest.mark.parametrize("tile_log2,threads", [(None, "2"), (None, "8"), ("10", "8"), ("11", "4")])
def test_pecaller_cli_outputs(tmp_path, tile_log2, threads, monkeypatch):
    """(tile_log2 = 11 also sets PECALLER_POST_CAP=3: the list of the columns with a posterior that is not 1 is too short for every
    tile, and the call is made again with the size the library asked for; tile_log2 = 10: the stream merge and the device calls take the fixture's 6,000 positions in six ranges of 1,024 -- records
    of a stream on both sides of a range boundary, streams without a record in a range, the last range partly empty; threads = 8:
    seven threads walk the 20 streams and format the rows, 2: one does)"""
    assert os.path.exists(EXE), "build with make -C pecaller_amd/csrc"
    if tile_log2:
        monkeypatch.setenv("PECALLER_TILE_LOG2", tile_log2)
    if tile_log2 == "11":
        monkeypatch.setenv("PECALLER_POST_CAP", "3")
    z = np.load(os.path.join(fx.GOLD, "pecall_sites.npz"))
    names = [str(x) for x in z["names"]]
    reads, pos, pad = z["reads"], z["pos"], int(z["pad"][0])
    _, seqs = refio.read_fasta(os.path.join(fx.GOLD, "g1.fa.gz"))
    shutil.copy(os.path.join(fx.GOLD, "g1.sdx"), tmp_path / "g1.sdx")
    with gzip.open(tmp_path / "g1.seq", "wb", compresslevel=1) as f:
        f.write(np.concatenate(seqs).tobytes())
    run = tmp_path / "run"
    run.mkdir()
    for s, nm in enumerate(names):              # the generator's files, padding columns included (the .dist counts them)
        recs = [struct.pack("<I6H", int(pos[i]), *[int(x) for x in reads[i, s]]) for i in range(len(pos)) if reads[i, s].sum() > 0]
        recs += [struct.pack("<I6H", int(pos[-1]) + 1 + k, 20, 0, 0, 0, 0, 0) for k in range(pad)]
        with gzip.open(run / ("%s.pileup.gz" % nm), "wb", compresslevel=1) as f:
            f.write(b"".join(recs))
    subprocess.check_call([EXE, "pileup", str(tmp_path / "g1.sdx"), "20", "out", "0.95", "0.001", "n", threads, "n"], cwd=run, stdout=subprocess.DEVNULL)
    base = gzip.open(run / "out.base.gz", "rt").read().split("\n")
    cols = [c for c in base[0].split("\t")[3:] if c]
    assert sorted(cols) == sorted(names)
    perm = [names.index(c) for c in cols]
    f = fx.load()
    ref_order = [str(x) for x in z["columns"]]
    # ---- against the oracle in the order this directory gave
    r = reads[:, perm, :]
    call, p, typ, ac, _ = oracle_py.call_sites(r, f["dom"])
    rows = {int(x.split("\t")[1]): x for x in base[1:] if x}
    snp = open(run / "out.snp").read().split("\n")
    srows = {int(x.split("\t")[1]): x for x in snp[1:] if x}
    n_base = n_snp = 0
    for i, q in enumerate(pos):
        pos1 = int(q) + 1
        if f["dom"][i] > 3 or r[i].sum() == 0:
            assert pos1 not in rows
            continue
        n_base += 1
        assert rows[pos1] == fx.base_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i]), pos1
        if typ[i] > 0:
            n_snp += 1
            assert srows[pos1] == fx.snp_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i], typ[i], ac[i]), pos1
        else:
            assert pos1 not in srows
    assert n_base == len(f["base_rows"]) and n_snp > 100
    piles = gzip.open(run / "out.piles.gz", "rt").read().split("\n")
    prow = {int(x.split("\t")[1]): x for x in piles[1:] if x}
    assert set(prow) == set(srows)
    k = next(iter(sorted(prow)))
    i = int(k - 1 - int(pos[0]))
    assert prow[k] == "chr1\t%d\t%s" % (k, chr(f["ref"][i])) + "".join("\t%d" % v for v in r[i].ravel())
    # ---- the dispatcher's statistics do not depend on the callers: the reference's .dist, column for column
    got = [x.split("\t") for x in open(run / "out.dist").read().split("\n")]
    exp = [x.split("\t") for x in open(os.path.join(fx.GOLD, "pecall_sites.dist.txt")).read().split("\n")]
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g[0] == e[0] and len(g) == len(e)
        if len(g) > 1:
            assert dict(zip(got[0][1:], g[1:])) == dict(zip(exp[0][1:], e[1:])), g[0]
    # ---- same column order as the reference's run: its text, row for row
    if cols == ref_order:
        for pos1, row in f["base_rows"].items():
            assert rows[pos1] == row
        for pos1, row in f["snp_rows"].items():
            assert srows[pos1] == row


def test_pecaller_cli_100_samples(tmp_path):
    """more than 64 pileup files in the directory (the reference takes any number: pecaller.c:251-257, 495-515): pecaller_hip's rows
    against the oracle in the directory's order and, where the order is the fixture's, the reference's own text"""
    assert os.path.exists(EXE), "build with make -C pecaller_amd/csrc"
    z = np.load(os.path.join(fx.GOLD, "pecall_wide.npz"))
    names = [str(x) for x in z["names"]]
    reads, pos, pad = z["reads"], z["pos"], int(z["pad"][0])
    _, seqs = refio.read_fasta(os.path.join(fx.GOLD, "g1.fa.gz"))
    shutil.copy(os.path.join(fx.GOLD, "g1.sdx"), tmp_path / "g1.sdx")
    with gzip.open(tmp_path / "g1.seq", "wb", compresslevel=1) as f:
        f.write(np.concatenate(seqs).tobytes())
    run = tmp_path / "run"
    run.mkdir()
    for s, nm in enumerate(names):
        recs = [struct.pack("<I6H", int(pos[i]), *[int(x) for x in reads[i, s]]) for i in range(len(pos)) if reads[i, s].sum() > 0]
        recs += [struct.pack("<I6H", int(pos[-1]) + 1 + k, 20, 0, 0, 0, 0, 0) for k in range(pad)]
        with gzip.open(run / ("%s.pileup.gz" % nm), "wb", compresslevel=1) as f:
            f.write(b"".join(recs))
    subprocess.check_call([EXE, "pileup", str(tmp_path / "g1.sdx"), "105", "out", "0.95", "0.001", "n", "8", "n"], cwd=run, stdout=subprocess.DEVNULL)
    base = gzip.open(run / "out.base.gz", "rt").read().split("\n")
    cols = [c for c in base[0].split("\t")[3:] if c]
    assert sorted(cols) == sorted(names) and len(cols) == 100
    perm = [names.index(c) for c in cols]
    f = fx.load("pecall_wide")
    r = reads[:, perm, :]
    call, p, typ, ac, _ = oracle_py.call_sites(r, f["dom"])
    rows = {int(x.split("\t")[1]): x for x in base[1:] if x}
    snp = open(run / "out.snp").read().split("\n")
    srows = {int(x.split("\t")[1]): x for x in snp[1:] if x}
    n_base = n_snp = 0
    for i, q in enumerate(pos):
        pos1 = int(q) + 1
        if f["dom"][i] > 3 or r[i].sum() == 0:
            assert pos1 not in rows
            continue
        n_base += 1
        assert rows[pos1] == fx.base_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i]), pos1
        if typ[i] > 0:
            n_snp += 1
            assert srows[pos1] == fx.snp_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i], typ[i], ac[i]), pos1
    assert n_base == len(f["base_rows"]) and n_snp > 30
    if cols == [str(x) for x in z["columns"]]:
        for pos1, row in f["base_rows"].items():
            assert rows[pos1] == row
        for pos1, row in f["snp_rows"].items():
            assert srows[pos1] == row


def test_pecaller_cli_with_pedigree(tmp_path):
    """use_pedfile = y: the ped file parsed by the host program, DENOVO_ rows"""
    z = np.load(os.path.join(fx.GOLD, "pecall_ped.npz"))
    names = [str(x) for x in z["names"]]
    reads, pos = z["reads"], z["pos"]
    _, seqs = refio.read_fasta(os.path.join(fx.GOLD, "g1.fa.gz"))
    shutil.copy(os.path.join(fx.GOLD, "g1.sdx"), tmp_path / "g1.sdx")
    with gzip.open(tmp_path / "g1.seq", "wb", compresslevel=1) as f:
        f.write(np.concatenate(seqs).tobytes())
    run = tmp_path / "run"
    run.mkdir()
    for s, nm in enumerate(names):
        recs = [struct.pack("<I6H", int(pos[i]), *[int(x) for x in reads[i, s]]) for i in range(len(pos)) if reads[i, s].sum() > 0]
        with gzip.open(run / ("%s.pileup.gz" % nm), "wb", compresslevel=1) as f:
            f.write(b"".join(recs))
    dad, mom, sex = z["dad"], z["mom"], z["sex"]
    with open(tmp_path / "ped.txt", "w") as f:
        for i, nm in enumerate(names):
            f.write("fam\t%s\t%s\t%s\t%d\n" % (nm, names[dad[i]] if dad[i] >= 0 else "0", names[mom[i]] if mom[i] >= 0 else "0", sex[i]))
    subprocess.check_call([EXE, "pileup", str(tmp_path / "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "y", str(tmp_path / "ped.txt"), "1e-06"],
                          cwd=run, stdout=subprocess.DEVNULL)
    base = gzip.open(run / "out.base.gz", "rt").read().split("\n")
    cols = [c for c in base[0].split("\t")[3:] if c]
    perm = [names.index(c) for c in cols]
    inv = {old: new for new, old in enumerate(perm)}
    ped = dict(dad=np.array([inv[int(dad[o])] if dad[o] >= 0 else -1 for o in perm]), mom=np.array([inv[int(mom[o])] if mom[o] >= 0 else -1 for o in perm]),
               sex=np.array([int(sex[o]) for o in perm]), order=[inv[o] for o in range(len(names))], denovo_rate=1e-6)
    f = fx.load("pecall_ped")
    r = reads[:, perm, :]
    call, p, typ, ac, _ = oracle_py.call_sites(r, f["dom"], ped=ped)
    den = oracle_py.call_sites.denovo
    rows = {int(x.split("\t")[1]): x for x in base[1:] if x}
    srows = {int(x.split("\t")[1]): x for x in open(run / "out.snp").read().split("\n")[1:] if x}
    n_den = 0
    for i, q in enumerate(pos):
        pos1 = int(q) + 1
        if f["dom"][i] > 3 or r[i].sum() == 0:
            continue
        assert rows[pos1] == fx.base_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i]), pos1
        if typ[i] > 0:
            assert srows[pos1] == fx.snp_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i], typ[i], ac[i], den[i]), pos1
            n_den += den[i] > 0
    assert n_den >= 10
    if cols == [str(x) for x in z["columns"]]:
        for pos1, row in f["snp_rows"].items():
            assert srows[pos1] == row


@pytest.mark.parametrize("range_min", [None, "64"])
def test_pecaller_cli_guide_mode(tmp_path, range_min, monkeypatch):
    """a BED guide file: every position of the intervals is called (uncovered ones too), chrY / chrMT columns haploid; the
    device caller itself on the same columns first (the + 16 chromosome flag).  range_min = 64: stretches of 64 positions and more
    of an interval go through the parallel walk of the streams (by default from 4,096 on), the rest through the per-column scan --
    same rows, same <out>.dist"""
    from pecaller_amd.pecall import PecallDev
    if range_min:
        monkeypatch.setenv("PECALLER_GUIDE_RANGE_MIN", range_min)
        monkeypatch.setenv("PECALLER_TILE_LOG2", "10")        # (and tiles of 1,024 columns: stretches cut by a full tile)
    f = fx.load_guide()
    dev = PecallDev(0)
    got = dev.call_sites(f["reads"], f["dom"], chrom=f["chrom"])
    exp = oracle_py.call_sites(f["reads"], f["dom"], chrom=f["chrom"])
    assert np.array_equal(got[0], exp[0]) and np.max(np.abs(got[1] - exp[1])) <= 1e-6 and np.array_equal(got[2], exp[2])
    dev.close()
    z = f["z"]
    names = f["names"]
    _, seqs = refio.read_fasta(os.path.join(fx.GOLD, "g1.fa.gz"))
    shutil.copy(os.path.join(fx.GOLD, "pecall_guide.sdx"), tmp_path / "g1.sdx")
    with gzip.open(tmp_path / "g1.seq", "wb", compresslevel=1) as fh:
        fh.write(np.concatenate(seqs).tobytes())
    run = tmp_path / "run"
    run.mkdir()
    tail = int(z["tail"][0])
    for s, nm in enumerate(names):
        recs = [struct.pack("<I6H", int(z["pos"][i]), *[int(x) for x in z["reads"][i, s]]) for i in range(len(z["pos"])) if z["reads"][i, s].sum() > 0]
        recs += [struct.pack("<I6H", tail + k, 20, 0, 0, 0, 0, 0) for k in range(40)]
        with gzip.open(run / ("%s.pileup.gz" % nm), "wb", compresslevel=1) as fh:
            fh.write(b"".join(recs))
    subprocess.check_call([EXE, "pileup", str(tmp_path / "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "n", os.path.join(fx.GOLD, "pecall_guide.bed")],
                          cwd=run, stdout=subprocess.DEVNULL)
    base = gzip.open(run / "out.base.gz", "rt").read().split("\n")
    cols = [c for c in base[0].split("\t")[3:] if c]
    rows = {(x.split("\t")[0], int(x.split("\t")[1])): x for x in base[1:] if x}
    srows = {(x.split("\t")[0], int(x.split("\t")[1])): x for x in open(run / "out.snp").read().split("\n")[1:] if x}
    assert len(rows) == len(f["base_rows"])
    perm = [f["columns"].index(c) for c in cols]           # this directory's order relative to the fixture's
    call, p, typ, ac, _ = oracle_py.call_sites(f["reads"][:, perm, :], f["dom"], chrom=f["chrom"])
    for i, (c, p1, refch) in enumerate(f["key"]):
        if f["dom"][i] > 3:
            continue
        assert rows[(c, p1)] == fx.base_row(c, p1, refch, call[i], p[i]), (c, p1)
        if typ[i] > 0:
            assert srows[(c, p1)] == fx.snp_row(c, p1, refch, call[i], p[i], typ[i], ac[i]), (c, p1)
    got_d = [x.split("\t") for x in open(run / "out.dist").read().split("\n")]
    exp_d = [x.split("\t") for x in open(os.path.join(fx.GOLD, "pecall_guide.dist.txt")).read().split("\n")]
    assert len(got_d) == len(exp_d)
    for g, e in zip(got_d, exp_d):
        assert g[0] == e[0]
        if len(g) > 1:
            assert dict(zip(got_d[0][1:], g[1:])) == dict(zip(exp_d[0][1:], e[1:])), g[0]
    if cols == f["columns"]:
        assert rows == f["base_rows"] and srows == f["snp_rows"]


@pytest.mark.parametrize("range_min", [None, "64"])
def test_pecaller_cli_guide_mode_ends_with_the_last_stream(tmp_path, range_min, monkeypatch):
    """the BED goes on behind the last record of every pileup stream: the reference's loop runs while a stream is open
    (pecaller.c:952), so the column at which the last stream ends is the last row and <out>.dist counts the positions up to it --
    through the per-column scan and through the parallel walk of a stretch (range_min = 64), which has to cut the stretch there.
    Fixture: tests/golden/make_golden_pecall_guide_end.py (the unmodified reference on the guide fixture's records up to chrMT:300)"""
    import json
    if range_min:
        monkeypatch.setenv("PECALLER_GUIDE_RANGE_MIN", range_min)
        monkeypatch.setenv("PECALLER_TILE_LOG2", "10")
    cut = json.load(open(os.path.join(fx.GOLD, "pecall_guide_end.json")))["cut"]
    z = np.load(os.path.join(fx.GOLD, "pecall_guide.npz"))
    lens, cn, _ = refio.read_sdx(os.path.join(fx.GOLD, "pecall_guide.sdx"))
    starts = np.concatenate([[0], np.cumsum(np.array(lens) + 15)])
    limit = int(starts[cn.index("chrMT")]) + cut - 1
    _, seqs = refio.read_fasta(os.path.join(fx.GOLD, "g1.fa.gz"))
    shutil.copy(os.path.join(fx.GOLD, "pecall_guide.sdx"), tmp_path / "g1.sdx")
    with gzip.open(tmp_path / "g1.seq", "wb", compresslevel=1) as fh:
        fh.write(np.concatenate(seqs).tobytes())
    run = tmp_path / "run"
    run.mkdir()
    for s, nm in enumerate([str(x) for x in z["names"]]):
        recs = [struct.pack("<I6H", int(z["pos"][i]), *[int(x) for x in z["reads"][i, s]]) for i in range(len(z["pos"]))
                if z["reads"][i, s].sum() > 0 and int(z["pos"][i]) <= limit]
        with gzip.open(run / ("%s.pileup.gz" % nm), "wb", compresslevel=1) as fh:
            fh.write(b"".join(recs))
    subprocess.check_call([EXE, "pileup", str(tmp_path / "g1.sdx"), "20", "out", "0.95", "0.001", "n", "2", "n", os.path.join(fx.GOLD, "pecall_guide.bed")],
                          cwd=run, stdout=subprocess.DEVNULL)
    base = gzip.open(run / "out.base.gz", "rt").read().split("\n")
    exp = gzip.open(os.path.join(fx.GOLD, "pecall_guide_end.base.txt.gz"), "rt").read().split("\n")
    cols = [c for c in base[0].split("\t")[3:] if c]
    ecols = [c for c in exp[0].split("\t")[3:] if c]
    got_keys = sorted((x.split("\t")[0], int(x.split("\t")[1])) for x in base[1:] if x)
    exp_keys = sorted((x.split("\t")[0], int(x.split("\t")[1])) for x in exp[1:] if x)
    assert got_keys == exp_keys                                 # 902 rows, the last one chrMT:300
    assert max(p for c, p in got_keys if c == "chrMT") == cut
    if cols == ecols:
        assert sorted(x for x in base[1:] if x) == [x for x in exp[1:] if x]
    got_d = [x.split("\t") for x in open(run / "out.dist").read().split("\n")]
    exp_d = [x.split("\t") for x in open(os.path.join(fx.GOLD, "pecall_guide_end.dist.txt")).read().split("\n")]
    assert len(got_d) == len(exp_d)
    for g, e in zip(got_d, exp_d):
        assert g[0] == e[0]
        if len(g) > 1:
            assert dict(zip(got_d[0][1:], g[1:])) == dict(zip(exp_d[0][1:], e[1:])), g[0]


@pytest.mark.parametrize("serial_from_start", [False, True])
def test_pecaller_cli_takes_streams_that_are_not_ascending(tmp_path, serial_from_start, monkeypatch):
    """two records swapped, one moved ten columns back and one position twice in two of the eight pileup streams: the reference's
    dispatcher takes the lowest pending position whatever the order (pecaller.c:865-923), so the late and the repeated records become
    columns of their own.  pecaller_hip's parallel walk of the streams notices the first record out of order, says so, and starts
    over with the reference's serial merge (or is told to use it from the start): the unmodified reference's rows and .dist
    (tests/golden/make_golden_pecall_unordered.py)"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("mk_unordered", os.path.join(fx.GOLD, "make_golden_pecall_unordered.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    if serial_from_start:
        monkeypatch.setenv("PECALLER_SERIAL_MERGE", "1")
    monkeypatch.setenv("PECALLER_TILE_LOG2", "10")
    z = np.load(os.path.join(fx.GOLD, "pecall_sites.npz"))
    names = [str(x) for x in z["names"]]
    dist_spec = json.load(open(os.path.join(fx.GOLD, "pecall_unordered.json")))
    _, seqs = refio.read_fasta(os.path.join(fx.GOLD, "g1.fa.gz"))
    shutil.copy(os.path.join(fx.GOLD, "g1.sdx"), tmp_path / "g1.sdx")
    with gzip.open(tmp_path / "g1.seq", "wb", compresslevel=1) as f:
        f.write(np.concatenate(seqs).tobytes())
    run = tmp_path / "run"
    run.mkdir()
    for s, nm in enumerate(names):
        with gzip.open(run / ("%s.pileup.gz" % nm), "wb", compresslevel=1) as f:
            f.write(b"".join(mk.stream_records(z, s, dist_spec)))
    out = subprocess.run([EXE, "pileup", str(tmp_path / "g1.sdx"), "20", "out", "0.95", "0.001", "n", "8", "n"], cwd=run, stdout=subprocess.PIPE, check=True).stdout
    assert (b"starting over with the serial merge" in out) == (not serial_from_start)
    last = int(z["pos"][-1]) + 1
    keep = lambda rows: sorted(x for x in rows if x and int(x.split("\t")[1]) <= last)
    base = gzip.open(run / "out.base.gz", "rt").read().split("\n")
    exp = gzip.open(os.path.join(fx.GOLD, "pecall_unordered.base.txt.gz"), "rt").read().split("\n")
    got_rows, exp_rows = keep(base[1:]), [x for x in exp[1:] if x]
    assert [x.split("\t")[:2] for x in got_rows] == [x.split("\t")[:2] for x in exp_rows]       # 5,910 rows, three positions twice
    cols = [c for c in base[0].split("\t")[3:] if c]
    if cols == [c for c in exp[0].split("\t")[3:] if c]:
        assert got_rows == exp_rows
        snp = open(run / "out.snp").read().split("\n")
        assert keep(snp[1:]) == [x for x in open(os.path.join(fx.GOLD, "pecall_unordered.snp.txt")).read().split("\n")[1:] if x]
    got_d = [x.split("\t") for x in open(run / "out.dist").read().split("\n")]
    exp_d = [x.split("\t") for x in open(os.path.join(fx.GOLD, "pecall_unordered.dist.txt")).read().split("\n")]
    assert len(got_d) == len(exp_d)
    for g, e in zip(got_d, exp_d):
        assert g[0] == e[0]
        if len(g) > 1:
            assert dict(zip(got_d[0][1:], g[1:])) == dict(zip(exp_d[0][1:], e[1:])), g[0]
