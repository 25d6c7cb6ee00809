"""The two text tools behind the caller (pecaller_amd/csrc/merge_indel_snp_main.c, snp_to_vcf_main.c) against what the
reference's merge_indel_snp.pl and snp_to_vcf print for the same files (tests/golden/downstream/, made by
tests/golden/make_golden_downstream.py from the unmodified reference tools).  Byte for byte; CPU only."""
import gzip
import os
import shutil
import subprocess
import pytest
import refio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DOWN = os.path.join(GOLD, "downstream")


@pytest.fixture(scope="module")
def tools(tmp_path_factory):
    d = tmp_path_factory.mktemp("downstream")
    exe = {}
    for name in ("merge_indel_snp", "snp_to_vcf"):
        exe[name] = str(d / name)
        subprocess.check_call(["gcc", "-O2", "-Wall", "-Wno-unused-result", "-o", exe[name],
                               os.path.join(ROOT, "pecaller_amd", "csrc", name + "_main.c"), "-lz"])
    # the genome as index_genome_whole lays it out: 15 filler bytes after every contig
    shutil.copy(os.path.join(GOLD, "g1.sdx"), str(d / "g1.sdx"))
    names, contigs = refio.read_fasta(os.path.join(GOLD, "g1.fa.gz"))
    with gzip.open(str(d / "g1.seq"), "wb") as f:
        for c in contigs:
            f.write(c.tobytes() + b"N" * 15)
    return exe, d


@pytest.mark.parametrize("case", ["sites", "ped", "multi"])
def test_merge_indel_snp_writes_the_reference_scripts_file(tools, case):
    exe, d = tools
    out = str(d / (case + ".merged"))
    r = subprocess.run([exe["merge_indel_snp"], str(d / "g1.sdx"), os.path.join(DOWN, case + ".snp.txt"),
                        os.path.join(DOWN, case + "_indel"), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(os.path.join(DOWN, case + ".merged.txt"), "rb").read()
    # the one position no sample's file holds is reported, as the script reports it
    assert r.stdout.count(b"This is impossible.  We fail to find insertion") == 1


@pytest.mark.parametrize("case", ["sites", "ped", "multi"])
def test_snp_to_vcf_prints_the_reference_programs_text(tools, case):
    exe, d = tools
    r = subprocess.run([exe["snp_to_vcf"], "g1.sdx", os.path.join(DOWN, case + ".merged.txt"), "0.9"], cwd=str(d),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    got = b"\n".join(l for l in r.stdout.split(b"\n") if not l.startswith(b"##fileDate="))
    assert got == open(os.path.join(DOWN, case + ".vcf.txt"), "rb").read()
    assert r.stdout.split(b"\n")[1].startswith(b"##fileDate=20")


def test_merge_indel_snp_argument_and_file_errors(tools, tmp_path):
    exe, d = tools
    assert subprocess.run([exe["merge_indel_snp"], "a", "b"], stdout=subprocess.PIPE).returncode == 1
    r = subprocess.run([exe["merge_indel_snp"], str(d / "g1.sdx"), os.path.join(DOWN, "sites.snp.txt"), str(tmp_path), str(tmp_path / "o")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"Can not open" in r.stderr
    # a deletion on a contig the .sdx does not name ends snp_to_vcf, as in the reference (snp_to_vcf.c:481-485)
    bad = tmp_path / "bad.snp"
    bad.write_text("Fragment\tPosition\tReference\tAlleles\tAllele_Counts\tType\ts0\t\nchrZ\t100\tA\tA,-2\t3,3\tDEL\tE\t1\n")
    r = subprocess.run([exe["snp_to_vcf"], "g1.sdx", str(bad), "0.9"], cwd=str(d), stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"Failed to find chrom = chrZ" in r.stdout
