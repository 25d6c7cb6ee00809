"""PECaller per-site caller: the oracle (oracle/pecall_site_oracle.c) against the text the reference itself printed"""
import numpy as np
import pytest
import oracle_py
import pecall_sites_fixture as fx


def test_site_oracle_matches_reference_text():
    f = fx.load()
    call, p, typ, ac, npass = oracle_py.call_sites(f["reads"], f["dom"])
    n_base = n_snp = 0
    bad = []
    for i, pos in enumerate(f["pos"]):
        pos1 = int(pos) + 1                      # every site of the fixture lies in the first contig
        exp = f["base_rows"].get(pos1)
        if exp is None:
            continue                              # no sample had a read there, or a non-ACGT reference base
        got = fx.base_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i])
        n_base += 1
        if got != exp:
            bad.append((pos1, exp, got))
        exp_s = f["snp_rows"].get(pos1)
        if typ[i] > 0:
            n_snp += 1
            got_s = fx.snp_row("chr1", pos1, chr(f["ref"][i]), call[i], p[i], typ[i], ac[i])
            if got_s != exp_s:
                bad.append((pos1, exp_s, got_s))
        elif exp_s is not None:
            bad.append((pos1, exp_s, None))
    assert n_base == len(f["base_rows"]) and n_snp == len(f["snp_rows"]), (n_base, len(f["base_rows"]), n_snp, len(f["snp_rows"]))
    assert not bad, bad[:3]
    assert npass.max() >= 2                      # the fixture exercises the alpha re-estimation
