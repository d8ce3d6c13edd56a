"""Butterfly search (--butterfly-search: segment_juncs.cpp:4178-4249, :1698-2049), CPU: the oracle's restatement
against what the search is for, and the CPU build of the kernel logic (thj_cov_core.h bf_*, through tests/hostsim) against the oracle.
No reference vector exists for this path (no test of the reference reaches it): parity unpinned, as for the coverage search."""
import pytest

import orc
from cov_util import CASES, butterfly_case, load


def _tuples(a):
    return {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in a}


@pytest.mark.parametrize("name", CASES)
def test_oracle_finds_junctions_of_the_fixture(name):
    """on the coverage-search fixtures (reads spliced over real GT-AG introns, the reads themselves as the unmapped set) the search proposes
    junctions, and nearly all of them are junctions the reference's own run recorded for the case by its other searches"""
    c = load(name)
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    g = orc.Genome(seqs)
    got = _tuples(orc.butterfly_search(g, c["hits"], c["ium"], c["cov"]["min_intron"], c["cov"]["max_intron"]))
    ids = {n: i + 1 for i, n in enumerate(c["names"])}
    known = {(ids[t[0]], int(t[1]), int(t[2]), 1 if t[3][0] == "-" else 0) for t in (l.split("\t") for l in c["expected"].splitlines())}
    assert len(got) >= 10
    assert len(got & known) >= len(got) - 1
    for (_, left, right, _) in got:
        assert c["cov"]["min_intron"] < right - left < c["cov"]["max_intron"]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("cap", [5000000, 7, 1])
def test_kernel_logic_reproduces_oracle_on_fixture(name, cap):
    import sim
    c = load(name)
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    g = orc.Genome(seqs)
    args = (c["cov"]["min_intron"], c["cov"]["max_intron"], cap)
    want = _tuples(orc.butterfly_search(g, c["hits"], c["ium"], *args))
    assert len(want) == min(cap, len(want)) and len(want) > 0
    assert sim.butterfly_search(seqs, c["hits"], c["ium"], *args) == want


@pytest.mark.parametrize("seed", range(500, 560))
def test_kernel_logic_matches_oracle_on_seeded_cases(seed):
    """planted forward and reverse-strand introns, reads across them in both orientations, islands at contig starts, dropped windows at
    contig ends, windows that abut, sites without a mer, N runs, repeated exon ends; with and without the cap"""
    import sim
    seqs, h, ium, args = butterfly_case(seed)
    folded = [orc.fold_genome_char(s) for s in seqs]
    g = orc.Genome(folded)
    for cap in (5000000, 2):
        assert sim.butterfly_search(folded, h, ium, *args, cap) == _tuples(orc.butterfly_search(g, h, ium, *args, cap)), cap


def test_seeded_cases_find_both_strands():
    n, anti = 0, 0
    for seed in range(500, 530):
        seqs, h, ium, args = butterfly_case(seed)
        g = orc.Genome([orc.fold_genome_char(s) for s in seqs])
        j = _tuples(orc.butterfly_search(g, h, ium, *args))
        n += len(j)
        anti += sum(1 for x in j if x[3])
    assert n > 30 and 0 < anti < n
