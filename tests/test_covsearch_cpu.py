"""Coverage search, CPU: the oracle's restatement (oracle/covsearch_oracle.c) against the
fixtures under tests/golden_cov/ (outputs of the survey-stage scratch build of the reference, see oracle/README.md)."""
import copy

import pytest

import orc
from cov_util import CASES, juncs_text, load
from tophat_amd.batch import merge_events


def _tuples(a):
    return {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in a}


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_coverage_search_fixture(name):
    c = load(name)
    g = orc.Genome([None if s is None else orc.fold_genome_char(s) for s in c["seqs"]])
    ev = None
    for side, b in c["seg_batches"]:
        p = copy.copy(c["p"])
        p.read_side = side
        e = orc.segjuncs(p, g, b)
        ev = e if ev is None else merge_events(ev, e)
    seg = _tuples(ev.juncs)
    assert juncs_text(seg, c["names"]) == c["expected_seg_only"]
    cov = _tuples(orc.coverage_search(g, c["hits"], c["ium"], c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"]))
    assert len(cov - seg) >= 3, "the case must have junctions only the coverage search finds"
    assert juncs_text(seg | cov, c["names"]) == c["expected"]


def test_cap_keeps_the_lowest_skip_counts():
    """max_cov_juncs (segment_juncs.cpp:56, :1611-1621): the set ordered by skip count keeps its smallest elements"""
    c = load(CASES[0])
    g = orc.Genome([orc.fold_genome_char(s) for s in c["seqs"]])
    full = _tuples(orc.coverage_search(g, c["hits"], c["ium"], c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"]))
    capped = _tuples(orc.coverage_search(g, c["hits"], c["ium"], c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"], max_juncs=5))
    assert len(capped) <= 5 and capped <= full and len(full) > 5


@pytest.mark.parametrize("name", CASES)
def test_kernel_logic_reproduces_coverage_search_fixture(name):
    """the CPU build of the coverage-search kernels (thj_cov_core.h through tests/hostsim) == the oracle"""
    import sim
    c = load(name)
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    g = orc.Genome(seqs)
    args = (c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"])
    assert sim.coverage_search(seqs, c["hits"], c["ium"], *args) == _tuples(orc.coverage_search(g, c["hits"], c["ium"], *args))


@pytest.mark.parametrize("seed", range(400, 430))
def test_kernel_logic_matches_oracle_on_seeded_cases(seed):
    """islands at contig starts / ends, coverage runs around the length threshold, several contigs, short reads in the
    unmapped set, N runs"""
    import sim
    from cov_util import edge_case
    seqs, h, ium, args = edge_case(seed)
    folded = [orc.fold_genome_char(s) for s in seqs]
    g = orc.Genome(folded)
    assert sim.coverage_search(folded, h, ium, *args) == _tuples(orc.coverage_search(g, h, ium, *args))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_short_read_spanning_fixture(name):
    """long_spanning_reads on reads of one and two segments (the shapes the coverage search exists for): the oracle and
    the CPU build of the stitch tiers against the scratch build's records"""
    import numpy as np
    import sim
    from tophat_amd.batch import Events, JUNC_DTYPE, events_to_span_inputs
    from tophat_amd.params import Params
    c = load(name)
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    g = orc.Genome(seqs)
    ids = {n: i + 1 for i, n in enumerate(c["names"])}
    j = [(ids[t[0]], int(t[1]), int(t[2]), 1 if t[3][0] == "-" else 0) for t in (l.split("\t") for l in c["expected"].splitlines())]
    ev = Events(np.array(j, dtype=JUNC_DTYPE), np.zeros(0, dtype=JUNC_DTYPE), [], {})
    juncs, ins = events_to_span_inputs(ev)
    p = Params(segment_length=c["p"].segment_length)
    for sd, sb in c["span_batches"].items():
        alns = orc.spanning(p, g, sb, juncs, ins)
        assert [a.sam_fields(int(sb.read_id[a.read_idx]), c["names"]) for a in alns] == c["exp_span"][sd]
        for mode in (0, 1, 2):
            got, status = sim.spanning(p, seqs, sb, juncs, ins, mode)
            got.sort(key=lambda a: a.read_idx)
            assert got == alns, "mode %d" % mode


@pytest.mark.parametrize("cap", [1, 3, 7, 15])
def test_kernel_logic_cap_matches_oracle(cap):
    """max_cov_juncs: the kernel logic's skip counts and cut (smallest by (skip count, junction)) == the oracle's"""
    import sim
    for name in CASES:
        c = load(name)
        seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
        g = orc.Genome(seqs)
        args = (c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"])
        want = _tuples(orc.coverage_search(g, c["hits"], c["ium"], *args, max_juncs=cap))
        assert len(want) == cap
        assert sim.coverage_search(seqs, c["hits"], c["ium"], *args, max_juncs=cap) == want
