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
    import numpy as np
    import sim
    from tophat_amd.batch import HIT_DTYPE
    rng = np.random.default_rng(seed)
    n_contigs = int(rng.integers(1, 4))
    lens = [int(x) for x in rng.integers(200, 6000, size=n_contigs)]
    seqs = ["".join(rng.choice(list("ACGT"), size=n)) for n in lens]
    if seed % 3 == 0:                        # N runs
        s = list(seqs[0]); a = int(rng.integers(0, max(1, lens[0] - 60))); s[a:a + 50] = "N" * len(s[a:a + 50]); seqs[0] = "".join(s)
    min_cov = int(rng.choice([20, 18, 10, 8]))
    hits = []
    for k, n in enumerate(lens):             # islands: runs of hits of assorted lengths, some abutting, some at the contig ends
        for _ in range(int(rng.integers(2, 12))):
            ln = int(rng.choice([min_cov - 2, min_cov - 1, min_cov, min_cov + 1, 25, 40, 75]))
            left = int(rng.choice([0, 1, 2, max(0, n - ln), max(0, n - ln - 1), int(rng.integers(0, max(1, n - ln)))]))
            right = min(n, left + ln)
            if right > left:
                hits.append((k + 1, left, right, 0, 0, 0, min(255, right - left)))
    # unmapped reads: spliced reads across random pairs of positions (so that some donor/acceptor pairs are extendable)
    ium = []
    for _ in range(int(rng.integers(20, 200))):
        k = int(rng.integers(0, n_contigs)); s = seqs[k]; n = lens[k]
        a = int(rng.integers(0, max(1, n - 40))); b = int(rng.integers(a, n))
        la = int(rng.integers(5, 28))
        r = (s[max(0, a - la):a] + s[b:b + 40])[:int(rng.choice([9, 12, 25, 32, 50]))]
        if r:
            ium.append(r.replace("N", "A") if seed % 2 else r)
    h = np.array(hits, dtype=HIT_DTYPE)
    folded = [orc.fold_genome_char(s) for s in seqs]
    g = orc.Genome(folded)
    args = (min_cov, int(rng.choice([1, 20, 50])), int(rng.choice([300, 2000, 20000])))
    assert sim.coverage_search(folded, h, ium, *args) == _tuples(orc.coverage_search(g, h, ium, *args))
