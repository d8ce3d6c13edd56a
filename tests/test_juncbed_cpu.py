"""CPU: the junction consensus of tophat_reports -- oracle/juncbed_oracle.c -- pinned by the
reference's own regression cases: the recorded accepted_hits.sam reduced to a JunctionSet and printed must be the recorded
junctions.bed, byte for byte, in all nine cases (four have a junction; five are the header line alone)."""
import os

import numpy as np
import pytest

import orc
import ref_regression as rr


@pytest.mark.parametrize("case", rr.CASES)
def test_recorded_alignments_reduce_to_the_recorded_junctions_bed(case):
    recs = rr.recorded_alignment_records(case)
    js = orc.junction_consensus(orc.jrecs_from_tuples(recs))
    assert orc.junctions_bed(js, ["fake"]) == open(os.path.join(rr.GOLD, case, "junctions.bed")).read()
    assert (len(js) == 1) == (case in rr.SPLICE_CASES)


def test_filters():
    """accept_if_valid / knockout_shadow_junctions / the final extent filter on hand-made records"""
    M, N = 1, 11

    def rec(left, a, gap, b, anti=False, ref=1):
        return (ref, left, anti, [(M, a), (N, gap), (M, b)])
    # anchors: a junction seen only with a 7-base right anchor is not accepted (min_anchor_len 8) ...
    js = orc.junction_consensus(orc.jrecs_from_tuples([rec(100, 20, 500, 7)]))
    assert len(js) == 0
    # ... but one read with a long anchor makes it valid, and the short-anchored read then counts as support
    js = orc.junction_consensus(orc.jrecs_from_tuples([rec(100, 20, 500, 7), rec(90, 30, 500, 12)]))
    assert [(int(j["left"]), int(j["right"]), int(j["support"]), int(j["left_extent"]), int(j["right_extent"])) for j in js] == [(119, 620, 2, 30, 12)]
    # introns over 50 kb need two supporting reads and anchors over 12
    long1 = [rec(100, 20, 60000, 20)]
    assert len(orc.junction_consensus(orc.jrecs_from_tuples(long1))) == 0
    assert len(orc.junction_consensus(orc.jrecs_from_tuples(long1 * 2))) == 1
    assert len(orc.junction_consensus(orc.jrecs_from_tuples([rec(100, 20, 60000, 12)] * 2))) == 0
    # shadow: the same intron (within the anchor length) on the other strand with more support knocks the weaker one out,
    # and the alignments on it with it
    strong = [rec(100, 20, 500, 20, anti=False)] * 3
    weak = [rec(103, 20, 497, 20, anti=True)]
    js = orc.junction_consensus(orc.jrecs_from_tuples(strong + weak))
    assert [(int(j["left"]), int(j["antisense"]), int(j["support"])) for j in js] == [(119, 0, 3)]
    # two junctions in one alignment: both counted; a record with one filtered junction contributes nothing
    two = (1, 100, False, [(M, 20), (N, 300), (M, 30), (N, 400), (M, 25)])
    js = orc.junction_consensus(orc.jrecs_from_tuples([two]))
    assert [(int(j["left"]), int(j["right"])) for j in js] == [(119, 420), (449, 850)]
    bad = (1, 100, False, [(M, 20), (N, 300), (M, 30), (N, 400), (M, 5)])
    assert len(orc.junction_consensus(orc.jrecs_from_tuples([bad]))) == 0
    # deletions and insertions move the coordinate the way the cigar walk says (junctions.cpp:78-92)
    d = (2, 1000, True, [(M, 10), (5, 3), (M, 10), (N, 100), (M, 15), (3, 2), (M, 9)])
    js = orc.junction_consensus(orc.jrecs_from_tuples([d]))
    assert [(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"]), int(j["left_extent"]), int(j["right_extent"])) for j in js] == \
        [(2, 1022, 1123, 1, 10, 15)]
    assert orc.junctions_bed(js, ["a", "b"]).split("\n")[1] == "b\t1013\t1138\tJUNC00000001\t1\t-\t1013\t1138\t255,0,0\t2\t10,15\t0,110"


def test_fusion_alignment_walk_by_hand():
    """junctions_from_spliced_hit (junctions.cpp:19-92) on fusion alignments, values worked out by hand: a junction behind an FF / FR / RF
    fusion lies on the second contig at the position the F op names; pieces that run down the genome (m, n) give right = j + 1,
    left = j - length with the extents swapped; an RR fusion neither jumps nor switches contigs (it has no case in the reference)"""
    M, m, N, n, FR, RR = 1, 2, 11, 12, 8, 10
    recs = [(1, 1000, False, [(M, 30), (N, 200), (M, 20), (FR, 9000), (m, 25), (n, 300), (m, 22)], 2)] * 2
    got = orc.junction_consensus(orc.jrecs_from_tuples(recs))
    assert [tuple(int(x) for x in r)[:7] for r in got.tolist()] == [(1, 1029, 1230, 0, 30, 20, 2), (2, 8675, 8976, 0, 22, 25, 2)]
    recs = [(2, 4000, False, [(m, 20), (n, 150), (m, 30), (RR, 700), (m, 30), (n, 90), (m, 20)], 1)] * 2
    got = orc.junction_consensus(orc.jrecs_from_tuples(recs))
    # j: 4000 -> 3980 -n150-> (3830, 3981) -> 3830 -> 3800 [RR: nothing] -> 3770 -n90-> (3680, 3771); both on the FIRST contig
    assert [tuple(int(x) for x in r)[:7] for r in got.tolist()] == [(2, 3680, 3771, 0, 20, 30, 2), (2, 3830, 3981, 0, 30, 20, 2)]
