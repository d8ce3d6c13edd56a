"""Known-answer tests from the reference's own regression cases -- all nine of them (tests/golden_ref/, see
tests/ref_regression.py): the CPU oracle, and the CPU build of the kernel logic, against what the reference's authors
recorded.  What each case pins is listed in oracle/README.md."""
import copy

import pytest

import orc
import ref_regression as rr
import sim
from tophat_amd.batch import merge_events

# (records checked, of them with a gap) per side: every one of them must come out of long_spanning_reads
RECORDED = {"test_SimpleSplicing": {"left": (986, 64)}, "test_SimpleIndel": {"left": (991, 117)}, "test_IndelWithErrors": {"left": (1921, 227)},
            "test_Paired": {"left": (970, 104), "right": (974, 116)}, "test_3Segment": {"left": (950, 84), "right": (958, 100)},
            "test_ReverseComplementSplicing": {"left": (982, 60)}, "test_ReverseComplementIndel": {"left": (984, 110)},
            "test_IndelLowerCase": {"left": (991, 117)}, "test_Indel_1": {"left": (2, 2)}}


def _orc_juncs_db(genome):
    og = orc.Genome([orc.fold_genome_char(genome)])
    return lambda names, jf, inf, df, read_len, min_anchor: orc.juncs_db_text(names, og, jf, inf, df, None, read_len, min_anchor)


def _load(case, tmp_path):
    import os
    genome = "".join(l.strip() for l in open(os.path.join(rr.GOLD, case, "genome.fa")) if not l.startswith(">")).upper()
    return rr.load(case, tmp_path, _orc_juncs_db(genome))


def side_params(c):
    out = []
    for sd, side in (("left", 1), ("right", 2)):
        if sd in c["sides"]:
            p = copy.copy(c["p"])
            p.read_side = side
            out.append((sd, p))
    return out


@pytest.mark.parametrize("case", rr.CASES)
def test_segment_juncs_on_the_recorded_cases(case, tmp_path):
    """the recorded junction (where the case has one) is among the potential junctions segment_juncs reports, on the oracle
    and on the CPU build of the kernel logic, which agree event for event (junctions, deletions, insertions)"""
    c = _load(case, tmp_path)
    seq = orc.fold_genome_char(c["genome"])
    og = orc.Genome([seq])
    ev = ev2 = None
    for sd, p in side_params(c):
        b = c["sides"][sd]["seg_batch"]
        e, e2 = orc.segjuncs(p, og, b), sim.segjuncs(p, [seq], b)
        ev = e if ev is None else merge_events(ev, e)
        ev2 = e2 if ev2 is None else merge_events(ev2, e2)
    got = sorted((int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ev.juncs)
    if case in rr.SPLICE_CASES:
        assert c["recorded_juncs"] == [rr.SPLICE_CASES[case]] and rr.SPLICE_CASES[case] in got
    if case in ("test_SimpleSplicing", "test_ReverseComplementSplicing"):
        assert got == c["recorded_juncs"]           # nothing else is even proposed
    if case == "test_3Segment":                     # three 8-base segments: the v2.1.2 indel search runs (segment_juncs.cpp:2856)
        assert len(ev.deletions) > 0 and len(ev.insertions) > 0
    assert [tuple(j) for j in ev2.juncs] == [tuple(j) for j in ev.juncs]
    assert [tuple(j) for j in ev2.deletions] == [tuple(j) for j in ev.deletions] and ev2.insertions == ev.insertions


@pytest.mark.parametrize("case", rr.CASES)
def test_long_spanning_reads_reproduces_the_recorded_alignments(case, tmp_path):
    c = _load(case, tmp_path)
    seq = orc.fold_genome_char(c["genome"])
    og = orc.Genome([seq])
    for sd, _p in side_params(c):
        sb = c["sides"][sd]["span_batch"]
        alns = orc.spanning(c["p"], og, sb, c["span_juncs"], c["span_ins"])
        assert rr.check_recorded_alignments(c, alns, sd) == RECORDED[case][sd]
        # the CPU build of the kernel logic: record for record what the oracle says, in every tier arrangement
        for mode in (0, 1, 2):
            got, status = sim.spanning(c["p"], [seq], sb, c["span_juncs"], c["span_ins"], mode)
            got.sort(key=lambda a: a.read_idx)
            assert status[1] == 0 and status[2] == 0
            assert got == alns, "%s mode %d" % (sd, mode)
