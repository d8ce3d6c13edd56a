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
        # The recorded deletions.bed / insertions.bed of the case are empty (none of the potential indels survives to the report), so
        # there is no reference value for them.  What is asserted: the values the oracle, the CPU build of the kernels (below) and the HIP
        # path (test_ref_regression_gpu.py) all produce, frozen here so that a change to find_insertions_and_deletions' restatement
        # (segment_juncs.cpp:2807-2942, :2470-2627) shows up as a failure and not as "still some indels".
        assert [tuple(int(x) for x in j)[:3] for j in ev.deletions] == [(1, 122, 124), (1, 383, 387), (1, 389, 393), (1, 416, 419)]
        assert ev.insertions == [(1, 66, "CG"), (1, 244, "GAA"), (1, 256, "TG"), (1, 383, "GTC"), (1, 385, "A"), (1, 401, "CCT"), (1, 439, "GAC"), (1, 441, "GTC")]
        # each is what a split alignment of some read explains: deleting [left + 1, right) from the genome / inserting the bases after
        # `left` makes a 16-base read piece match with at most the mismatches its two segment hits had (2 + 2)
        g = c["genome"]
        pieces = set()
        for sd in c["sides"]:
            for r in c["sides"][sd]["reads"].values():
                for q in (r, r.translate(rr._COMP)[::-1]):
                    pieces.update(q[k:k + 16] for k in range(0, len(q) - 15))

        def explained(edited, around):
            return any(sum(a != b for a, b in zip(edited[s:s + 16], pc)) <= 4 for pc in pieces for s in range(max(0, around - 15), around + 1) if s + 16 <= len(edited))
        for (_r, l, rgt, _a) in (tuple(int(x) for x in j) for j in ev.deletions):
            assert explained(g[:l + 1] + g[rgt:], l)
        for (_r, l, sq) in ev.insertions:
            assert explained(g[:l + 1] + sq + g[l + 1:], l)
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
