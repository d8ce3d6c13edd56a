"""Known-answer tests from the reference's own regression cases (tests/golden_ref/, see tests/ref_regression.py):
the CPU oracle -- and the CPU build of the kernel logic -- against what the reference's authors recorded."""
import pytest

import orc
import ref_regression as rr
import sim


def _orc_juncs_db(genome):
    og = orc.Genome([orc.fold_genome_char(genome)])
    return lambda names, jf, inf, df, read_len, min_anchor: orc.juncs_db_text(names, og, jf, inf, df, None, read_len, min_anchor)


def _load(case, tmp_path):
    import os
    genome = "".join(l.strip() for l in open(os.path.join(rr.GOLD, case, "genome.fa")) if not l.startswith(">")).upper()
    return rr.load(case, tmp_path, _orc_juncs_db(genome))


def test_segment_juncs_finds_the_recorded_junction(tmp_path):
    c = _load("test_SimpleSplicing", tmp_path)
    og = orc.Genome([orc.fold_genome_char(c["genome"])])
    ev = orc.segjuncs(c["p"], og, c["seg_batch"])
    got = sorted((int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ev.juncs)
    assert got == c["recorded_juncs"] == [(1, 63, 138, 0)]           # fake:45-154, blocks 19,16, '+'
    assert len(ev.deletions) == 0 and len(ev.insertions) == 0
    # the CPU build of the kernel logic agrees
    ev2 = sim.segjuncs(c["p"], [orc.fold_genome_char(c["genome"])], c["seg_batch"])
    assert [tuple(j) for j in ev2.juncs] == [tuple(j) for j in ev.juncs]


@pytest.mark.parametrize("case", rr.CASES)
def test_long_spanning_reads_reproduces_the_recorded_alignments(case, tmp_path):
    c = _load(case, tmp_path)
    og = orc.Genome([orc.fold_genome_char(c["genome"])])
    alns = orc.spanning(c["p"], og, c["span_batch"], c["span_juncs"], c["span_ins"])
    n, gapped = rr.check_recorded_alignments(c, alns)
    assert n == {"test_SimpleSplicing": 986, "test_SimpleIndel": 991, "test_IndelWithErrors": 1921}[case]
    assert gapped == {"test_SimpleSplicing": 64, "test_SimpleIndel": 117, "test_IndelWithErrors": 227}[case]
    # the CPU build of the kernel logic: record for record what the oracle says, in every tier arrangement
    for mode in (0, 1, 2):
        got, status = sim.spanning(c["p"], [orc.fold_genome_char(c["genome"])], c["span_batch"], c["span_juncs"], c["span_ins"], mode)
        got.sort(key=lambda a: a.read_idx)
        assert status[1] == 0 and status[2] == 0
        assert got == alns, "mode %d" % mode
