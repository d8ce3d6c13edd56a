"""CPU: oracle and kernel logic against the committed fixtures (tests/golden/, see make_golden.py for provenance)."""
import copy

import pytest

import orc
import sim
from golden_util import CASES, events_text, fusions_text, load
from tophat_amd.batch import events_to_span_inputs, merge_events


@pytest.mark.parametrize("name", CASES)
def test_oracle_and_kernel_logic_reproduce_fixture(name, tmp_path):
    c = load(name)
    seqs = [None if s is None else orc.fold_genome_char(s) for s in c["seqs"]]
    g = orc.Genome(seqs)
    ev = ev2 = fus = fus2 = None
    for side, b in c["seg_batches"]:
        p = copy.copy(c["p"])
        p.read_side = side
        e = orc.segjuncs(p, g, b)
        e2 = sim.segjuncs(p, seqs, b)
        ev = e if ev is None else merge_events(ev, e)
        ev2 = e2 if ev2 is None else merge_events(ev2, e2)
        if c["fusion"]:
            f = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist, c["fusion_ignore"])
            f2 = sim.fusions(p, seqs, b, c["fusion_ignore"])
            fus = f if fus is None else orc.merge_fusions(fus, f)
            fus2 = f2 if fus2 is None else orc.merge_fusions(fus2, f2)
    exp = dict(c["exp"])
    exp_fus = exp.pop("fusions", None)
    assert events_text(ev, c["names"], tmp_path) == exp
    assert events_text(ev2, c["names"], tmp_path) == exp
    if c["fusion"]:
        assert fusions_text(fus, ev.juncs, c["names"], tmp_path) == exp_fus
        assert fusions_text(fus2, ev.juncs, c["names"], tmp_path) == exp_fus
    juncs, ins = events_to_span_inputs(ev)
    for sd, sb in c["span_batches"].items():
        want = c["exp_span"][sd]
        got = [a.sam_fields(int(sb.read_id[a.read_idx]), c["names"]) for a in orc.spanning(c["p"], g, sb, juncs, ins)]
        assert got == want
        got2, st = sim.spanning(c["p"], seqs, sb, juncs, ins)
        assert [a.sam_fields(int(sb.read_id[a.read_idx]), c["names"]) for a in got2] == want


def fusion_span_inputs(c):
    """junction / insertion / fusion sets of a --fusion-search fixture as long_spanning_reads loads them from the list files"""
    import numpy as np
    from tophat_amd.host import JUNC_DTYPE
    d, ref_ids = c["dir"], c["ref_ids"]
    rows = set()
    for line in open(d + "/expected.juncs"):
        t = line.split("\t")
        rows.add((ref_ids[t[0]], int(t[1]), int(t[2]), 1 if t[3].strip() == "-" else 0))
    for line in open(d + "/expected.deletions"):
        t = line.split("\t")
        rows.add((ref_ids[t[0]], int(t[1]) - 1, int(t[2]), 0))
    ins = set()
    for line in open(d + "/expected.insertions"):
        t = line.rstrip("\n").split("\t")
        ins.add((ref_ids[t[0]], int(t[1]), t[3]))
    juncs = np.array(sorted(rows), dtype=JUNC_DTYPE) if rows else np.zeros(0, dtype=JUNC_DTYPE)
    return juncs, sorted(ins, key=lambda x: (x[0], x[1], len(x[2]))), orc.read_fusions_file(d + "/expected.fusions", ref_ids)


def span_records(c, sd, sb, alns):
    out = []
    for a in alns:
        rid = int(sb.read_id[a.read_idx])
        out += [tuple(str(x) for x in r) for r in a.sam_records(rid, c["names"], c["sides"][sd]["reads"][rid], c["sides"][sd]["quals"][rid])]
    return out


from golden_util import FUSION_SPAN_CASES  # noqa: E402


@pytest.mark.parametrize("name", FUSION_SPAN_CASES)
def test_fusion_spanning_oracle_reproduces_fixture(name):
    """long_spanning_reads --fusion-search: fused segment hits from the fusion contigs of the junction database, all four
    directions, two-record XF output -- record for record (SEQ, QUAL and tags included) what the scratch build wrote"""
    c = load(name)
    g = orc.Genome(c["seqs"])
    juncs, ins, fus = fusion_span_inputs(c)
    n_fused = 0
    for sd, sb in c["span_batches"].items():
        alns = orc.spanning_fusion(c["p"], g, sb, juncs, ins, fus, True)
        n_fused += sum(1 for a in alns if a.is_fusion())
        assert span_records(c, sd, sb, alns) == c["exp_span_full"][sd]
    assert n_fused >= 20


@pytest.mark.parametrize("name", FUSION_SPAN_CASES)
def test_fusion_tier_logic_reproduces_fixture(name):
    """the kernel headers of the fusion tier (thj_span_fusion.h behind tier 0) compiled for the CPU, on the same fixtures"""
    c = load(name)
    juncs, ins, fus = fusion_span_inputs(c)
    p = copy.copy(c["p"])
    p.fusion_search = 1
    for sd, sb in c["span_batches"].items():
        for skip0 in (False, True):
            alns, st = sim.spanning_fusion(p, c["seqs"], sb, juncs, ins, fus, skip0)
            assert st[1] == 0
            assert span_records(c, sd, sb, alns) == c["exp_span_full"][sd]
