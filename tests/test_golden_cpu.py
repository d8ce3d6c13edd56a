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
