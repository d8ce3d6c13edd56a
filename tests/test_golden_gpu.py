"""GPU: the HIP path through the C ABI reproduces the committed fixtures (both stages, device-to-device sets)."""
import copy

import pytest

import orc
from golden_util import CASES, events_text, fusions_text, load
from tophat_amd import host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_hip_path_reproduces_fixture(name, tmp_path):
    c = load(name)
    seqs = [None if s is None else orc.fold_genome_char(s) for s in c["seqs"]]
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs, base = [], 0
        for side, b in c["seg_batches"]:
            p = copy.copy(c["p"])
            p.read_side = side
            runs.append((p, ctx.upload_batch(b, ordinal_base=base)))
            base += b.n_reads
        ev = ctx.segjuncs(runs)
        exp = dict(c["exp"])
        exp_fus = exp.pop("fusions", None)
        assert events_text(ev, c["names"], tmp_path) == exp
        if c["fusion"]:
            fus = ctx.fusions(runs, c["fusion_ignore"])
            assert fusions_text(fus, ev.juncs, c["names"], tmp_path) == exp_fus
        ctx.span_sets_from_segjuncs()
        for sd, sb in c["span_batches"].items():
            got = ctx.spanning(c["p"], [ctx.upload_span_batch(sb)])
            assert [a.sam_fields(int(sb.read_id[a.read_idx]), c["names"]) for a in got] == c["exp_span"][sd]


from golden_util import FUSION_SPAN_CASES  # noqa: E402


@pytest.mark.parametrize("name", FUSION_SPAN_CASES)
def test_hip_path_fusion_spanning_reproduces_fixture(name):
    """long_spanning_reads --fusion-search on the device (thj_k_stitch_fusion behind tier 0): the records the scratch build wrote,
    two-record XF output included"""
    from test_golden_cpu import fusion_span_inputs, span_records
    c = load(name)
    juncs, ins, fus = fusion_span_inputs(c)
    p = copy.copy(c["p"])
    p.fusion_search = 1
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(c["seqs"]))
        ctx.upload_span_sets(juncs, ins)
        ctx.upload_span_fusions(fus)
        n_fused = 0
        for sd, sb in c["span_batches"].items():
            alns = ctx.spanning(p, [ctx.upload_span_batch(sb)])
            n_fused += sum(1 for a in alns if a.is_fusion())
            assert span_records(c, sd, sb, alns) == c["exp_span_full"][sd]
            # and with fusion search off the same batch gives what the plain tiers give the oracle
            p0 = copy.copy(c["p"])
            want0 = orc.spanning_fusion(p0, orc.Genome(c["seqs"]), sb, juncs, ins, fus, False)
            assert ctx.spanning(p0, [ctx.upload_span_batch(sb)]) == want0
        assert n_fused >= 20
