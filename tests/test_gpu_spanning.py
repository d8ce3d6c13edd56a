"""GPU parity: thj_k_stitch through the C ABI against the CPU oracle (exact records, exact order)."""
import pytest

import orc
from tophat_amd import host
from tophat_amd.batch import build_seg_batch
from tophat_amd.params import Params
from test_hostsim_spanning import SPAN_CASES, span_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", SPAN_CASES, ids=lambda c: "seed%d_rl%d_L%d" % (c["seed"], c["read_len"], c["seg_len"]))
def test_spanning_matches_oracle(cfg):
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, n_reads=max(800, cfg.get("n_reads", 0)))
    want = orc.spanning(p, g, sb, juncs, ins)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(juncs, ins)
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
    assert len(want) > 100
    assert got == want


def test_pipeline_on_device_sets(tmp_path):
    """segment_juncs tables feed the stitch kernel device-to-device; same records as going through
    the host lists (what the .juncs/.deletions/.insertions files carry)."""
    cfg = SPAN_CASES[3]
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, n_reads=max(800, cfg.get("n_reads", 0)))
    want = orc.spanning(p, g, sb, juncs, ins)
    recs = [[h for h in seg if not any(o == 11 and n > p.max_report_intron for o, n in h[9])] for seg in case.seg_recs["left"]]
    b = build_seg_batch(recs, case.reads["left"])
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ev = ctx.segjuncs([(p, ctx.upload_batch(b))])
        ctx.span_sets_from_segjuncs()
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
        assert got == want
        # and the segment_juncs tables still work afterwards
        ev2 = ctx.segjuncs([(p, ctx.upload_batch(b))])
        assert ev2.juncs.tolist() == ev.juncs.tolist()


def test_two_batches_keep_read_order():
    cfg = SPAN_CASES[0]
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, n_reads=600)
    want = orc.spanning(p, g, sb, juncs, ins)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(juncs, ins)
        h = ctx.upload_span_batch(sb)
        a = ctx.spanning(p, [h])
        b = ctx.spanning(p, [h])        # reset + rerun: identical
    assert a == want and b == want


def test_many_joined_alignments_per_read_gpu():
    """30 joined alignments per read (a 30-copy tandem repeat): multihit tier, overflow pool ordering"""
    import numpy as np
    from test_hostsim_spanning import repeat_span_batch
    from tophat_amd.batch import JUNC_DTYPE
    seq, sb = repeat_span_batch()
    p = Params()
    want = orc.spanning(p, orc.Genome([seq]), sb, np.zeros(0, dtype=JUNC_DTYPE), [])
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        ctx.upload_span_sets(np.zeros(0, dtype=JUNC_DTYPE), [])
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
    assert len(want) == 30 * sb.n_reads and got == want
