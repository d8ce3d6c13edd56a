"""GPU: the exchange step of the multi-GPU path without the collective -- two contexts run stage 1 on two shards of the
reads, each folds the other's sorted event keys in (thj_segjuncs_merge_keys_async / _merge_insertions_async, what
bench.py does with the all-gathered rows) and both must end with the single-context result."""
import copy

import pytest

import orc
from tophat_amd import host
from tophat_amd.batch import build_seg_batch
from tophat_amd.params import Params
from tophat_amd.synth import make_case

pytestmark = pytest.mark.gpu


def _shard(seg_recs, reads, lo, hi):
    return [[h for h in seg if lo <= h[0] < hi] for seg in seg_recs], {k: v for k, v in reads.items() if lo <= k < hi}


def test_cross_merge_of_two_contexts_equals_one():
    case = make_case(seed=77, paired=False, read_len=100, seg_len=25, n_reads=3000, boundary_bias=0.5, indel_frac=0.3,
                     contig_lens=(60000, 30000), genes_per_contig=12)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    ids = sorted(case.reads["left"])
    cut = ids[len(ids) // 2]
    p = Params(read_side=1)
    whole = build_seg_batch(case.seg_recs["left"], case.reads["left"])
    parts = [build_seg_batch(*_shard(case.seg_recs["left"], case.reads["left"], lo, hi)) for lo, hi in ((0, cut), (cut, 1 << 31))]
    with host.Context(0) as one:
        one.upload_genome(host.pack_genome(seqs))
        want = one.segjuncs([(p, one.upload_batch(whole))])
    with host.Context(0) as a, host.Context(0) as b:
        ctxs = (a, b)
        cnts = []
        base = 0
        for ctx, part in zip(ctxs, parts):
            ctx.upload_genome(host.pack_genome(seqs))
            ctx.reset()
            ctx.run(p, ctx.upload_batch(part, ordinal_base=base))
            base += part.n_reads
            cnts.append(ctx.finish())
        assert 0 < cnts[0].n_juncs < len(want.juncs) or 0 < cnts[1].n_juncs < len(want.juncs)
        state = []
        for ctx in ctxs:                      # sorted keys of this rank, still on the device
            ctx.sync()
            state.append((ctx.device_keys(0), ctx.device_keys(1), ctx.device_insertions()))
        for me, other in ((0, 1), (1, 0)):
            (jp, jn), (dp, dn), (ik, iv, inn) = state[other]
            ctxs[me].merge_keys(0, jp, jn)
            ctxs[me].merge_keys(1, dp, dn)
            ctxs[me].merge_insertions(ik, iv, inn)
        got = []
        for ctx in ctxs:
            ctx.sync()
        for ctx in ctxs:
            got.append(ctx.download(ctx.finish()))
    for g in got:
        assert g.juncs.tolist() == want.juncs.tolist()
        assert g.deletions.tolist() == want.deletions.tolist()
        assert g.insertions == want.insertions
    assert len(want.juncs) > 20 and len(want.insertions) > 0
