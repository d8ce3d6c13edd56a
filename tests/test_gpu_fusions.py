"""GPU parity: thj_k_fusion through the C ABI against the CPU oracle (exact FusionSimpleSet: keys, counts, edit distances)."""
import pytest

import orc
from tophat_amd import host
from test_hostsim_fusions import FUSION_CASES, fusion_batches

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", FUSION_CASES, ids=lambda c: "seed%d_%s_rl%d" % (c["seed"], "pe" if c["paired"] else "se", c["read_len"]))
def test_fusions_match_oracle(cfg):
    case, batches = fusion_batches(cfg, n_reads=500)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    want = None
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs = []
        for p, b in batches:
            f = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist)
            want = f if want is None else orc.merge_fusions(want, f)
            runs.append((p, ctx.upload_batch(b)))
        got = ctx.fusions(runs)
    assert len(want) > 30
    assert [tuple(int(x[k]) for k in ("ref_id1", "ref_id2", "left", "right", "dir", "count", "edit_dist")) for x in got] == \
           [tuple(int(x[k]) for k in ("ref_id1", "ref_id2", "left", "right", "dir", "count", "edit_dist")) for x in want]


def test_fusion_ignore_chromosomes_gpu():
    cfg = FUSION_CASES[1]
    case, batches = fusion_batches(cfg, n_reads=500)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    want = None
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs = []
        for p, b in batches:
            f = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist, ignore_ref_ids=[2])
            want = f if want is None else orc.merge_fusions(want, f)
            runs.append((p, ctx.upload_batch(b)))
        got = ctx.fusions(runs, ignore_ref_ids=[2])
        assert got.tolist() == want.tolist() and len(want) > 10
        assert len(ctx.fusions(runs)) > len(want)          # the ignore set is cleared by passing none
