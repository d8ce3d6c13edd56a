"""GPU parity: thj_k_fusion through the C ABI against the CPU oracle (exact FusionSimpleSet: keys, counts, edit distances)."""
import numpy as np
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


def test_fusion_event_buffer_grows(monkeypatch):
    """the raw candidate events of a pass pile up until thj_fusion_finish reduces them; the buffer grows ahead of the count (round 6: configs[3] at
    full size had 9.9 M of them in a buffer of 1 M).  THJ_FUSION_CAP=1024 and the same batches twelve times over in one pass: more raw events than
    the buffer first held, every count twelve times the oracle's -- and a single batch that does not fit is done again with the room its count asked for"""
    import ctypes as C
    cfg = FUSION_CASES[0]
    case, batches = fusion_batches(cfg, n_reads=500)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    want = None
    for p, b in batches:
        f = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist)
        want = f if want is None else orc.merge_fusions(want, f)
    raw = int(sum(int(x["count"]) for x in want))
    assert 40 < raw < 500
    monkeypatch.setenv("THJ_FUSION_CAP", "1024")
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ups = [(p, ctx.upload_batch(b)) for p, b in batches]
        host._check(ctx.lib, ctx.lib.thj_fusion_reset_async(ctx._ctx), "thj_fusion_reset_async")
        for _ in range(12):
            for p, b in ups:
                cp = p.as_ctypes()
                host._check(ctx.lib, ctx.lib.thj_fusion_run_async(ctx._ctx, C.byref(cp), C.byref(b) if isinstance(b, host.CSegBatch) else b), "thj_fusion_run_async")
                ctx.sync()                   # (the count has arrived when the next call looks: the growth is then certain, not a matter of timing)
        n = C.c_int64()
        host._check(ctx.lib, ctx.lib.thj_fusion_finish(ctx._ctx, C.byref(n)), "thj_fusion_finish")
        assert 12 * raw > 1024 and n.value == len(want)
        got = np.zeros(max(1, n.value), dtype=host.FUSION_DTYPE)
        host._check(ctx.lib, ctx.lib.thj_fusion_download(ctx._ctx, host._ptr(got)), "thj_fusion_download")
    key = lambda x, m: tuple(int(x[k]) for k in ("ref_id1", "ref_id2", "left", "right", "dir")) + (int(x["count"]) * m, int(x["edit_dist"]))
    assert [key(x, 1) for x in got[:len(want)]] == [key(x, 12) for x in want]
    monkeypatch.setenv("THJ_FUSION_CAP", "64")
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ups = [(p, ctx.upload_batch(b)) for p, b in batches]
        # a single batch with more candidates than the buffer holds: thj_fusion_finish enlarges it to the count the pass reached and answers
        # THJ_ERETRY, the pass runs again (Context.fusions does; round 6: bench.py --fusion-search hands over 10 M pairs as one batch)
        host._check(ctx.lib, ctx.lib.thj_fusion_reset_async(ctx._ctx), "thj_fusion_reset_async")
        for p, b in ups:
            cp = p.as_ctypes()
            host._check(ctx.lib, ctx.lib.thj_fusion_run_async(ctx._ctx, C.byref(cp), C.byref(b) if isinstance(b, host.CSegBatch) else b), "thj_fusion_run_async")
        n = C.c_int64()
        assert ctx.lib.thj_fusion_finish(ctx._ctx, C.byref(n)) == -7 and b"run the pass again" in ctx.lib.thj_last_error()
        got2 = ctx.fusions(ups)
        assert [key(x, 1) for x in got2] == [key(x, 1) for x in want]
    monkeypatch.setenv("THJ_FUSION_CAP", "64")
    with host.Context(0) as ctx:                          # ... and without the first failed attempt
        ctx.upload_genome(host.pack_genome(seqs))
        ups = [(p, ctx.upload_batch(b)) for p, b in batches]
        got3 = ctx.fusions(ups)
        assert [key(x, 1) for x in got3] == [key(x, 1) for x in want]


def test_fusions_of_a_repeat_family():
    """reads of a 41-copy repeat family (bench.py's mix): every hit of the first segment pairs with every hit of the last one, k x k
    candidates a read, the copies further apart than --fusion-min-dist; and for a third of the reads only the first segment maps, so the
    mate-anchored part runs over k x k x 2 x k triples.  thj_k_fusion takes such reads a pair (a triple) a thread, the workgroup on one read
    after the other (round 6; thj_fusion_block.h -- the same code runs on the CPU in test_hostsim_fusions) -- the FusionSimpleSet must come
    out the same: keys, counts, smallest edit distances."""
    from test_hostsim_fusions import family_fusion_batches
    strs, batches, heavy_pairs, heavy_mates = family_fusion_batches()
    assert heavy_pairs > 20 and heavy_mates > 20
    og = orc.Genome(strs)
    want = None
    runs = []
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(strs))
        for p, sb in batches:
            f = orc.fusions(p, og, sb, p.fusion_anchor_length, p.fusion_min_dist)
            want = f if want is None else orc.merge_fusions(want, f)
            runs.append((p, ctx.upload_batch(sb)))
        got = ctx.fusions(runs)
    cols = ("ref_id1", "ref_id2", "left", "right", "dir", "count", "edit_dist")
    assert int(sum(int(x["count"]) for x in want)) > 1000
    assert [tuple(int(x[k]) for k in cols) for x in got] == [tuple(int(x[k]) for k in cols) for x in want]
