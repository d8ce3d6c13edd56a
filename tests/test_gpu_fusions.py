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
    candidates a read, the copies further apart than --fusion-min-dist.  thj_k_fusion takes such reads a pair a thread, the workgroup on
    one read after the other (round 6; before: the read's own thread, with a queue of 1 024 for a tile's 7 800 pairs) -- the
    FusionSimpleSet must come out the same: keys, counts, smallest edit distances."""
    import torch  # noqa: F401
    from bench import sample_segbatch
    from tophat_amd.params import Params
    from tophat_amd.synth import make_device_workload, make_scale_genome
    seqs, genes = make_scale_genome(1, [4_000_000], 3000, intron_max=1500, exon_len=300)
    S, copies = 40_000, 41
    for k in range(1, copies):
        seqs[0][k * S:(k + 1) * S] = seqs[0][:S]
    fam = (genes[:, 3] + 300 + 1000 < S)
    uniq = genes[:, 1] >= copies * S + 1000
    genes = genes[fam | uniq]
    strs = [s.tobytes().decode() for s in seqs]
    n = 3000
    w = make_device_workload(9, seqs, genes, None, n, "cpu", exon_len=300, multi_frac=0.3, dup_shift=S, max_copies=copies, fusion_frac=0.02)
    og = orc.Genome(strs)
    want = None
    runs = []
    heavy_mates = 0
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(strs))
        for sd, side in (("left", 1), ("right", 2)):
            p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20, fusion_min_dist=30000)
            sb = sample_segbatch(w[sd], n)
            # ... and for every third read only the first segment maps: no partner among the read's own hits, so the mate-anchored part runs
            # (find_fusions :3117-3202) -- for a family read over k x k (hit, mate hit) pairs and then k first-segment hits per pseudo-hit; the
            # workgroup takes such reads a triple a thread
            so = sb.seg_off.astype(np.int64)
            keep = np.ones(len(sb.hits), dtype=bool)
            cnt = []
            for r in range(0, n, 3):
                keep[so[r * sb.nseg + 1]:so[(r + 1) * sb.nseg]] = False
            for k in range(n * sb.nseg):
                cnt.append(int(keep[so[k]:so[k + 1]].sum()))
            sb.hits = sb.hits[keep]
            sb.seg_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
            first = np.array(cnt[0::sb.nseg][0::3], dtype=np.int64)
            nm = np.diff(sb.mate_off.astype(np.int64))[0::3]
            heavy_mates += int((first * first * nm >= 256).sum())
            f = orc.fusions(p, og, sb, p.fusion_anchor_length, p.fusion_min_dist)
            want = f if want is None else orc.merge_fusions(want, f)
            runs.append((p, ctx.upload_batch(sb)))
        got = ctx.fusions(runs)
    assert heavy_mates > 20
    cols = ("ref_id1", "ref_id2", "left", "right", "dir", "count", "edit_dist")
    cells = (w["left"]["seg_off"][1:] - w["left"]["seg_off"][:-1]).reshape(n, 4)
    assert int(((cells[:, 0] >= 8) & (cells[:, 3] >= 8)).sum()) > 20                 # reads with 64 or more pairs, up to 41 x 41
    assert int(sum(int(x["count"]) for x in want)) > 1000
    assert [tuple(int(x[k]) for k in cols) for x in got] == [tuple(int(x[k]) for k in cols) for x in want]
