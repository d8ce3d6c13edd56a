"""GPU, the shape of BASELINE.json configs[3] at a few million pairs: 2 x 150 bp reads (six segments), 2 % of the pairs with a
chimeric left read whose two parts come from different genes -- other contigs and either strand included --, segment_juncs
--fusion-search followed by long_spanning_reads --fusion-search.  Size-independent properties:

* the planted fusions are found (stage 1) and the chimeric reads are joined through them (stage 2: thj_k_stitch_fusion);
* sample parity: stage-1 fusions of the first reads are among the full run's; the stage-2 records of the first reads equal
  the oracle's given the full junction / fusion sets;
* idempotence, and two half batches concatenate to the full batch's records;
* structure: one fusion op per fusion alignment, its second contig valid, every CIGAR spans the read;
* the reads that are not chimeric come out exactly as they do with fusion search off.
"""
import numpy as np
import pytest
import torch

import orc
from bench import CHR20_LEN, cbatch_from_tensors, sample_segbatch, sample_spanbatch, span_cbatch_from_tensors
from test_gpu_fullsize_properties import half
from test_scale_workload_cpu import fusion_list_from_events
from tophat_amd import host
from tophat_amd.batch import events_to_span_inputs
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT
from tophat_amd.synth import make_device_workload, make_scale_genome

pytestmark = pytest.mark.gpu
PAIRS = 2_000_000


def test_fullsize_fusion_search():
    dev = torch.device("cuda", 0)
    seqs, genes = make_scale_genome(1, [CHR20_LEN // 2, CHR20_LEN // 4, CHR20_LEN // 4], 20000, exon_len=300)
    strs = [s.tobytes().decode() for s in seqs]
    w = make_device_workload(100, seqs, genes, None, PAIRS, dev, exon_len=300, read_len=150, fusion_frac=0.02)
    torch.cuda.synchronize()
    fz = w["left"]["fusion_reads"].cpu().numpy()
    assert 0.015 * PAIRS < len(fz) < 0.025 * PAIRS
    stream = torch.cuda.Stream(device=dev)
    kw = dict(inner_dist_mean=50, inner_dist_std_dev=20, fusion_min_dist=100000)
    pl, pr = Params(read_side=READ_LEFT, **kw), Params(read_side=READ_RIGHT, **kw)
    with host.Context(0, stream=stream.cuda_stream) as ctx:
        ctx.upload_genome(host.pack_genome(strs))
        ctx.configure(1 << 22, 1 << 20)
        runs = [(pl, cbatch_from_tensors(w["left"], 0)), (pr, cbatch_from_tensors(w["right"], PAIRS))]
        ctx.reset()
        for p, cb in runs:
            ctx.run(p, cb)
        ev = ctx.download(ctx.finish())
        fus = ctx.fusions(runs)
        fkeys = {(int(x["ref_id1"]), int(x["ref_id2"]), int(x["left"]), int(x["right"]), int(x["dir"])) for x in fus}
        assert len(fkeys) > 0.8 * len(fz)
        assert any(k[0] != k[1] for k in fkeys) and {k[4] for k in fkeys} >= {7, 8}          # inter-contig, more than one direction
        assert ctx.fusions(runs).tolist() == fus.tolist()                                     # idempotence
        og = orc.Genome(strs)
        m = 60_000
        fs = orc.fusions(pl, og, sample_segbatch(w["left"], m), pl.fusion_anchor_length, pl.fusion_min_dist)
        assert {(int(x["ref_id1"]), int(x["ref_id2"]), int(x["left"]), int(x["right"]), int(x["dir"])) for x in fs} <= fkeys

        # ---- stage 2
        ctx.span_sets_from_segjuncs()
        fl = fusion_list_from_events(fus)
        ctx.upload_span_fusions(fl)
        p2 = Params(fusion_search=1, fusion_min_dist=100000)
        sp = span_cbatch_from_tensors(w["left"], ctx)
        ctx.span_reset()
        ctx.span_run(p2, sp)
        a = ctx.span_download(ctx.span_finish()).copy()
        ops, lens = a["cigar"] >> 28, a["cigar"] & 0x0FFFFFFF
        isf = np.isin(ops, (7, 8, 9, 10))
        nfo = isf.sum(1)
        assert set(np.unique(nfo).tolist()) <= {0, 1}
        fa = a[nfo == 1]
        assert len(np.unique(fa["read_idx"])) > 0.8 * len(fz) and set(np.unique(fa["read_idx"]).tolist()) <= set(fz.tolist())
        assert ((fa["cigar"][:, 15] >= 1) & (fa["cigar"][:, 15] <= len(strs))).all()         # ref_id2 rides in the last slot
        lens_r = np.where(np.arange(16)[None, :] < a["n_cigar"][:, None], lens, 0)
        assert ((lens_r * np.isin(ops, (1, 2, 3, 4, 13))).sum(1) == 150).all()
        key = a["read_idx"].astype(np.int64) * 65536 + a["order"]
        assert (np.diff(key) > 0).all()
        n_lean, n_multi, _ = ctx.span_tier_counts()
        # the chimeric reads went through the fusion kernel -- all but the few whose two parts lie on one strand of one contig within
        # an intron's reach: their only chain is compatible the plain way and fails the same way with fusion search on (tier 1 keeps them)
        assert n_multi >= 0.98 * len(fz)
        # sample parity against the oracle with the full sets
        juncs, ins = events_to_span_inputs(ev)
        want = orc.spanning_fusion(p2, og, sample_spanbatch(w["left"], m), juncs, ins, fl, True)
        assert sum(1 for x in want if x.is_fusion()) > 0.8 * int((fz < m).sum())
        resolver = host.span_md_resolver(strs, [sample_spanbatch(w["left"], m)])
        assert host.alns_from_array(a[a["read_idx"] < m], resolver) == want
        # idempotence
        ctx.span_reset()
        ctx.span_run(p2, sp)
        assert ctx.span_download(ctx.span_finish()).tobytes() == a.tobytes()
        # the fusion list handed over on the device (thj_span_fusions_from_segjuncs) instead of downloaded and uploaded
        assert ctx.fusion_search(runs) == len(fus)
        ctx.span_fusions_from_segjuncs()
        ctx.span_reset()
        ctx.span_run(p2, sp)
        assert ctx.span_download(ctx.span_finish()).tobytes() == a.tobytes()
        assert ctx.fusions(runs).tolist() == fus.tolist()                                     # ... and it still comes down afterwards
        # shard merge
        hs = [half(w["left"], k) for k in (0, 1)]
        parts = []
        for k in (0, 1):
            ctx.span_reset()
            ctx.span_run(p2, span_cbatch_from_tensors(hs[k], ctx))
            parts.append(ctx.span_download(ctx.span_finish()).copy())
        parts[1]["read_idx"] += hs[0]["n_reads"]
        assert np.concatenate(parts).tobytes() == a.tobytes()
        # fusion search off: the other reads' records are the same
        ctx.span_reset()
        ctx.span_run(Params(), sp)
        a0 = ctx.span_download(ctx.span_finish()).copy()
        keep = ~np.isin(a["read_idx"], fz)
        keep0 = ~np.isin(a0["read_idx"], fz)
        assert a[keep].tobytes() == a0[keep0].tobytes()
