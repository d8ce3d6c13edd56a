"""GPU, BASELINE.json configs[1] at FULL size (10 M pairs vs a chr20-sized genome): size-independent properties.

* sample parity: the first 100 k reads of each side through the oracle; their junction set must be a subset of the
  full run's set, and the stage-2 records of those reads must be identical to the oracle's given the FULL junction set;
* shard-merge: running the batch as two halves into the same tables gives the same event sets (set semantics of
  segment_juncs.cpp:4911-4916), and stage 2 over two half batches concatenates to the same records;
* idempotence: a second pass over the same inputs reproduces the same sets and records;
* sortedness / structure: keys strictly increasing; records ordered by (read, rank), every CIGAR spans the read,
  NM/XM bookkeeping consistent; planted junctions recovered.
"""

import os

import numpy as np
import pytest
import torch

import orc
from bench import CHR20_LEN, GRCH38_LENS, cbatch_from_tensors, sample_segbatch, sample_spanbatch, span_cbatch_from_tensors
from tophat_amd import host
from tophat_amd.batch import events_to_span_inputs, merge_events
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT
from tophat_amd.synth import make_device_workload, make_scale_genome

pytestmark = pytest.mark.gpu
# (read length, pairs): configs[1] at full size, and the other read shapes of BASELINE.json's configs (76 bp = 3
# segments, 150 bp = 6 segments -> the 8-segment kernel variants, 50 bp = 2 segments) at a few million pairs
# ... and configs[1]'s shape with 15 % multihit reads (the second half of the genome a copy of the first: tiers 2 / 3)
SHAPES = [(100, 10_000_000, 0.0), (150, 2_000_000, 0.0), (50, 4_000_000, 0.0), (76, 2_000_000, 0.0), (100, 3_000_000, 0.15)]
# THJ_FULLSIZE_PAIRS=N: one extra run of configs[1]'s shape at N pairs (developer runs beyond the default sizes)
if os.environ.get("THJ_FULLSIZE_PAIRS"):
    SHAPES.append((100, int(os.environ["THJ_FULLSIZE_PAIRS"]), 0.0))
# ... and config 3's genome (25 contigs with the GRCh38 lengths, 3.09 Gb: block indices past 2^31 bases, contig boundaries,
# ref ids > 1); THJ_FULLSIZE_GRCH38=0 skips it (it needs ~10 GB of host memory for the genome text and the oracle's copy)
if os.environ.get("THJ_FULLSIZE_GRCH38", "1") == "1":
    SHAPES.append((100, 4_000_000, 0.0, "grch38"))
# ... and configs[4] as written: 2 x 50 bp reads (two segments: the mate-anchored rescue does the work), --max-intron-length
# 500000 with introns planted up to 499 999 bases -- the widest windows the path allows
SHAPES.append((50, 4_000_000, 0.0, "chr20", 499_999))
# ... and bench.py's default mix: 5 % of the pairs from a 41-copy repeat family (2..41 hits per segment: the reads a wave shares in
# stage 1, tier 3's shared pass in stage 2), 3 % deletion reads
SHAPES.append((100, 3_000_000, 0.05, "chr20", 200_000, 41))


def half(w, which):
    """first / second half of a device workload as a new workload dict (views + re-based CSR)"""
    n = w["n_reads"] // 2
    lo, hi = (0, n) if which == 0 else (n, w["n_reads"])
    nseg, W = w["nseg"], w["W"]
    out = dict(n_reads=hi - lo, nseg=nseg, W=W, qual_stride=w["qual_stride"])
    for off_key, data_key in (("seg_off", "hits"), ("span_off", "span_hits")):
        off = w[off_key][lo * nseg:hi * nseg + 1]
        base = int(off[0])
        out[off_key] = (off - base).contiguous()
        out[data_key] = w[data_key][base:int(off[-1])].contiguous()
    mo = w["mate_off"][lo:hi + 1]
    out["mate_off"] = (mo - int(mo[0])).contiguous()
    out["mate_hits"] = w["mate_hits"][int(mo[0]):int(mo[-1])].contiguous()
    out["planes"] = w["planes"][lo * 3 * W:hi * 3 * W].contiguous()
    out["read_len"] = w["read_len"][lo:hi].contiguous()
    out["quals"] = w["quals"][lo * w["qual_stride"]:hi * w["qual_stride"]].contiguous()
    return out


@pytest.fixture(scope="module", params=SHAPES,
                ids=lambda s: "%dbp_%dMpairs%s%s%s%s" % (s[0], s[1] // 1_000_000, "_multihit" if s[2] else "",
                                                         "_grch38" if len(s) > 3 and s[3] == "grch38" else "", "_intron%dk" % (s[4] // 1000) if len(s) > 4 else "",
                                                         "_family%d" % s[5] if len(s) > 5 else ""))
def world(request):
    read_len, pairs, multi_frac = request.param[:3]
    dev = torch.device("cuda", 0)
    intron_max = request.param[4] if len(request.param) > 4 else 200000
    if len(request.param) > 3 and request.param[3] == "grch38":
        seqs, genes = make_scale_genome(1, GRCH38_LENS, 300000, exon_len=300, intron_max=intron_max)
    else:
        seqs, genes = make_scale_genome(1, [CHR20_LEN], 20000, exon_len=300, intron_max=intron_max)
    dup_shift = 0
    max_copies = request.param[5] if len(request.param) > 5 else 2
    if max_copies > 2:
        dup_shift = len(seqs[0]) // 96
        for k in range(1, max_copies):
            seqs[0][k * dup_shift:(k + 1) * dup_shift] = seqs[0][:dup_shift]
        genes = genes[((genes[:, 0] == 0) & (genes[:, 3] + 300 + 1000 < dup_shift)) | (genes[:, 1] >= max_copies * dup_shift + 1000)]
    elif multi_frac > 0:
        dup_shift = len(seqs[0]) // 2
        seqs[0][dup_shift:2 * dup_shift] = seqs[0][:dup_shift]
        genes = genes[(genes[:, 0] == 0) & (genes[:, 3] + 300 + 1000 < dup_shift)]
    strs = [s.tobytes().decode() for s in seqs]
    w = make_device_workload(100, seqs, genes, None, pairs, dev, exon_len=300, read_len=read_len, multi_frac=multi_frac, dup_shift=dup_shift,
                             max_copies=max_copies, indel_frac=0.03 if max_copies > 2 else 0.0)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    ctx = host.Context(0, stream=stream.cuda_stream)
    ctx.upload_genome(host.pack_genome(strs))
    ctx.configure(1 << 22, 1 << 20)
    yield dict(ctx=ctx, w=w, strs=strs, genes=genes, stream=stream, read_len=read_len, pairs=pairs, multi_frac=multi_frac, intron_max=intron_max, max_copies=max_copies)
    ctx.close()
    del w
    torch.cuda.empty_cache()


def run_stage1(ctx, batches):
    ctx.reset()
    for p, cb in batches:
        ctx.run(p, cb)
    return ctx.download(ctx.finish())


def test_fullsize_properties(world):
    ctx, w, strs, genes = world["ctx"], world["w"], world["strs"], world["genes"]
    PAIRS, read_len = world["pairs"], world["read_len"]
    pl = Params(read_side=READ_LEFT, inner_dist_mean=50, inner_dist_std_dev=20)
    pr = Params(read_side=READ_RIGHT, inner_dist_mean=50, inner_dist_std_dev=20)
    full = [(pl, cbatch_from_tensors(w["left"], 0)), (pr, cbatch_from_tensors(w["right"], PAIRS))]
    ev = run_stage1(ctx, full)
    assert ev.stats["overflow_blocks"] == 0 or world["max_copies"] > 2
    # sortedness: strictly increasing in Junction::operator< order
    j = ev.juncs
    keys = [tuple(int(x[k]) for k in ("ref_id", "left", "right", "antisense")) for x in j]
    assert keys == sorted(set(keys))
    # planted introns recovered
    truth = {(int(g[0]) + 1, int(g[2]) - 1, int(g[3])) for g in genes}
    found = {(k[0], k[1], k[2]) for k in keys}
    assert len(found & truth) > (0.95 if read_len >= 100 else 0.5) * len(truth)
    if world["intron_max"] > 400_000:        # configs[4]: the longest planted introns are there and they are found
        assert max(k[2] - k[1] for k in truth) > 450_000
        assert max(k[2] - k[1] for k in found & truth) > 400_000

    # idempotence
    ev2 = run_stage1(ctx, full)
    assert ev2.juncs.tolist() == ev.juncs.tolist() and ev2.insertions == ev.insertions

    # shard-merge: four half batches into the same tables
    hl, hr = [half(w["left"], k) for k in (0, 1)], [half(w["right"], k) for k in (0, 1)]
    n0 = hl[0]["n_reads"]
    parts = [(pl, cbatch_from_tensors(hl[0], 0)), (pl, cbatch_from_tensors(hl[1], n0)),
             (pr, cbatch_from_tensors(hr[0], PAIRS)), (pr, cbatch_from_tensors(hr[1], PAIRS + n0))]
    ev3 = run_stage1(ctx, parts)
    assert ev3.juncs.tolist() == ev.juncs.tolist() and ev3.deletions.tolist() == ev.deletions.tolist()
    assert ev3.insertions == ev.insertions

    # sample parity, stage 1: the sample's events are a subset of the full run's
    og = orc.Genome(strs)
    m = 100_000
    es = merge_events(orc.segjuncs(pl, og, sample_segbatch(w["left"], m)), orc.segjuncs(pr, og, sample_segbatch(w["right"], m)))
    assert set(map(tuple, es.juncs.tolist())) <= set(map(tuple, ev.juncs.tolist()))

    # ---- stage 2 with the full junction set, device to device
    run_stage1(ctx, full)
    ctx.span_sets_from_segjuncs()
    p2 = Params()
    recs = {}
    for sd in ("left", "right"):
        ctx.span_reset()
        ctx.span_run(p2, span_cbatch_from_tensors(w[sd]))
        n = ctx.span_finish()
        a = ctx.span_download(n)
        recs[sd] = a
        # structure: ordered by (read, rank); every CIGAR spans the read; NM bookkeeping
        key = a["read_idx"].astype(np.int64) * 65536 + a["order"]
        assert (np.diff(key) > 0).all()
        ops, lens = a["cigar"] >> 28, a["cigar"] & 0x0FFFFFFF
        rlen = (lens * np.isin(ops, (1, 3, 13))).sum(1)
        assert (rlen == read_len).all()
        assert (a["XM"] <= a["mismatches"]).all() and (a["mismatches"] <= 2).all()
        assert n > (0.8 if read_len <= 100 else 0.6) * PAIRS
        if world["multi_frac"]:          # the multihit tier took its share and produced second records
            assert ctx.span_tier_counts()[1] > 0.5 * world["multi_frac"] * PAIRS
            assert int((a["order"] > 0).sum()) > 0.5 * world["multi_frac"] * PAIRS
        # sample parity, stage 2: records of the first m reads equal the oracle's, given the full junction set
        juncs, ins = events_to_span_inputs(ev)
        want = orc.spanning(p2, og, sample_spanbatch(w[sd], m), juncs, ins)
        got = host.alns_from_array(a[a["read_idx"] < m])
        assert got == want
    # shard-merge, stage 2: two half batches concatenate to the same records (read_idx is per batch)
    for sd, hs in (("left", hl), ("right", hr)):
        ctx.span_reset()
        ctx.span_run(p2, span_cbatch_from_tensors(hs[0], ctx))   # the halves also carry the optional dense hit-head array
        a0 = ctx.span_download(ctx.span_finish()).copy()
        ctx.span_reset()
        ctx.span_run(p2, span_cbatch_from_tensors(hs[1], ctx))
        a1 = ctx.span_download(ctx.span_finish()).copy()
        a1["read_idx"] += hs[0]["n_reads"]
        both = np.concatenate([a0, a1])
        assert both.tobytes() == recs[sd].tobytes()
