"""GPU, butterfly search at scale (2 x 50 bp pairs in two segments against a chr20-sized genome, the shape the coverage-search family is
for): thj_butterfly_run after a coverage search of the same pass equals the oracle on the same hits and unmapped reads; device time reported."""
import time

import numpy as np
import pytest
import torch

import orc
from bench import CHR20_LEN, cbatch_from_tensors, sample_segbatch
from tophat_amd import host
from tophat_amd.batch import HIT_DTYPE
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT
from tophat_amd.synth import make_device_workload, make_scale_genome

pytestmark = pytest.mark.gpu


def _tuples(a):
    return {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in a}


def test_butterfly_search_at_scale_matches_oracle():
    PAIRS, N_IUM = 1_000_000, 200_000
    dev = torch.device("cuda", 0)
    seqs, genes = make_scale_genome(1, [CHR20_LEN], 20000, exon_len=300)
    strs = [s.tobytes().decode() for s in seqs]
    w = make_device_workload(100, seqs, genes, None, PAIRS, dev, exon_len=300, read_len=50)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    with host.Context(0, stream=stream.cuda_stream) as ctx:
        ctx.upload_genome(host.pack_genome(strs))
        ctx.configure(1 << 22, 1 << 20)
        ctx.reset()
        ctx.covsearch_reset()
        for side, base in (("left", 0), ("right", PAIRS)):       # the coverage map only: no segment search in this pass
            ctx.covsearch_add_hits(cbatch_from_tensors(w[side], base))
        for sd in ("left", "right"):
            ctx.covsearch_add_reads_device(N_IUM, w[sd]["W"], w[sd]["planes"].data_ptr(), w[sd]["read_len"].data_ptr())
        ctx.covsearch_run(20, 50, 20000)
        n_cov = ctx.covsearch_finish()
        ctx.sync()
        t0 = time.time()
        found = ctx.butterfly_run(50, 20000)
        dt = time.time() - t0
        got = _tuples(ctx.download(ctx.finish()).juncs)
    hits = np.concatenate([w[sd]["hits"].cpu().numpy().view(HIT_DTYPE).reshape(-1) for sd in ("left", "right")])
    ium = []
    for sd in ("left", "right"):
        sb = sample_segbatch(w[sd], N_IUM)
        ium += [sb.read_seq(r) for r in range(sb.n_reads)]
    og = orc.Genome(strs)
    t1 = time.time()
    want = _tuples(orc.butterfly_search(og, hits, ium, 50, 20000))
    t_orc = time.time() - t1
    cov = _tuples(orc.coverage_search(og, hits, ium, 20, 50, 20000))
    print("butterfly search: %d junctions (coverage search %d); device %.1f ms (%d hits, %d unmapped reads, %d bp), oracle %.1f s" % (
        found, n_cov, 1e3 * dt, len(hits), len(ium), CHR20_LEN, t_orc))
    assert found == len(want) and found > 1000
    assert got == want | cov
    # capped: the shortest introns survive (skip count = intron length), ties by junction order
    with host.Context(0, stream=stream.cuda_stream) as ctx:
        ctx.upload_genome(host.pack_genome(strs))
        ctx.configure(1 << 22, 1 << 20)
        ctx.reset()
        ctx.covsearch_reset()
        for side, base in (("left", 0), ("right", PAIRS)):
            ctx.covsearch_add_hits(cbatch_from_tensors(w[side], base))
        for sd in ("left", "right"):
            ctx.covsearch_add_reads_device(N_IUM, w[sd]["W"], w[sd]["planes"].data_ptr(), w[sd]["read_len"].data_ptr())
        assert ctx.butterfly_run(50, 20000, 500) == 500
        got_cap = _tuples(ctx.download(ctx.finish()).juncs)
    assert got_cap == _tuples(orc.butterfly_search(og, hits, ium, 50, 20000, 500))
