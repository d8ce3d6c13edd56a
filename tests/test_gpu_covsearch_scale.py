"""GPU, coverage search at the scale of BASELINE.json's config 5 shard (2 x 50 bp pairs in two segments against a
chr20-sized genome): the device result equals the oracle's on the same hits and unmapped reads, with the time of the
device pass reported."""
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


def test_coverage_search_at_scale_matches_oracle():
    PAIRS, N_IUM = 2_000_000, 500_000
    dev = torch.device("cuda", 0)
    seqs, genes = make_scale_genome(1, [CHR20_LEN], 20000, exon_len=300)
    strs = [s.tobytes().decode() for s in seqs]
    w = make_device_workload(100, seqs, genes, None, PAIRS, dev, exon_len=300, read_len=50)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    pl = Params(read_side=READ_LEFT, inner_dist_mean=50, inner_dist_std_dev=20)
    pr = Params(read_side=READ_RIGHT, inner_dist_mean=50, inner_dist_std_dev=20)
    with host.Context(0, stream=stream.cuda_stream) as ctx:
        ctx.upload_genome(host.pack_genome(strs))
        ctx.configure(1 << 22, 1 << 20)
        batches = [(pl, cbatch_from_tensors(w["left"], 0)), (pr, cbatch_from_tensors(w["right"], PAIRS))]
        for rep in range(2):                      # second pass: steady state (buffers allocated)
            ctx.reset()
            ctx.covsearch_reset()
            for p, cb in batches:
                ctx.run(p, cb)
                ctx.covsearch_add_hits(cb)
            ctx.sync()
            t0 = time.time()
            for sd in ("left", "right"):          # the first N_IUM reads of each side play the unmapped reads
                ctx.covsearch_add_reads_device(N_IUM, w[sd]["W"], w[sd]["planes"].data_ptr(), w[sd]["read_len"].data_ptr())
            ctx.covsearch_run(20, 50, 20000)
            found = ctx.covsearch_finish()
            dt = time.time() - t0
            ev = ctx.download(ctx.finish())
        got = {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ev.juncs}
    # the oracle on the same inputs: every hit of both sides, the same reads as text
    hits = np.concatenate([w[sd]["hits"].cpu().numpy().view(HIT_DTYPE).reshape(-1) for sd in ("left", "right")])
    ium = []
    for sd in ("left", "right"):
        sb = sample_segbatch(w[sd], N_IUM)
        ium += [sb.read_seq(r) for r in range(sb.n_reads)]
    og = orc.Genome(strs)
    t1 = time.time()
    cov = orc.coverage_search(og, hits, ium, 20, 50, 20000)
    t_orc = time.time() - t1
    want_cov = {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in cov}
    print("coverage search: %d junctions; device %.1f ms (%d hits, %d unmapped reads, %d bp), oracle %.1f s" % (
        found, 1e3 * dt, len(hits), len(ium), CHR20_LEN, t_orc))
    assert found == len(want_cov) and found > 1000
    assert want_cov <= got
    # what is in the set beyond the coverage junctions comes from the segment search: none of it may be missing either
    ctx2 = None
    with host.Context(0, stream=stream.cuda_stream) as ctx:
        ctx.upload_genome(host.pack_genome(strs))
        ctx.configure(1 << 22, 1 << 20)
        ctx.reset()
        for p, cb in batches:
            ctx.run(p, cb)
        seg = {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ctx.download(ctx.finish()).juncs}
    assert got == seg | want_cov
