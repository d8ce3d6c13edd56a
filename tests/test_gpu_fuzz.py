"""GPU: the adversarial batches of test_fuzz_cpu.py through the real kernels and the C ABI (several tiles per batch:
LDS staging, work lists, rescue list, task queue, the three stitch tiers and their worklists) against the oracle."""
import os

import numpy as np
import pytest

import orc
from test_fuzz_cpu import rand_genome, rand_seg_batch, rand_span_batch
from tophat_amd import host
from tophat_amd.batch import JUNC_DTYPE
from tophat_amd.params import Params
from util import assert_events_equal

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("THJ_FUZZ_SEEDS", "12"))


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_segment_juncs_gpu(seed):
    rng = np.random.default_rng(9000 + seed)
    seqs = rand_genome(rng, int(rng.integers(1, 4)))
    L = int(rng.choice([20, 25, 25, 32, 33, 47, 50, 64]))
    nseg = int(rng.choice([2, 3, 4, 6])) if L <= 32 else int(rng.choice([2, 3]))
    paired = bool(seed % 2)
    b = rand_seg_batch(rng, seqs, 900, L, nseg, paired)
    p = Params(segment_length=L, read_side=1 + seed % 2, library_type=int(rng.choice([0, 0, 1, 2, 3])),
               min_segment_intron=int(rng.choice([10, 50])), max_segment_intron=int(rng.choice([400, 5000, 500000])),
               max_insertion_length=int(rng.choice([1, 3, 6])), max_deletion_length=int(rng.choice([1, 3, 10])),
               inner_dist_mean=int(rng.choice([0, 30, 50])), inner_dist_std_dev=int(rng.choice([5, 20, 60])),
               segment_mismatches=int(rng.choice([0, 2, 3])), fusion_min_dist=int(rng.choice([50, 1000])))
    g = orc.Genome(seqs)
    want = orc.segjuncs(p, g, b)
    wf = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        db = ctx.upload_batch(b)
        got = ctx.segjuncs([(p, db)])
        gf = ctx.fusions([(p, db)])
    assert_events_equal(got, want, "seed %d" % seed)
    for k in ("windows", "indel_pairs", "rescue_pairs"):
        assert got.stats[k] == want.stats[k], k
    assert gf.tolist() == wf.tolist()


def span_fuzz_case(seed):
    """-> (contig strings, SpanBatch, Params, junction array): one adversarial long_spanning_reads batch"""
    rng = np.random.default_rng(9500 + seed)
    seqs = rand_genome(rng, int(rng.integers(1, 3)))
    L = int(rng.choice([20, 25, 25, 40]))
    nseg = int(rng.choice([1, 2, 3, 4, 6]))
    if L * (nseg + 1) > 256:          # reads stay within the 256 bp the device path supports
        nseg = 256 // L - 1
    sb = rand_span_batch(rng, seqs, 900, L, nseg)
    p = Params(segment_length=L, max_insertion_length=int(rng.choice([1, 3])), max_deletion_length=int(rng.choice([1, 3, 10])),
               min_report_intron=int(rng.choice([10, 50])), max_report_intron=int(rng.choice([300, 5000, 500000])),
               read_mismatches=int(rng.choice([2, 4])), read_edit_dist=int(rng.choice([2, 5])), read_gap_length=int(rng.choice([2, 3])))
    juncs = set()
    h = sb.hits
    for k in range(0, len(h) - 1):
        a, b_ = h[k], h[k + 1]
        if a["ref_id"] != b_["ref_id"]:
            continue
        ra = int(a["left"]) + sum(int(c & 0x0FFFFFFF) for c in a["cigar"] if (c >> 28) in (1, 5, 11))
        for d in (-2, 0, 1):
            l_, r_ = ra - 1 + d, int(b_["left"]) + d
            if r_ > l_ + 1 and l_ >= 0:
                juncs.add((int(a["ref_id"]), l_, r_, int(rng.integers(0, 2))))
    jl = sorted(juncs)
    ja = np.array(jl, dtype=JUNC_DTYPE) if jl else np.zeros(0, dtype=JUNC_DTYPE)
    return seqs, sb, p, ja


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_long_spanning_reads_gpu(seed):
    seqs, sb, p, ja = span_fuzz_case(seed)
    want = orc.spanning(p, orc.Genome(seqs), sb, ja, [])
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(ja, [])
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
    assert got == want
