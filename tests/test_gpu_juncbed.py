"""GPU: the junction consensus through the C ABI -- thj_juncbed_* -- against the recorded
junctions.bed of the reference's nine regression cases, against the oracle on seeded long_spanning_reads results left
resident on the device, and on hand-made filter cases."""
import os

import numpy as np
import pytest

import orc
import ref_regression as rr
from test_hostsim_spanning import SPAN_CASES, span_inputs
from tophat_amd import host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", rr.CASES)
def test_recorded_alignments_give_the_recorded_junctions_bed(case):
    recs = rr.recorded_alignment_records(case)
    genome = "".join(l.strip() for l in open(os.path.join(rr.GOLD, case, "genome.fa")) if not l.startswith(">")).upper()
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([orc.fold_genome_char(genome)]))
        ctx.juncbed_reset()
        a = host.aln_array_from_tuples(recs)
        half = len(a) // 2
        ctx.juncbed_add_records(a[:half])              # in two pieces: the reduce does not care how records arrive
        ctx.juncbed_add_records(a[half:])
        js = ctx.juncbed_finish(8)
    assert host.junctions_bed_text(js, ["fake"]) == open(os.path.join(rr.GOLD, case, "junctions.bed")).read()


@pytest.mark.parametrize("cfg", SPAN_CASES, ids=lambda c: "seed%d_rl%d_L%d" % (c["seed"], c["read_len"], c["seg_len"]))
def test_resident_spanning_records_reduce_like_the_oracle(cfg):
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, cfg.get("n_reads", 400))
    want_alns = orc.spanning(p, g, sb, juncs, ins)
    want = orc.junction_consensus(orc.jrecs_from_alns(want_alns))
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(juncs, ins)
        got_alns = ctx.spanning(p, [ctx.upload_span_batch(sb)])
        assert got_alns == want_alns
        ctx.juncbed_reset()
        ctx.juncbed_add_span()                          # straight from the slots the stitch kernels wrote
        js = ctx.juncbed_finish(8)
        js2 = ctx.juncbed_finish(8)                     # finishing twice changes nothing
    assert js.tolist() == js2.tolist()
    assert [tuple(int(x) for x in r) for r in js.tolist()] == [tuple(int(x) for x in r) for r in want.tolist()]
    assert len(want) > 3
    assert host.junctions_bed_text(js, case.names) == orc.junctions_bed(want, case.names)


def test_filters_on_the_device():
    from test_juncbed_cpu import test_filters as _unused      # noqa: F401  (the same cases, see there for what each one shows)
    M, N = 1, 11

    def rec(left, a, gap, b, anti=False, ref=1):
        return (ref, left, anti, [(M, a), (N, gap), (M, b)])
    cases = [
        [rec(100, 20, 500, 7)],
        [rec(100, 20, 500, 7), rec(90, 30, 500, 12)],
        [rec(100, 20, 60000, 20)], [rec(100, 20, 60000, 20)] * 2, [rec(100, 20, 60000, 12)] * 2,
        [rec(100, 20, 500, 20, anti=False)] * 3 + [rec(103, 20, 497, 20, anti=True)],
        [rec(100, 20, 500, 20, anti=False)] * 2 + [rec(103, 20, 497, 20, anti=True)] * 2,       # equal support: both stay
        [(1, 100, False, [(M, 20), (N, 300), (M, 30), (N, 400), (M, 25)])],
        [(1, 100, False, [(M, 20), (N, 300), (M, 30), (N, 400), (M, 5)])],
        [(2, 1000, True, [(M, 10), (5, 3), (M, 10), (N, 100), (M, 15), (3, 2), (M, 9)])],
        [rec(3, 9, 100, 30), rec(5, 9, 98, 30, anti=True), rec(5, 9, 98, 30, anti=True)],       # left < anchor: no shadow test at all
        [],
    ]
    rng = np.random.default_rng(3)
    many = []
    for _ in range(3000):                                # a crowd of near-coincident junctions on both strands
        l0 = int(rng.integers(50, 400))
        many.append(rec(l0, int(rng.integers(5, 40)), int(rng.integers(60, 90)), int(rng.integers(5, 40)), anti=bool(rng.integers(0, 2)), ref=int(rng.integers(1, 3))))
    cases.append(many)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(["ACGT" * 5000, "TTGCA" * 4000]))
        for recs in cases:
            ctx.juncbed_reset()
            ctx.juncbed_add_records(host.aln_array_from_tuples(recs))
            got = ctx.juncbed_finish(8)
            want = orc.junction_consensus(orc.jrecs_from_tuples(recs))
            assert [tuple(int(x) for x in r) for r in got.tolist()] == [tuple(int(x) for x in r) for r in want.tolist()], recs[:3]
    assert len(want) > 50


def test_table_overflow_is_loud_and_recoverable():
    M, N = 1, 11
    recs = [(1, 100 + 3 * k, False, [(M, 20), (N, 200), (M, 20)]) for k in range(3000)]
    a = host.aln_array_from_tuples(recs)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(["ACGT" * 5000]))
        ctx.juncbed_configure(1)                         # rounds up to the minimum table
        ctx.juncbed_reset()
        ctx.juncbed_add_records(np.concatenate([a] * 30) if False else a)
        js = ctx.juncbed_finish(8)
        assert len(js) == 3000
