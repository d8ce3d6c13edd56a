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


def _fusion_cases():
    """fusion alignments that also hold splices: pieces running up (M, N, D) and down (m, n, d) the genome on either side of the fusion
    op, every direction, junctions before and behind the fusion (the ones behind belong to the second contig -- except after an RR
    fusion, which junctions_from_spliced_hit has no case for, junctions.cpp:77-85); plain spliced records among them"""
    M, m, I, D, d, N, n = 1, 2, 3, 5, 6, 11, 12
    FF, FR, RF, RR = 7, 8, 9, 10
    return [
        (1, 1000, False, [(M, 30), (N, 200), (M, 20), (FF, 5000), (M, 50)], 2),
        (1, 1000, False, [(M, 30), (N, 200), (M, 20), (FF, 5000), (M, 25), (N, 300), (M, 25)], 2),
        (1, 1000, True, [(M, 30), (N, 200), (M, 20), (FR, 9000), (m, 25), (n, 300), (m, 25)], 2),
        (2, 4000, False, [(m, 20), (n, 150), (m, 30), (RF, 700), (M, 30), (N, 90), (M, 20)], 1),
        (2, 4000, False, [(m, 20), (n, 150), (m, 30), (RR, 700), (m, 30), (n, 90), (m, 20)], 1),
        (1, 2000, False, [(M, 12), (D, 2), (M, 10), (N, 500), (M, 28), (FF, 3000), (M, 20), (I, 1), (M, 29)], 2),
        (1, 1000, False, [(M, 30), (N, 200), (M, 70)]), (1, 1000, False, [(M, 30), (N, 200), (M, 70)]),
        (2, 5025, False, [(M, 25), (N, 300), (M, 25)]),
        (2, 8675, True, [(M, 25), (N, 300), (M, 25)]),
    ]


def test_fusion_alignments_on_the_device():
    recs = _fusion_cases() * 3
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(["ACGT" * 5000, "TTGCA" * 4000]))
        ctx.juncbed_reset()
        ctx.juncbed_add_records(host.aln_array_from_tuples(recs))
        got = ctx.juncbed_finish(8)
    want = orc.junction_consensus(orc.jrecs_from_tuples(recs))
    assert [tuple(int(x) for x in r) for r in got.tolist()] == [tuple(int(x) for x in r) for r in want.tolist()]
    assert len(want) >= 8 and {int(r["ref_id"]) for r in want} == {1, 2}


def test_thj_junctions_reads_fusion_bams(tmp_path):
    """the executable on a BAM that holds fusion alignments in their two-record XF:Z form (bwt_map.cpp:2047-2083): the XF re-parse of
    BAMHitFactory (bwt_map.cpp:1208-1318) -> the same junctions.bed as the oracle's consensus of the alignments themselves"""
    import subprocess
    from locked_make import locked_make
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    locked_make(os.path.join(here, "hostio"))
    names, lens = ["chrA", "chrB"], [20000, 20000]
    seqs = ["ACGT" * 5000, "TTGCA" * 4000]
    open(tmp_path / "ref.fa", "w").write("".join(">%s\n%s\n" % (n_, s_) for n_, s_ in zip(names, seqs)))
    open(tmp_path / "hdr.sam", "w").write("@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n_, l_) for n_, l_ in zip(names, lens)))
    letters = {1: "M", 2: "m", 3: "I", 4: "i", 5: "D", 6: "d", 11: "N", 12: "n"}
    recs = _fusion_cases() * 2
    lines = []
    for k, rec in enumerate(recs):
        ref, left, anti, cig = rec[:4]
        xs = "XS:A:%s" % ("-" if anti else "+")
        if len(rec) > 4:
            # F carries the position on the second contig + 1; the two records' own columns are the two pieces (their exact CIGARs do not matter here)
            cg = "".join("%d%s" % ((ln + 1, "F") if op in (7, 8, 9, 10) else (ln, letters[op])) for op, ln in cig)
            rl = sum(ln for op, ln in cig if op in (1, 2, 3, 4))
            xf = "%s-%s %d %s %s %s" % (names[ref - 1], names[rec[4] - 1], left + 1, cg, "A" * rl, "I" * rl)
            lines.append("%d\t0\t%s\t%d\t255\t%dM\t%s\t%s\tNM:i:0\t%s\tXF:Z:1 %s" % (k + 1, names[ref - 1], left + 1, 10, "A" * 10, "I" * 10, xs, xf))
            lines.append("%d\t0\t%s\t%d\t255\t%dM\t%s\t%s\tNM:i:0\t%s\tXF:Z:2 %s" % (k + 1, names[rec[4] - 1], 7, 10, "A" * 10, "I" * 10, xs, xf))
        else:
            cg = "".join("%d%s" % (ln, letters[op]) for op, ln in cig)
            rl = sum(ln for op, ln in cig if op in (1, 3))
            lines.append("%d\t0\t%s\t%d\t255\t%s\t%s\t%s\tNM:i:0\t%s" % (k + 1, names[ref - 1], left + 1, cg, "A" * rl, "I" * rl, xs))
    open(tmp_path / "recs.sam", "w").write("\n".join(lines) + "\n")
    subprocess.check_call([os.path.join(here, "hostio", "hostio_check"), "sam2bam", str(tmp_path / "hdr.sam"), str(tmp_path / "recs.sam"), str(tmp_path / "in.bam")])
    subprocess.check_call([os.path.join(root, "tophat_amd", "bin", "thj_junctions"), "--sam-header", str(tmp_path / "hdr.sam"), str(tmp_path / "ref.fa"),
                           str(tmp_path / "junctions.bed"), str(tmp_path / "in.bam")], stderr=subprocess.DEVNULL)
    want = orc.junctions_bed(orc.junction_consensus(orc.jrecs_from_tuples(recs)), names)
    assert open(tmp_path / "junctions.bed").read() == want
    assert want.count("\n") >= 8
