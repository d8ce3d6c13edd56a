"""Coverage search, GPU: the HIP path through the C ABI against the fixtures under
tests/golden_cov/ and, on seeded cases, against the oracle."""
import copy

import numpy as np
import pytest

import orc
from cov_util import CASES, juncs_text, load
from tophat_amd import host
from tophat_amd.batch import HIT_DTYPE, build_seg_batch, hit_tuple_to_struct, merge_events
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT
from tophat_amd.synth import make_case

pytestmark = pytest.mark.gpu


def _tuples(a):
    return {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in a}


@pytest.mark.parametrize("name", CASES)
def test_hip_coverage_search_reproduces_fixture(name):
    c = load(name)
    seqs = [None if s is None else orc.fold_genome_char(s) for s in c["seqs"]]
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs, base = [], 0
        for side, b in c["seg_batches"]:
            p = copy.copy(c["p"])
            p.read_side = side
            runs.append((p, ctx.upload_batch(b, ordinal_base=base)))
            base += b.n_reads
        ev, found = ctx.segjuncs_with_coverage_search(runs, c["ium"], c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"])
    assert juncs_text(_tuples(ev.juncs), c["names"]) == c["expected"]
    g = orc.Genome(seqs)
    assert found == len(orc.coverage_search(g, c["hits"], c["ium"], c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"]))


@pytest.mark.parametrize("seed", range(300, 300 + int(__import__("os").environ.get("THJ_COV_SEEDS", "12"))))
def test_hip_coverage_search_matches_oracle(seed):
    """seeded cases: contig ends inside islands, several contigs, N runs, odd segment lengths and intron bounds"""
    rng = np.random.default_rng(seed)
    seg_len = int(rng.choice([20, 25, 25, 30]))
    paired = bool(seed % 2)
    try:
        case = make_case(seed=seed, paired=paired, read_len=2 * seg_len, seg_len=seg_len, n_reads=int(rng.integers(400, 1500)),
                         contig_lens=tuple(int(x) for x in rng.integers(8000, 30000, size=int(rng.integers(1, 4)))),
                         genes_per_contig=int(rng.integers(2, 10)), spliced_seg_frac=0.0, n_frac=float(rng.choice([0.0, 0.1])))
    except IndexError:
        pytest.skip("the generator could not place a gene for this seed")
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    min_ci, max_ci = int(rng.choice([50, 60, 100])), int(rng.choice([20000, 5000, 1500]))
    min_cov = min(20, seg_len - 2)
    p0 = Params(segment_length=seg_len, inner_dist_mean=50, inner_dist_std_dev=20)
    want, hits, ium, batches = None, [], [], []
    for sd, side in (("left", READ_LEFT), ("right", READ_RIGHT)):
        if sd not in case.reads:
            continue
        other = "right" if sd == "left" else "left"
        b = build_seg_batch(case.seg_recs[sd], case.reads[sd], case.full_recs[other], case.seg_recs[other][-1], include_top0=True) if paired \
            else build_seg_batch(case.seg_recs[sd], case.reads[sd], include_top0=True)
        p = copy.copy(p0)
        p.read_side = side
        batches.append((p, b))
        e = orc.segjuncs(p, g, b)
        want = e if want is None else merge_events(want, e)
        for recs in case.seg_recs[sd]:
            hits += [hit_tuple_to_struct(h) for h in recs]
        ium += [case.reads[sd][rid] for rid in sorted(case.reads[sd])]
    cov = orc.coverage_search(g, np.array(hits, dtype=HIT_DTYPE), ium, min_cov, min_ci, max_ci)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs, base = [], 0
        for p, b in batches:
            runs.append((p, ctx.upload_batch(b, ordinal_base=base)))
            base += b.n_reads
        ev, found = ctx.segjuncs_with_coverage_search(runs, ium, min_cov, min_ci, max_ci)
    assert _tuples(ev.juncs) == _tuples(want.juncs) | _tuples(cov)
    assert found == len(cov)


@pytest.mark.parametrize("name", CASES)
def test_segment_juncs_executable_with_coverage_search(name, tmp_path):
    """the drop-in executable the way tophat.py runs it for short reads: no --no-coverage-search, --ium-reads given"""
    import os
    import subprocess
    from cov_util import GOLD
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(GOLD, name)
    opts = open(os.path.join(d, "options.txt")).read().split("\n")
    argv = opts[0].split()
    kv = dict(x.split("=") for x in opts[1].split())
    paired = kv["paired"] == "1"
    nseg = len([f for f in os.listdir(d) if f.startswith("left_seg")])
    sides = ("left", "right") if paired else ("left",)
    out = {k: str(tmp_path / ("out." + k)) for k in ("juncs", "insertions", "deletions", "fusions")}
    cmd = [os.path.join(root, "tophat_amd", "bin", "segment_juncs"), "--no-microexon-search", "--segment-length", kv["segment_length"],
           "--sam-header", os.path.join(d, "hdr.sam")] + argv + ["--ium-reads", ",".join(os.path.join(d, "%s.fq" % sd) for sd in sides),
           os.path.join(d, "ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"]]
    for sd in sides:
        cmd += [os.path.join(d, "%s.fq" % sd), os.path.join(d, "%s_map.sam" % sd), ",".join(os.path.join(d, "%s_seg%d.sam" % (sd, k + 1)) for k in range(nseg))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Performing coverage-search" in r.stderr
    for k in ("juncs", "insertions", "deletions"):
        assert open(out[k]).read() == open(os.path.join(d, "expected." + k)).read(), k
    # long_spanning_reads on the lists just written (reads of one / two segments)
    from tophat_amd.bamio import read_bam
    for sd in sides:
        bam = str(tmp_path / ("span_%s.bam" % sd))
        r = subprocess.run([os.path.join(root, "tophat_amd", "bin", "long_spanning_reads"), "--segment-length", kv["segment_length"],
                            "--sam-header", os.path.join(d, "hdr.sam"), os.path.join(d, "ref.fa"), os.path.join(d, "%s.fq" % sd),
                            out["juncs"], out["insertions"], out["deletions"], "/dev/null", bam,
                            ",".join(os.path.join(d, "%s_seg%d.sam" % (sd, k + 1)) for k in range(nseg))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        _, recs = read_bam(bam)
        want = [tuple(l.rstrip("\n").split("\t")) for l in open(os.path.join(d, "expected.span_%s.sam" % sd))]
        assert [tuple(str(x) for x in rec) for rec in recs] == want
    # without unmapped reads the coverage search is skipped (segment_juncs.cpp:4978-4982): the segment search's set
    cmd2 = [c for c in cmd]
    i = cmd2.index("--ium-reads")
    del cmd2[i:i + 2]
    r = subprocess.run(cmd2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out["juncs"]).read() == open(os.path.join(d, "expected.seg_only.juncs")).read()


def test_two_shards_merged_on_the_device_equal_one():
    """thj_covsearch_device_state / thj_covsearch_merge_async: the left side's hits and reads in one context, the right
    side's in another, merged -> the same junctions as everything in one context"""
    c = load("pe50_cov")
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    n_left = sum(1 for _ in open(__import__("os").path.join(c["dir"], "left.fq"))) // 4
    ium = [c["ium"][:n_left], c["ium"][n_left:]]
    args = (c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"])
    with host.Context(0) as a, host.Context(0) as b:
        for ctx in (a, b):
            ctx.upload_genome(host.pack_genome(seqs))
            ctx.reset()
            ctx.covsearch_reset()
        base = 0
        for ctx, (side, sb), reads in zip((a, b), c["seg_batches"], ium):
            p = copy.copy(c["p"])
            p.read_side = side
            ctx.covsearch_add_hits(ctx.upload_batch(sb, ordinal_base=base))       # coverage only: no segment search here
            ctx.covsearch_add_reads(reads)
            base += sb.n_reads
        b.sync()
        bits, nw, sizes, keys, vals, n_ext = b.covsearch_device_state()
        a.covsearch_merge(bits, sizes, keys, vals, n_ext)
        a.covsearch_run(*args)
        found = a.covsearch_finish()
        got = _tuples(a.download(a.finish()).juncs)
    g = orc.Genome(seqs)
    want = _tuples(orc.coverage_search(g, c["hits"], c["ium"], *args))
    assert got == want and found == len(want)


def _device_coverage_only(seqs, hits, ium, args, max_cov_juncs=5000000):
    """coverage search alone on the device: the hits go up as a one-segment batch that is never run through the segment search"""
    from tophat_amd.batch import SegBatch
    n = len(hits)
    b = SegBatch(1, np.arange(1, n + 1, dtype=np.uint32), np.arange(0, n + 1, dtype=np.int64) * 10,
                 np.frombuffer(b"ACGTACGTAC" * max(1, n), dtype=np.uint8)[:10 * n].copy(), np.arange(0, n + 1, dtype=np.uint32), hits)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.reset()
        ctx.covsearch_reset()
        if n:
            ctx.covsearch_add_hits(ctx.upload_batch(b))
        ctx.covsearch_add_reads(ium)
        ctx.covsearch_run(*args)
        found = ctx.covsearch_finish(max_cov_juncs)
        return _tuples(ctx.download(ctx.finish()).juncs), found


@pytest.mark.parametrize("seed", range(400, 430))
def test_hip_coverage_search_edge_cases(seed):
    from cov_util import edge_case
    seqs, h, ium, args = edge_case(seed)
    folded = [orc.fold_genome_char(s) for s in seqs]
    want = _tuples(orc.coverage_search(orc.Genome(folded), h, ium, *args))
    got, found = _device_coverage_only(folded, h, ium, args)
    assert got == want and found == len(want)


def test_hip_coverage_search_empty_inputs_and_cap():
    from cov_util import edge_case
    seqs, h, ium, args = edge_case(404)
    folded = [orc.fold_genome_char(s) for s in seqs]
    assert _device_coverage_only(folded, h[:0], ium, args) == (set(), 0)          # no hits: no islands
    assert _device_coverage_only(folded, h, [], args) == (set(), 0)               # no unmapped reads: nothing is extendable
    assert _device_coverage_only(folded, h, ["ACGT", "A" * 9], args) == (set(), 0)   # reads too short for a 10-mer seed
    # the max_cov_juncs cut (segment_juncs.cpp:56, :1611-1621): the smallest by (skip count, junction), as the oracle makes it
    for name in CASES:
        c = load(name)
        seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
        args = (c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"])
        g = orc.Genome(seqs)
        for cap in (1, 5, 12):
            want = _tuples(orc.coverage_search(g, c["hits"], c["ium"], *args, max_juncs=cap))
            got, found = _device_coverage_only(seqs, c["hits"], c["ium"], args, max_cov_juncs=cap)
            assert got == want and found == cap == len(want)
