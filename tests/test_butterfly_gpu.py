"""Butterfly search (--butterfly-search), GPU: thj_butterfly_run through the C ABI and the segment_juncs executable against the oracle."""
import copy
import os
import subprocess

import numpy as np
import pytest

import orc
from cov_util import CASES, GOLD, butterfly_case, juncs_text, load
from tophat_amd import host
from tophat_amd.batch import HIT_DTYPE, SegBatch

pytestmark = pytest.mark.gpu


def _tuples(a):
    return {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in a}


def _hits_batch(hits):
    """the hits as a batch of one-segment reads (one hit each): what thj_covsearch_add_hits_async needs of a batch is its hit array"""
    n = len(hits)
    return SegBatch(nseg=1, read_id=np.arange(1, n + 1, dtype=np.uint32), read_off=np.arange(0, 25 * (n + 1), 25, dtype=np.int64),
                    bases=np.full(25 * n, ord("A"), dtype=np.uint8), seg_off=np.arange(0, n + 1, dtype=np.uint32), hits=np.ascontiguousarray(hits, dtype=HIT_DTYPE))


def _device_butterfly(seqs, hits, ium, min_intron, max_intron, cap=5000000, with_coverage_search=None):
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        b = ctx.upload_batch(_hits_batch(hits))
        ctx.reset()
        ctx.covsearch_reset()
        ctx.covsearch_add_hits(b)
        ctx.covsearch_add_reads(ium)
        if with_coverage_search is not None:
            ctx.covsearch_run(with_coverage_search, min_intron, max_intron)
            ctx.covsearch_finish()
        found = ctx.butterfly_run(min_intron, max_intron, cap)
        ev = ctx.download(ctx.finish())
    return _tuples(ev.juncs), found


@pytest.mark.parametrize("name", CASES)
def test_hip_butterfly_search_reproduces_oracle_on_fixture(name):
    c = load(name)
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    g = orc.Genome(seqs)
    args = (c["cov"]["min_intron"], c["cov"]["max_intron"])
    want = _tuples(orc.butterfly_search(g, c["hits"], c["ium"], *args))
    got, found = _device_butterfly(seqs, c["hits"], c["ium"], *args)
    assert got == want and found == len(want) and len(want) > 0
    for cap in (7, 1):
        got, found = _device_butterfly(seqs, c["hits"], c["ium"], *args, cap)
        assert got == _tuples(orc.butterfly_search(g, c["hits"], c["ium"], *args, cap)) and found == cap
    # after a coverage search in the same pass (shared buffers): both searches' junctions
    cov = _tuples(orc.coverage_search(g, c["hits"], c["ium"], c["cov"]["min_cov_length"], *args))
    got, found = _device_butterfly(seqs, c["hits"], c["ium"], *args, with_coverage_search=c["cov"]["min_cov_length"])
    assert got == want | cov and found == len(want)


@pytest.mark.parametrize("seed", range(500, 500 + int(os.environ.get("THJ_BF_SEEDS", "40"))))
def test_hip_butterfly_search_matches_oracle(seed):
    seqs, h, ium, args = butterfly_case(seed)
    folded = [orc.fold_genome_char(s) for s in seqs]
    g = orc.Genome(folded)
    for cap in (5000000, 2):
        want = _tuples(orc.butterfly_search(g, h, ium, *args, cap))
        got, found = _device_butterfly(folded, h, ium, *args, cap)
        assert got == want and found == len(want), cap


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("coverage", [False, True])
def test_segment_juncs_executable_with_butterfly_search(name, coverage, tmp_path):
    """--butterfly-search with and without the coverage search beside it: the junction file = segment search + the searches asked for"""
    c = load(name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(GOLD, name)
    opts = open(os.path.join(d, "options.txt")).read().split("\n")
    argv = opts[0].split()
    kv = dict(x.split("=") for x in opts[1].split())
    sides = ("left", "right") if kv["paired"] == "1" else ("left",)
    nseg = len([f for f in os.listdir(d) if f.startswith("left_seg")])
    out = {k: str(tmp_path / ("out." + k)) for k in ("juncs", "insertions", "deletions", "fusions")}
    cmd = [os.path.join(root, "tophat_amd", "bin", "segment_juncs"), "--butterfly-search", "--no-microexon-search"] + ([] if coverage else ["--no-coverage-search"]) + \
          ["--segment-length", kv["segment_length"], "--sam-header", os.path.join(d, "hdr.sam")] + argv + \
          ["--ium-reads", ",".join(os.path.join(d, "%s.fq" % sd) for sd in sides), os.path.join(d, "ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"]]
    for sd in sides:
        cmd += [os.path.join(d, "%s.fq" % sd), os.path.join(d, "%s_map.sam" % sd), ",".join(os.path.join(d, "%s_seg%d.sam" % (sd, k + 1)) for k in range(nseg))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Performing butterfly-search" in r.stderr and ("Performing coverage-search" in r.stderr) == coverage
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    g = orc.Genome(seqs)
    ids = {n: i + 1 for i, n in enumerate(c["names"])}
    base = open(os.path.join(d, "expected.juncs" if coverage else "expected.seg_only.juncs")).read()
    want = {(ids[t[0]], int(t[1]), int(t[2]), 1 if t[3][0] == "-" else 0) for t in (l.split("\t") for l in base.splitlines())}
    want |= _tuples(orc.butterfly_search(g, c["hits"], c["ium"], c["cov"]["min_intron"], c["cov"]["max_intron"]))
    assert open(out["juncs"]).read() == juncs_text(want, c["names"])
    # without unmapped reads neither search runs (segment_juncs.cpp:4978-4982)
    i = cmd.index("--ium-reads")
    r = subprocess.run(cmd[:i] + cmd[i + 2:], capture_output=True, text=True)
    assert r.returncode == 0 and "butterfly-search" not in r.stderr
    assert open(out["juncs"]).read() == open(os.path.join(d, "expected.seg_only.juncs")).read()


def test_unmapped_reads_from_bam_on_the_device_equal_the_host_stream(tmp_path):
    """--ium-reads as unaligned BAM files (what tophat.py passes): the device-side ingest of those files (thj_covsearch_add_reads_bam) gives the
    coverage and butterfly searches the same table as the host's ReadStream does"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = os.path.join(root, "tools", "bin", "thj_gen")
    d = str(tmp_path)
    subprocess.check_call([gen, "--out", d, "--pairs", "60000", "--read-len", "50", "--genome-len", "4000000", "--introns", "1500"], stdout=subprocess.DEVNULL)
    f = lambda n: os.path.join(d, n)      # noqa: E731
    segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k)) for k in (1, 2)) for sd in ("left", "right")}
    outs = {}
    for mode, env in (("device", {}), ("host", {"THJ_HOST_INGEST": "1"})):
        out = {k: f("%s.%s" % (mode, k)) for k in ("juncs", "insertions", "deletions", "fusions")}
        r = subprocess.run([os.path.join(root, "tophat_amd", "bin", "segment_juncs"), "--butterfly-search", "--no-microexon-search", "--segment-length", "25", "--sam-header", f("hdr.sam"),
                            "--inner-dist-mean", "50", "--inner-dist-std-dev", "20", "--ium-reads", f("left_reads.bam") + "," + f("right_reads.bam"),
                            f("ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"], f("left_reads.bam"), f("left_map.bam"), segs["left"],
                            f("right_reads.bam"), f("right_map.bam"), segs["right"]], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Performing coverage-search" in r.stderr and "Performing butterfly-search" in r.stderr
        outs[mode] = open(out["juncs"]).read()
    assert outs["device"] == outs["host"] and outs["device"].count("\n") > 100


def _rechunk_bgzf(src, dst, chunk):
    """the BAM stream of `src` cut into BGZF members of `chunk` inflated bytes, wherever that falls: records straddle members (a writer
    other than samtools' bam_write1 may do that)"""
    import gzip
    import struct
    import zlib
    raw = gzip.open(src, "rb").read()
    with open(dst, "wb") as f:
        for a in list(range(0, len(raw), chunk)) + [None]:
            piece = b"" if a is None else raw[a:a + chunk]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = co.compress(piece) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(piece), len(piece)))


def test_unmapped_reads_bam_with_straddling_records_takes_the_host_stream(tmp_path):
    """a piece the device-side ingest declines (records across member borders) is read by the host stream, past the records the device took"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = os.path.join(root, "tools", "bin", "thj_gen")
    d = str(tmp_path)
    subprocess.check_call([gen, "--out", d, "--pairs", "20000", "--read-len", "50", "--genome-len", "4000000", "--introns", "1500"], stdout=subprocess.DEVNULL)
    f = lambda n: os.path.join(d, n)      # noqa: E731
    _rechunk_bgzf(f("left_reads.bam"), f("left_ium.bam"), 50021)
    segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k)) for k in (1, 2)) for sd in ("left", "right")}
    outs = {}
    for mode, ium in (("straddling", f("left_ium.bam") + "," + f("right_reads.bam")), ("plain", f("left_reads.bam") + "," + f("right_reads.bam"))):
        out = {k: f("%s.%s" % (mode, k)) for k in ("juncs", "insertions", "deletions", "fusions")}
        r = subprocess.run([os.path.join(root, "tophat_amd", "bin", "segment_juncs"), "--butterfly-search", "--no-microexon-search", "--segment-length", "25", "--sam-header", f("hdr.sam"),
                            "--inner-dist-mean", "50", "--inner-dist-std-dev", "20", "--ium-reads", ium,
                            f("ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"], f("left_reads.bam"), f("left_map.bam"), segs["left"],
                            f("right_reads.bam"), f("right_map.bam"), segs["right"]], capture_output=True, text=True, env=dict(os.environ, THJ_TIMING="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = open(out["juncs"]).read()
    assert outs["straddling"] == outs["plain"] and outs["plain"].count("\n") > 50
