"""GPU: the drop-in executables (tophat_amd/bin/) on the fixture files, text-SAM and BAM inputs:
segment.juncs/.insertions/.deletions byte-identical to the expected files, spanning BAM record-identical
and its uncompressed byte stream identical to the reference's BAM (GBamRecord encoding, common.cpp:1000-1173)."""
import gzip
import os
import re
import subprocess

import pytest

from golden_util import CASES, GOLD
from tophat_amd.bamio import read_bam, write_bam_from_sam

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tophat_amd", "bin")


def _inputs(d, tmp_path, as_bam):
    def conv(p):
        if not as_bam:
            return p
        out = str(tmp_path / (os.path.basename(p)[:-4] + ".bam"))
        write_bam_from_sam(p, out)
        return out
    files = sorted(os.listdir(d))
    nseg = len([f for f in files if re.fullmatch(r"left_seg\d+\.sam", f)])
    paired = "right.fq" in files
    r = dict(left_segs=[conv(os.path.join(d, "left_seg%d.sam" % (k + 1))) for k in range(nseg)], left_map=conv(os.path.join(d, "left_map.sam")))
    if paired:
        r.update(right_segs=[conv(os.path.join(d, "right_seg%d.sam" % (k + 1))) for k in range(nseg)], right_map=conv(os.path.join(d, "right_map.sam")))
    return r, paired


@pytest.mark.parametrize("as_bam,batch_reads,threads", [(False, None, None), (True, None, None), (True, "37", "3"), (False, "1", "1")],
                         ids=["sam", "bam", "bam_batches_of_37", "sam_batches_of_1_single_thread"])
@pytest.mark.parametrize("name", CASES)
def test_dropin_binaries_reproduce_fixture(name, as_bam, batch_reads, threads, tmp_path):
    env = dict(os.environ)
    if batch_reads:
        env["THJ_BATCH_READS"] = batch_reads          # many batches: ordinals, writer hand-over, `.index` continuity
    if threads:
        env["THJ_HOST_THREADS"] = threads
    assert os.path.exists(os.path.join(BIN, "segment_juncs")), "run __graft_entry__.build() first"
    d = os.path.join(GOLD, name)
    opts = open(os.path.join(d, "options.txt")).read().split("\n")
    argv = opts[0].split()
    seglen = dict(x.split("=") for x in opts[1].split())["segment_length"]
    inp, paired = _inputs(d, tmp_path, as_bam)
    out = {k: str(tmp_path / ("out." + k)) for k in ("juncs", "insertions", "deletions", "fusions")}
    cmd = [os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", seglen,
           "--sam-header", os.path.join(d, "hdr.sam"), "-p", "1"] + argv + \
          [os.path.join(d, "ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"],
           os.path.join(d, "left.fq"), inp["left_map"], ",".join(inp["left_segs"])]
    if paired:
        cmd += [os.path.join(d, "right.fq"), inp["right_map"], ",".join(inp["right_segs"])]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in ("juncs", "insertions", "deletions"):
        assert open(out[k]).read() == open(os.path.join(d, "expected." + k)).read(), k
    if "--fusion-search" in argv:
        assert open(out["fusions"]).read() == open(os.path.join(d, "expected.fusions")).read()
        return
    for sd in (("left", "right") if paired else ("left",)):
        bam = str(tmp_path / ("span_%s.bam" % sd))
        cmd = [os.path.join(BIN, "long_spanning_reads"), "--segment-length", seglen, "--sam-header", os.path.join(d, "hdr.sam"),
               os.path.join(d, "ref.fa"), os.path.join(d, "%s.fq" % sd), out["juncs"], out["insertions"], out["deletions"], "/dev/null",
               bam, ",".join(inp["%s_segs" % sd])]
        sp = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.startswith("%s_seg" % sd) and f.endswith(".to_spliced.sam")]
        if sp:          # junction-db segment maps (SplicedBAMHitFactory path)
            if as_bam:
                conv = []
                for f in sp:
                    o_ = str(tmp_path / (os.path.basename(f)[:-4] + ".bam"))
                    write_bam_from_sam(f, o_)
                    conv.append(o_)
                sp = conv
            cmd.append(",".join(sp))
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        _, recs = read_bam(bam)
        want = [tuple(l.rstrip("\n").split("\t")) for l in open(os.path.join(d, "expected.span_%s.sam" % sd))]
        assert [tuple(str(x) for x in rec) for rec in recs] == want
        # byte-level: identical uncompressed BAM stream (header + every record's encoding)
        assert gzip.open(bam, "rb").read() == gzip.open(os.path.join(d, "expected.span_%s.bam" % sd), "rb").read()
        assert os.path.exists(bam + ".index")


def test_unsupported_modes_fail_loudly(tmp_path):
    d = os.path.join(GOLD, CASES[0])
    r = subprocess.run([os.path.join(BIN, "segment_juncs"), "--segment-length", "25", "--sam-header", os.path.join(d, "hdr.sam"),
                        os.path.join(d, "ref.fa"), "a", "b", "c", "d", os.path.join(d, "left.fq"), os.path.join(d, "left_map.sam"),
                        os.path.join(d, "left_seg1.sam")], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "not supported" in r.stderr      # coverage search is on by default for a bare binary run
    r = subprocess.run([os.path.join(BIN, "segment_juncs"), "--no-such-option"], capture_output=True, text=True)
    assert r.returncode == 1


def test_long_spanning_reads_parts(tmp_path):
    """-p N: <base>{0..N-1}.bam, each with its `.index` (long_spanning_reads.cpp:3056-3064); cut at read boundaries,
    concatenation == the single-file output"""
    name = "se100"
    d = os.path.join(GOLD, name)
    seglen = dict(x.split("=") for x in open(os.path.join(d, "options.txt")).read().split("\n")[1].split())["segment_length"]
    inp, _ = _inputs(d, tmp_path, False)
    out = str(tmp_path / "span.bam")
    cmd = [os.path.join(BIN, "long_spanning_reads"), "-p", "3", "--segment-length", seglen, "--sam-header", os.path.join(d, "hdr.sam"),
           os.path.join(d, "ref.fa"), os.path.join(d, "left.fq"), os.path.join(d, "expected.juncs"), os.path.join(d, "expected.insertions"),
           os.path.join(d, "expected.deletions"), "/dev/null", out, ",".join(inp["left_segs"])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got, names_seen = [], []
    for k in range(3):
        part = str(tmp_path / ("span%d.bam" % k))
        assert os.path.exists(part) and os.path.exists(part + ".index")
        _, recs = read_bam(part)
        recs = [tuple(str(x) for x in rec) for rec in recs]
        assert recs, "empty part"
        if names_seen:
            assert recs[0][0] != names_seen[-1]          # a read never straddles two parts
        names_seen.append(recs[-1][0])
        got += recs
    want = [tuple(l.rstrip("\n").split("\t")) for l in open(os.path.join(d, "expected.span_left.sam"))]
    assert got == want
