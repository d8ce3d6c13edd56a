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
    # a bare run has the microexon search on (common.cpp:138; the coverage search too when --ium-reads names reads): built, the program runs
    assert r.returncode == 0 and "Performing microexon-search" in r.stderr
    r = subprocess.run([os.path.join(BIN, "segment_juncs"), "--butterfly-search", "--segment-length", "25", "--sam-header", os.path.join(d, "hdr.sam"),
                        os.path.join(d, "ref.fa"), "a", "b", "c", "d", os.path.join(d, "left.fq"), os.path.join(d, "left_map.sam"),
                        os.path.join(d, "left_seg1.sam")], capture_output=True, text=True, cwd=str(tmp_path))
    # built since round 4; without --ium-reads neither it nor the coverage search runs (segment_juncs.cpp:4978-4982)
    assert r.returncode == 0 and "butterfly-search" not in r.stderr
    r = subprocess.run([os.path.join(BIN, "segment_juncs"), "--color", "--segment-length", "25", "--sam-header", os.path.join(d, "hdr.sam"),
                        os.path.join(d, "ref.fa"), "a", "b", "c", "d", os.path.join(d, "left.fq"), os.path.join(d, "left_map.sam"),
                        os.path.join(d, "left_seg1.sam")], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "not supported" in r.stderr      # colour space: the mode that is not built
    r = subprocess.run([os.path.join(BIN, "segment_juncs"), "--no-such-option"], capture_output=True, text=True)
    assert r.returncode == 1


def test_long_spanning_reads_parts_fall_back_to_one_file_on_small_inputs(tmp_path):
    """-p N on inputs whose indexes are too small for N ranges: one thread, one file (long_spanning_reads.cpp:2991-2993,
    utils.cpp:75-80) -- which tophat.py looks for first (tophat.py:3772-3779)"""
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
    assert os.path.exists(out) and os.path.exists(out + ".index") and not os.path.exists(str(tmp_path / "span0.bam"))
    _, recs = read_bam(out)
    want = [tuple(l.rstrip("\n").split("\t")) for l in open(os.path.join(d, "expected.span_left.sam"))]
    assert [tuple(str(x) for x in rec) for rec in recs] == want


def index_positions(bam):
    """[(read id, position in the inflated BAM stream)] of the `.index` lines: `read_id \\t virtual offset` with the BGZF member's file
    offset in the high bits (GBamWriter::write, common.h:562-606).  Two files with the same stream and the same positions index
    the same records, wherever their members are cut."""
    data = open(bam, "rb").read()
    at_of, off, pos = {}, 0, 0
    while off < len(data):
        bsize = int.from_bytes(data[off + 16:off + 18], "little") + 1
        at_of[off] = pos
        pos += int.from_bytes(data[off + bsize - 4:off + bsize], "little")
        off += bsize
    out = []
    for line in open(bam + ".index"):
        rid, voff = (int(x) for x in line.split())
        out.append((rid, at_of[voff >> 16] + (voff & 0xFFFF)))
    return out


def _gen_case(tmp_path, pairs=60000):
    d = str(tmp_path / "gen")
    if os.path.exists(os.path.join(d, "ref.fa")):
        return d
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", str(pairs), "--genome-len", "3000000",
                           "--introns", "1200", "--threads", "8"], stdout=subprocess.DEVNULL)
    return d


def _run_both(d, tmp_path, tag, env_extra, p_arg=()):
    """segment_juncs, then long_spanning_reads on the left side -> (text outputs, path of the spanning BAM)"""
    env = dict(os.environ, **env_extra)
    out = {k: str(tmp_path / ("%s.%s" % (tag, k))) for k in ("juncs", "insertions", "deletions", "fusions")}
    segs = {sd: ",".join(os.path.join(d, "%s_seg%d.bam" % (sd, k)) for k in (1, 2, 3, 4)) for sd in ("left", "right")}
    cmd = [os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", "25", "--sam-header",
           os.path.join(d, "hdr.sam"), "--inner-dist-mean", "50", "--inner-dist-std-dev", "20", os.path.join(d, "ref.fa"), out["juncs"],
           out["insertions"], out["deletions"], out["fusions"], os.path.join(d, "left_reads.bam"), os.path.join(d, "left_map.bam"), segs["left"],
           os.path.join(d, "right_reads.bam"), os.path.join(d, "right_map.bam"), segs["right"]]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    bam = str(tmp_path / ("%s.span.bam" % tag))
    cmd = [os.path.join(BIN, "long_spanning_reads")] + list(p_arg) + ["--segment-length", "25", "--sam-header", os.path.join(d, "hdr.sam"),
           os.path.join(d, "ref.fa"), os.path.join(d, "left_reads.bam"), out["juncs"], out["insertions"], out["deletions"], "/dev/null", bam, segs["left"]]
    r2 = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    return {k: open(v).read() for k, v in out.items()}, bam, r.stderr + r2.stderr


def test_results_do_not_depend_on_shards_or_workers(tmp_path):
    """the same files through one shard / one worker and through many: identical text outputs, identical BAM stream and .index"""
    d = _gen_case(tmp_path)
    one, bam1, log1 = _run_both(d, tmp_path, "one", {"THJ_SHARDS": "1", "THJ_WORKERS": "1"})
    many, bam2, log2 = _run_both(d, tmp_path, "many", {"THJ_SHARDS": "13", "THJ_WORKERS": "5", "THJ_BATCH_READS": "3000"})
    assert "13 left + 13 right read-id shards" in log2 and "1 left + 1 right read-id shards" in log1
    assert one == many and one["juncs"].count("\n") > 500
    assert gzip.open(bam1, "rb").read() == gzip.open(bam2, "rb").read()
    # (the BAM members are cut at shard ends when they are made on the device, so the virtual offsets differ; what they point at does not)
    assert index_positions(bam1) == index_positions(bam2) and len(index_positions(bam1)) > 20


def test_eight_contexts_with_the_exchange_step_equal_one(tmp_path):
    """configs[2]'s rehearsal in small: eight ranks as eight contexts on the device -- the 8-way shard dealing, the exchange
    step over the loopback transport with eight sections -- against one context: identical event files, identical BAM stream.  With event tables
    made small (THJ_TABLE_CAPS) every rank's own deletions fit its table and the eight sections together do not: the table fills up while
    the gathered keys are merged, is reported full (not walked slot by slot: round 6), grows to what the headers ask for and the merge is repeated."""
    d = str(tmp_path / "gen8")
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", "60000", "--genome-len", "3000000", "--introns", "1200",
                           "--threads", "8", "--indel-frac", "0.03"], stdout=subprocess.DEVNULL)
    one, bam1, log1 = _run_both(d, tmp_path, "c1", {"THJ_CTX_PER_GPU": "1", "THJ_TIMING": "1"})
    eight, bam8, log8 = _run_both(d, tmp_path, "c8", {"THJ_CTX_PER_GPU": "8", "THJ_SHARDS": "16", "THJ_TIMING": "1"})
    small, bam8s, log8s = _run_both(d, tmp_path, "c8s", {"THJ_CTX_PER_GPU": "8", "THJ_SHARDS": "16", "THJ_TABLE_CAPS": "4096,512", "THJ_TIMING": "1"})
    n_del = one["deletions"].count("\n")
    assert n_del > 600 and one["juncs"].count("\n") > 500
    assert one == eight and one == small
    assert gzip.open(bam1, "rb").read() == gzip.open(bam8, "rb").read() == gzip.open(bam8s, "rb").read()
    assert "a table filled up while the ranks' keys were merged" in log8s and "a table filled up" not in log8


def test_packed_genome_cache(tmp_path):
    """segment_juncs packs the reference and leaves the blocks beside its outputs; long_spanning_reads -- and a second segment_juncs --
    map that file instead of parsing the FASTA: the same results as with the cache off; a FASTA that has changed since (other
    modification time) is parsed again and the cache rewritten"""
    import glob
    import shutil
    d0 = _gen_case(tmp_path)
    d = str(tmp_path / "own")
    shutil.copytree(d0, d)                       # (its own copy: the FASTA's modification time is changed below)
    off, bam_off, log_off = _run_both(d, tmp_path, "off", {"THJ_GENOME_CACHE": "0", "THJ_TIMING": "1"})
    assert not glob.glob(str(tmp_path / ".thj2bit.*")) and "packed-genome cache" not in log_off
    on, bam_on, log_on = _run_both(d, tmp_path, "on", {"THJ_TIMING": "1"})
    caches = glob.glob(str(tmp_path / ".thj2bit.*"))
    assert len(caches) == 1 and os.path.getsize(caches[0]) > 3000000 // 4
    assert log_on.count("reference taken from the packed-genome cache") == 1          # long_spanning_reads took it, segment_juncs wrote it
    again, bam_again, log_again = _run_both(d, tmp_path, "again", {"THJ_TIMING": "1"})
    assert log_again.count("reference taken from the packed-genome cache") == 2
    assert off == on == again and off["juncs"].count("\n") > 500
    assert gzip.open(bam_off, "rb").read() == gzip.open(bam_on, "rb").read() == gzip.open(bam_again, "rb").read()
    t = os.path.getmtime(caches[0])
    os.utime(os.path.join(d, "ref.fa"), (1000000000, 1000000000))
    stale, bam_stale, log_stale = _run_both(d, tmp_path, "stale", {"THJ_TIMING": "1"})
    assert log_stale.count("reference taken from the packed-genome cache") == 1 and stale == off      # parsed once, written anew, taken by the second program
    assert os.path.getmtime(caches[0]) >= t and len(glob.glob(str(tmp_path / ".thj2bit.*"))) == 1


def test_process_and_compressor_choices_do_not_change_the_results(tmp_path):
    """the output hand-off to a child process (THJ_HANDOFF=1; one process is the default), zlib instead of the writer's own DEFLATE (THJ_BGZF_LEVEL), no
    page-locked staging (THJ_NO_STAGING): the same event files and the same BAM stream"""
    d = _gen_case(tmp_path)
    ref, bam0, _ = _run_both(d, tmp_path, "dflt", {})
    stream = gzip.open(bam0, "rb").read()
    assert len(stream) > 1000000
    for tag, env in (("handoff", {"THJ_HANDOFF": "1"}), ("zlib", {"THJ_BGZF_LEVEL": "6"}), ("nostage", {"THJ_NO_STAGING": "1", "THJ_CTX_PER_GPU": "1"})):
        got, bam, _ = _run_both(d, tmp_path, tag, env)
        assert got == ref, tag
        assert gzip.open(bam, "rb").read() == stream, tag
    # the own compressor's file is about the size zlib's level 1 would give (and smaller than twice level 6's)
    assert os.path.getsize(bam0) < 2 * os.path.getsize(str(tmp_path / "zlib.span.bam"))


def test_device_bam_output_equals_the_host_writers(tmp_path):
    """long_spanning_reads building its BAM records and deflating its BGZF members on the device (thj_span_bam_encode, thj_bgzf_deflate:
    the default when reads and maps are BAM) against the host encoder + compressor (THJ_HOST_BAM=1): the same BAM stream, `.index` lines
    that point at the same records, every member a valid BGZF block (gzip reads the file), about the same size"""
    d = _gen_case(tmp_path, pairs=80000)
    dev, bam_dev, log_dev = _run_both(d, tmp_path, "dev", {"THJ_SHARDS": "5"})
    hst, bam_hst, log_hst = _run_both(d, tmp_path, "hst", {"THJ_SHARDS": "5", "THJ_HOST_BAM": "1"})
    one, bam_one, log_one = _run_both(d, tmp_path, "one", {"THJ_SHARDS": "1"})
    assert "made on the device for 5 shards, on the host for 0" in log_dev and "made on the device" not in log_hst
    assert "made on the device for 1 shard, on the host for 0" in log_one
    ref = gzip.open(bam_hst, "rb").read()
    assert len(ref) > 1000000
    assert gzip.open(bam_dev, "rb").read() == ref and gzip.open(bam_one, "rb").read() == ref
    assert index_positions(bam_dev) == index_positions(bam_hst) == index_positions(bam_one)
    assert os.path.getsize(bam_dev) < 1.30 * os.path.getsize(bam_hst)
    # every member's CRC-32 and ISIZE are right (gzip checks them per member), and the file ends with the BGZF EOF marker
    assert open(bam_dev, "rb").read()[-28:] == open(bam_hst, "rb").read()[-28:]
    # the runs are reproducible: the device deflater's output does not depend on scheduling
    again, bam_again, _ = _run_both(d, tmp_path, "again", {"THJ_SHARDS": "5"})
    assert open(bam_again, "rb").read() == open(bam_dev, "rb").read()


def test_device_ingest_equals_host_ingest(tmp_path):
    """segment_juncs reading its BAM inputs on the device (BGZF inflate + record parse + merge by read id in HBM) against the
    host readers: identical event files -- with one shard and with many, paired-end with mate maps"""
    d = _gen_case(tmp_path, pairs=80000)
    dev, bam_dev, log_dev = _run_both(d, tmp_path, "dev", {"THJ_SHARDS": "7", "THJ_WORKERS": "3"})
    hst, bam_hst, log_hst = _run_both(d, tmp_path, "hst", {"THJ_SHARDS": "7", "THJ_WORKERS": "3", "THJ_HOST_INGEST": "1"})
    one, bam_one, _ = _run_both(d, tmp_path, "one", {"THJ_SHARDS": "1", "THJ_WORKERS": "1"})
    assert "reading on the host" not in log_dev
    assert dev == hst == one and dev["juncs"].count("\n") > 500
    # long_spanning_reads: segment maps parsed on the device, reads on the host -> the same BAM stream and .index
    ref = gzip.open(bam_hst, "rb").read()
    assert gzip.open(bam_dev, "rb").read() == ref and gzip.open(bam_one, "rb").read() == ref and len(ref) > 1000000
    assert index_positions(bam_dev) == index_positions(bam_hst)


def test_long_spanning_reads_parts(tmp_path):
    """-p N: <base>{0..N-1}.bam, each with its `.index` (long_spanning_reads.cpp:3056-3064), cut where calculate_offsets cuts;
    their concatenation == the single-file output"""
    d = _gen_case(tmp_path)
    _, bam1, _ = _run_both(d, tmp_path, "one", {})
    _, bam3, _ = _run_both(d, tmp_path, "three", {}, p_arg=("-p", "3"))
    _, whole = read_bam(bam1)
    whole = list(whole)
    got, last = [], None
    for k in range(3):
        part = bam3[:-4] + "%d.bam" % k
        assert os.path.exists(part) and os.path.exists(part + ".index") and not os.path.exists(bam3)
        _, recs = read_bam(part)
        recs = list(recs)
        assert len(recs) > len(whole) // 6, "a part is far smaller than a third"
        if last is not None:
            assert recs[0][0] != last                    # a read never straddles two parts
        last = recs[-1][0]
        got += recs
    assert got == whole


from golden_util import FUSION_SPAN_CASES  # noqa: E402


@pytest.mark.parametrize("threads", [None, "1"], ids=["default_threads", "single_thread"])
@pytest.mark.parametrize("name", FUSION_SPAN_CASES)
def test_long_spanning_reads_fusion_search(name, threads, tmp_path):
    """long_spanning_reads --fusion-search as a drop-in: the .fusions list, fused segment hits from the fusion contigs of the
    junction database (BAM maps: the SAM hit factory of the reference drops them), two-record XF output -- every record and the
    uncompressed BAM stream identical to what the scratch build wrote"""
    env = dict(os.environ)
    if threads:
        env["THJ_HOST_THREADS"] = threads
    d = os.path.join(GOLD, name)
    opts = open(os.path.join(d, "options.txt")).read().split("\n")
    argv = opts[0].split()
    seglen = dict(x.split("=") for x in opts[1].split())["segment_length"]
    nseg = len([f for f in os.listdir(d) if f.startswith("left_seg") and f.endswith(".to_spliced.sam")])
    n_xf = 0
    for sd in ("left", "right"):
        sp = []
        for k in range(nseg):
            o_ = str(tmp_path / ("%s_seg%d.to_spliced.bam" % (sd, k + 1)))
            write_bam_from_sam(os.path.join(d, "%s_seg%d.to_spliced.sam" % (sd, k + 1)), o_)
            sp.append(o_)
        bam = str(tmp_path / ("span_%s.bam" % sd))
        cmd = [os.path.join(BIN, "long_spanning_reads"), "--segment-length", seglen, "--sam-header", os.path.join(d, "hdr.sam")] + argv + [
            os.path.join(d, "ref.fa"), os.path.join(d, "%s.fq" % sd), os.path.join(d, "expected.juncs"), os.path.join(d, "expected.insertions"),
            os.path.join(d, "expected.deletions"), os.path.join(d, "expected.fusions"), bam,
            ",".join(os.path.join(d, "%s_seg%d.sam" % (sd, k + 1)) for k in range(nseg)), ",".join(sp)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        _, recs = read_bam(bam)
        want = [tuple(l.rstrip("\n").split("\t")) for l in open(os.path.join(d, "expected.span_%s.sam" % sd))]
        assert [tuple(str(x) for x in rec) for rec in recs] == want
        n_xf += sum(1 for rec in recs if any(str(x).startswith("XF:Z:") for x in rec))
        assert gzip.open(bam, "rb").read() == gzip.open(os.path.join(d, "expected.span_%s.bam" % sd), "rb").read()
    assert n_xf >= 40


def test_repeated_runs_are_identical_on_the_mix(tmp_path):
    """SURVEY 8(d)'s mix through both executables (a repeat family of up to 41 copies whose mates carry up to 82 hits, deletion reads:
    every kernel of stage 1's pipeline and the packed multihit tier have work) six times over: the same event files and the same bytes
    in the spanning BAM every time -- nothing depends on how the streams, the waves drawing from lists or the workers interleave"""
    d = str(tmp_path / "mix")
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", "60000", "--genome-len", "52000000", "--introns", "16000",
                           "--multihit-frac", "0.08", "--max-copies", "41", "--indel-frac", "0.03", "--threads", "8"], stdout=subprocess.DEVNULL)
    ref, bam0, _ = _run_both(d, tmp_path, "r0", {"THJ_SHARDS": "7"})
    first = open(bam0, "rb").read()
    assert ref["deletions"].count("\n") > 200 and ref["juncs"].count("\n") > 500 and len(first) > 1000000
    for k in range(1, 6):
        got, bam, _ = _run_both(d, tmp_path, "r%d" % k, {"THJ_SHARDS": "7"})
        assert got == ref, k
        assert open(bam, "rb").read() == first, k


def test_reads_of_ten_segments_through_both_executables(tmp_path):
    """2 x 250 bp at --segment-length 25 (ten segments, four plane words: tophat.py:3486-3492 cuts reads that way): the device-side ingest
    hands such shards to the host readers, the kernels take them -- event files byte for byte and every spanning record as the oracle has them"""
    import pathlib
    import orc
    from golden_util import events_text
    from tophat_amd.bamio import read_bam
    from tophat_amd.batch import build_seg_batch, build_span_batch, events_to_span_inputs, merge_events
    from tophat_amd.params import Params
    from tophat_amd.samtext import parse_header, parse_sam_hits, read_fasta, read_fastq
    d = str(tmp_path / "long")
    nseg = 10
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", "6000", "--genome-len", "4000000", "--introns", "1200",
                           "--read-len", "250", "--exon-len", "450", "--text", "--threads", "8"], stdout=subprocess.DEVNULL)
    f = lambda n: os.path.join(d, n)      # noqa: E731
    segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k + 1)) for k in range(nseg)) for sd in ("left", "right")}
    out = {k: f("out." + k) for k in ("juncs", "insertions", "deletions", "fusions")}
    r = subprocess.run([os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", "25", "--sam-header",
                        f("hdr.sam"), "--inner-dist-mean", "50", "--inner-dist-std-dev", "20", f("ref.fa"), out["juncs"], out["insertions"],
                        out["deletions"], out["fusions"], f("left_reads.bam"), f("left_map.bam"), segs["left"], f("right_reads.bam"),
                        f("right_map.bam"), segs["right"]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names, _ = parse_header(f("hdr.sam"))
    _, fa_seqs = read_fasta(f("ref.fa"))
    seqs = [orc.fold_genome_char(s) for s in fa_seqs]
    ref_ids = {n: i + 1 for i, n in enumerate(names)}
    og = orc.Genome(seqs)
    sides = {}
    for sd in ("left", "right"):
        sides[sd] = dict(reads=read_fastq(f("%s.fq" % sd)),
                         segs=[list(parse_sam_hits(f("%s_seg%d.sam" % (sd, k + 1)), ref_ids, 500000)) for k in range(nseg)],
                         full=list(parse_sam_hits(f("%s_map.sam" % sd), ref_ids, 500000)))
    want = None
    for sd, side, other in (("left", 1, "right"), ("right", 2, "left")):
        b = build_seg_batch(sides[sd]["segs"], sides[sd]["reads"], sides[other]["full"], sides[other]["segs"][-1])
        assert b.nseg == nseg
        e = orc.segjuncs(Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20), og, b)
        want = e if want is None else merge_events(want, e)
    assert len(want.juncs) > 300
    wt = events_text(want, names, pathlib.Path(d))
    for k in ("juncs", "insertions", "deletions"):
        assert open(out[k]).read() == wt[k], k
    jj, ii = events_to_span_inputs(want)
    n_rec = 0
    for sd in ("left", "right"):
        bam = f("span_%s.bam" % sd)
        r = subprocess.run([os.path.join(BIN, "long_spanning_reads"), "--segment-length", "25", "--sam-header", f("hdr.sam"), f("ref.fa"),
                            f("%s_reads.bam" % sd), out["juncs"], out["insertions"], out["deletions"], "/dev/null", bam, segs[sd]], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        quals = {k: "I" * len(v) for k, v in sides[sd]["reads"].items()}
        sb = build_span_batch(sides[sd]["segs"], sides[sd]["reads"], quals)
        alns = orc.spanning(Params(), og, sb, jj, ii)
        wrecs = [tuple(str(x) for x in a.sam_fields(int(sb.read_id[a.read_idx]), names)) for a in alns]
        _, recs = read_bam(bam)
        grecs = [tuple(str(x) for x in (rr[0], rr[1], rr[2], rr[3], rr[5]) + tuple(rr[8:])) for rr in recs]
        n_rec += len(grecs)
        assert grecs == wrecs, sd
    assert n_rec > 3000
