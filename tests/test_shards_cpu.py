"""CPU: read-id shards of the drop-in executables' ingest (tophat_amd/csrc/host/thj_hostio.h: load_index,
calculate_offsets, calculate_offsets_from_ids = utils.cpp:22-170; HitStream / ReadStream opened at an offset with an id
range).  The ingest loop of segment_juncs run shard after shard must see exactly the reads, hits and mate hits of the
unsharded loop, in the same order, whatever the number of shards -- on BAM inputs with their .index side files and on the
text twins (FASTQ + SAM, offsets found by probing)."""
import os
import subprocess

import pytest

from locked_make import locked_make

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(HERE, "hostio", "hostio_check")
GEN = os.path.join(ROOT, "tools", "bin", "thj_gen")


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    locked_make(os.path.join(HERE, "hostio"))
    if not os.path.exists(GEN):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    d = str(tmp_path_factory.mktemp("gen"))
    subprocess.check_call([GEN, "--out", d, "--pairs", "60000", "--genome-len", "3000000", "--introns", "1200", "--text", "--threads", "4"],
                          stdout=subprocess.DEVNULL)
    return d


def _run(d, n, side, mate, ext):
    reads = os.path.join(d, "%s_reads.bam" % side) if ext == "bam" else os.path.join(d, "%s.fq" % side)
    segs = ",".join(os.path.join(d, "%s_seg%d.%s" % (side, k, ext)) for k in (1, 2, 3, 4))
    out = subprocess.check_output([EXE, "shardmerge", str(n), reads, os.path.join(d, "%s_map.%s" % (mate, ext)), segs], text=True).split()
    return int(out[0]), out[1:]


@pytest.mark.parametrize("ext", ["bam", "sam"])
def test_sharded_ingest_equals_unsharded(case, ext):
    for side, mate in (("left", "right"), ("right", "left")):
        used1, whole = _run(case, 1, side, mate, ext)
        assert used1 == 1 and int(whole[0]) > 50000 and int(whole[2]) > 10000
        for n in (2, 3, 7, 16):
            used, got = _run(case, n, side, mate, ext)
            assert got == whole, (side, ext, n)
            assert used > 1, "the plan fell back to one shard (%s, %d asked)" % (ext, n)


def test_bam_and_text_twins_agree(case):
    assert _run(case, 4, "left", "right", "bam")[1] == _run(case, 4, "left", "right", "sam")[1]


def test_index_files_point_at_group_starts(case):
    """every .index line `read_id \\t voffset` must be the BGZF address of the first record of that read"""
    import struct
    import zlib
    p = os.path.join(case, "left_seg2.bam")
    data = open(p, "rb").read()
    lines = [l.split() for l in open(p + ".index")]
    assert len(lines) > 20
    for rid, voff in lines[:: max(1, len(lines) // 25)]:
        voff = int(voff)
        caddr, within = voff >> 16, voff & 0xFFFF
        raw = b""
        off = caddr
        while len(raw) < within + 200 and off < len(data):
            bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
            raw += zlib.decompress(data[off + 18:off + bsize - 8], -15)
            off += bsize
        l_rn = raw[within + 4 + 8]
        qname = raw[within + 4 + 32:within + 4 + 32 + l_rn - 1].decode()
        assert qname.split("|")[0] == rid
