"""GPU: the BAM writer's device side through the C ABI -- thj_bgzf_deflate's members against the CPU build of the same source
(tests/hostsim/bamout_sim.cpp: byte for byte) and against zlib (they inflate to the member, CRC-32 as zlib computes it).
thj_span_bam_encode is covered where its input comes from: tests/test_gpu_binaries.py runs long_spanning_reads with the device
writer against the host writer."""
import ctypes as C
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from locked_make import locked_make  # noqa: E402
from test_bamout_sim_cpu import bam_like_stream  # noqa: E402

from tophat_amd.host import Context, ThjError, bgzf_plan_cuts  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sim():
    d = os.path.join(HERE, "hostsim")
    locked_make(d)
    l = C.CDLL(os.path.join(d, "libbamoutsim.so"))
    l.deflate_sim_member.restype = C.c_int
    return l


def sim_deflate(sim, data):
    out = np.zeros(65536, np.uint8)
    res = np.zeros(3, np.uint32)
    assert sim.deflate_sim_member(C.c_char_p(data), C.c_uint32(len(data)), C.c_void_p(out.ctypes.data), C.c_void_p(res.ctypes.data)) == 0
    return out[:int(res[0])].tobytes(), int(res[1]), int(res[2])


def test_members_match_the_cpu_build_and_inflate(sim):
    rng = np.random.default_rng(21)
    recs = bam_like_stream(rng, 2500)
    pieces = [b"".join(recs)]
    text = (b"long_spanning_reads writes BAM records; " * 2000)[:65536]
    members = [b"x", b"ab", bytes(65536), text, text[:4097], rng.integers(0, 4, 30000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(),
               bytes(range(256)) * 7]
    for n in (1, 3, 63, 64, 65, 1023, 1025, 4096, 65535):
        members.append(rng.integers(65, 70, n, dtype=np.uint8).tobytes())
    ends = bgzf_plan_cuts([len(r) for r in recs])
    stream = pieces[0]
    for m in members:
        stream += m
        ends.append(len(stream))
    ctx = Context()
    ctx.bam_stream_upload(stream)
    assert ctx.bam_stream_download(len(stream)) == stream
    comp, crcs = ctx.bgzf_deflate(ends)
    assert len(comp) == len(ends) > 15
    tot_raw = tot_comp = 0
    for k, (a, b) in enumerate(zip([0] + ends, ends)):
        d = stream[a:b]
        assert crcs[k] == (zlib.crc32(d) & 0xFFFFFFFF)
        z = zlib.decompressobj(-15)
        assert z.decompress(comp[k]) == d and z.eof and z.unused_data == b""
        want, wcrc, st = sim_deflate(sim, d)
        assert st == 0 and comp[k] == want, "member %d differs from the CPU build" % k
        if k < len(ends) - len(members):
            tot_raw += len(d); tot_comp += len(comp[k])
    assert tot_comp < 0.30 * tot_raw             # BAM records: about zlib level 1's ratio


def test_many_small_members_take_several_launches(sim):
    rng = np.random.default_rng(22)
    recs = bam_like_stream(rng, 2300)
    stream = b"".join(recs)
    ends = list(np.cumsum([len(r) for r in recs]))            # one record per member: 2300 members > one launch's 1024
    ctx = Context()
    ctx.bam_stream_upload(stream)
    comp, crcs = ctx.bgzf_deflate([int(e) for e in ends])
    for k, (a, b) in enumerate(zip([0] + ends, ends)):
        assert zlib.decompress(comp[k], -15) == stream[a:b] and crcs[k] == (zlib.crc32(stream[a:b]) & 0xFFFFFFFF)
    for k in (0, 1023, 1024, 2299):
        assert comp[k] == sim_deflate(sim, stream[(ends[k - 1] if k else 0):ends[k]])[0]


def test_bad_members_are_refused():
    ctx = Context()
    rng = np.random.default_rng(23)
    noise = rng.integers(0, 256, 65536, dtype=np.uint8).tobytes()
    ctx.bam_stream_upload(noise + b"abc")
    with pytest.raises(ThjError, match="does not fit a BGZF block"):      # THJ_EFALLBACK: the caller replays on the host, where bgzf.c's shrink rule lives
        ctx.bgzf_deflate([65536, 65539])
    with pytest.raises(ThjError, match="larger than 64 KiB"):
        ctx.bgzf_deflate([65539])
    with pytest.raises(ThjError, match="empty"):
        ctx.bgzf_deflate([10, 10])
    with pytest.raises(ThjError, match="outside the encoded stream"):
        ctx.bgzf_deflate([10, 70000])
    comp, _ = ctx.bgzf_deflate([60000, 65539])
    assert zlib.decompress(comp[0], -15) == noise[:60000] and zlib.decompress(comp[1], -15) == noise[60000:] + b"abc"
