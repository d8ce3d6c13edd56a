"""GPU: device-side ingest.  thj_bgzf_inflate against zlib: BGZF members of real BAM files written
by the host I/O layer, and hand-made DEFLATE streams of every block type (stored, fixed, dynamic), compression level and
edge (empty input, one byte, 64 KiB of one byte, incompressible data, long-distance matches, codes longer than the direct
lookup tables), plus corrupt input."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from tophat_amd import host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Block(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint32), ("reserved", C.c_uint32)]


def inflate(ctx, payloads):
    comp = b"".join(payloads)
    n = len(payloads)
    blocks = (Block * n)()
    off = 0
    for k, p in enumerate(payloads):
        blocks[k].in_off, blocks[k].in_len = off, len(p)
        off += len(p)
    out = np.zeros(n << 16, dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    buf = np.frombuffer(comp + b"\0", dtype=np.uint8)
    rc = ctx.lib.thj_bgzf_inflate(ctx._ctx, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(comp)), blocks, C.c_int64(n),
                                  out.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), 0)
    assert rc == 0, ctx.lib.thj_last_error()
    return [None if lens[k] == 0xFFFFFFFF else out[k << 16:(k << 16) + int(lens[k])].tobytes() for k in range(n)]


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def test_handmade_streams():
    rng = np.random.default_rng(1)
    datas = [b"", b"A", b"A" * 65536, bytes(rng.integers(0, 256, size=65536, dtype=np.uint8)),
             bytes(rng.integers(0, 4, size=65536, dtype=np.uint8)), (b"ACGTTGCA" * 9000)[:65536],
             bytes(rng.integers(0, 256, size=300, dtype=np.uint8)) * 200,                      # long-distance matches (up to 32 KiB back)
             b"".join(bytes([i % 251]) * (i % 37 + 1) for i in range(3000))[:65536]]
    # a skewed alphabet: Huffman codes well beyond the 10-bit direct table
    p = np.array([2.0 ** -k for k in range(1, 41)]); p /= p.sum()
    datas.append(bytes(rng.choice(40, size=60000, p=p).astype(np.uint8)))
    # short periods (match distances 1..9, every length up to the maximum): the decoder copies these as wider periods
    per = []
    while sum(map(len, per)) < 65536:
        pl, rl = int(rng.integers(1, 10)), int(rng.integers(3, 400))
        per.append((bytes(rng.integers(0, 256, size=pl, dtype=np.uint8)) * (rl // pl + 1))[:rl])
    datas.append(b"".join(per)[:65536])
    payloads, want = [], []
    for d in datas:
        for level in (0, 1, 6, 9):
            payloads.append(raw_deflate(d, level)); want.append(d)
        payloads.append(raw_deflate(d, 6, zlib.Z_FIXED)); want.append(d)
        payloads.append(raw_deflate(d, 6, zlib.Z_HUFFMAN_ONLY)); want.append(d)
    with host.Context(0) as ctx:
        got = inflate(ctx, payloads)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "stream %d (%d bytes in, %d out)" % (k, len(payloads[k]), len(w))


def uneven_token_density(rng):
    """a Huffman-only block whose first bits hold one token a bit (12 000 literals of one value, a 1-bit code) and whose rest holds one in
    eleven: cut into 64 equal bit segments, the lanes of the dense part have ~700 tokens each -- more than a lane's slot of thj_k_huffp
    (inf2::SLOT_TOKENS = 640) holds, so the block takes the storing pass instead of the copy from the slots"""
    return bytes([7]) * 12000 + bytes((rng.integers(0, 250, size=4300) + 6).astype(np.uint8))


def test_a_lane_with_more_tokens_than_its_slot_holds():
    rng = np.random.default_rng(9)
    d = uneven_token_density(rng)
    payloads = [raw_deflate(d, 6, zlib.Z_HUFFMAN_ONLY), raw_deflate(d[::-1], 6, zlib.Z_HUFFMAN_ONLY), raw_deflate(d, 6)]
    want = [d, d[::-1], d]
    assert len(payloads[0]) * 8 // 64 > 640          # bits per segment = tokens per dense segment
    with host.Context(0) as ctx:
        got = inflate(ctx, payloads)
    assert got == want


def test_corrupt_streams_are_flagged_not_trusted():
    good = raw_deflate(b"the quick brown fox jumps over the lazy dog " * 500)
    bad = [good[:len(good) // 2],                          # truncated
           b"\x07" + good[1:],                              # reserved block type 3
           good[:40] + bytes(40) + good[80:],               # damaged middle
           b""]                                             # nothing at all
    with host.Context(0) as ctx:
        got = inflate(ctx, bad + [good])
    assert got[-1] == b"the quick brown fox jumps over the lazy dog " * 500
    assert got[0] is None and got[1] is None and got[3] is None
    assert got[2] is None or got[2] != b"the quick brown fox jumps over the lazy dog " * 500


def bgzf_payloads(path, limit=None):
    data = open(path, "rb").read()
    out, isize, off = [], [], 0
    while off < len(data) and (limit is None or len(out) < limit):
        assert data[off:off + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
        out.append(data[off + 18:off + bsize - 8])
        isize.append(struct.unpack_from("<I", data, off + bsize - 4)[0])
        off += bsize
    return out, isize


def test_bam_files_inflate_like_zlib(tmp_path):
    d = str(tmp_path / "gen")
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", "40000", "--genome-len", "3000000",
                           "--introns", "1200", "--threads", "8"], stdout=subprocess.DEVNULL)
    with host.Context(0) as ctx:
        for fn in ("left_seg1.bam", "right_map.bam", "left_reads.bam"):
            pl, isize = bgzf_payloads(os.path.join(d, fn))
            got = inflate(ctx, pl)
            assert len(pl) > 10
            for k, p in enumerate(pl):
                w = zlib.decompress(p, -15)
                assert len(w) == isize[k] and got[k] == w, (fn, k)


def test_a_launch_of_many_members_goes_through_in_pieces(tmp_path, monkeypatch):
    """the token scratch between the two inflate kernels is bounded: a launch of more members than a piece holds (THJ_INFLATE_CHUNK, 8192
    by default) runs piece by piece over the same scratch -- the same bytes"""
    d = str(tmp_path / "gen")
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", "20000", "--genome-len", "3000000",
                           "--introns", "1200", "--threads", "8"], stdout=subprocess.DEVNULL)
    pl, isize = bgzf_payloads(os.path.join(d, "left_reads.bam"))
    assert len(pl) > 20
    monkeypatch.setenv("THJ_INFLATE_CHUNK", "7")
    with host.Context(0) as ctx:
        got = inflate(ctx, pl)
    for k, p in enumerate(pl):
        assert got[k] == zlib.decompress(p, -15), k


def test_two_kernel_inflater_mixed_launch():
    """One launch holding members of every kind at once -- several dynamic blocks per member, fixed and stored blocks, members of more
    symbols than the token stream keeps (those and the stored ones take the one-lane kernel), empty members, corrupt ones -- in an
    order that spreads them over the lanes of the lane-per-member kernel."""
    import random
    rng = random.Random(7)
    payloads, want = [], []
    for k in range(330):
        kind = k % 11
        n = rng.choice((1, 40, 3000, 30000, 65536))
        if kind < 4:
            words = [bytes(rng.randrange(97, 123) for _ in range(rng.randrange(2, 9))) for _ in range(rng.randrange(3, 300))]
            d = b" ".join(rng.choice(words) for _ in range(n // 4 + 1))[:n]
        elif kind < 6:
            d = bytes(rng.randrange(rng.randrange(2, 256)) for _ in range(n))
        elif kind < 8:
            piece = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 500)))
            d = (piece * (n // len(piece) + 1))[:n]
        else:
            d = bytes(min(255, int(rng.expovariate(rng.choice((0.02, 0.1, 0.5))))) for _ in range(n))
        c = zlib.compressobj(rng.choice((0, 1, 6, 9)) if kind != 10 else 6, zlib.DEFLATED, -15, rng.choice((1, 3, 8, 9)),
                             rng.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED)))
        p = c.compress(d) + c.flush()
        if kind == 10 and len(p) > 20:                       # a damaged member among the good ones
            p = p[:len(p) // 2]
            d = None
        payloads.append(p); want.append(d)
    with host.Context(0) as ctx:
        got = inflate(ctx, payloads)
    for k, (g, w) in enumerate(zip(got, want)):
        if w is None:
            assert g is None or g != w
        else:
            assert g == w, "member %d (%d bytes in, %d out)" % (k, len(payloads[k]), len(w))


def test_members_written_by_the_references_bgzf_c():
    """fixed inputs made by REFERENCE code (tests/golden/ref_samtools/, samtools 0.1.18's bgzf.c at zlib's default level, see mint.py)"""
    import json
    ref = os.path.join(ROOT, "tests", "golden", "ref_samtools")
    man = json.load(open(os.path.join(ref, "MANIFEST.json")))
    files = [man["bgzf_members"]["file"]] + sorted(k + ".samtools.bam" for k in man["cases"])
    with host.Context(0) as ctx:
        for fn in files:
            pl, isize = bgzf_payloads(os.path.join(ref, fn))
            got = inflate(ctx, pl)
            for k, p in enumerate(pl):
                w = zlib.decompress(p, -15)
                assert len(w) == isize[k]
                if isize[k]:
                    assert got[k] == w, (fn, k)
