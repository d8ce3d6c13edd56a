"""The device inflater's logic on the CPU (tests/hostsim/inflate_sim.cpp): the lane logic of thj_k_huff compiled as it is, and
thj_k_lz's batch algorithm restated, against zlib -- dynamic, fixed and multi-block streams, long codes, run-length data, every
start alignment, BGZF members as samtools' bgzf.c writes them (zlib's default level) and as this build's own compressor does,
and the cases the fast path must hand to the one-lane kernel (stored blocks, corrupt streams)."""
import ctypes as C
import os
import random
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from locked_make import locked_make  # noqa: E402

ROOT = os.path.dirname(HERE)
TOKCAP = 20480
FALLBACK = 0xFFFFFFFF


@pytest.fixture(scope="module")
def lib():
    d = os.path.join(HERE, "hostsim")
    locked_make(d)
    l = C.CDLL(os.path.join(d, "libinflatesim.so"))
    l.inflate_sim_huff.restype = C.c_int
    l.inflate_sim_lz.restype = C.c_int64
    return l


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for i in range(0, len(data), flush_every):
        out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH if i % (2 * flush_every) else zlib.Z_BLOCK if hasattr(zlib, "Z_BLOCK") else zlib.Z_FULL_FLUSH)
    return out + c.flush()


PASSES = []          # segment passes per member of the wave-per-member design (statistics)


def run(lib, comp, skew=0):
    tokens = np.zeros(TOKCAP + 8, dtype=np.uint32)
    ntok, outp = C.c_uint32(0), C.c_uint32(0)
    cb = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
    rc = lib.inflate_sim_huff(cb, C.c_uint32(len(comp)), C.c_uint32(skew), C.c_void_p(tokens.ctypes.data), C.byref(ntok), C.byref(outp))
    # the wave-per-member kernel's logic must produce the very same token stream
    tokens2 = np.zeros(TOKCAP + 8, dtype=np.uint32)
    ntok2, outp2, passes = C.c_uint32(0), C.c_uint32(0), C.c_int64(0)
    rc2 = lib.inflate_sim_huffp(cb, C.c_uint32(len(comp)), C.c_uint32(skew), C.c_void_p(tokens2.ctypes.data), C.byref(ntok2), C.byref(outp2), C.byref(passes))
    assert rc2 == rc, (rc, rc2)
    if not rc:
        assert ntok2.value == ntok.value and outp2.value == outp.value and (tokens2[:ntok.value] == tokens[:ntok.value]).all()
        PASSES.append(passes.value)
    if rc:
        assert ntok.value == FALLBACK
        return None, 0
    out = np.full(65536 + 64, 0xCC, dtype=np.uint8)
    rounds = C.c_int64(0)
    n = lib.inflate_sim_lz(C.c_void_p(tokens.ctypes.data), C.c_uint32(ntok.value), C.c_void_p(out.ctypes.data), C.byref(rounds))
    assert n == outp.value, (n, outp.value)
    assert (out[n:] == 0xCC).all(), "wrote past the end"
    return out[:n].tobytes(), ntok.value


def bam_like(rng, n):
    """records that resemble BAM: a counter, packed random bases, a run of one quality, short tags"""
    out = bytearray()
    i = rng.randrange(10 ** 6)
    while len(out) < n:
        i += 1
        out += struct.pack("<iiBBHHHiiii", 120, 0, 8, 255, 4681, 1, rng.choice((0, 16)), 100, -1, -1, 0)
        out += (str(i) + "\0").encode() + struct.pack("<I", 100 << 4)
        out += bytes(rng.randrange(256) for _ in range(50)) + bytes([40]) * 100 + b"NMC" + bytes([rng.randrange(3)]) + b"MDZ100\0"
    return bytes(out[:n])


def cases(rng):
    yield "empty", b"", {}
    yield "one byte", b"x", {}
    yield "text", (b"the quick brown fox jumps over the lazy dog. " * 2000)[:65536], {}
    yield "random", bytes(rng.randrange(256) for _ in range(30000)), {}
    yield "random 4 symbols", bytes(rng.choice(b"ACGT") for _ in range(65536)), {}
    yield "run of one byte", b"I" * 65536, {}
    yield "period 3", b"abc" * 21000, {}
    yield "skewed alphabet (long codes)", bytes(min(255, int(rng.expovariate(0.08))) for _ in range(65536)), {}
    yield "many symbols, 15-bit codes", b"".join(bytes([k]) * max(1, 2 ** (k % 14)) for k in range(256))[:65536], {}
    yield "bam-like", bam_like(rng, 65536), {}
    yield "bam-like level 1", bam_like(rng, 65536), {"level": 1}
    yield "bam-like level 9", bam_like(rng, 65536), {"level": 9}
    yield "fixed codes", bam_like(rng, 20000), {"strategy": zlib.Z_FIXED}
    yield "huffman only", bam_like(rng, 20000), {"strategy": zlib.Z_HUFFMAN_ONLY}
    yield "rle", bam_like(rng, 65536), {"strategy": zlib.Z_RLE}
    yield "several blocks", bam_like(rng, 65536), {"flush_every": 9000}
    yield "small memLevel (many blocks)", bam_like(rng, 65536), {"mem": 1}
    yield "far matches", (bytes(rng.randrange(256) for _ in range(32000)) * 3)[:65536], {}


def test_streams_against_zlib(lib):
    rng = random.Random(5)
    for name, data, kw in cases(rng):
        comp = raw_deflate(data, **kw)
        assert zlib.decompress(comp, -15) == data
        for skew in (0, 1, 7, 15) if len(data) > 1000 else range(16):
            got, ntok = run(lib, comp, skew)
            if got is None:
                # only what the fast path may refuse: stored blocks, or more tokens than it keeps
                assert kw.get("level") == 0 or len(data) > TOKCAP - 2 or b"\x00" in comp[:1] or name in ("random", "far matches", "small memLevel (many blocks)", "empty", "one byte"), name
                continue
            assert got == data, (name, skew)


def test_fuzz_against_zlib(lib):
    rng = random.Random(11)
    n_fast = 0
    for it in range(int(os.environ.get("THJ_INFLATE_SIM_SEEDS", "150"))):
        kind = rng.randrange(5)
        n = rng.choice((1, 2, 3, 50, 300, 5000, 40000, 65536))
        if kind == 0:
            data = bam_like(rng, n)
        elif kind == 1:
            k = rng.randrange(2, 200)
            data = bytes(rng.randrange(k) for _ in range(n))
        elif kind == 2:
            piece = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 400)))
            data = (piece * (n // len(piece) + 1))[:n]
        elif kind == 3:
            data = bytes(min(255, int(rng.expovariate(rng.choice((0.02, 0.1, 0.5))))) for _ in range(n))
        else:
            words = [bytes(rng.randrange(97, 123) for _ in range(rng.randrange(2, 9))) for _ in range(rng.randrange(3, 300))]
            data = b" ".join(rng.choice(words) for _ in range(n // 4 + 1))[:n]
        comp = raw_deflate(data, level=rng.choice((1, 4, 6, 9)), strategy=rng.choice((zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_FIXED, zlib.Z_RLE)),
                           mem=rng.choice((8, 8, 9, 3)), flush_every=rng.choice((0, 0, 0, 3000)))
        got, ntok = run(lib, comp, rng.randrange(16))
        if got is not None:
            assert got == data, it
            n_fast += 1
    assert n_fast > 100


def test_lanes_whose_tokens_do_not_fit_their_slots(lib):
    """thj_k_huffp keeps a lane's tokens in the lane's slot until the lanes before it have counted theirs; a segment with more tokens
    than a slot holds sends the block through a storing pass instead (inflate_sim_set_slot: the slot made small, so that ordinary
    streams take that pass) -- the same tokens either way"""
    lib.inflate_sim_set_slot(C.c_uint32(40))
    try:
        test_streams_against_zlib(lib)
        rng = random.Random(17)
        for n in (300, 5000, 65536):
            data = bam_like(rng, n)
            got, _ = run(lib, raw_deflate(data), rng.randrange(16))
            assert got == data
    finally:
        lib.inflate_sim_set_slot(C.c_uint32(0))


def test_a_lane_with_more_tokens_than_a_real_slot_holds(lib):
    """the same with the slots at their real size: a block whose first bits hold a token a bit (tests/test_gpu_ingest.py has its twin)"""
    rng = random.Random(9)
    data = bytes([7]) * 12000 + bytes(rng.randrange(6, 256) for _ in range(4300))
    for d in (data, data[::-1]):
        comp = raw_deflate(d, strategy=zlib.Z_HUFFMAN_ONLY)
        assert len(comp) * 8 // 64 > 640
        got, _ = run(lib, comp, 3)
        assert got == d


def test_refusals(lib):
    rng = random.Random(3)
    data = bam_like(rng, 30000)
    assert run(lib, raw_deflate(data, level=0))[0] is None                      # stored blocks
    comp = bytearray(raw_deflate(data))
    assert run(lib, bytes(comp[:len(comp) // 2]))[0] in (None,)                 # truncated: runs off the end
    bad = 0
    for k in range(40):                                                         # bit flips: refused, or decoded to something of legal size -- never a crash
        c2 = bytearray(comp)
        c2[rng.randrange(len(c2))] ^= 1 << rng.randrange(8)
        got, _ = run(lib, bytes(c2))
        bad += got is None
    assert bad > 0
    assert run(lib, b"\x07")[0] is None                                         # BTYPE 3


def test_bgzf_members_of_generated_bam(lib, tmp_path):
    gen = os.path.join(ROOT, "tools", "bin", "thj_gen")
    if not os.path.exists(gen):
        pytest.skip("tools/bin/thj_gen not built")
    subprocess.check_call([gen, "--out", str(tmp_path), "--pairs", "3000", "--genome-len", "2000000", "--introns", "300", "--threads", "2"], stdout=subprocess.DEVNULL)
    n = 0
    for f in ("left_seg1.bam", "left_map.bam", "left_reads.bam"):
        d = open(os.path.join(tmp_path, f), "rb").read()
        off = 0
        while off < len(d):
            bsize = struct.unpack_from("<H", d, off + 16)[0] + 1
            comp = d[off + 18:off + bsize - 8]
            want = zlib.decompress(comp, -15)
            if want:
                got, ntok = run(lib, comp, (off + 18) & 15)
                assert got == want, (f, off)
                n += 1
            off += bsize
    assert n > 10
