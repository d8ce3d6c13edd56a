"""The BAM writer's device side on the CPU (tests/hostsim/bamout_sim.cpp): the deflate kernel's source compiled as it is and run one
workgroup at a time under tests/hostsim/simt.h -- its streams must inflate (zlib, and this build's own device inflater's CPU build)
to the member's bytes, with the member's CRC-32 -- and the per-record BAM encoder against an independent restatement of
print_bamhit's record (bwt_map.cpp:1888-2093, GBamRecord / add_aux common.cpp:1005-1173) written from the BAM format."""
import ctypes as C
import os
import struct
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from locked_make import locked_make  # noqa: E402

from tophat_amd.host import ALN_DTYPE, bgzf_plan_cuts  # noqa: E402
from tophat_amd.bamio import parse_bam_record  # noqa: E402

ST_OK, ST_TOO_BIG = 0, 1


@pytest.fixture(scope="module", params=["libbamoutsim.so", "libbamoutsim_exact.so"], ids=["shannon_slack_lengths", "huffman_lengths"])
def lib(request):
    d = os.path.join(HERE, "hostsim")
    locked_make(d)
    l = C.CDLL(os.path.join(d, request.param))
    l.deflate_sim_member.restype = C.c_int
    l.bamenc_sim_records.restype = C.c_int64
    return l


@pytest.fixture(scope="module")
def inflater():
    d = os.path.join(HERE, "hostsim")
    locked_make(d)
    l = C.CDLL(os.path.join(d, "libinflatesim.so"))
    l.inflate_sim_huffp.restype = C.c_int
    l.inflate_sim_lz.restype = C.c_int64
    return l


def deflate(lib, data):
    out = np.zeros(65536, np.uint8)
    res = np.zeros(3, np.uint32)
    assert lib.deflate_sim_member(C.c_char_p(data), C.c_uint32(len(data)), C.c_void_p(out.ctypes.data), C.c_void_p(res.ctypes.data)) == 0
    clen, crc, st = (int(x) for x in res)
    return out[:clen].tobytes(), crc, st


def check(lib, data):
    comp, crc, st = deflate(lib, data)
    assert st == ST_OK
    assert crc == (zlib.crc32(data) & 0xFFFFFFFF)
    d = zlib.decompressobj(-15)
    assert d.decompress(comp) == data and d.eof and d.unused_data == b""       # exactly one complete stream, nothing after it
    return comp


def bam_like_stream(rng, n, first_id=1000000):
    out = []
    for i in range(n):
        name = str(first_id + i).encode() + b"\0"
        seq = rng.integers(0, 4, 100)
        nib = np.array([1, 2, 4, 8], np.uint8)[seq]
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
        mm = int(rng.random() < 0.2)
        md = (b"100" if not mm else b"37A62") + b"\0"
        spliced = rng.random() < 0.25
        cig = [(100 << 4)] if not spliced else [(60 << 4), (int(rng.integers(100, 5000)) << 4) | 3, (40 << 4)]
        body = struct.pack("<iiIIiiii", 0, int(rng.integers(0, 60000000)), (4681 << 16) | (255 << 8) | len(name), len(cig), 100, -1, -1, 0)
        body += name + b"".join(struct.pack("<I", c) for c in cig) + packed + bytes([40]) * 100
        body += b"ASC\0XMC" + bytes([mm]) + b"XOC\0XGC\0MDZ" + md + b"NMC" + bytes([mm]) + (b"XSA+" if spliced else b"")
        out.append(struct.pack("<I", len(body)) + body)
    return out


def test_members_of_every_shape_inflate_to_themselves(lib):
    rng = np.random.default_rng(11)
    text = (b"the quick brown fox jumps over the lazy dog. " * 1500)[:65536]
    cases = [b"x", b"ab", b"abc", b"aaaa", b"hello hello hello hello", bytes(65536), bytes([7]) * 1000, text, text[:4095], text[:4097],
             bytes(range(256)) * 3, rng.integers(0, 4, 30000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(),
             rng.integers(0, 2, 65536, dtype=np.uint8).tobytes()]
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 4096, 65535, 65536):
        cases.append(rng.integers(65, 70, n, dtype=np.uint8).tobytes())
    for d in cases:
        check(lib, d)


def test_long_matches_and_runs_cross_the_waves_slices(lib):
    # 258-byte matches, matches cut at a slice's end (4 KiB slices for a full member), the history of the slice before
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 256, 700, dtype=np.uint8).tobytes()
    d = (unit * 100)[:65536]
    c = check(lib, d)
    assert len(c) < 9000                       # one literal copy of the unit per slice at the worst
    d2 = bytes([1]) * 5000 + bytes([2]) * 300 + unit + bytes([3]) * 60000
    assert len(check(lib, d2[:65536])) < 2500


def test_a_skewed_alphabet_is_length_limited_to_15_bits(lib):
    # Fibonacci counts: the unrestricted Huffman tree is as deep as the alphabet is large
    fib = [1, 1]
    while len(fib) < 22:
        fib.append(fib[-1] + fib[-2])
    data = b"".join(bytes([k]) * f for k, f in enumerate(fib))
    rng = np.random.default_rng(2)
    data = bytes(rng.permutation(np.frombuffer(data, np.uint8)))
    assert 40000 < len(data) <= 65536
    check(lib, data)


def test_incompressible_members_are_refused(lib):
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 65536, dtype=np.uint8).tobytes()
    comp, crc, st = deflate(lib, data)
    assert st == ST_TOO_BIG                     # does not fit 64 KiB with its envelope: the caller replays on the host (bgzf.c shrinks the member)
    assert crc == (zlib.crc32(data) & 0xFFFFFFFF)
    check(lib, data[:60000])                    # this one does


def test_bam_records_compress_about_as_zlib_level_1(lib, inflater):
    rng = np.random.default_rng(7)
    recs = bam_like_stream(rng, 900)
    stream = b"".join(recs)
    cuts = bgzf_plan_cuts([len(r) for r in recs])
    assert cuts[-1] == len(stream) and all(b - a <= 65536 for a, b in zip([0] + cuts, cuts))
    total = z1 = 0
    for a, b in zip([0] + cuts, cuts):
        d = stream[a:b]
        comp = check(lib, d)
        total += len(comp)
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        z1 += len(co.compress(d) + co.flush())
        # the device inflater (its CPU build) reads the device deflater's members
        tokens = np.zeros(20480, np.uint32); ntok = C.c_uint32(); outp = C.c_uint32(); passes = C.c_int64()
        assert inflater.inflate_sim_huffp(C.c_char_p(comp), C.c_uint32(len(comp)), C.c_uint32(0), C.c_void_p(tokens.ctypes.data), C.byref(ntok), C.byref(outp),
                                          C.byref(passes)) == 0
        out = np.zeros(65536, np.uint8); rounds = C.c_int64()
        n = inflater.inflate_sim_lz(C.c_void_p(tokens.ctypes.data), ntok, C.c_void_p(out.ctypes.data), C.byref(rounds))
        assert n == len(d) and out[:n].tobytes() == d
    assert total < 1.12 * z1


# ---------------------------------------------------------------- the record encoder

BAMOP = {1: 0, 2: 0, 3: 1, 4: 1, 5: 2, 6: 2, 11: 3, 12: 3, 13: 4}


def reg2bin(beg, end):
    end -= 1
    for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return off + (beg >> sh)
    return 0


def int_tag(tag, x):
    if x < 0:
        return tag + (b"c" + struct.pack("<b", x) if x >= -127 else b"s" + struct.pack("<h", x) if x >= -32767 else b"i" + struct.pack("<i", x))
    return tag + (b"C" + struct.pack("<B", x) if x <= 255 else b"S" + struct.pack("<H", x) if x <= 65535 else b"I" + struct.pack("<I", x))


def expected_record(a, name, seq, qual, tid):
    """the record print_bamhit writes for alignment `a` of the read (name, seq: letters, qual: phred bytes)"""
    cig = [(int(c) >> 28, int(c) & 0x0FFFFFFF) for c in a["cigar"][:int(a["n_cigar"])]]
    anti = bool(a["flags"] & 1)
    if anti:
        seq = "".join({"A": "T", "C": "G", "G": "C", "T": "A"}.get(ch, "N") for ch in reversed(seq))
        qual = qual[::-1]
    nt16 = {ch: i for i, ch in enumerate("=ACMGRSVTWYHKDBN")}
    nib = [nt16.get(ch, 15) for ch in seq] + [0]
    packed = bytes((nib[2 * k] << 4) | nib[2 * k + 1] for k in range((len(seq) + 1) // 2))
    pos = int(a["left"])
    end = pos + sum(l for op, l in cig if BAMOP[op] in (0, 2, 3))
    indel = sum(l for op, l in cig if 3 <= op <= 6)
    spliced = any(op in (11, 12) for op, _ in cig)
    nm = name.encode() + b"\0"
    body = struct.pack("<iiIIiiii", tid, pos, (reg2bin(pos, end) << 16) | (255 << 8) | len(nm), ((0x10 if anti else 0) << 16) | len(cig), len(seq), -1, -1, 0)
    body += nm + b"".join(struct.pack("<I", (l << 4) | BAMOP[op]) for op, l in cig) + packed + qual
    body += int_tag(b"AS", int(a["AS"])) + int_tag(b"XM", int(a["XM"])) + int_tag(b"XO", int(a["XO"])) + int_tag(b"XG", int(a["XG"]))
    body += b"MDZ" + bytes(a["md"])[:int(a["md_len"])] + b"\0" + int_tag(b"NM", int(a["mismatches"]) + indel)
    if spliced:
        body += b"XSA" + (b"-" if a["flags"] & 4 else b"+")
    return struct.pack("<I", len(body)) + body


def read_record(name, seq, qual):
    """an unaligned read as the reads BAM holds it (bam_import of the prepared reads: flag 4, no cigar)"""
    nt16 = {ch: i for i, ch in enumerate("=ACMGRSVTWYHKDBN")}
    nib = [nt16.get(ch, 15) for ch in seq] + [0]
    packed = bytes((nib[2 * k] << 4) | nib[2 * k + 1] for k in range((len(seq) + 1) // 2))
    nm = name.encode() + b"\0"
    body = struct.pack("<iiIIiiii", -1, -1, (4680 << 16) | len(nm), 4 << 16, len(seq), -1, -1, 0) + nm + packed + qual + b"ZTZextra\0"
    return struct.pack("<I", len(body)) + body


def test_records_are_what_print_bamhit_writes(lib):
    rng = np.random.default_rng(9)
    reads, infl, loc = [], b"", []
    for r in range(300):
        n = int(rng.choice([100, 99, 75, 51, 101]))
        seq = "".join(rng.choice(list("ACGTN"), p=[0.24, 0.24, 0.24, 0.24, 0.04]) for _ in range(n))
        qual = bytes(int(x) for x in rng.integers(2, 41, n))
        name = str(1 + r * 7) if r % 11 else " %d" % (r + 5)           # atol skips leading blanks
        reads.append((name, seq, qual))
        infl += b"\xEE" * int(rng.integers(0, 5))                     # records do not start aligned
        loc.append(len(infl))
        infl += read_record(name, seq, qual)
    alns = np.zeros(1000, dtype=ALN_DTYPE)
    want = []
    for i in range(len(alns)):
        a = alns[i]
        r = int(rng.integers(0, len(reads)))
        n = len(reads[r][1])
        kind = int(rng.integers(0, 6))
        if kind == 0:
            cig = [(1, n)]
        elif kind == 1:
            cig = [(1, 30), (11, int(rng.integers(50, 400000))), (1, n - 30)]
        elif kind == 2:
            cig = [(2, 20), (12, 700), (2, n - 20)]
        elif kind == 3:
            cig = [(1, 40), (3, 3), (1, n - 43)]
        elif kind == 4:
            cig = [(1, 25), (5, 2), (1, 10), (11, 900), (1, 5), (4, 1), (1, n - 41)]
        else:
            cig = [(13, 4), (1, n - 4)]
        a["read_idx"] = r; a["ref_id"] = int(rng.integers(1, 4)); a["left"] = int(rng.integers(0, 200000000))
        a["flags"] = int(rng.choice([0, 1, 4, 5]))
        a["mismatches"] = int(rng.integers(0, 4)); a["n_cigar"] = len(cig)
        a["AS"] = int(rng.choice([0, -6, -127, -128, -300, 12])); a["XM"] = int(rng.integers(0, 4)); a["XO"] = int(rng.integers(0, 2)); a["XG"] = int(rng.integers(0, 3))
        md = ("%d" % n if not a["mismatches"] else "10A5^CT%d" % (n - 16)).encode()
        a["md_len"] = len(md); a["md"] = md
        a["cigar"][:len(cig)] = [(op << 28) | l for op, l in cig]
        want.append(expected_record(a, reads[r][0], reads[r][1], reads[r][2], [5, 0, 2][int(a["ref_id"]) - 1]))
    loc = np.asarray(loc, np.uint32); tid = np.asarray([5, 0, 2], np.int32)
    sizes = np.zeros(len(alns), np.uint32); rids = np.zeros(len(alns), np.int64)
    inflb = np.frombuffer(infl + b"\0" * 8, np.uint8)

    def run(out):
        return lib.bamenc_sim_records(C.c_void_p(alns.ctypes.data), C.c_int64(len(alns)), C.c_void_p(inflb.ctypes.data), C.c_void_p(loc.ctypes.data),
                                      C.c_void_p(tid.ctypes.data), C.c_void_p(sizes.ctypes.data), C.c_void_p(rids.ctypes.data), None if out is None else C.c_void_p(out.ctypes.data))
    total = run(None)
    assert total == sum(len(w) for w in want)
    out = np.full(total + 16, 0xCD, np.uint8)
    assert run(out) == total and bytes(out[total:]) == b"\xCD" * 16
    assert out[:total].tobytes() == b"".join(want)
    assert [int(x) for x in sizes] == [len(w) for w in want]
    assert [int(x) for x in rids] == [int(reads[int(a["read_idx"])][0]) for a in alns]
    # and they are well-formed BAM: the repo's reader takes them
    rec = parse_bam_record(want[1][4:], ["c0", "c1", "c2", "c3", "c4", "c5"])
    assert rec is not None
    # fusion alignments, MD strings left to the host and reads of another length are not the device encoder's
    for edit in ("fusion", "md", "len"):
        b = alns[:3].copy()
        if edit == "fusion":
            b[1]["cigar"][1] = (8 << 28) | 1000
        elif edit == "md":
            b[1]["md_len"] = 255
        else:
            b[1]["cigar"][0] = (1 << 28) | 7; b[1]["n_cigar"] = 1
        assert lib.bamenc_sim_records(C.c_void_p(b.ctypes.data), C.c_int64(3), C.c_void_p(inflb.ctypes.data), C.c_void_p(loc.ctypes.data), C.c_void_p(tid.ctypes.data),
                                      C.c_void_p(sizes.ctypes.data), C.c_void_p(rids.ctypes.data), None) == -1


def test_plan_cuts_follows_bam_write1s_flush_rule():
    assert bgzf_plan_cuts([]) == []
    assert bgzf_plan_cuts([100]) == [100]
    assert bgzf_plan_cuts([40000, 30000]) == [40000, 70000]                       # the second record does not fit: a new member
    assert bgzf_plan_cuts([30000, 30000, 5536]) == [65536]                         # exactly full: flushed by bgzf_write
    assert bgzf_plan_cuts([30000, 30000, 5537]) == [60000, 65537]
    assert bgzf_plan_cuts([10, 70000, 10]) == [10, 65546, 70020]                   # a record larger than a block spills over
