"""CPU: the host I/O layer of the drop-in executables (tophat_amd/csrc/host/thj_hostio.h) through tests/hostio:
the threaded BGZF/BAM writer (+ its `.index` side file, common.h:562-606) and the threaded record readers."""
import os
import struct
import subprocess

from locked_make import locked_make
import zlib

import pytest

from golden_util import CASES, GOLD
from tophat_amd.bamio import read_bam, write_bam_from_sam
from tophat_amd.samtext import parse_sam_hits

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "hostio", "hostio_check")


@pytest.fixture(scope="module")
def exe():
    locked_make(os.path.join(HERE, "hostio"))
    return EXE


def bgzf_blocks(path):
    """[(file offset, uncompressed bytes)] of every BGZF member"""
    data = open(path, "rb").read()
    out, off = [], 0
    while off < len(data):
        assert data[off:off + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
        raw = zlib.decompress(data[off + 18:off + bsize - 8], -15)
        assert struct.unpack_from("<I", data, off + bsize - 4)[0] == len(raw)
        assert struct.unpack_from("<I", data, off + bsize - 8)[0] == (zlib.crc32(raw) & 0xFFFFFFFF)
        out.append((off, raw))
        off += bsize
    return out


def test_bam_writer_is_thread_and_batch_invariant(exe, tmp_path):
    n = 60000
    a, b, c = (str(tmp_path / x) for x in ("a.bam", "b.bam", "c.bam"))
    subprocess.check_call([exe, "write", a, str(n), str(n)], env=dict(os.environ, THJ_HOST_THREADS="1"))
    subprocess.check_call([exe, "write", b, str(n), "777"], env=dict(os.environ, THJ_HOST_THREADS="7"))
    subprocess.check_call([exe, "write", c, str(n), "12345"], env=dict(os.environ, THJ_HOST_THREADS="3"))
    # the shard workers' path: batches planned in order, deflated independently (here in reverse order), committed in order
    d, e = str(tmp_path / "d.bam"), str(tmp_path / "e.bam")
    subprocess.check_call([exe, "write", d, str(n), "777", "planned"])
    subprocess.check_call([exe, "write", e, str(n), "5", "planned-fail"])      # ... and the writer falls back to write_encoded half way
    ref = open(a, "rb").read()
    assert open(b, "rb").read() == ref and open(c, "rb").read() == ref
    assert open(d, "rb").read() == ref and open(e, "rb").read() == ref
    idx = open(a + ".index").read()
    assert open(b + ".index").read() == idx and open(c + ".index").read() == idx
    assert open(d + ".index").read() == idx and open(e + ".index").read() == idx
    # a well-formed BAM: every record comes back, in order
    names, recs = read_bam(a)
    recs = list(recs)
    assert names == ["chr1", "chr2"] and len(recs) == n
    assert [int(r[0]) for r in recs] == [1 + (i // 2) * 3 for i in range(n)]
    # BGZF structure as samtools-0.1.18 writes it: the header in a member of its own (bam_header_write ends with bgzf_flush),
    # then members of at most 64 KiB that hold whole records only (bgzf_flush_try before every record) and are as full as that
    # allows, then the empty EOF member
    blocks = bgzf_blocks(a)
    assert len(blocks[-1][1]) == 0
    assert blocks[0][1][:4] == b"BAM\x01" and len(blocks[0][1]) < 200
    sizes = []
    for _, raw in blocks[1:-1]:
        assert 0 < len(raw) <= 65536
        p = 0
        while p < len(raw):
            bs = struct.unpack_from("<i", raw, p)[0]
            p += 4 + bs
        assert p == len(raw), "a record straddles two members"
        sizes.append(len(raw))
    assert all(sz > 65536 - 400 for sz in sizes[:-1])
    # .index: `read_id \\t virtual offset`; every offset is the start of the first record of that read
    by_off = {off: raw for off, raw in blocks}
    lines = [l.split("\t") for l in idx.strip().split("\n")]
    assert len(lines) > 20
    stream = b"".join(raw for _, raw in blocks)
    starts = {}
    pos = 0
    for off, raw in blocks:
        starts[off] = pos
        pos += len(raw)
    last = 0
    for rid, voff in lines:
        rid, voff = int(rid), int(voff)
        assert rid > last
        last = rid
        coff, uoff = voff >> 16, voff & 0xFFFF
        assert coff in by_off
        at = starts[coff] + uoff
        bs, = struct.unpack_from("<i", stream, at)
        l_rn = stream[at + 4 + 8]
        qname = stream[at + 4 + 32:at + 4 + 32 + l_rn - 1].decode()
        assert int(qname) == rid and 32 < bs < 1000
        # ... and it is the FIRST record of that read: the one before it has another name
        prev_names = [int(r[0]) for r in recs[max(0, (rid - 1) // 3 * 2 - 2):(rid - 1) // 3 * 2]]
        assert all(x != rid for x in prev_names)


@pytest.mark.parametrize("name", CASES[:2])
def test_threaded_hit_reader_sam_and_bam(exe, name, tmp_path):
    d = os.path.join(GOLD, name)
    sam = os.path.join(d, "left_seg1.sam")
    bam = str(tmp_path / "left_seg1.bam")
    write_bam_from_sam(sam, bam)
    o1 = subprocess.check_output([exe, "hits", sam]).split()
    o2 = subprocess.check_output([exe, "hits", bam]).split()
    assert o1 == o2
    class AnyRef(dict):
        def __missing__(self, k):
            return 1
    hits = list(parse_sam_hits(sam, AnyRef()))
    assert int(o1[1]) == len(hits) and int(o1[0]) == len({h[0] for h in hits})


def test_threaded_read_stream(exe):
    d = os.path.join(GOLD, CASES[0])
    fq = open(os.path.join(d, "left.fq")).read().split("\n")
    recs = {int(fq[i][1:].split()[0]): (fq[i + 1], fq[i + 3]) for i in range(0, len(fq) - 3, 4)}
    ids = sorted(recs)[::7]
    out = subprocess.check_output([exe, "reads", os.path.join(d, "left.fq")] + [str(i) for i in ids]).decode().strip().split("\n")
    assert [tuple(l.split()) for l in out] == [(str(i), recs[i][0], recs[i][1]) for i in ids]


def test_fasta_loader_awkward_input(exe, tmp_path):
    """get_seqs (segment_juncs.cpp:64-88): names cut at the first blank, bases folded to ACGTN; CRLF, lower case, IUPAC
    codes, blank lines, an empty record and a missing final newline"""
    recs = [("chrA desc here", "acgtNNRYacgt" * 50), ("chrB", ""), ("chrC\tx", "ACGT" * 300000 + "nnnn"), ("chrD", "GATTACA")]
    txt = ""
    for name, seq in recs:
        txt += ">" + name + "\r\n"
        for i in range(0, len(seq), 61):
            txt += seq[i:i + 61] + ("\r\n" if name.startswith("chrA") else "\n")
        txt += "\n"
    fa = str(tmp_path / "awkward.fa")
    open(fa, "w", newline="").write(txt.rstrip("\n"))

    def cks(s):
        x = 0
        for c in s:
            x = (x * 131 + ord(c)) & 0xFFFFFFFFFFFFFFFF
        return x
    fold = lambda s: "".join(c.upper() if c.upper() in "ACGT" else "N" for c in s)
    exp = ["%s %d %d" % (n.split()[0], len(seq), cks(fold(seq))) for n, seq in recs]
    for threads in ("1", "5"):
        out = subprocess.check_output([exe, "fasta", fa], env=dict(os.environ, THJ_HOST_THREADS=threads)).decode().strip().split("\n")
        assert out == exp


def test_fast_deflate_members_inflate_with_zlib(exe, tmp_path):
    """thj_fastdeflate.h (the BAM writer's default compressor): every stream it makes is inflated by zlib to the input; inputs it
    declines (it must never write past a member's room) are left to zlib by the writer."""
    import zlib
    import numpy as np
    rng = np.random.default_rng(3)
    recs = []
    for i in range(40000):                       # BAM-record-like bytes: counters, short names, 4-bit bases, flat and noisy qualities
        ln = int(rng.integers(50, 151))
        recs.append(struct.pack("<iiIIiiii", 100 + ln, int(rng.integers(0, 3)), int(rng.integers(0, 1 << 26)), 0x12340000 | (i & 0xFF), 1, ln, -1, -1)
                    + str(1 + i // 2).encode() + b"\0" + bytes(rng.integers(0, 256, size=(ln + 1) // 2, dtype=np.uint8))
                    + (bytes([40]) * ln if i & 1 else bytes(rng.integers(2, 42, size=ln, dtype=np.uint8))) + b"NMC\x01MDZ" + str(ln).encode() + b"\0")
    raw = b"".join(recs)
    ins = [raw[i:i + 65280] for i in range(0, len(raw), 65280)]
    p = np.array([2.0 ** -k for k in range(1, 41)]); p /= p.sum()
    ins += [b"A" * 64, b"A" * 65536, bytes(rng.integers(0, 256, size=65536, dtype=np.uint8)), bytes(rng.integers(0, 4, size=65536, dtype=np.uint8)),
            (b"ACGTTGCA" * 9000)[:65536], bytes(rng.integers(0, 256, size=300, dtype=np.uint8)) * 200,
            b"".join(bytes([i % 251]) * (i % 37 + 1) for i in range(3000))[:65536], bytes(range(64)), b"ab" * 40, raw[:100], raw[5:70], raw[:7], b"",
            bytes(rng.choice(40, size=60000, p=p).astype(np.uint8)),                      # a skewed alphabet: codes that need the length limit
            bytes(rng.choice(256, size=65536, p=np.r_[[0.5], np.full(255, 0.5 / 255)]).astype(np.uint8))]
    fi, fo = str(tmp_path / "in"), str(tmp_path / "out")
    with open(fi, "wb") as f:
        for b in ins:
            f.write(struct.pack("<I", len(b))); f.write(b)
    subprocess.check_call([exe, "fdz", fi, fo])
    o = open(fo, "rb").read()
    pos = declined = 0
    tin = tout = 0
    for k, b in enumerate(ins):
        cl, = struct.unpack_from("<I", o, pos); pos += 4
        if cl == 0xFFFFFFFF:
            declined += 1
            continue
        assert cl <= 65536 - 26
        assert zlib.decompress(o[pos:pos + cl], -15) == b, "stream %d (%d bytes)" % (k, len(b))
        pos += cl
        if k < len(ins) - 15:
            tin += len(b); tout += cl
    assert pos == len(o)
    assert declined <= 2                          # only the incompressible 64 KiB blocks
    assert tout < 0.6 * tin                       # and it does compress records


@pytest.mark.parametrize("variant", ["planned-device", "planned-device-fail"])
def test_bam_writer_takes_batches_deflated_elsewhere(exe, tmp_path, variant):
    """every other batch arrives as the device path delivers it (BamWriter::plan_device / wrap_member: members ready, no record bytes):
    the BAM stream inside is the one the sequential writer makes, members end at the batch's ends in addition, and the `.index`
    offsets point at the first record of their read; -fail: a member of a host batch does not fit and the writer replays"""
    n = 60000
    a, d = str(tmp_path / "a.bam"), str(tmp_path / "d.bam")
    subprocess.check_call([exe, "write", a, str(n), str(n)])
    subprocess.check_call([exe, "write", d, str(n), "777", variant])
    A, D = bgzf_blocks(a), bgzf_blocks(d)
    stream = b"".join(raw for _, raw in D)
    assert stream == b"".join(raw for _, raw in A)
    assert len(D) > len(A) and len(D[-1][1]) == 0
    for _, raw in D[1:-1]:                                  # whole records only
        p = 0
        while p < len(raw):
            p += 4 + struct.unpack_from("<i", raw, p)[0]
        assert p == len(raw)
    # record starts in the inflated stream, by name
    hdr = len(D[0][1])
    starts, p = [], hdr
    while p < len(stream):
        bs, = struct.unpack_from("<i", stream, p)
        l_rn = stream[p + 12]
        starts.append((p, int(stream[p + 36:p + 36 + l_rn - 1])))
        p += 4 + bs
    first_of = {}
    for at, rid in starts:
        first_of.setdefault(rid, at)
    at_of, pos = {}, 0
    for off, raw in D:
        at_of[off] = pos
        pos += len(raw)
    lines = [l.split("\t") for l in open(d + ".index").read().strip().split("\n")]
    assert len(lines) > 20
    for rid, voff in lines:
        rid, voff = int(rid), int(voff)
        assert at_of[voff >> 16] + (voff & 0xFFFF) == first_of[rid]
    # the same reads are indexed as in the sequential writer's file (the rule counts records, not bytes)
    assert [l.split("\t")[0] for l in open(a + ".index").read().strip().split("\n")] == [x[0] for x in lines]
