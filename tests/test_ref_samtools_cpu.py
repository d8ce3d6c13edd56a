"""Vectors made by REFERENCE code (tests/golden/ref_samtools/, minted by mint.py with the reference's vendored samtools 0.1.18 compiled
from /root/reference -- oracle/ref_samtools.mk; nothing of it is needed here): MD / NM as bam_fillmd1_core recomputes them (the
reference's own regression check, regression_test.py:96-110), the BAM byte stream as sam_read1 + bam_write1 write it, and BGZF
members as bgzf.c writes them.  Rows B5 (NM), B6 (MD) and B8 (record bytes) of SURVEY 8(a) at 2x100 / 2x76 / 1x150 bp."""
import ctypes as C
import gzip
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF = os.path.join(GOLD, "ref_samtools")
MAN = json.load(open(os.path.join(REF, "MANIFEST.json")))
KEYS = sorted(MAN["cases"])


def records(key):
    case, side = key.split(".span_")
    return [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, case, "expected.span_%s.sam" % side)) if l.strip() and not l.startswith("@")]


def test_manifest_matches_the_files():
    import hashlib
    assert len(KEYS) >= 9
    for k in KEYS:
        assert hashlib.sha256(open(os.path.join(REF, k + ".samtools.bam"), "rb").read()).hexdigest() == MAN["cases"][k]["samtools_bam_sha256"]
        assert MAN["cases"][k]["samtools_reads_this_builds_bam_to_the_same_text"]


@pytest.mark.parametrize("key", KEYS)
def test_md_and_nm_are_what_samtools_calmd_computes(key):
    """the MD:Z and NM:i the spanning records carry (reproduced bit for bit by the oracle, the CPU build of the kernels and the HIP path in the
    golden tests) against bam_fillmd1_core's"""
    want = {}
    for l in open(os.path.join(REF, key + ".calmd.tsv")):
        q, fl, rn, pos, md, nm = l.rstrip("\n").split("\t")
        want.setdefault((q, fl, rn, pos), []).append((md, nm))
    n = n_over_n = 0
    for c in records(key):
        tags = {t[:2]: t[5:] for t in c[8:]}
        if "XF" in tags:
            continue
        md_nm = want[(c[0], c[1], c[2], c[3])].pop(0)
        if "N" in md_nm[0].upper():
            # the alignment lies over N in the genome: samtools counts N against N as a mismatch (bam_md.c:40), TopHat as a match
            # (bwt_map.cpp:2349-2465 counts them apart) -- the one place the two disagree by design; the reference's own
            # regression genomes hold no N
            n_over_n += 1
            n += 1
            continue
        # samtools writes a deleted or mismatched base as the FASTA has it; TopHat upper-cases (bam_fillmd1_core itself compares case-blind, bam_md.c:110)
        assert (tags["MD"].upper(), tags["NM"]) == (md_nm[0].upper(), md_nm[1]), (key, c[0], c[3])
        n += 1
    assert n == MAN["cases"][key]["calmd_rows"] and n - n_over_n > 20


def bam_stream(path):
    return gzip.open(path, "rb").read()


@pytest.mark.parametrize("key", KEYS)
def test_bam_bytes_are_what_samtools_writes(key, tmp_path):
    """BamWriter::encode (bin, packed CIGAR, 4-bit bases, qualities, smallest-int tags, Z strings) and the header against sam_read1 + bam_write1"""
    from locked_make import locked_make
    locked_make(os.path.join(HERE, "hostio"))
    case, side = key.split(".span_")
    out = str(tmp_path / "ours.bam")
    subprocess.check_call([os.path.join(HERE, "hostio", "hostio_check"), "sam2bam", os.path.join(GOLD, case, "hdr.sam"),
                           os.path.join(GOLD, case, "expected.span_%s.sam" % side), out])
    ours, theirs = bam_stream(out), bam_stream(os.path.join(REF, key + ".samtools.bam"))
    assert ours == theirs


def members(path):
    d = open(path, "rb").read()
    off = 0
    while off < len(d):
        bsize = struct.unpack_from("<H", d, off + 16)[0] + 1
        yield off + 18, d[off + 18:off + bsize - 8], struct.unpack_from("<I", d, off + bsize - 4)[0]
        off += bsize


def test_inflate_logic_on_members_written_by_bgzf_c():
    """the device inflater's logic (CPU build, tests/hostsim/inflate_sim.cpp) on members the reference's bgzf.c wrote"""
    from locked_make import locked_make
    locked_make(os.path.join(HERE, "hostsim"))
    lib = C.CDLL(os.path.join(HERE, "hostsim", "libinflatesim.so"))
    lib.inflate_sim_huffp.restype = C.c_int
    lib.inflate_sim_lz.restype = C.c_int64
    n = 0
    for fn in [MAN["bgzf_members"]["file"]] + [k + ".samtools.bam" for k in KEYS[:3]]:
        for at, comp, isize in members(os.path.join(REF, fn)):
            if not isize:
                continue
            want = zlib.decompress(comp, -15)
            assert len(want) == isize
            tokens = np.zeros(20480 + 8, dtype=np.uint32)
            ntok, outp, passes = C.c_uint32(0), C.c_uint32(0), C.c_int64(0)
            cb = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
            rc = lib.inflate_sim_huffp(cb, C.c_uint32(len(comp)), C.c_uint32(at & 15), C.c_void_p(tokens.ctypes.data), C.byref(ntok), C.byref(outp), C.byref(passes))
            assert rc == 0 and outp.value == isize
            out = np.zeros(65536 + 64, dtype=np.uint8)
            got = lib.inflate_sim_lz(C.c_void_p(tokens.ctypes.data), C.c_uint32(ntok.value), C.c_void_p(out.ctypes.data), None)
            assert got == isize and out[:isize].tobytes() == want
            n += 1
    assert n >= 5
