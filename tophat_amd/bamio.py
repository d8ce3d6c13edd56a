"""Tiny BAM reader (BGZF = concatenated gzip members) used by tests to read the
spanning BAM back as SAM-like tuples.  Test/fixture helper; the drop-in binaries
have their own C++ BAM I/O."""
from __future__ import annotations

import gzip
import struct
from typing import Iterator, List, Tuple

_CIG = "MIDNSHP=X"
_SEQ = "=ACMGRSVTWYHKDBN"


def read_bam(path: str) -> Tuple[List[str], Iterator[tuple]]:
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\x01", "not a BAM file"
    l_text, = struct.unpack_from("<i", data, 4)
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, off)
    off += 4
    names = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, off)
        off += 4
        names.append(data[off:off + l_name - 1].decode())
        off += l_name + 4
    recs = []
    while off < len(data):
        bs, = struct.unpack_from("<i", data, off)
        off += 4
        recs.append(parse_bam_record(data[off:off + bs], names))
        off += bs
    return names, recs


def parse_bam_record(rec: bytes, names: List[str]) -> tuple:
    """one BAM record (without its block_size word) -> (qname, flag, rname, pos, mapq, cigar, seq, qual, tags...)"""
    tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
    p = 32
    qname = rec[p:p + l_rn - 1].decode()
    p += l_rn
    cig = struct.unpack_from("<%dI" % n_cig, rec, p)
    p += 4 * n_cig
    cigar = "".join("%d%s" % (c >> 4, _CIG[c & 0xF]) for c in cig) or "*"
    sb = rec[p:p + (l_seq + 1) // 2]
    p += (l_seq + 1) // 2
    seq = "".join(_SEQ[(sb[i >> 1] >> (4 if (i & 1) == 0 else 0)) & 0xF] for i in range(l_seq))
    qual = "".join(chr(min(q, 93) + 33) for q in rec[p:p + l_seq])
    p += l_seq
    tags = []
    while p < len(rec):
        tag = rec[p:p + 2].decode()
        ty = chr(rec[p + 2])
        p += 3
        if ty == "A":
            tags.append("%s:A:%s" % (tag, chr(rec[p])))
            p += 1
        elif ty in "cCsSiI":
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
            v, = struct.unpack_from(fmt, rec, p)
            p += struct.calcsize(fmt)
            tags.append("%s:i:%d" % (tag, v))
        elif ty == "Z":
            e = rec.index(b"\0", p)
            tags.append("%s:Z:%s" % (tag, rec[p:e].decode()))
            p = e + 1
        else:
            raise ValueError("unsupported aux type %s" % ty)
    return (qname, flag, names[tid] if tid >= 0 else "*", pos + 1, mapq, cigar, seq, qual) + tuple(tags)


def write_bam_from_sam(sam_path: str, bam_path: str) -> None:
    """SAM text (with @SQ header) -> BAM, enough of the format for segment / read maps used as test inputs."""
    import zlib
    names, lens, text, recs = [], [], "", []
    with open(sam_path) as f:
        for line in f:
            if line.startswith("@"):
                text += line
                if line.startswith("@SQ"):
                    d = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                    names.append(d["SN"])
                    lens.append(int(d["LN"]))
            elif line.strip():
                recs.append(line.rstrip("\n").split("\t"))
    tid = {n: i for i, n in enumerate(names)}
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(names)))
    for n, l in zip(names, lens):
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    for t in recs:
        qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = t[:11]
        ops = []
        if cigar != "*":
            import re
            ops = [(int(n), _CIG.index(o)) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", cigar)]
        body = bytearray()
        body += qname.encode() + b"\0"
        for n, o in ops:
            body += struct.pack("<I", (n << 4) | o)
        sb = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            sb[i >> 1] |= _SEQ.index(ch if ch in _SEQ else "N") << (4 if (i & 1) == 0 else 0)
        body += sb + bytes(ord(c) - 33 for c in qual)
        for tag in t[11:]:
            k, ty, v = tag.split(":", 2)
            if ty == "i":
                body += k.encode() + b"i" + struct.pack("<i", int(v))
            elif ty == "A":
                body += k.encode() + b"A" + v.encode()
            else:
                body += k.encode() + b"Z" + v.encode() + b"\0"
        rt = tid.get(rname, -1)
        mt = rt if rnext == "=" else tid.get(rnext, -1)
        core = struct.pack("<iiBBHHHiiii", rt, int(pos) - 1, len(qname) + 1, int(mapq), 4680, len(ops), int(flag), len(seq), mt,
                           int(pnext) - 1, int(tlen))
        out += struct.pack("<i", len(core) + len(body)) + core + body
    # BGZF framing
    with open(bam_path, "wb") as f:
        data = bytes(out)
        for off in list(range(0, len(data), 0xFF00)) + [None]:
            chunk = b"" if off is None else data[off:off + 0xFF00]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                    struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
