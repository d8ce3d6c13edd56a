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
        rec = data[off:off + bs]
        off += bs
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
        recs.append((qname, flag, names[tid] if tid >= 0 else "*", pos + 1, mapq, cigar, seq, qual) + tuple(tags))
    return names, recs
