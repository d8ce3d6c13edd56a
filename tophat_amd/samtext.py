"""Minimal SAM-text reader/writer for tests and fixtures.

The record -> hit rules follow BAMHitFactory::get_hit_from_buf
(bwt_map.cpp:1101-1452): qname `id|offset:seg:nseg` gives insert_id (atoi) and
the end() flag (`seg+1 == nseg`; no '|' means end, :1120-1143); mismatches =
NM minus indel lengths (:1362-1365) kept in an unsigned char; edit_dist =
mismatches + gap length (:1430-1431); a record whose mate maps to another
contig is dropped (:1409-1415); an N op longer than max_report_intron drops the
record (:1341-1345).  Unmapped records (tid < 0) are skipped: the segment maps
the pipeline feeds this path hold mapped records only.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

from .batch import HitRec

_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


def parse_header(path: str) -> Tuple[List[str], List[int]]:
    names, lens = [], []
    with open(path) as f:
        for line in f:
            if not line.startswith("@"):
                break
            if line.startswith("@SQ"):
                d = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                names.append(d["SN"])
                lens.append(int(d["LN"]))
    return names, lens


def parse_sam_hits(path: str, ref_ids: Dict[str, int], max_report_intron: int = 500000) -> Iterator[HitRec]:
    with open(path) as f:
        for line in f:
            if line.startswith("@") or not line.strip():
                continue
            t = line.rstrip("\n").split("\t")
            qname, flag, rname, pos, _mapq, cigar, rnext = t[0], int(t[1]), t[2], int(t[3]), t[4], t[5], t[6]
            end = True
            pipe = qname.rfind("|")
            if pipe >= 0:
                tag = qname[pipe + 1:]
                if ":" in tag:
                    m = re.match(r"(\d+):(\d+):(\d+)", tag)
                    if m:
                        end = int(m.group(2)) + 1 == int(m.group(3))
                qname = qname[:pipe]
            m = re.match(r"\s*[+-]?\d+", qname)
            rid = int(m.group(0)) if m else 0          # atoi
            if rname == "*" or (flag & 4):
                continue
            nm = 0
            xs_minus = False
            for tag in t[11:]:
                if tag.startswith("NM:i:"):
                    nm = int(tag[5:])
                elif tag.startswith("XS:A:"):
                    xs_minus = tag[5:6] == "-"
            ops = []
            spliced = False
            mism = nm & 0xFF
            right = pos - 1
            read_len = 0
            gap = 0
            ok = True
            for n, op in _CIG.findall(cigar):
                n = int(n)
                if n <= 0:
                    ok = False
                    break
                opcode = {"M": 1, "I": 3, "D": 5, "N": 11, "S": 13, "H": 14, "P": 15}.get(op)
                if opcode is not None and op != "H":
                    ops.append((opcode, n))
                if op == "N":
                    spliced = True
                if op == "M":
                    right += n
                    read_len += n
                elif op == "I":
                    read_len += n
                    gap += n
                    mism = (mism - n) & 0xFF
                elif op == "D":
                    right += n
                    gap += n
                    mism = (mism - n) & 0xFF
                elif op == "N":
                    if n > max_report_intron:
                        ok = False
                        break
                    right += n
                elif op == "S":
                    read_len += n
                elif op in "HP":
                    pass
                else:
                    ok = False
                    break
            if not ok:
                continue
            if rnext not in ("*", "=", rname):
                continue
            yield (rid, ref_ids[rname], pos - 1, right, bool(flag & 0x10), end, mism, (mism + gap) & 0xFF, read_len,
                   ops, xs_minus and spliced)   # antisense_splice only kept for spliced hits (bwt_map.cpp:1419-1448)


def read_fastq(path: str) -> Dict[int, str]:
    out: Dict[int, str] = {}
    with open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            if not h.strip():
                continue
            seq = f.readline().strip()
            f.readline()
            f.readline()
            name = h[1:].split()[0]
            out[int(name)] = seq.replace(".", "N")      # reads.cpp:119
    return out


def read_fasta(path: str) -> Tuple[List[str], List[str]]:
    names: List[str] = []
    seqs: List[List[str]] = []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                names.append(re.split(r"[ \t\r]", line[1:].strip(), 1)[0])   # segment_juncs.cpp:73-77
                seqs.append([])
            elif names:
                seqs[-1].append(line.strip())
    return names, ["".join(s) for s in seqs]


def md_nm(ref: str, rs: str) -> Tuple[int, str]:
    md, run, nm = "", 0, 0
    for r, q in zip(ref, rs):
        if r == q:
            run += 1
        else:
            md += str(run) + r
            run = 0
            nm += 1
    return nm, md + str(run)


def sam_header(names: Sequence[str], lens: Sequence[int]) -> str:
    return "@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(names, lens))
