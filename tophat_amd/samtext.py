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


# ---------------------------------------------------------------------------------------------
# junction-db ("spliced") segment maps: SplicedBAMHitFactory::get_hit_from_buf + spliceCigar +
# getBAMmismatches (bwt_map.cpp:1469-1770, :681-883, :410-475).  Python mirror of the C++ host parser
# (tophat_amd/csrc/host/thj_hostio.h: parse_spliced_hit), intron / deletion / insertion entries.
# ---------------------------------------------------------------------------------------------

def _cigar_add(c, op):
    if op[1] <= 0:
        return
    if c and c[-1][0] == op[0]:          # bwt_map.cpp:672-678: extends the previous op AND appends (sic)
        c[-1] = (c[-1][0], c[-1][1] + op[1])
    c.append(op)


_FUSION_CODES = (7, 8, 9, 10)          # FUSION_FF, FUSION_FR, FUSION_RF, FUSION_RR (bwt_map.h:36-55)
_LOWER = {1: 2, 3: 4, 5: 6, 11: 12}    # MATCH -> mATCH, INS -> iNS, DEL -> dEL, REF_SKIP -> rEF_SKIP


def _splice_cigar(cigar, mism, left, spl_start, spl_len, spl_code, min_anchor_len):
    """spliceCigar (bwt_map.cpp:681-865).  For the fusion codes the pieces before an rf / rr break and after an fr / rr break
    come out in their lower-case (reversed) forms."""
    INS, DEL, SKIP, MATCH, PAD, SOFT = 3, 5, 11, 1, 15, 13
    fus = spl_code in _FUSION_CODES
    low_before = spl_code in (9, 10)
    low_after = spl_code in (8, 10)
    out = []
    spl_ofs = spl_start - left
    if fus:
        spl_ofs = abs(spl_ofs)
    spl_ofs_end = spl_ofs + (spl_len if spl_code == INS else 0)
    gapop = (spl_code, spl_len)
    ref_ofs = read_ofs = 0
    spl_mm = 0
    xfound = False
    if spl_ofs_end > 0:
        for (op, ln) in cigar:
            prev_read, cur_ofs = read_ofs, ref_ofs
            if op == MATCH:
                ref_ofs += ln
                read_ofs += ln
                for o in range(cur_ofs, ref_ofs):
                    r = prev_read + (o - cur_ofs)
                    m = 0 <= r < len(mism) and mism[r]
                    if spl_code == INS:
                        spl_mm += 1 if (spl_ofs <= o < spl_ofs_end and m) else 0
                    else:
                        spl_mm += 1 if (abs(spl_ofs - o) < min_anchor_len and m) else 0
            elif op in (DEL, SKIP, PAD):
                ref_ofs += ln
            elif op in (SOFT, INS):
                read_ofs += ln
            if cur_ofs >= spl_ofs_end or ref_ofs <= spl_ofs:
                if cur_ofs == spl_ofs_end and spl_code != INS and op != INS:
                    xfound = True
                    _cigar_add(out, gapop)
                o2 = op
                if (xfound and low_after) or (not xfound and low_before):
                    o2 = _LOWER.get(op, op)
                _cigar_add(out, (o2, ln))
            elif spl_code == INS:
                xfound = True
                if spl_ofs > cur_ofs:
                    _cigar_add(out, (op, spl_ofs - cur_ofs))
                if spl_ofs < 0:
                    if gapop[1] + spl_ofs > 0:
                        _cigar_add(out, (gapop[0], gapop[1] + spl_ofs))
                else:
                    _cigar_add(out, gapop)
                if ref_ofs > spl_ofs_end:
                    _cigar_add(out, (op, ref_ofs - spl_ofs_end))
            else:
                xfound = True
                _cigar_add(out, (_LOWER.get(op, op) if low_before else op, spl_ofs - cur_ofs))
                _cigar_add(out, gapop)
                _cigar_add(out, (_LOWER.get(op, op) if low_after else op, ref_ofs - spl_ofs))
    if spl_ofs_end <= 0:
        left = left - spl_len if spl_code == INS else left + spl_len
        out = list(cigar)
    ok = len(out) >= len(cigar) + 2 and out[0][0] in (1, 2) and out[-1][0] in (1, 2)
    return ok, out, left, spl_mm


def parse_spliced_sam_hits(path: str, ref_ids: Dict[str, int], max_report_intron: int = 500000,
                           min_anchor_len: int = 8) -> Iterator[HitRec]:
    with open(path) as f:
        for line in f:
            if line.startswith("@") or not line.strip():
                continue
            t = line.rstrip("\n").split("\t")
            qname, flag, rname, pos, cigar, rnext, seq = t[0], int(t[1]), t[2], int(t[3]), t[5], t[6], t[9]
            end = True
            pipe = qname.rfind("|")
            if pipe >= 0:
                m = re.match(r"(\d+):(\d+):(\d+)", qname[pipe + 1:])
                if m:
                    end = int(m.group(2)) + 1 == int(m.group(3))
                qname = qname[:pipe]
            rid = int(re.match(r"\s*[+-]?\d+", qname).group(0))
            if rname == "*" or (flag & 4):
                continue
            ops, ok = [], True
            for n, o in _CIG.findall(cigar):
                n = int(n)
                code = {"M": 1, "I": 3, "D": 5, "N": 11, "S": 13, "P": 15}.get(o)
                if o == "H":
                    continue
                if n <= 0 or code is None or (o == "N" and n > max_report_intron):
                    ok = False
                    break
                ops.append((code, n))
            if not ok or rnext not in ("*", "=", rname):
                continue
            mism = [False] * len(seq)
            num_mm = 0
            md = next((x[5:] for x in t[11:] if x.startswith("MD:Z:")), None)
            if md is not None:
                bi = 0
                for tok in re.findall(r"\d+|\^[A-Za-z]+|[A-Za-z]", md):
                    if tok[0].isdigit():
                        bi += int(tok)
                    elif tok[0] == "^":
                        bi += len(tok) - 1
                    else:
                        num_mm += 1
                        if bi < len(mism):
                            mism[bi] = True
                        bi += 1
            toks = rname.split("|")
            ne = len(toks) - 6
            if ne < 0:
                continue
            contig = "|".join(toks[:ne + 1])
            st = [x for x in toks[ne + 2].split("-") if x]
            if len(st) != 2:
                continue
            jtype, jstrand = toks[ne + 4], toks[ne + 5]
            lsp = int(st[0])
            ref_id2, flipped = None, False
            antisense = bool(flag & 0x10)
            if jtype == "ins":
                left = int(toks[ne + 1]) + pos - 1
                if left > lsp:
                    continue
                ok, spl, left, spl_mm = _splice_cigar(ops, mism, left, lsp + 1, len(st[1]), 3, min_anchor_len)
                if not ok:
                    continue
                num_mm -= spl_mm
            else:
                if jstrand not in ("ff", "fr", "rf", "rr", "rev", "fwd"):
                    continue
                fus = jtype == "fus"
                # bwt_map.cpp:1672-1677: on rf / rr fusion contigs the first piece runs down the genome
                if fus and jstrand in ("rf", "rr"):
                    left = int(toks[ne + 1]) - (pos - 1)
                else:
                    left = int(toks[ne + 1]) + pos - 1
                if jtype == "del":
                    code = 5
                elif fus:
                    code = {"ff": 7, "fr": 8, "rf": 9}.get(jstrand, 10)
                else:
                    code = 11
                gap_len = int(st[1]) if fus else int(st[1]) - lsp - 1
                if code in (9, 10):
                    lsp -= 1
                    if left <= lsp:
                        continue
                else:
                    lsp += 1
                    if left >= lsp:
                        continue
                ok, spl, left, spl_mm = _splice_cigar(ops, mism, left, lsp, gap_len, code, min_anchor_len)
                if not ok:
                    continue
                if fus:
                    cs = contig.split("-")
                    if len(cs) != 2:
                        continue
                    contig, ref_id2 = cs[0], ref_ids[cs[1]]
                    if jstrand in ("rf", "rr"):
                        antisense = not antisense
                        flipped = True
            gap = sum(n for o, n in spl if o in (3, 4, 5, 6))
            right = left
            for o, n in spl:
                if o in (1, 5, 11):
                    right += n
                elif o in (2, 6, 12):
                    right -= n
                elif o in _FUSION_CODES:
                    right = n
            rl = sum(n for o, n in spl if o in (1, 2, 3, 4, 13))
            rec = (rid, ref_ids[contig], left, right, antisense, end, num_mm & 0xFF, (num_mm + gap) & 0xFF, rl, spl, jstrand == "rev")
            if ref_id2 is not None:
                rec += (ref_id2, flipped)
            yield rec
