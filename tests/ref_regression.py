"""The reference's own regression cases as known-answer tests (fixtures: tests/golden_ref/, minted by
make_ref_regression.py from /root/reference/tests/regression_tests).  Test infrastructure only.

The recorded tophat_out/ came out of the whole pipeline; the two programs of the hot path sit in its middle, between
bowtie runs that are not available here.  `load()` supplies what bowtie would (`bowtie -n 2 -k 40`, run.log of the
cases): every placement of every 12-base segment on the genome with at most two mismatches, by exhaustive search, and
-- for the long_spanning_reads stage -- every such placement on the junction database `juncs_db 3 12` builds from the
recorded junctions / insertions / deletions (the step tophat.py runs between the two programs).  What the cases then
pin:

  * segment_juncs: the junction set found == the introns of the recorded junctions.bed (test_SimpleSplicing);
  * long_spanning_reads: every record of the recorded accepted_hits.sam that long_spanning_reads can be the source of
    (all of them, in fact) appears in its output with the same strand, POS, CIGAR and NM.

Not pinned by them: segment_juncs' indel search (24-base reads have two segments; v2.1.2 searches indels from three,
segment_juncs.cpp:2856 -- the recorded indels came from an older release's closure search), anything paired-end, fusions.
The recorded runs predate --read-mismatches / --read-gap-length / --read-edit-dist (common.cpp:123-125), so the indel
cases run with those limits opened up (3-base insertions exceed the default gap length of 2)."""
from __future__ import annotations

import os
import re

import numpy as np

from tophat_amd.batch import Events, JUNC_DTYPE, build_seg_batch, build_span_batch, cigar_string, events_to_span_inputs
from tophat_amd.params import Params
from tophat_amd.samtext import md_nm, parse_spliced_sam_hits

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden_ref")
CASES = ["test_SimpleSplicing", "test_SimpleIndel", "test_IndelWithErrors"]
SEG_LEN = 12
_COMP = str.maketrans("ACGTN", "TGCAN")


def _placements(target: str, query: str, max_mm: int = 2):
    """(pos, mismatches) of every placement of query on target with at most max_mm mismatches"""
    if len(target) < len(query):
        return []
    t = np.frombuffer(target.encode(), dtype=np.uint8)
    w = np.lib.stride_tricks.sliding_window_view(t, len(query))
    mm = (w != np.frombuffer(query.encode(), dtype=np.uint8)).sum(axis=1)
    return [(int(p), int(mm[p])) for p in np.nonzero(mm <= max_mm)[0]]


def _segments(read: str):
    ns = len(read) // SEG_LEN
    return [read[s * SEG_LEN:(s + 1) * SEG_LEN] if s < ns - 1 else read[s * SEG_LEN:] for s in range(ns)]


def load(case: str, tmp_path, juncs_db_text):
    """`juncs_db_text(names, juncs_file, ins_file, del_file, read_len, min_anchor)` -> FASTA text of the junction database
    (the oracle's, or the product's juncs_db: the caller decides which one it is testing with)."""
    d = os.path.join(GOLD, case)
    genome = "".join(l.strip() for l in open(os.path.join(d, "genome.fa")) if not l.startswith(">")).upper()
    name = open(os.path.join(d, "genome.fa")).readline()[1:].split()[0]
    reads, ids = {}, {}
    for k, l in enumerate(open(os.path.join(d, "reads.tsv"))):
        nm, seq = l.rstrip("\n").split("\t")
        ids[nm] = k + 1                      # prep_reads numbers the reads in input order
        reads[k + 1] = seq.upper()
    nseg = max(len(r) for r in reads.values()) // SEG_LEN
    # -- segments against the genome
    seg_recs = [[] for _ in range(nseg)]
    for rid in sorted(reads):
        segs = _segments(reads[rid])
        for s, seg in enumerate(segs):
            for anti in (False, True):
                q = seg.translate(_COMP)[::-1] if anti else seg
                for pos, mm in _placements(genome, q):
                    seg_recs[s].append((rid, 1, pos, pos + len(q), anti, s == len(segs) - 1, mm, mm, len(q)))
    # -- the recorded events, as the coordinate files segment_juncs writes (A13)
    juncs, dels, ins = [], [], []
    for l in open(os.path.join(d, "junctions.bed")):
        if not l.startswith("track"):
            t = l.split("\t")
            bs = [int(x) for x in t[10].split(",")]
            juncs.append((1, int(t[1]) + bs[0] - 1, int(t[2]) - bs[1], 1 if t[5] == "-" else 0))
    for l in open(os.path.join(d, "deletions.bed")):
        if not l.startswith("track"):
            t = l.split("\t")
            dels.append((1, int(t[1]) - 1, int(t[2]), 0))
    for l in open(os.path.join(d, "insertions.bed")):
        if not l.startswith("track"):
            t = l.split("\t")
            ins.append((1, int(t[1]), t[3].upper()))
    files = {k: str(tmp_path / ("%s.%s" % (case, k))) for k in ("juncs", "insertions", "deletions")}
    open(files["juncs"], "w").write("".join("%s\t%d\t%d\t%s\n" % (name, l, r, "-" if a else "+") for (_, l, r, a) in juncs))
    open(files["deletions"], "w").write("".join("%s\t%d\t%d\n" % (name, l + 1, r) for (_, l, r, _a) in dels))
    open(files["insertions"], "w").write("".join("%s\t%d\t%d\t%s\n" % (name, l, l, q) for (_, l, q) in ins))
    # -- segments against the junction database (tophat.py: juncs_db <min_anchor 3> <segment length>, then bowtie)
    db = juncs_db_text([name], files["juncs"], files["insertions"], files["deletions"], SEG_LEN, 3)
    contigs = []
    for blk in db.split(">")[1:]:
        cn, cs = blk.split("\n", 1)
        contigs.append((cn.strip(), cs.replace("\n", "").upper()))
    spliced = []
    for s in range(nseg):
        path = str(tmp_path / ("%s.seg%d.to_spliced.sam" % (case, s + 1)))
        with open(path, "w") as f:
            for rid in sorted(reads):
                segs = _segments(reads[rid])
                if s >= len(segs):
                    continue
                for anti in (False, True):
                    q = segs[s].translate(_COMP)[::-1] if anti else segs[s]
                    for cn, cs in contigs:
                        for pos, _mm in _placements(cs, q):
                            nm_, md = md_nm(cs[pos:pos + len(q)], q)
                            f.write("%d|%d:%d:%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                                rid, s * SEG_LEN, s, len(segs), 16 if anti else 0, cn, pos + 1, len(q), q, "I" * len(q), nm_, md))
        spliced.append(path)
    p = Params(segment_length=SEG_LEN)
    if case != "test_SimpleSplicing":
        p.read_mismatches, p.read_gap_length, p.read_edit_dist = 4, 3, 7        # see the module docstring
    spl_recs = [list(parse_spliced_sam_hits(f, {name: 1}, p.max_report_intron, p.min_anchor_len)) for f in spliced]
    quals = {rid: "I" * len(r) for rid, r in reads.items()}
    ev = Events(np.array(juncs, dtype=JUNC_DTYPE) if juncs else np.zeros(0, dtype=JUNC_DTYPE),
                np.array(dels, dtype=JUNC_DTYPE) if dels else np.zeros(0, dtype=JUNC_DTYPE), sorted(ins), {})
    span_juncs, span_ins = events_to_span_inputs(ev)
    expected = []
    for l in open(os.path.join(d, "accepted_hits.tsv")):
        qn, flag, pos, cigar, nm = l.rstrip("\n").split("\t")
        expected.append((ids[qn], int(flag) & 16, int(pos), cigar, int(nm)))
    return dict(p=p, names=[name], genome=genome, reads=reads, seg_recs=seg_recs, spliced_sam=spliced, seg_batch=build_seg_batch(seg_recs, reads),
                span_batch=build_span_batch(seg_recs, reads, quals, spl_recs), span_juncs=span_juncs, span_ins=span_ins,
                recorded_juncs=sorted(juncs), recorded_dels=sorted(dels), recorded_ins=sorted(ins), expected=expected, files=files)


def record_keys(alns, span_batch):
    """{read id: {(strand flag, POS, CIGAR, NM)}} of a long_spanning_reads result (tophat_amd.batch.Aln list)"""
    out = {}
    for a in alns:
        indel = sum(c & 0x0FFFFFFF for c in a.cigar if (c >> 28) in (3, 4, 5, 6))
        out.setdefault(int(span_batch.read_id[a.read_idx]), set()).add(
            (16 if a.antisense else 0, a.left + 1, cigar_string(a.cigar), a.mismatches + indel))
    return out


def check_recorded_alignments(case_data, alns):
    """every recorded accepted hit is among `alns`; returns (records checked, gapped records checked)"""
    ours = record_keys(alns, case_data["span_batch"])
    missing = [e for e in case_data["expected"] if e[1:] not in ours.get(e[0], set())]
    assert not missing, "%d recorded alignments not reproduced, e.g. %s (ours for that read: %s)" % (
        len(missing), missing[0], sorted(ours.get(missing[0][0], ())))
    gapped = sum(1 for e in case_data["expected"] if re.search("[NDI]", e[3]))
    return len(case_data["expected"]), gapped


def write_program_inputs(case_data, tmp_path):
    """the files tophat.py would hand to the two programs: ref.fa, hdr.sam, reads.fq (numbered reads), one id-sorted
    SAM-text map per segment; -> dict of paths"""
    name, genome, reads = case_data["names"][0], case_data["genome"], case_data["reads"]
    f = {k: str(tmp_path / k) for k in ("ref.fa", "hdr.sam", "reads.fq", "left_map.sam")}
    open(f["ref.fa"], "w").write(">%s\n%s\n" % (name, genome))
    hdr = "@HD\tVN:1.0\tSO:unsorted\n@SQ\tSN:%s\tLN:%d\n" % (name, len(genome))
    open(f["hdr.sam"], "w").write(hdr)
    open(f["left_map.sam"], "w").write(hdr)
    open(f["reads.fq"], "w").write("".join("@%d\n%s\n+\n%s\n" % (rid, r, "I" * len(r)) for rid, r in sorted(reads.items())))
    segs = []
    for s, recs in enumerate(case_data["seg_recs"]):
        path = str(tmp_path / ("left_seg%d.sam" % (s + 1)))
        with open(path, "w") as out:
            out.write(hdr)
            for (rid, _ref, left, right, anti, _end, _mm, _ed, rl) in recs:
                segs_of = _segments(reads[rid])
                q = segs_of[s].translate(_COMP)[::-1] if anti else segs_of[s]
                nm_, md = md_nm(genome[left:right], q)
                out.write("%d|%d:%d:%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                    rid, s * SEG_LEN, s, len(segs_of), 16 if anti else 0, name, left + 1, rl, q, "I" * rl, nm_, md))
        segs.append(path)
    f["segs"] = segs
    return f
