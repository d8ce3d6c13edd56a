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
# case -> segment length of the recorded command line (24-base reads: two 12-base or three 8-base segments)
CASE_SEG_LEN = {"test_SimpleSplicing": 12, "test_SimpleIndel": 12, "test_IndelWithErrors": 12, "test_Paired": 12, "test_3Segment": 8,
                "test_ReverseComplementSplicing": 12, "test_ReverseComplementIndel": 12, "test_IndelLowerCase": 12, "test_Indel_1": 12}
CASES = list(CASE_SEG_LEN)
# what each case pins: see oracle/README.md
SPLICE_CASES = {"test_SimpleSplicing": (1, 63, 138, 0), "test_ReverseComplementSplicing": (1, 63, 138, 0),
                "test_Paired": (1, 122, 280, 0), "test_3Segment": (1, 122, 280, 0)}
SEG_LEN = 12          # the single-end cases of round 1 (kept for callers that only run those)
_COMP = str.maketrans("ACGTN", "TGCAN")


def _placements(target: str, query: str, max_mm: int = 2):
    """(pos, mismatches) of every placement of query on target with at most max_mm mismatches"""
    if len(target) < len(query):
        return []
    t = np.frombuffer(target.encode(), dtype=np.uint8)
    w = np.lib.stride_tricks.sliding_window_view(t, len(query))
    mm = (w != np.frombuffer(query.encode(), dtype=np.uint8)).sum(axis=1)
    return [(int(p), int(mm[p])) for p in np.nonzero(mm <= max_mm)[0]]


def _segments(read: str, seg_len: int = SEG_LEN):
    ns = len(read) // seg_len
    return [read[s * seg_len:(s + 1) * seg_len] if s < ns - 1 else read[s * seg_len:] for s in range(ns)]


def _both_strands(target, seg, k_max=40):
    """bowtie -n 2 -k 40 -m 40 (run.log of the cases): every placement on either strand with <= 2 mismatches, none at all when
    there are more than 40 of them"""
    out = []
    for anti in (False, True):
        q = seg.translate(_COMP)[::-1] if anti else seg
        out += [(pos, mm, anti, q) for pos, mm in _placements(target, q)]
    return out if len(out) <= k_max else []


def load(case: str, tmp_path, juncs_db_text):
    """`juncs_db_text(names, juncs_file, ins_file, del_file, read_len, min_anchor)` -> FASTA text of the junction database
    (the oracle's, or the product's juncs_db: the caller decides which one it is testing with).
    Single-end cases: the left side only.  c["sides"][side] holds the per-side inputs; the keys of round 1 (seg_batch,
    span_batch, expected, ...) stay as aliases of the left side's."""
    d = os.path.join(GOLD, case)
    L = CASE_SEG_LEN[case]
    genome = "".join(l.strip() for l in open(os.path.join(d, "genome.fa")) if not l.startswith(">")).upper()
    name = open(os.path.join(d, "genome.fa")).readline()[1:].split()[0]
    side_files = {"left": "reads.tsv"}
    if os.path.exists(os.path.join(d, "reads_right.tsv")):
        side_files["right"] = "reads_right.tsv"
    paired = len(side_files) == 2
    ids = {}
    sides = {}
    for sd, fn in side_files.items():
        reads = {}
        for k, l in enumerate(open(os.path.join(d, fn))):
            nm, seq = l.rstrip("\n").split("\t")
            ids[nm] = k + 1                      # prep_reads numbers the reads in input order; mates share the number
            reads[k + 1] = seq.upper()           # lower-case input (test_IndelLowerCase) is upper-cased by prep_reads
        nseg = max(len(r) for r in reads.values()) // L
        seg_recs = [[] for _ in range(nseg)]
        full_recs = []
        for rid in sorted(reads):
            segs = _segments(reads[rid], L)
            for s_, seg in enumerate(segs):
                for pos, mm, anti, q in _both_strands(genome, seg):
                    seg_recs[s_].append((rid, 1, pos, pos + len(q), anti, s_ == len(segs) - 1, mm, mm, len(q)))
            for pos, mm, anti, q in _both_strands(genome, reads[rid]):           # the whole-read map (the mate's, in find_gaps)
                full_recs.append((rid, 1, pos, pos + len(q), anti, True, mm, mm, len(q)))
        sides[sd] = dict(reads=reads, nseg=nseg, seg_recs=seg_recs, full_recs=full_recs)
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
    ins_files = ["insertions.bed"] + (["input_insertions.bed"] if os.path.exists(os.path.join(d, "input_insertions.bed")) else [])
    for fn in ins_files:                          # recorded + user-supplied (tophat --insertions, test_Indel_1)
        for l in open(os.path.join(d, fn)):
            if not l.startswith("track") and l.strip():
                t = l.split("\t")
                if (1, int(t[1]), t[3].upper()) not in ins:
                    ins.append((1, int(t[1]), t[3].upper()))
    files = {k: str(tmp_path / ("%s.%s" % (case, k))) for k in ("juncs", "insertions", "deletions")}
    open(files["juncs"], "w").write("".join("%s\t%d\t%d\t%s\n" % (name, l, r, "-" if a else "+") for (_, l, r, a) in juncs))
    open(files["deletions"], "w").write("".join("%s\t%d\t%d\n" % (name, l + 1, r) for (_, l, r, _a) in dels))
    open(files["insertions"], "w").write("".join("%s\t%d\t%d\t%s\n" % (name, l, l, q) for (_, l, q) in ins))
    # -- segments against the junction database (tophat.py: juncs_db <min_anchor 3> <segment length>, then bowtie)
    db = juncs_db_text([name], files["juncs"], files["insertions"], files["deletions"], L, 3)
    contigs = []
    for blk in db.split(">")[1:]:
        cn, cs = blk.split("\n", 1)
        contigs.append((cn.strip(), cs.replace("\n", "").upper()))
    p = Params(segment_length=L)
    if paired:
        p.inner_dist_mean, p.inner_dist_std_dev = 50, 20          # tophat -r 50 (command.txt), default deviation
    if case not in ("test_SimpleSplicing", "test_ReverseComplementSplicing", "test_Paired", "test_3Segment"):
        p.read_mismatches, p.read_gap_length, p.read_edit_dist = 4, 3, 7        # see the module docstring
    ev = Events(np.array(juncs, dtype=JUNC_DTYPE) if juncs else np.zeros(0, dtype=JUNC_DTYPE),
                np.array(dels, dtype=JUNC_DTYPE) if dels else np.zeros(0, dtype=JUNC_DTYPE), sorted(ins), {})
    span_juncs, span_ins = events_to_span_inputs(ev)
    expected = {sd: [] for sd in sides}
    for l in open(os.path.join(d, "accepted_hits.tsv")):
        qn, flag, pos, cigar, nm = l.rstrip("\n").split("\t")[:5]
        sd = "right" if (int(flag) & 0x80) else "left"
        expected[sd].append((ids[qn], int(flag) & 16, int(pos), cigar, int(nm)))
    for sd, S in sides.items():
        reads, nseg = S["reads"], S["nseg"]
        spliced = []
        for s_ in range(nseg):
            path = str(tmp_path / ("%s.%s.seg%d.to_spliced.sam" % (case, sd, s_ + 1)))
            with open(path, "w") as f:
                for rid in sorted(reads):
                    segs = _segments(reads[rid], L)
                    if s_ >= len(segs):
                        continue
                    rows = []
                    for cn, cs in contigs:
                        rows += [(cn, cs) + x for x in _both_strands(cs, segs[s_], k_max=1 << 30)]
                    if len(rows) > 40:
                        rows = []
                    for cn, cs, pos, _mm, anti, q in sorted(rows, key=lambda x: (x[4], [c[0] for c in contigs].index(x[0]), x[2])):
                        nm_, md = md_nm(cs[pos:pos + len(q)], q)
                        f.write("%d|%d:%d:%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                            rid, s_ * L, s_, len(segs), 16 if anti else 0, cn, pos + 1, len(q), q, "I" * len(q), nm_, md))
            spliced.append(path)
        spl_recs = [list(parse_spliced_sam_hits(f, {name: 1}, p.max_report_intron, p.min_anchor_len)) for f in spliced]
        quals = {rid: "I" * len(r) for rid, r in reads.items()}
        other = "right" if sd == "left" else "left"
        S["spliced_sam"] = spliced
        S["seg_batch"] = build_seg_batch(S["seg_recs"], reads, sides[other]["full_recs"], sides[other]["seg_recs"][-1]) if paired \
            else build_seg_batch(S["seg_recs"], reads)
        S["span_batch"] = build_span_batch(S["seg_recs"], reads, quals, spl_recs)
        S["expected"] = expected[sd]
    left = sides["left"]
    return dict(p=p, L=L, paired=paired, names=[name], genome=genome, sides=sides, span_juncs=span_juncs, span_ins=span_ins,
                recorded_juncs=sorted(juncs), recorded_dels=sorted(dels), recorded_ins=sorted(ins), files=files,
                reads=left["reads"], seg_recs=left["seg_recs"], spliced_sam=left["spliced_sam"], seg_batch=left["seg_batch"],
                span_batch=left["span_batch"], expected=left["expected"])


def record_keys(alns, span_batch):
    """{read id: {(strand flag, POS, CIGAR, NM)}} of a long_spanning_reads result (tophat_amd.batch.Aln list)"""
    out = {}
    for a in alns:
        indel = sum(c & 0x0FFFFFFF for c in a.cigar if (c >> 28) in (3, 4, 5, 6))
        out.setdefault(int(span_batch.read_id[a.read_idx]), set()).add(
            (16 if a.antisense else 0, a.left + 1, cigar_string(a.cigar), a.mismatches + indel))
    return out


def check_recorded_alignments(case_data, alns, side="left"):
    """every recorded accepted hit of that side is among `alns`; returns (records checked, gapped records checked)"""
    S = case_data["sides"][side]
    ours = record_keys(alns, S["span_batch"])
    missing = [e for e in S["expected"] if e[1:] not in ours.get(e[0], set())]
    assert not missing, "%d recorded alignments not reproduced, e.g. %s (ours for that read: %s)" % (
        len(missing), missing[0], sorted(ours.get(missing[0][0], ())))
    gapped = sum(1 for e in S["expected"] if re.search("[NDI]", e[3]))
    return len(S["expected"]), gapped


def write_program_inputs(case_data, tmp_path):
    """the files tophat.py would hand to the two programs: ref.fa, hdr.sam, and per side reads (numbered), the whole-read map
    and one id-sorted SAM-text map per segment; -> dict of paths (f[side] = dict(reads, map, segs))"""
    name, genome, L = case_data["names"][0], case_data["genome"], case_data["L"]
    f = {k: str(tmp_path / k) for k in ("ref.fa", "hdr.sam")}
    open(f["ref.fa"], "w").write(">%s\n%s\n" % (name, genome))
    hdr = "@HD\tVN:1.0\tSO:unsorted\n@SQ\tSN:%s\tLN:%d\n" % (name, len(genome))
    open(f["hdr.sam"], "w").write(hdr)
    for sd, S in case_data["sides"].items():
        reads = S["reads"]
        g = dict(reads=str(tmp_path / ("%s.fq" % sd)), map=str(tmp_path / ("%s_map.sam" % sd)), segs=[])
        open(g["reads"], "w").write("".join("@%d\n%s\n+\n%s\n" % (rid, r, "I" * len(r)) for rid, r in sorted(reads.items())))
        with open(g["map"], "w") as out:
            out.write(hdr)
            for (rid, _ref, left, right, anti, _end, _mm, _ed, rl) in S["full_recs"]:
                q = reads[rid].translate(_COMP)[::-1] if anti else reads[rid]
                nm_, md = md_nm(genome[left:right], q)
                out.write("%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (rid, 16 if anti else 0, name, left + 1, rl, q, "I" * rl, nm_, md))
        for s_, recs in enumerate(S["seg_recs"]):
            path = str(tmp_path / ("%s_seg%d.sam" % (sd, s_ + 1)))
            with open(path, "w") as out:
                out.write(hdr)
                for (rid, _ref, left, right, anti, _end, _mm, _ed, rl) in recs:
                    segs_of = _segments(reads[rid], L)
                    q = segs_of[s_].translate(_COMP)[::-1] if anti else segs_of[s_]
                    nm_, md = md_nm(genome[left:right], q)
                    out.write("%d|%d:%d:%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                        rid, s_ * L, s_, len(segs_of), 16 if anti else 0, name, left + 1, rl, q, "I" * rl, nm_, md))
            g["segs"].append(path)
        f[sd] = g
    return f


def recorded_alignment_records(case: str):
    """the recorded accepted hits as junction-consensus input records: [(ref_id, left, antisense_splice, [(op, len) ...])]"""
    out = []
    opmap = {"M": 1, "I": 3, "D": 5, "N": 11, "S": 13}
    for l in open(os.path.join(GOLD, case, "accepted_hits.tsv")):
        t = l.rstrip("\n").split("\t")
        cig = [(opmap[o], int(n)) for n, o in re.findall(r"(\d+)([MIDNS])", t[3])]
        out.append((1, int(t[2]) - 1, t[5] == "-", cig))
    return out
