"""Host-side batch model for the segment_juncs / long_spanning_reads hot path.

A `SegBatch` is the sequence of `hits_for_read` vectors that the reference's
stream synchroniser hands to its per-read finders, flattened to SoA arrays:

* reference: look_for_hit_group / process_next_hit_group
  (segment_juncs.cpp:3823-4123) walk `nseg` id-sorted hit streams from the last
  segment backwards.  Tracing the recursion (including the EOF call with
  insert_id 0, whose observation_order is VMAXINT32, bwt_map.h:556-561) shows
  that every read id that has a hit in ANY segment stream is visited exactly
  once, in increasing id order, with the hits of each segment it has;
  find_insertions_and_deletions + find_gaps run for a visited read unless its
  highest mapped segment is segment 0 (`curr_file > 0`, :3994-4012), and the
  partner streams consumed by find_gaps (:3321-3348) reduce to an id join.
  `build_seg_batch` implements exactly that visiting set.

Layouts (all little-endian, C-contiguous):
  HIT_DTYPE   16 B/hit  {ref_id u32, left i32, right i32, flags u8, edit_dist u8,
                         mismatches u8, read_len u8}   == thj_hit (include/thj.h)
  seg_off     u32[n_reads*nseg+1]  CSR over hits, index r*nseg+s
  mate_off    u32[n_reads+1]       CSR over mate_hits
  read_off    i64[n_reads+1]       into `bases` (ASCII, as ReadStream returns it)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

HIT_DTYPE = np.dtype([
    ("ref_id", "<u4"), ("left", "<i4"), ("right", "<i4"),
    ("flags", "u1"), ("edit_dist", "u1"), ("mismatches", "u1"), ("read_len", "u1"),
])
assert HIT_DTYPE.itemsize == 16

HIT_ANTISENSE = 1
HIT_END = 2


@dataclass
class SegBatch:
    nseg: int
    read_id: np.ndarray            # u32[n]
    read_off: np.ndarray           # i64[n+1]
    bases: np.ndarray              # u8[...]
    seg_off: np.ndarray            # u32[n*nseg+1]
    hits: np.ndarray               # HIT_DTYPE[...]
    mate_off: Optional[np.ndarray] = None   # u32[n+1]
    mate_hits: Optional[np.ndarray] = None  # HIT_DTYPE[...]

    @property
    def n_reads(self) -> int:
        return int(self.read_id.shape[0])

    def read_seq(self, r: int) -> str:
        return self.bases[self.read_off[r]:self.read_off[r + 1]].tobytes().decode()

    def seg_hits(self, r: int, s: int) -> np.ndarray:
        k = r * self.nseg + s
        return self.hits[self.seg_off[k]:self.seg_off[k + 1]]

    def select(self, mask: np.ndarray) -> "SegBatch":
        """the sub-batch of the reads where mask is True, order kept (a shard of the reads)"""
        idx = np.flatnonzero(mask)
        read_off, seg_off, mate_off = [0], [0], [0]
        bases, hits, mates = [], [], []
        for r in idx:
            bases.append(self.bases[self.read_off[r]:self.read_off[r + 1]])
            read_off.append(read_off[-1] + len(bases[-1]))
            for s in range(self.nseg):
                h = self.seg_hits(int(r), s)
                hits.append(h)
                seg_off.append(seg_off[-1] + len(h))
            if self.mate_off is not None:
                m = self.mate_hits[self.mate_off[r]:self.mate_off[r + 1]]
                mates.append(m)
                mate_off.append(mate_off[-1] + len(m))
        cat = lambda parts, like: np.concatenate(parts) if parts else like[:0]     # noqa: E731
        return SegBatch(self.nseg, self.read_id[idx], np.asarray(read_off, dtype=np.int64), cat(bases, self.bases),
                        np.asarray(seg_off, dtype=np.uint32), cat(hits, self.hits),
                        None if self.mate_off is None else np.asarray(mate_off, dtype=np.uint32),
                        None if self.mate_off is None else cat(mates, self.mate_hits))


# One parsed alignment record of a segment / read map, before grouping.
# (read_id, ref_id, left, right, antisense, end, mismatches, edit_dist, read_len)
HitRec = tuple   # + optional (cigar [(op,len)...], antisense_splice) for long_spanning_reads


def hit_tuple_to_struct(h: HitRec) -> tuple:
    _, ref_id, left, right, anti, end, mm, ed, rl = h[:9]
    return (ref_id, left, right, (HIT_ANTISENSE if anti else 0) | (HIT_END if end else 0),
            ed & 0xFF, mm & 0xFF, min(rl, 255))


def group_by_id(recs: Iterable[HitRec]) -> Dict[int, List[HitRec]]:
    """HitStream::next_read_hits (bwt_map.h:1155-1220): consecutive records with
    equal insert_id form one group.  Files are id-sorted, so a dict keyed by id
    preserving file order within the id is equivalent."""
    out: Dict[int, List[HitRec]] = {}
    for h in recs:
        out.setdefault(h[0], []).append(h)
    return out


def build_seg_batch(seg_recs: Sequence[Iterable[HitRec]],
                    reads: Dict[int, str],
                    mate_map_recs: Optional[Iterable[HitRec]] = None,
                    mate_lastseg_recs: Optional[Iterable[HitRec]] = None, include_top0: bool = False) -> SegBatch:
    """seg_recs[s] = records of segment-s map in file order; reads = id -> sequence.

    mate_map_recs / mate_lastseg_recs are the mate side's full-read map and
    last-segment map (the `partner_hit_stream` / `seg_partner_hit_stream` of
    find_gaps): the mate group is the full-read group when one exists for the id,
    otherwise the last-segment group (segment_juncs.cpp:3324-3348)."""
    nseg = len(seg_recs)
    groups = [group_by_id(r) for r in seg_recs]
    ids = sorted(set().union(*[g.keys() for g in groups])) if groups else []
    mate_full = group_by_id(mate_map_recs) if mate_map_recs is not None else {}
    mate_last = group_by_id(mate_lastseg_recs) if mate_lastseg_recs is not None else {}
    have_mates = mate_map_recs is not None or mate_lastseg_recs is not None

    read_id: List[int] = []
    read_off = [0]
    bases = bytearray()
    seg_off = [0]
    hits: List[tuple] = []
    mate_off = [0]
    mate_hits: List[tuple] = []
    for rid in ids:
        if rid == 0:
            continue  # insert_id 0 is "no group" (bwt_map.h:1174-1176)
        top = max(s for s in range(nseg) if rid in groups[s])
        if top == 0 and not include_top0:
            continue  # only find_fusions runs for these (segment_juncs.cpp:3994-4028); event-neutral for the other finders
        if rid not in reads:
            raise KeyError("could not get read# %d from stream" % rid)  # :3352-3356
        read_id.append(rid)
        seq = reads[rid]
        bases += seq.encode()
        read_off.append(len(bases))
        for s in range(nseg):
            for h in groups[s].get(rid, ()):
                hits.append(hit_tuple_to_struct(h))
            seg_off.append(len(hits))
        if have_mates:
            mg = mate_full.get(rid) or mate_last.get(rid) or ()
            for h in mg:
                mate_hits.append(hit_tuple_to_struct(h))
            mate_off.append(len(mate_hits))
    b = SegBatch(
        nseg=nseg,
        read_id=np.asarray(read_id, dtype=np.uint32),
        read_off=np.asarray(read_off, dtype=np.int64),
        bases=np.frombuffer(bytes(bases), dtype=np.uint8).copy(),
        seg_off=np.asarray(seg_off, dtype=np.uint32),
        hits=np.array(hits, dtype=HIT_DTYPE) if hits else np.zeros(0, dtype=HIT_DTYPE),
    )
    if have_mates:
        b.mate_off = np.asarray(mate_off, dtype=np.uint32)
        b.mate_hits = np.array(mate_hits, dtype=HIT_DTYPE) if mate_hits else np.zeros(0, dtype=HIT_DTYPE)
    return b


# ----------------------------------------------------------------- events

JUNC_DTYPE = np.dtype([("ref_id", "<u4"), ("left", "<u4"), ("right", "<u4"), ("antisense", "<u4")])


@dataclass
class Events:
    """Sorted-unique outputs of one segment_juncs pass (one side, one batch)."""
    juncs: np.ndarray                       # JUNC_DTYPE, junctions.h:39-57 order
    deletions: np.ndarray                   # JUNC_DTYPE (antisense always 0)
    insertions: List[Tuple[int, int, str]] = field(default_factory=list)  # (ref_id, left, seq)
    stats: Dict[str, int] = field(default_factory=dict)


def merge_events(a: Events, b: Events) -> Events:
    """Merge in the order the reference inserts into its sets: `a` first.
    Junction/deletion = set union (segment_juncs.cpp:4911-4916); Insertion's
    ordering ignores sequence content (insertions.h:52-67) so the earlier one
    of equal (ref,left,len) survives."""
    def u(x, y):
        if len(x) == 0:
            return y.copy()
        if len(y) == 0:
            return x.copy()
        z = np.concatenate([x, y])
        z = np.unique(z)
        order = np.lexsort((z["antisense"], z["right"], z["left"], z["ref_id"]))
        return z[order]
    seen = {}
    for ins in list(a.insertions) + list(b.insertions):
        k = (ins[0], ins[1], len(ins[2]))
        if k not in seen:
            seen[k] = ins
    ins = [seen[k] for k in sorted(seen)]
    st = dict(a.stats)
    for k, v in b.stats.items():
        st[k] = st.get(k, 0) + v
    return Events(u(a.juncs, b.juncs), u(a.deletions, b.deletions), ins, st)


def write_segment_files(ev: Events, ref_names: Sequence[str],
                        juncs_path: str, ins_path: str, del_path: str, fus_path: Optional[str] = None) -> None:
    """The four text outputs, formats of segment_juncs.cpp:5035-5095."""
    with open(juncs_path, "w") as f:
        for j in ev.juncs:
            f.write("%s\t%d\t%d\t%c\n" % (ref_names[j["ref_id"] - 1], np.int32(j["left"]), np.int32(j["right"]),
                                          "-" if j["antisense"] else "+"))
    with open(del_path, "w") as f:
        for j in ev.deletions:
            f.write("%s\t%d\t%d\n" % (ref_names[j["ref_id"] - 1], np.int32(j["left"]) + 1, np.int32(j["right"])))
    with open(ins_path, "w") as f:
        for (ref, left, seq) in ev.insertions:
            f.write("%s\t%d\t%d\t%s\n" % (ref_names[ref - 1], left, left, seq))
    if fus_path:
        open(fus_path, "w").close()


# ------------------------------------------------------- long_spanning_reads

SPAN_HIT_DTYPE = np.dtype([
    ("ref_id", "<u4"), ("left", "<i4"),
    ("flags", "u1"), ("mismatches", "u1"), ("edit_dist", "u1"), ("n_cigar", "u1"),
    ("cigar", "<u4", (5,)),
])
assert SPAN_HIT_DTYPE.itemsize == 32
HIT_ANTISENSE_SPLICE = 4
HIT_FUSED = 16                  # the hit's cigar holds a fusion op
HIT_STRAND_FLIPPED = 8          # hit on an rf / rr fusion contig: antisense_align is the opposite of the record's strand flag
CIG_FUSION_FF, CIG_FUSION_FR, CIG_FUSION_RF, CIG_FUSION_RR = 7, 8, 9, 10
FUSION_OPS = (7, 8, 9, 10)

CIG_MATCH, CIG_INS, CIG_DEL, CIG_REF_SKIP, CIG_SOFT_CLIP = 1, 3, 5, 11, 13   # bwt_map.h:36-55
CIG_CHARS = {1: "M", 2: "m", 3: "I", 4: "i", 5: "D", 6: "d", 7: "F", 8: "F", 9: "F", 10: "F", 11: "N", 12: "n", 13: "S"}


def cig_pack(op: int, length: int) -> int:
    return ((op & 0xF) << 28) | (length & 0x0FFFFFFF)


def cigar_string(cigar: Sequence[int]) -> str:
    """print_bamhit / GBamRecord::set_cigar: letters are upper-cased in the BAM record"""
    return "".join("%d%s" % (c & 0x0FFFFFFF, CIG_CHARS[c >> 28].upper()) for c in cigar)


@dataclass
class SpanBatch:
    """Per-read segment hit lists for long_spanning_reads: for each read that has a hit in the
    first segment map, segment s holds the contig hits then the spliced hits of that segment
    (long_spanning_reads.cpp:2706-2765 and :87-163)."""
    nseg: int
    read_id: np.ndarray            # u32[n]
    read_off: np.ndarray           # i64[n+1]
    bases: np.ndarray              # u8
    quals: np.ndarray              # u8 (phred+33), same offsets
    seg_off: np.ndarray            # u32[n*nseg+1]
    hits: np.ndarray               # SPAN_HIT_DTYPE

    @property
    def n_reads(self) -> int:
        return int(self.read_id.shape[0])


def span_hit_struct(h: HitRec) -> tuple:
    _, ref_id, left, _right, anti, end, mm, ed, _rl = h[:9]
    cigar = h[9] if len(h) > 9 else [(CIG_MATCH, _rl)]
    asp = bool(h[10]) if len(h) > 10 else False
    fused = any(o in FUSION_OPS for (o, _n) in cigar)
    if len(cigar) > (4 if fused else 5):
        raise ValueError("segment hit with %d CIGAR ops (device path supports <= 5, <= 4 for a fused hit)" % len(cigar))
    cig = [cig_pack(o, n) for (o, n) in cigar] + [0] * (5 - len(cigar))
    flipped = False
    if fused:
        cig[4] = int(h[11])        # ref_id2 rides in the last cigar slot
        flipped = bool(h[12])
    return (ref_id, left, (HIT_ANTISENSE if anti else 0) | (HIT_END if end else 0) | (HIT_ANTISENSE_SPLICE if asp else 0) |
            (HIT_STRAND_FLIPPED if flipped else 0) | (HIT_FUSED if fused else 0), mm & 0xFF, ed & 0xFF, len(cigar), cig)


def build_span_batch(seg_recs: Sequence[Iterable[HitRec]], reads: Dict[int, str], quals: Dict[int, str],
                     spliced_recs: Optional[Sequence[Iterable[HitRec]]] = None) -> SpanBatch:
    nseg = len(seg_recs)
    groups = [group_by_id(r) for r in seg_recs]
    sgroups = [group_by_id(r) for r in spliced_recs] if spliced_recs else [{} for _ in range(nseg)]
    ids = sorted(set(groups[0].keys()) | set(sgroups[0].keys())) if nseg else []
    read_id, read_off, seg_off, hits = [], [0], [0], []
    bases, qs = bytearray(), bytearray()
    for rid in ids:
        if rid == 0:
            continue
        if rid not in reads:
            raise KeyError("could not get read # %d from stream" % rid)      # long_spanning_reads.cpp:2832-2836
        read_id.append(rid)
        bases += reads[rid].encode()
        qs += quals[rid].encode()
        read_off.append(len(bases))
        for s in range(nseg):
            for h in list(groups[s].get(rid, ())) + list(sgroups[s].get(rid, ())):
                hits.append(span_hit_struct(h))
            seg_off.append(len(hits))
    return SpanBatch(nseg, np.asarray(read_id, dtype=np.uint32), np.asarray(read_off, dtype=np.int64),
                     np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.frombuffer(bytes(qs), dtype=np.uint8).copy(),
                     np.asarray(seg_off, dtype=np.uint32),
                     np.array(hits, dtype=SPAN_HIT_DTYPE) if hits else np.zeros(0, dtype=SPAN_HIT_DTYPE))


@dataclass
class Aln:
    """One output record of long_spanning_reads, the fields print_bamhit writes (bwt_map.cpp:1888-2093)."""
    read_idx: int
    ref_id: int
    left: int
    antisense: bool
    antisense_splice: bool
    mismatches: int
    edit_dist: int
    cigar: Tuple[int, ...]
    AS: int
    XM: int
    XO: int
    XG: int
    MD: str
    ref_id2: int = 0           # second contig of a fusion alignment (0: none)

    def _tags(self):
        indel = sum(c & 0x0FFFFFFF for c in self.cigar if (c >> 28) in (3, 4, 5, 6))
        tags = ["AS:i:%d" % self.AS, "XM:i:%d" % self.XM, "XO:i:%d" % self.XO, "XG:i:%d" % self.XG,
                "MD:Z:%s" % self.MD, "NM:i:%d" % (self.mismatches + indel)]
        if any((c >> 28) in (11, 12) for c in self.cigar):
            tags.append("XS:A:%s" % ("-" if self.antisense_splice else "+"))
        return tags

    def sam_fields(self, read_id: int, ref_names: Sequence[str]) -> tuple:
        """(QNAME, FLAG, RNAME, POS, CIGAR, tags...) as in the BAM record"""
        return (str(read_id), 16 if self.antisense else 0, ref_names[self.ref_id - 1], self.left + 1,
                cigar_string(self.cigar)) + tuple(self._tags())

    def is_fusion(self) -> bool:
        return any((c >> 28) in FUSION_OPS for c in self.cigar)

    def sam_records(self, read_id: int, ref_names: Sequence[str], read_seq: str, read_qual: str) -> list:
        """The record(s) print_bamhit writes (bwt_map.cpp:1888-2093) as (QNAME, FLAG, RNAME, POS, CIGAR, SEQ, QUAL, tags...):
        one for a plain alignment, two for a fusion alignment (extract_partial_hits, :2148-2347), each carrying the whole
        alignment in XF:Z."""
        seq, qual = read_seq, read_qual
        if self.antisense:
            seq = seq.translate(_RC)[::-1]
            qual = qual[::-1]
        if not self.is_fusion():
            f = self.sam_fields(read_id, ref_names)
            return [f[:5] + (seq, qual) + f[5:]]
        ops = [(c >> 28, c & 0x0FFFFFFF) for c in self.cigar]
        fi = next(i for i, (o, _n) in enumerate(ops) if o in FUSION_OPS)
        fdir = ops[fi][0]
        right = self.left
        left_part_len = 0
        fusion_left = fusion_right = -1
        for i, (o, n) in enumerate(ops):
            if o in (1, 11, 5):
                right += n
            elif o in (2, 12, 6):
                right -= n
            elif o in FUSION_OPS:
                fusion_left = right - 1 if o in (7, 8) else right + 1
                fusion_right = right = n
            if i < fi and o in (1, 2, 3, 4):
                left_part_len += n
        up = {1: "M", 2: "M", 3: "I", 4: "I", 5: "D", 6: "D", 11: "N", 12: "N"}
        first = ops[:fi] if fdir in (7, 8) else ops[:fi][::-1]
        second = ops[fi + 1:] if fdir in (7, 9) else ops[fi + 1:][::-1]
        cigar1 = "".join("%d%s" % (n, up[o]) for o, n in first)
        cigar2 = "".join("%d%s" % (n, up[o]) for o, n in second)
        seq1, qual1 = seq[:left_part_len], qual[:left_part_len]
        seq2, qual2 = seq[left_part_len:], qual[left_part_len:]
        if fdir in (9, 10):
            seq1, qual1 = seq1.translate(_RC)[::-1], qual1[::-1]
        if fdir in (8, 10):
            seq2, qual2 = seq2.translate(_RC)[::-1], qual2[::-1]
        left1 = self.left if fdir in (7, 8) else fusion_left
        left2 = fusion_right if fdir in (7, 9) else right + 1
        n1, n2 = ref_names[self.ref_id - 1], ref_names[self.ref_id2 - 1]
        full = "".join("%d%s" % (n + 1 if o in FUSION_OPS else n, CIG_CHARS[o]) for o, n in ops)
        xf = "%s-%s %d %s %s %s" % (n1, n2, self.left + 1, full, seq, qual)
        flag = 16 if self.antisense else 0
        tags = self._tags()
        return [(str(read_id), flag, n1, left1 + 1, cigar1, seq1, qual1) + tuple(tags) + ("XF:Z:1 " + xf,),
                (str(read_id), flag, n2, left2 + 1, cigar2, seq2, qual2) + tuple(tags) + ("XF:Z:2 " + xf,)]


_RC = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def events_to_span_inputs(ev: Events):
    """Junction + deletion sets merged the way long_spanning_reads loads them
    (long_spanning_reads.cpp:2897-2944: a deletion line `left+1, right` becomes Junction(left, right, '+'))
    and the insertion list."""
    j = ev.juncs
    if len(ev.deletions):
        j = np.concatenate([j, ev.deletions])
    if len(j):
        j = np.unique(j)
        j = j[np.lexsort((j["antisense"], j["right"], j["left"], j["ref_id"]))]
    return j, list(ev.insertions)
