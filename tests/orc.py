"""ctypes binding of the CPU oracle (oracle/liborc.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from locked_make import locked_make
from typing import List, Optional, Sequence

import numpy as np

from tophat_amd.batch import Events, JUNC_DTYPE, SegBatch
from tophat_amd.params import Params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")


def _lib():
    so = os.path.join(ORC_DIR, "liborc.so")
    srcs = [os.path.join(ORC_DIR, f) for f in os.listdir(ORC_DIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        locked_make(ORC_DIR)
    return C.CDLL(so)


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "segment_length", "segment_mismatches", "min_segment_intron", "max_segment_intron",
        "max_insertion_length", "max_deletion_length", "max_seg_multihits", "inner_dist_mean",
        "inner_dist_std_dev", "library_type", "bowtie2", "read_side")]


class OrcGenome(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("seq", C.POINTER(C.c_char_p)), ("len", C.POINTER(C.c_int64))]


class OrcBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("nseg", C.c_int32), ("read_id", C.c_void_p), ("read_off", C.c_void_p),
                ("bases", C.c_void_p), ("seg_off", C.c_void_p), ("hits", C.c_void_p),
                ("mate_off", C.c_void_p), ("mate_hits", C.c_void_p)]


class OrcIns(C.Structure):
    _fields_ = [("ref_id", C.c_uint32), ("left", C.c_uint32), ("seq", C.c_char * 16), ("prio", C.c_uint64)]


class OrcEvents(C.Structure):
    _fields_ = [("juncs", C.c_void_p), ("n_juncs", C.c_int64), ("deletions", C.c_void_p), ("n_deletions", C.c_int64),
                ("insertions", C.POINTER(OrcIns)), ("n_insertions", C.c_int64),
                ("n_windows", C.c_int64), ("n_indel_pairs", C.c_int64), ("n_rescue_pairs", C.c_int64)]


def orc_params(p: Params) -> OrcParams:
    o = OrcParams()
    for n, _ in OrcParams._fields_:
        setattr(o, n, int(getattr(p, n)))
    return o


class Genome:
    """ASCII genome for the oracle.  seqs[i] is ref_id i+1; None = no FASTA record."""

    def __init__(self, seqs: Sequence[Optional[str]]):
        self.seqs = list(seqs)
        self._bufs = [None if s is None else s.encode() for s in self.seqs]
        n = len(self.seqs)
        self._arr = (C.c_char_p * n)(*self._bufs)
        self._len = (C.c_int64 * n)(*[0 if s is None else len(s) for s in self.seqs])
        self.c = OrcGenome(n, C.cast(self._arr, C.POINTER(C.c_char_p)), C.cast(self._len, C.POINTER(C.c_int64)))


def fold_genome_char(s: str) -> str:
    """char -> Dna5 as SeqAn does when loading the FASTA: acgt upper-cased, all else N."""
    return "".join(c if c in "ACGT" else "N" for c in s.upper())


def segjuncs(p: Params, g: Genome, b: SegBatch) -> Events:
    lib = _lib()
    ob = OrcBatch()
    ob.n_reads, ob.nseg = b.n_reads, b.nseg
    keep = [np.ascontiguousarray(b.read_id, dtype=np.uint32), np.ascontiguousarray(b.read_off, dtype=np.int64),
            np.ascontiguousarray(b.bases, dtype=np.uint8), np.ascontiguousarray(b.seg_off, dtype=np.int64),
            np.ascontiguousarray(b.hits)]
    ob.read_id, ob.read_off, ob.bases, ob.seg_off, ob.hits = [a.ctypes.data for a in keep]
    if b.mate_off is not None:
        keep += [np.ascontiguousarray(b.mate_off, dtype=np.int64), np.ascontiguousarray(b.mate_hits)]
        ob.mate_off, ob.mate_hits = keep[-2].ctypes.data, keep[-1].ctypes.data
    ev = OrcEvents()
    op = orc_params(p)
    rc = lib.orc_segjuncs_batch(C.byref(op), C.byref(g.c), C.byref(ob), C.byref(ev))
    assert rc == 0

    def jarr(ptr, n):
        if n == 0:
            return np.zeros(0, dtype=JUNC_DTYPE)
        buf = (C.c_char * (n * 16)).from_address(ptr)
        return np.frombuffer(buf, dtype=JUNC_DTYPE).copy()

    out = Events(jarr(ev.juncs, ev.n_juncs), jarr(ev.deletions, ev.n_deletions),
                 [(int(ev.insertions[i].ref_id), int(ev.insertions[i].left), ev.insertions[i].seq.decode())
                  for i in range(ev.n_insertions)],
                 {"windows": int(ev.n_windows), "indel_pairs": int(ev.n_indel_pairs),
                  "rescue_pairs": int(ev.n_rescue_pairs)})
    lib.orc_events_free(C.byref(ev))
    return out


# ---------------------------------------------------------------- long_spanning_reads

class OrcSpanParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "segment_length", "max_insertion_length", "max_deletion_length", "min_report_intron", "max_report_intron",
        "max_seg_multihits", "read_mismatches", "read_gap_length", "read_edit_dist", "bowtie2",
        "bowtie2_max_penalty", "bowtie2_min_penalty", "bowtie2_penalty_for_N",
        "bowtie2_read_gap_open", "bowtie2_read_gap_cont", "bowtie2_ref_gap_open", "bowtie2_ref_gap_cont")]


class OrcSpanBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("nseg", C.c_int32), ("read_off", C.c_void_p), ("bases", C.c_void_p),
                ("quals", C.c_void_p), ("seg_off", C.c_void_p), ("hits", C.c_void_p)]


class OrcInsIn(C.Structure):
    _fields_ = [("ref_id", C.c_uint32), ("left", C.c_uint32), ("seq", C.c_char * 16)]


class OrcAln(C.Structure):
    _fields_ = [("read_idx", C.c_int32), ("ref_id", C.c_uint32), ("left", C.c_int32),
                ("antisense", C.c_uint8), ("antisense_splice", C.c_uint8), ("mismatches", C.c_uint8), ("edit_dist", C.c_uint8),
                ("n_cigar", C.c_int32), ("cigar", C.c_uint32 * 24),
                ("AS", C.c_int32), ("XM", C.c_int32), ("XO", C.c_int32), ("XG", C.c_int32), ("md", C.c_char * 96)]


def _md(lib, raw: bytes) -> str:
    """a record's MD: in the record, or -- longer than its field -- in the oracle's per-thread pool (thj_oracle.h)"""
    if raw[:1] != b"\x01":
        return raw.decode()
    lib.orc_long_md.restype = C.c_char_p
    lib.orc_long_md.argtypes = [C.c_int64]
    return lib.orc_long_md(int(raw[1:])).decode()


def spanning(p: Params, g: Genome, b, juncs: np.ndarray, insertions) -> list:
    """-> list of tophat_amd.batch.Aln in output order"""
    from tophat_amd.batch import Aln
    lib = _lib()
    op = OrcSpanParams()
    for n, _ in OrcSpanParams._fields_:
        setattr(op, n, int(getattr(p, n)))
    ob = OrcSpanBatch()
    ob.n_reads, ob.nseg = b.n_reads, b.nseg
    keep = [np.ascontiguousarray(b.read_off, dtype=np.int64), np.ascontiguousarray(b.bases, dtype=np.uint8),
            np.ascontiguousarray(b.quals, dtype=np.uint8), np.ascontiguousarray(b.seg_off, dtype=np.int64),
            np.ascontiguousarray(b.hits)]
    ob.read_off, ob.bases, ob.quals, ob.seg_off, ob.hits = [a.ctypes.data for a in keep]
    j = np.ascontiguousarray(juncs, dtype=JUNC_DTYPE)
    ins = (OrcInsIn * max(1, len(insertions)))()
    for k, (ref, left, seq) in enumerate(insertions):
        ins[k].ref_id, ins[k].left, ins[k].seq = ref, left, seq.encode()
    out = C.POINTER(OrcAln)()
    n_out = C.c_int64()
    lib.orc_long_md_reset()
    rc = lib.orc_spanning_batch(C.byref(op), C.byref(g.c), C.byref(ob), C.c_void_p(j.ctypes.data), C.c_int64(len(j)),
                                ins, C.c_int64(len(insertions)), C.byref(out), C.byref(n_out))
    assert rc == 0
    res = []
    for k in range(n_out.value):
        a = out[k]
        res.append(Aln(a.read_idx, a.ref_id, a.left, bool(a.antisense), bool(a.antisense_splice), a.mismatches, a.edit_dist,
                       tuple(a.cigar[i] for i in range(a.n_cigar)), a.AS, a.XM, a.XO, a.XG, _md(lib, a.md)))
    lib.orc_free(out)
    return res


class OrcFAln(C.Structure):
    _fields_ = [("read_idx", C.c_int32), ("ref_id", C.c_uint32), ("ref_id2", C.c_uint32), ("left", C.c_int32),
                ("antisense", C.c_uint8), ("antisense_splice", C.c_uint8), ("mismatches", C.c_uint8), ("edit_dist", C.c_uint8),
                ("n_cigar", C.c_int32), ("cigar", C.c_uint32 * 32),
                ("AS", C.c_int32), ("XM", C.c_int32), ("XO", C.c_int32), ("XG", C.c_int32), ("md", C.c_char * 128)]


SPAN_FUSION_DTYPE = np.dtype([("ref_id1", "<u4"), ("ref_id2", "<u4"), ("left", "<u4"), ("right", "<u4"), ("dir", "<u4")])


def read_fusions_file(path: str, ref_ids) -> np.ndarray:
    """the .fusions list as long_spanning_reads loads it (long_spanning_reads.cpp:2998-3040), in Fusion::operator< order"""
    rows = set()
    for line in open(path):
        t = line.rstrip("\n").split("\t")
        if len(t) < 5:
            continue
        d = {"fr": 8, "rf": 9, "rr": 10}.get(t[4], 7)
        rows.add((ref_ids.get(t[0], 0), ref_ids.get(t[2], 0), int(t[1]) & 0xFFFFFFFF, int(t[3]) & 0xFFFFFFFF, d))
    return np.array(sorted(rows), dtype=SPAN_FUSION_DTYPE) if rows else np.zeros(0, dtype=SPAN_FUSION_DTYPE)


def spanning_fusion(p: Params, g: Genome, b, juncs: np.ndarray, insertions, fusions: np.ndarray, fusion_search: bool = True) -> list:
    """long_spanning_reads with its fusion branches (spanning_fusion_oracle.c) -> list of tophat_amd.batch.Aln"""
    from tophat_amd.batch import Aln
    lib = _lib()
    op = OrcSpanParams()
    for n, _ in OrcSpanParams._fields_:
        setattr(op, n, int(getattr(p, n)))
    ob = OrcSpanBatch()
    ob.n_reads, ob.nseg = b.n_reads, b.nseg
    keep = [np.ascontiguousarray(b.read_off, dtype=np.int64), np.ascontiguousarray(b.bases, dtype=np.uint8),
            np.ascontiguousarray(b.quals, dtype=np.uint8), np.ascontiguousarray(b.seg_off, dtype=np.int64),
            np.ascontiguousarray(b.hits)]
    ob.read_off, ob.bases, ob.quals, ob.seg_off, ob.hits = [a.ctypes.data for a in keep]
    j = np.ascontiguousarray(juncs, dtype=JUNC_DTYPE)
    ins = (OrcInsIn * max(1, len(insertions)))()
    for k, (ref, left, seq) in enumerate(insertions):
        ins[k].ref_id, ins[k].left, ins[k].seq = ref, left, seq.encode()
    f = np.ascontiguousarray(fusions, dtype=SPAN_FUSION_DTYPE)
    out = C.POINTER(OrcFAln)()
    n_out = C.c_int64()
    lib.orc_long_md_reset()
    rc = lib.orc_spanning_batch_fusion(C.byref(op), C.c_int(1 if fusion_search else 0), C.c_int(int(p.fusion_min_dist)), C.byref(g.c),
                                       C.byref(ob), C.c_void_p(j.ctypes.data), C.c_int64(len(j)), ins, C.c_int64(len(insertions)),
                                       C.c_void_p(f.ctypes.data), C.c_int64(len(f)), C.byref(out), C.byref(n_out))
    assert rc == 0
    res = []
    for k in range(n_out.value):
        a = out[k]
        cig = tuple(a.cigar[i] for i in range(a.n_cigar))
        fused = any((c >> 28) in (7, 8, 9, 10) for c in cig)
        res.append(Aln(a.read_idx, a.ref_id, a.left, bool(a.antisense), bool(a.antisense_splice), a.mismatches, a.edit_dist,
                       cig, a.AS, a.XM, a.XO, a.XG, _md(lib, a.md), a.ref_id2 if fused else 0))
    lib.orc_free(out)
    return res


def spanning_count(p: Params, g: Genome, b, juncs: np.ndarray, insertions) -> int:
    """Runs the spanning oracle and returns only the record count (bench.py's cpu_baseline leg:
    avoids building Python objects for millions of records)."""
    lib = _lib()
    op = OrcSpanParams()
    for n, _ in OrcSpanParams._fields_:
        setattr(op, n, int(getattr(p, n)))
    ob = OrcSpanBatch()
    ob.n_reads, ob.nseg = b.n_reads, b.nseg
    keep = [np.ascontiguousarray(b.read_off, dtype=np.int64), np.ascontiguousarray(b.bases, dtype=np.uint8),
            np.ascontiguousarray(b.quals, dtype=np.uint8), np.ascontiguousarray(b.seg_off, dtype=np.int64),
            np.ascontiguousarray(b.hits)]
    ob.read_off, ob.bases, ob.quals, ob.seg_off, ob.hits = [a.ctypes.data for a in keep]
    j = np.ascontiguousarray(juncs, dtype=JUNC_DTYPE)
    ins = (OrcInsIn * max(1, len(insertions)))()
    for k, (ref, left, seq) in enumerate(insertions):
        ins[k].ref_id, ins[k].left, ins[k].seq = ref, left, seq.encode()
    out = C.POINTER(OrcAln)()
    n_out = C.c_int64()
    lib.orc_long_md_reset()
    rc = lib.orc_spanning_batch(C.byref(op), C.byref(g.c), C.byref(ob), C.c_void_p(j.ctypes.data), C.c_int64(len(j)),
                                ins, C.c_int64(len(insertions)), C.byref(out), C.byref(n_out))
    assert rc == 0
    lib.orc_free(out)
    return int(n_out.value)


# ---------------------------------------------------------------- fusion search
FUSION_DTYPE = np.dtype([("ref_id1", "<u4"), ("ref_id2", "<u4"), ("left", "<u4"), ("right", "<u4"), ("dir", "<u4"),
                         ("count", "<u4"), ("edit_dist", "<u4"), ("skip", "<u4")])


def fusions(p: Params, g: Genome, b: SegBatch, fusion_anchor_length: int = 20, fusion_min_dist: int = 10000000,
            ignore_ref_ids=()) -> np.ndarray:
    lib = _lib()
    ob = OrcBatch()
    ob.n_reads, ob.nseg = b.n_reads, b.nseg
    keep = [np.ascontiguousarray(b.read_id, dtype=np.uint32), np.ascontiguousarray(b.read_off, dtype=np.int64),
            np.ascontiguousarray(b.bases, dtype=np.uint8), np.ascontiguousarray(b.seg_off, dtype=np.int64),
            np.ascontiguousarray(b.hits)]
    ob.read_id, ob.read_off, ob.bases, ob.seg_off, ob.hits = [a.ctypes.data for a in keep]
    if b.mate_off is not None:
        keep += [np.ascontiguousarray(b.mate_off, dtype=np.int64), np.ascontiguousarray(b.mate_hits)]
        ob.mate_off, ob.mate_hits = keep[-2].ctypes.data, keep[-1].ctypes.data
    op = orc_params(p)
    out = C.c_void_p()
    n = C.c_int64()
    ign = np.ascontiguousarray(list(ignore_ref_ids), dtype=np.uint32)
    rc = lib.orc_fusions_batch(C.byref(op), fusion_anchor_length, fusion_min_dist, C.byref(g.c), C.byref(ob),
                               C.c_void_p(ign.ctypes.data), len(ign), C.byref(out), C.byref(n))
    assert rc == 0
    if n.value == 0:
        return np.zeros(0, dtype=FUSION_DTYPE)
    a = np.frombuffer((C.c_char * (n.value * 32)).from_address(out.value), dtype=FUSION_DTYPE).copy()
    lib.orc_free(out)
    return a


def merge_fusions(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """merge_with(FusionSimpleSet&, ...) fusions.cpp:975-990: counts add, edit_dist = min"""
    d = {}
    for x in list(a) + list(b):
        k = (int(x["ref_id1"]), int(x["ref_id2"]), int(x["left"]), int(x["right"]), int(x["dir"]))
        if k in d:
            d[k] = (d[k][0] + int(x["count"]), min(d[k][1], int(x["edit_dist"])))
        else:
            d[k] = (int(x["count"]), int(x["edit_dist"]))
    out = np.zeros(len(d), dtype=FUSION_DTYPE)
    for i, k in enumerate(sorted(d)):
        out[i] = k + d[k] + (0,)
    return out


def fusion_filter(f: np.ndarray, juncs: np.ndarray) -> np.ndarray:
    lib = _lib()
    f = np.ascontiguousarray(f.copy())
    j = np.ascontiguousarray(juncs, dtype=JUNC_DTYPE)
    lib.orc_fusion_filter(C.c_void_p(f.ctypes.data), C.c_int64(len(f)), C.c_void_p(j.ctypes.data), C.c_int64(len(j)))
    return f


def write_fusions(f: np.ndarray, names, path: str):
    """segment_juncs.cpp:5160-5180"""
    DIR = {7: "ff", 8: "fr", 9: "rf", 10: "rr"}
    with open(path, "w") as fh:
        for x in f:
            if x["skip"]:
                continue
            fh.write("%s\t%d\t%s\t%d\t%s\n" % (names[x["ref_id1"] - 1], np.int32(x["left"]), names[x["ref_id2"] - 1], np.int32(x["right"]),
                                               DIR[int(x["dir"])]))


def juncs_db_text(names, g: Genome, juncs_file, ins_file, del_file, fus_file, read_len: int, min_anchor_len: int) -> str:
    """orc_juncs_db over the coordinate files, read the way juncs_db.cpp:298-470 reads them (std::set orders,
    first insertion of a (ref, left, length) wins, insertions with ambiguity codes dropped, deletion left - 1)."""
    lib = _lib()
    ids = {n: i + 1 for i, n in enumerate(names)}

    def lines(fn):
        if not fn or fn == "/dev/null":
            return []
        return [l.rstrip("\n").split("\t") for l in open(fn) if l.rstrip("\n")]
    jset = sorted({(ids[t[0]], int(t[1]), int(t[2]), 1 if t[3][0] == "-" else 0) for t in lines(juncs_file)})
    dset = sorted({(ids[t[0]], int(t[1]) - 1, int(t[2]), 0) for t in lines(del_file)})
    ins = {}
    for t in lines(ins_file):
        seq = t[3].upper()
        if any(c not in "ACGT" for c in seq):
            continue
        ins.setdefault((ids[t[0]], int(t[1]), len(seq)), seq)
    iset = sorted(ins.items())
    dirs = {"ff": 7, "fr": 8, "rf": 9, "rr": 10}
    fset = sorted({(ids[t[0]], ids[t[2]], int(t[1]), int(t[3]), dirs.get(t[4], 7)) for t in lines(fus_file)})
    ja = np.array(jset, dtype=JUNC_DTYPE) if jset else np.zeros(0, dtype=JUNC_DTYPE)
    da = np.array(dset, dtype=JUNC_DTYPE) if dset else np.zeros(0, dtype=JUNC_DTYPE)
    fa = np.zeros(len(fset), dtype=FUSION_DTYPE)
    for k, f in enumerate(fset):
        fa[k]["ref_id1"], fa[k]["ref_id2"], fa[k]["left"], fa[k]["right"], fa[k]["dir"] = f
    iref = np.array([k[0][0] for k in iset], dtype=np.uint32)
    ileft = np.array([k[0][1] for k in iset], dtype=np.uint32)
    iseq = (C.c_char_p * max(1, len(iset)))(*[k[1].encode() for k in iset])
    nm = (C.c_char_p * len(names))(*[n.encode() for n in names])
    lib.orc_juncs_db.restype = C.c_void_p
    out = lib.orc_juncs_db(C.byref(g.c), nm, read_len, min_anchor_len, C.c_void_p(ja.ctypes.data), C.c_int64(len(ja)),
                           C.c_void_p(da.ctypes.data), C.c_int64(len(da)), C.c_void_p(iref.ctypes.data), C.c_void_p(ileft.ctypes.data),
                           iseq, C.c_int64(len(iset)), C.c_void_p(fa.ctypes.data), C.c_int64(len(fa)))
    text = C.string_at(out).decode()
    lib.orc_free(C.c_void_p(out))
    return text


def coverage_search(g: Genome, hits: np.ndarray, ium_reads, min_cov_length: int = 20, min_intron: int = 50, max_intron: int = 20000,
                    max_juncs: int = 5000000) -> np.ndarray:
    """orc_coverage_search (segment_juncs.cpp:4268-4543): hits = HIT_DTYPE records of every segment map of both sides,
    ium_reads = sequences of the initially unmapped reads -> JUNC_DTYPE array in Junction order"""
    from tophat_amd.batch import HIT_DTYPE
    lib = _lib()
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    bases = "".join(ium_reads).encode()
    off = np.zeros(len(ium_reads) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in ium_reads], out=off[1:])
    out = C.c_void_p()
    n = C.c_int64()
    rc = lib.orc_coverage_search(C.byref(g.c), C.c_void_p(h.ctypes.data), C.c_int64(len(h)), C.c_char_p(bases), C.c_void_p(off.ctypes.data),
                                 C.c_int64(len(ium_reads)), min_cov_length, min_intron, max_intron, C.c_int64(max_juncs), C.byref(out), C.byref(n))
    assert rc == 0
    a = np.zeros(0, dtype=JUNC_DTYPE)
    if n.value:
        a = np.frombuffer((C.c_char * (n.value * 16)).from_address(out.value), dtype=JUNC_DTYPE).copy()
    lib.orc_free(out)
    return a


def butterfly_search(g: Genome, hits: np.ndarray, ium_reads, min_intron: int = 50, max_intron: int = 20000, max_juncs: int = 5000000) -> np.ndarray:
    """orc_butterfly_search (segment_juncs.cpp:4178-4249, :1698-2049): the inputs of coverage_search -> JUNC_DTYPE array in Junction order"""
    from tophat_amd.batch import HIT_DTYPE
    lib = _lib()
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    bases = "".join(ium_reads).encode()
    off = np.zeros(len(ium_reads) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in ium_reads], out=off[1:])
    out = C.c_void_p()
    n = C.c_int64()
    rc = lib.orc_butterfly_search(C.byref(g.c), C.c_void_p(h.ctypes.data), C.c_int64(len(h)), C.c_char_p(bases), C.c_void_p(off.ctypes.data),
                                  C.c_int64(len(ium_reads)), int(min_intron), int(max_intron), C.c_int64(max_juncs), C.byref(out), C.byref(n))
    assert rc == 0
    a = np.zeros(0, dtype=JUNC_DTYPE)
    if n.value:
        a = np.frombuffer((C.c_char * (n.value * 16)).from_address(out.value), dtype=JUNC_DTYPE).copy()
    lib.orc_free(out)
    return a


def _orc_batch(b: SegBatch):
    ob = OrcBatch()
    ob.n_reads, ob.nseg = b.n_reads, b.nseg
    keep = [np.ascontiguousarray(b.read_id, dtype=np.uint32), np.ascontiguousarray(b.read_off, dtype=np.int64),
            np.ascontiguousarray(b.bases, dtype=np.uint8), np.ascontiguousarray(b.seg_off, dtype=np.int64),
            np.ascontiguousarray(b.hits)]
    ob.read_id, ob.read_off, ob.bases, ob.seg_off, ob.hits = [a.ctypes.data for a in keep]
    return ob, keep


def microexon_search(p: Params, g: Genome, batches, min_anchor_len: int = 8, min_intron: int = 50, max_juncs: int = 5000000):
    """orc_microexon_search (segment_juncs.cpp:3737-3941): batches = [(SegBatch, side)] in visiting order (left side first)
    -> (JUNC_DTYPE array in Junction order, number of windows)"""
    lib = _lib()
    obs = [_orc_batch(b) for b, _ in batches]
    arr = (C.POINTER(OrcBatch) * max(1, len(obs)))(*[C.pointer(o[0]) for o in obs])
    sides = (C.c_int * max(1, len(obs)))(*[sd for _, sd in batches])
    op = orc_params(p)
    out = C.c_void_p()
    n, nw = C.c_int64(), C.c_int64()
    rc = lib.orc_microexon_search(C.byref(op), int(min_anchor_len), int(min_intron), C.c_int64(max_juncs), C.byref(g.c), arr, sides, len(obs),
                                  C.byref(out), C.byref(n), C.byref(nw))
    assert rc == 0
    a = np.zeros(0, dtype=JUNC_DTYPE)
    if n.value:
        a = np.frombuffer((C.c_char * (n.value * 16)).from_address(out.value), dtype=JUNC_DTYPE).copy()
    lib.orc_free(out)
    return a, nw.value


# ---- junction consensus of tophat_reports (juncbed_oracle.c)
JREC_DTYPE = np.dtype([("ref_id", "<u4"), ("left", "<i4"), ("antisense_splice", "u1"), ("n_cigar", "u1"), ("reserved", "<u2"), ("cigar", "<u4", 16)])
JSTAT_DTYPE = np.dtype([("ref_id", "<u4"), ("left", "<u4"), ("right", "<u4"), ("antisense", "<u4"), ("left_extent", "<u4"),
                        ("right_extent", "<u4"), ("support", "<u4"), ("reserved", "<u4")])


def jrecs_from_tuples(recs) -> np.ndarray:
    """[(ref_id, left, antisense_splice, [(op, len) ...][, ref_id2])] -> JREC_DTYPE array (a fusion record: at most 15 ops, its second
    contig in cigar[15] as in thj_aln)"""
    a = np.zeros(len(recs), dtype=JREC_DTYPE)
    for k, rec in enumerate(recs):
        ref, left, anti, cig = rec[:4]
        a[k]["ref_id"], a[k]["left"], a[k]["antisense_splice"], a[k]["n_cigar"] = ref, left, 1 if anti else 0, len(cig)
        for i, (op, ln) in enumerate(cig):
            a[k]["cigar"][i] = (op << 28) | ln
        if len(rec) > 4 and rec[4]:
            assert len(cig) <= 15
            a[k]["cigar"][15] = rec[4]
    return a


def jrecs_from_alns(alns) -> np.ndarray:
    """tophat_amd.batch.Aln list (a long_spanning_reads result) -> JREC_DTYPE array"""
    return jrecs_from_tuples([(a.ref_id, a.left, a.antisense_splice, [(c >> 28, c & 0x0FFFFFFF) for c in a.cigar], a.ref_id2) for a in alns])


def junction_consensus(jrecs: np.ndarray, min_anchor_len: int = 8) -> np.ndarray:
    lib = _lib()
    out = C.c_void_p()
    n = C.c_int64()
    a = np.ascontiguousarray(jrecs, dtype=JREC_DTYPE)
    rc = lib.orc_junction_consensus(C.c_void_p(a.ctypes.data), C.c_int64(len(a)), min_anchor_len, C.byref(out), C.byref(n))
    assert rc == 0
    res = np.ctypeslib.as_array((C.c_uint8 * (n.value * JSTAT_DTYPE.itemsize)).from_address(out.value)).view(JSTAT_DTYPE).copy() if n.value else \
        np.zeros(0, dtype=JSTAT_DTYPE)
    lib.orc_free(out)
    return res


def junctions_bed(jstats: np.ndarray, names) -> str:
    lib = _lib()
    lib.orc_junctions_bed.restype = C.c_void_p
    a = np.ascontiguousarray(jstats, dtype=JSTAT_DTYPE)
    nm = (C.c_char_p * len(names))(*[n.encode() for n in names])
    p = lib.orc_junctions_bed(C.c_void_p(a.ctypes.data), C.c_int64(len(a)), nm)
    s = C.string_at(p).decode()
    lib.orc_free(C.c_void_p(p))
    return s
