"""ctypes binding of the C ABI in include/thj.h (libthj_hip.so).

Host-side mirror used by bench.py, the tests and smoke(); the drop-in C++
binaries call the same ABI directly.  There is no fallback: if the HIP library
is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .batch import Events, HIT_DTYPE, JUNC_DTYPE, SegBatch
from .params import CParams, Params

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libthj_hip.so")

INS_CODE = "ACGTN"

FUSION_DTYPE = np.dtype([("ref_id1", "<u4"), ("ref_id2", "<u4"), ("left", "<u4"), ("right", "<u4"), ("dir", "<u4"),
                         ("count", "<u4"), ("edit_dist", "<u4"), ("skip", "<u4")])


class ThjError(RuntimeError):
    pass


class CSegBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("nseg", C.c_int32), ("words_per_plane", C.c_int32), ("reserved", C.c_int32),
                ("seg_off", C.c_void_p), ("hits", C.c_void_p), ("read_planes", C.c_void_p), ("read_len", C.c_void_p),
                ("mate_off", C.c_void_p), ("mate_hits", C.c_void_p), ("ordinal_base", C.c_uint32),
                ("reserved2", C.c_uint32)]


class CJunction(C.Structure):
    _fields_ = [("ref_id", C.c_uint32), ("left", C.c_uint32), ("right", C.c_uint32), ("antisense", C.c_uint32)]


class CInsertion(C.Structure):
    _fields_ = [("ref_id", C.c_uint32), ("left", C.c_uint32), ("seq", C.c_char * 8), ("prio", C.c_uint64)]


class CCounts(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_juncs", "n_deletions", "n_insertions", "n_windows", "n_indel_pairs",
                                         "n_rescue_pairs", "n_overflow_blocks", "n_hits_read")]


ABI_VERSION = 2
ABI_SYMBOLS = [
    "thj_params_default", "thj_last_error", "thj_version", "thj_abi_version", "thj_ctx_stream_info", "thj_ctx_probe_streams",
    "thj_device_count", "thj_ctx_create", "thj_ctx_destroy", "thj_ctx_sync", "thj_ctx_stream",
    "thj_genome_layout", "thj_genome_pack", "thj_genome_upload", "thj_genome_adopt",
    "thj_reads_pack",
    "thj_batch_upload", "thj_batch_free",
    "thj_segjuncs_configure", "thj_segjuncs_reset_async", "thj_segjuncs_run_async", "thj_segjuncs_run_pair_async",
    "thj_segjuncs_finish", "thj_segjuncs_download", "thj_segjuncs_device_keys",
    "thj_segjuncs_merge_keys_async", "thj_profile_segjuncs", "thj_profile_serial",
    "thj_segjuncs_device_insertions", "thj_segjuncs_merge_insertions_async",
    "thj_fusion_reset_async", "thj_fusion_set_ignored", "thj_fusion_run_async", "thj_genome_gather", "thj_fusion_finish", "thj_fusion_download",
    "thj_covsearch_reset_async", "thj_covsearch_add_hits_async", "thj_covsearch_add_reads", "thj_covsearch_run_async", "thj_covsearch_finish",
    "thj_covsearch_device_state", "thj_covsearch_merge_async", "thj_span_hit_heads_async",
    "thj_comm_unique_id", "thj_comm_create", "thj_comm_create_local", "thj_comm_destroy", "thj_comm_info",
    "thj_events_allgather_async", "thj_fusion_allgather", "thj_covsearch_allgather", "thj_span_fusions_upload", "thj_md_string2", "thj_ingest_span_batch", "thj_ingest_timing_report", "thj_pinned_alloc", "thj_pinned_free", "thj_pinned_drain", "thj_ctx_warm",
]

_lib = None


def load_lib(path: Optional[str] = None):
    """Loads libthj_hip.so; raises ThjError if it is not built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ThjError("HIP extension %s not built; run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    lib = C.CDLL(p)
    lib.thj_last_error.restype = C.c_char_p
    lib.thj_version.restype = C.c_char_p
    # the argument layouts this module was written against (include/thj.h: THJ_ABI_VERSION)
    if not hasattr(lib, "thj_abi_version") or lib.thj_abi_version() != ABI_VERSION:
        raise ThjError("%s has ABI revision %s, this module wants %d: rebuild (python -c 'import __graft_entry__ as g; g.build()')" % (
            p, lib.thj_abi_version() if hasattr(lib, "thj_abi_version") else "< 2", ABI_VERSION))
    if hasattr(lib, "thj_ctx_stream"):
        lib.thj_ctx_stream.restype = C.c_void_p
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int, what: str):
    if rc != 0:
        raise ThjError("%s failed (%d): %s" % (what, rc, lib.thj_last_error().decode()))


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else C.c_void_p(a.ctypes.data)


class PackedGenome:
    def __init__(self, blocks: np.ndarray, contig_blk: np.ndarray, lens: np.ndarray):
        self.blocks, self.contig_blk, self.lens = blocks, contig_blk, lens

    @property
    def n_blocks(self) -> int:
        return self.blocks.shape[0] // 4

    @property
    def n_contigs(self) -> int:
        return self.lens.shape[0]

    def decode_gpos(self, gpos: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """global base coordinate -> (ref_id 1-based, pos)"""
        starts = self.contig_blk[:-1].astype(np.int64) * 64
        idx = np.searchsorted(starts, gpos, side="right") - 1
        return (idx + 1).astype(np.uint32), (gpos - starts[idx]).astype(np.int64)


def pack_genome(seqs: Sequence[Optional[str]], lib=None) -> PackedGenome:
    lib = lib or load_lib()
    n = len(seqs)
    lens = np.array([0 if s is None else len(s) for s in seqs], dtype=np.int64)
    contig_blk = np.zeros(n + 1, dtype=np.uint32)
    nb = C.c_int64()
    _check(lib, lib.thj_genome_layout(n, _ptr(lens), _ptr(contig_blk), C.byref(nb)), "thj_genome_layout")
    blocks = np.zeros(nb.value * 4, dtype=np.uint64)
    bufs = [None if s is None else s.encode() for s in seqs]
    arr = (C.c_char_p * n)(*bufs)
    _check(lib, lib.thj_genome_pack(n, arr, _ptr(lens), _ptr(contig_blk), _ptr(blocks), nb), "thj_genome_pack")
    pg = PackedGenome(blocks, contig_blk, lens)
    pg.seqs = list(seqs)            # the host keeps the bases: MD strings the device leaves to it are rebuilt from them
    return pg


def words_per_plane(max_len: int) -> int:
    return max(1, (int(max_len) + 63) // 64)


def pack_reads(b: SegBatch, W: Optional[int] = None, lib=None) -> Tuple[np.ndarray, np.ndarray, int]:
    lib = lib or load_lib()
    n = b.n_reads
    lens_ = np.diff(b.read_off) if n else np.zeros(0, dtype=np.int64)
    if W is None:
        W = words_per_plane(int(lens_.max()) if n else 1)
    planes = np.zeros(n * 3 * W, dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint16)
    bases = np.ascontiguousarray(b.bases, dtype=np.uint8)
    off = np.ascontiguousarray(b.read_off, dtype=np.int64)
    _check(lib, lib.thj_reads_pack(C.c_int64(n), _ptr(off), _ptr(bases), W, _ptr(planes), _ptr(lens)), "thj_reads_pack")
    return planes, lens, W


def host_cbatch(b: SegBatch, ordinal_base: int = 0, lib=None):
    """CSegBatch whose pointers are HOST numpy arrays (+ the arrays, to keep them alive)."""
    planes, lens, W = pack_reads(b, lib=lib)
    keep = [np.ascontiguousarray(b.seg_off, dtype=np.uint32), np.ascontiguousarray(b.hits), planes, lens]
    cb = CSegBatch()
    cb.n_reads, cb.nseg, cb.words_per_plane = b.n_reads, b.nseg, W
    cb.seg_off, cb.hits, cb.read_planes, cb.read_len = [a.ctypes.data for a in keep]
    if b.mate_off is not None:
        keep += [np.ascontiguousarray(b.mate_off, dtype=np.uint32), np.ascontiguousarray(b.mate_hits)]
        cb.mate_off = keep[-2].ctypes.data
        cb.mate_hits = keep[-1].ctypes.data if len(keep[-1]) else keep[-2].ctypes.data
    cb.ordinal_base = ordinal_base
    n_mate = 0 if b.mate_hits is None else len(b.mate_hits)
    return cb, keep, len(b.hits), n_mate


COMM_ID_BYTES = 128
COMM_TRANSPORT = {0: "self", 1: "rccl", 2: "loopback"}


def comm_unique_id(lib=None) -> bytes:
    """ncclGetUniqueId through the C ABI: made by one rank, handed to the others by the caller."""
    lib = lib or load_lib()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib, lib.thj_comm_unique_id(buf), "thj_comm_unique_id")
    return bytes(buf)


class Comm:
    """One rank of the exchange step (thj_comm): bound to one Context."""

    def __init__(self, ctx: "Context", handle: C.c_void_p):
        self.ctx, self._h = ctx, handle

    @staticmethod
    def create(ctx: "Context", unique_id: Optional[bytes], n_ranks: int, rank: int) -> "Comm":
        h = C.c_void_p()
        idbuf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id) if unique_id is not None else None
        _check(ctx.lib, ctx.lib.thj_comm_create(ctx._ctx, idbuf, n_ranks, rank, C.byref(h)), "thj_comm_create")
        return Comm(ctx, h)

    @staticmethod
    def create_local(ctxs: Sequence["Context"]) -> List["Comm"]:
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c._ctx for c in ctxs])
        out = (C.c_void_p * n)()
        _check(ctxs[0].lib, ctxs[0].lib.thj_comm_create_local(arr, n, out), "thj_comm_create_local")
        return [Comm(ctxs[i], C.c_void_p(out[i])) for i in range(n)]

    def info(self):
        n, r, t = C.c_int32(), C.c_int32(), C.c_int32()
        st = (C.c_int64 * 4)()
        _check(self.ctx.lib, self.ctx.lib.thj_comm_info(self._h, C.byref(n), C.byref(r), C.byref(t), st), "thj_comm_info")
        return {"n_ranks": n.value, "rank": r.value, "transport": COMM_TRANSPORT[t.value], "steps": st[0], "repeats": st[1],
                "bytes_per_rank": st[2], "junction_capacity": st[3]}

    def events_allgather(self):
        _check(self.ctx.lib, self.ctx.lib.thj_events_allgather_async(self.ctx._ctx, self._h), "thj_events_allgather_async")

    def fusion_allgather(self) -> np.ndarray:
        """after Context.fusions(...) on every rank: the merged FusionSimpleSet (FUSION_DTYPE array), same on every rank"""
        n = C.c_int64()
        _check(self.ctx.lib, self.ctx.lib.thj_fusion_allgather(self.ctx._ctx, self._h, C.byref(n)), "thj_fusion_allgather")
        out = np.zeros(max(1, n.value), dtype=FUSION_DTYPE)
        _check(self.ctx.lib, self.ctx.lib.thj_fusion_download(self.ctx._ctx, _ptr(out)), "thj_fusion_download")
        return out[:n.value]

    def covsearch_allgather(self):
        _check(self.ctx.lib, self.ctx.lib.thj_covsearch_allgather(self.ctx._ctx, self._h), "thj_covsearch_allgather")

    def close(self):
        if self._h:
            self.ctx.lib.thj_comm_destroy(self._h)
            self._h = C.c_void_p()


def decode_ins_seq(code: int, length: int) -> str:
    return "".join(INS_CODE[(code >> (3 * k)) & 7] for k in range(length))


class Context:
    """One (host thread, device) context: resident genome, event tables, stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load_lib()
        self._ctx = C.c_void_p()
        _check(self.lib, self.lib.thj_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._ctx)),
               "thj_ctx_create")
        self.genome: Optional[PackedGenome] = None
        self._batches: List[C.c_void_p] = []

    def close(self):
        if self._ctx:
            for b in self._batches:
                self.lib.thj_batch_free(self._ctx, b)
            self._batches = []
            self.lib.thj_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        _check(self.lib, self.lib.thj_ctx_sync(self._ctx), "thj_ctx_sync")

    def upload_genome(self, g: PackedGenome):
        _check(self.lib, self.lib.thj_genome_upload(self._ctx, _ptr(g.blocks), C.c_int64(g.n_blocks), _ptr(g.contig_blk),
                                                    _ptr(g.lens), g.n_contigs), "thj_genome_upload")
        self.genome = g
        self.host_seqs = getattr(g, "seqs", None)

    def adopt_genome(self, d_blocks: int, g: PackedGenome):
        _check(self.lib, self.lib.thj_genome_adopt(self._ctx, C.c_void_p(d_blocks), C.c_int64(g.n_blocks),
                                                   _ptr(g.contig_blk), _ptr(g.lens), g.n_contigs), "thj_genome_adopt")
        self.genome = g

    def upload_batch(self, b: SegBatch, ordinal_base: int = 0) -> C.c_void_p:
        cb, keep, nh, nm = host_cbatch(b, ordinal_base, self.lib)
        out = C.c_void_p()
        _check(self.lib, self.lib.thj_batch_upload(self._ctx, C.byref(cb), C.c_int64(nh), C.c_int64(nm), C.byref(out)),
               "thj_batch_upload")
        self.sync()
        self._batches.append(out)
        return out

    def free_batch(self, h: C.c_void_p):
        self._batches = [b for b in self._batches if b.value != h.value]
        _check(self.lib, self.lib.thj_batch_free(self._ctx, h), "thj_batch_free")

    def configure(self, junc_capacity: int, indel_capacity: int):
        _check(self.lib, self.lib.thj_segjuncs_configure(self._ctx, C.c_int64(junc_capacity), C.c_int64(indel_capacity)),
               "thj_segjuncs_configure")

    def reset(self):
        _check(self.lib, self.lib.thj_segjuncs_reset_async(self._ctx), "thj_segjuncs_reset_async")

    def run(self, p: Params, batch):
        """batch: handle from upload_batch, or a CSegBatch of DEVICE pointers."""
        cp = p.as_ctypes()
        arg = C.byref(batch) if isinstance(batch, CSegBatch) else batch
        _check(self.lib, self.lib.thj_segjuncs_run_async(self._ctx, C.byref(cp), arg), "thj_segjuncs_run_async")

    def run_pair(self, p0: Params, batch0, p1: Params, batch1):
        """two batches (the two sides of a pass) as one call: see thj_segjuncs_run_pair_async"""
        c0, c1 = p0.as_ctypes(), p1.as_ctypes()
        a0 = C.byref(batch0) if isinstance(batch0, CSegBatch) else batch0
        a1 = C.byref(batch1) if isinstance(batch1, CSegBatch) else batch1
        _check(self.lib, self.lib.thj_segjuncs_run_pair_async(self._ctx, C.byref(c0), a0, C.byref(c1), a1), "thj_segjuncs_run_pair_async")

    def finish(self) -> CCounts:
        cnt = CCounts()
        _check(self.lib, self.lib.thj_segjuncs_finish(self._ctx, C.byref(cnt)), "thj_segjuncs_finish")
        return cnt

    def download(self, cnt: CCounts) -> Events:
        j = np.zeros(cnt.n_juncs, dtype=JUNC_DTYPE)
        d = np.zeros(cnt.n_deletions, dtype=JUNC_DTYPE)
        ins = (CInsertion * max(1, cnt.n_insertions))()
        _check(self.lib, self.lib.thj_segjuncs_download(self._ctx, _ptr(j), _ptr(d), ins), "thj_segjuncs_download")
        il = [(int(ins[k].ref_id), int(ins[k].left), ins[k].seq.decode()) for k in range(cnt.n_insertions)]
        stats = {"windows": cnt.n_windows, "indel_pairs": cnt.n_indel_pairs, "rescue_pairs": cnt.n_rescue_pairs,
                 "overflow_blocks": cnt.n_overflow_blocks, "hits_read": cnt.n_hits_read}
        return Events(j, d, il, stats)

    def segjuncs(self, runs: Sequence[Tuple[Params, object]]) -> Events:
        """reset; run every (params, batch); finish; download -- one segment_juncs pass."""
        self.reset()
        for p, b in runs:
            self.run(p, b)
        return self.download(self.finish())

    # ---- coverage search (thj_covsearch_*): call between reset() and finish() of a segment_juncs pass
    def covsearch_reset(self):
        _check(self.lib, self.lib.thj_covsearch_reset_async(self._ctx), "thj_covsearch_reset_async")

    def covsearch_add_hits(self, batch):
        arg = C.byref(batch) if isinstance(batch, CSegBatch) else batch
        _check(self.lib, self.lib.thj_covsearch_add_hits_async(self._ctx, arg), "thj_covsearch_add_hits_async")

    def covsearch_add_reads(self, ium_reads: Sequence[str]):
        n = len(ium_reads)
        if not n:
            return
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(r) for r in ium_reads], out=off[1:])
        bases = np.frombuffer("".join(ium_reads).encode(), dtype=np.uint8)
        W = words_per_plane(max(len(r) for r in ium_reads))
        planes = np.zeros(n * 3 * W, dtype=np.uint64)
        lens = np.zeros(n, dtype=np.uint16)
        _check(self.lib, self.lib.thj_reads_pack(C.c_int64(n), _ptr(off), _ptr(bases), W, _ptr(planes), _ptr(lens)), "thj_reads_pack")
        _check(self.lib, self.lib.thj_covsearch_add_reads(self._ctx, C.c_int64(n), W, _ptr(planes), _ptr(lens), 0), "thj_covsearch_add_reads")

    def covsearch_add_reads_device(self, n: int, W: int, planes_ptr: int, lens_ptr: int):
        """planes / lengths already on the device (thj_reads_pack layout)"""
        _check(self.lib, self.lib.thj_covsearch_add_reads(self._ctx, C.c_int64(n), W, C.c_void_p(planes_ptr), C.c_void_p(lens_ptr), 1),
               "thj_covsearch_add_reads")

    def covsearch_device_state(self):
        """-> (coverage bits ptr, n_words, sizes ptr, ext keys ptr, ext vals ptr, n_ext): device pointers of this rank's state"""
        bits, sizes, keys, vals = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nw, ne = C.c_int64(), C.c_int64()
        _check(self.lib, self.lib.thj_covsearch_device_state(self._ctx, C.byref(bits), C.byref(nw), C.byref(sizes), C.byref(keys), C.byref(vals),
                                                             C.byref(ne)), "thj_covsearch_device_state")
        return bits.value, nw.value, sizes.value, keys.value, vals.value, ne.value

    def covsearch_merge(self, bits_ptr: int, sizes_ptr: int, keys_ptr: int, vals_ptr: int, n_ext: int):
        _check(self.lib, self.lib.thj_covsearch_merge_async(self._ctx, C.c_void_p(bits_ptr), C.c_void_p(sizes_ptr), C.c_void_p(keys_ptr),
                                                            C.c_void_p(vals_ptr), C.c_int64(n_ext)), "thj_covsearch_merge_async")

    def covsearch_run(self, min_cov_length: int, min_intron: int = 50, max_intron: int = 20000):
        _check(self.lib, self.lib.thj_covsearch_run_async(self._ctx, min_cov_length, min_intron, max_intron), "thj_covsearch_run_async")

    def covsearch_finish(self, max_cov_juncs: int = 5000000) -> int:
        found = C.c_int64()
        _check(self.lib, self.lib.thj_covsearch_finish(self._ctx, C.c_int64(max_cov_juncs), C.byref(found)), "thj_covsearch_finish")
        return found.value

    def butterfly_run(self, min_intron: int = 50, max_intron: int = 20000, max_cov_juncs: int = 5000000) -> int:
        """--butterfly-search (segment_juncs.cpp:4178-4249, :1698-2049) from the coverage map and the extension table of the
        covsearch_add_* calls; after covsearch_finish when the coverage search runs too.  -> junctions added to the pass's set"""
        found = C.c_int64()
        _check(self.lib, self.lib.thj_butterfly_run(self._ctx, int(min_intron), int(max_intron), C.c_int64(max_cov_juncs), C.byref(found)), "thj_butterfly_run")
        return found.value

    # ---- microexon search (thj_microexon_*): candidates on the device, window merge on the host, per-window pairing on the device
    def microexon_reset(self):
        _check(self.lib, self.lib.thj_microexon_reset_async(self._ctx), "thj_microexon_reset_async")

    def microexon_collect(self, p: Params, batch, read_side: int):
        arg = C.byref(batch) if isinstance(batch, CSegBatch) else batch
        cp = p.as_ctypes()
        _check(self.lib, self.lib.thj_microexon_collect(self._ctx, C.byref(cp), arg, int(read_side)), "thj_microexon_collect")

    def microexon_candidates(self) -> np.ndarray:
        ptr, n = C.c_void_p(), C.c_int64()
        _check(self.lib, self.lib.thj_microexon_candidates(self._ctx, C.byref(ptr), C.byref(n)), "thj_microexon_candidates")
        a = np.zeros(0, dtype=MX_CAND_DTYPE)
        if n.value:
            a = np.frombuffer((C.c_char * (n.value * MX_CAND_DTYPE.itemsize)).from_address(ptr.value), dtype=MX_CAND_DTYPE).copy()
            C.CDLL(None).free(ptr)
        return a

    def microexon_run(self, windows: np.ndarray, strs: np.ndarray, str_len: np.ndarray, str_window: np.ndarray, min_intron: int = 50, library_type: int = 0,
                      max_juncs: int = 5000000) -> int:
        w = np.ascontiguousarray(windows, dtype=MX_WINDOW_DTYPE)
        a, b, c_ = np.ascontiguousarray(strs, dtype=np.uint64), np.ascontiguousarray(str_len, dtype=np.uint8), np.ascontiguousarray(str_window, dtype=np.uint32)
        found = C.c_int64()
        _check(self.lib, self.lib.thj_microexon_run(self._ctx, C.c_void_p(w.ctypes.data), C.c_int64(len(w)), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data),
                                                    C.c_void_p(c_.ctypes.data), C.c_int64(len(a)), int(min_intron), int(library_type), C.c_int64(max_juncs), C.byref(found)),
               "thj_microexon_run")
        return found.value

    def segjuncs_with_coverage_search(self, runs: Sequence[Tuple[Params, object]], ium_reads: Sequence[str], min_cov_length: int,
                                      min_intron: int = 50, max_intron: int = 20000, max_cov_juncs: int = 5000000):
        """One segment_juncs pass with the coverage search (segment_juncs.cpp:4268-4543) on top of the segment search:
        runs = (params, uploaded batch) of both sides, ium_reads = the initially unmapped reads (--ium-reads).
        -> (Events, junctions the coverage search found, counted with those the segment search also found)"""
        self.reset()
        self.covsearch_reset()
        for p, b in runs:
            self.run(p, b)
            self.covsearch_add_hits(b)
        self.covsearch_add_reads(ium_reads)
        self.covsearch_run(min_cov_length, min_intron, max_intron)
        found = self.covsearch_finish(max_cov_juncs)
        return self.download(self.finish()), found

    SJ_KERNELS = ("thj_k_sj_flat", "thj_k_sj_general<8, 256, true>", "thj_k_sj_general<32, 256, false>", "thj_k_segjuncs_shared",
                  "thj_k_segjuncs_rescue + thj_k_segjuncs_rescue_shared", "thj_k_sj_tasks_list", "thj_k_sj_rescue_scan + thj_k_sj_rescue_flat",
                  "thj_k_sj_tasks")

    def profile_serial(self, on: bool = True):
        """every kernel of the stage calls on the context's stream, one after the other: the profiles then give each kernel's own duration"""
        _check(self.lib, self.lib.thj_profile_serial(self._ctx, 1 if on else 0), "thj_profile_serial")

    def profile(self, enable: bool = True):
        """(ms per launch of each entry of SJ_KERNELS, runs, {reads and tasks of the lists}) since the last call"""
        ms = (C.c_double * 8)()
        st = (C.c_double * 4)()
        n = C.c_int64()
        _check(self.lib, self.lib.thj_profile_segjuncs(self._ctx, 1 if enable else 0, ms, C.byref(n), st), "thj_profile_segjuncs")
        return tuple(ms), n.value, {"many_reads": st[0], "mid_reads": st[1], "list_tasks": st[2], "flat_rescue_pairs": st[3]}

    def device_keys(self, kind: int) -> Tuple[int, int]:
        p = C.c_void_p()
        n = C.c_int64()
        _check(self.lib, self.lib.thj_segjuncs_device_keys(self._ctx, kind, C.byref(p), C.byref(n)),
               "thj_segjuncs_device_keys")
        return (p.value or 0), n.value

    def _fusion_pass(self, runs, ignore_ref_ids) -> int:
        """reset; thj_fusion_run_async for every (params, batch); finish -> number of fusions.  THJ_ERETRY at the finish (a batch had more
        raw candidates than the buffer held; it has been enlarged) runs the pass again."""
        runs = list(runs)
        ign = np.ascontiguousarray(list(ignore_ref_ids), dtype=np.uint32)
        for attempt in range(3):
            _check(self.lib, self.lib.thj_fusion_reset_async(self._ctx), "thj_fusion_reset_async")
            _check(self.lib, self.lib.thj_fusion_set_ignored(self._ctx, _ptr(ign) if len(ign) else None, len(ign)), "thj_fusion_set_ignored")
            for p, b in runs:
                cp = p.as_ctypes()
                arg = C.byref(b) if isinstance(b, CSegBatch) else b
                _check(self.lib, self.lib.thj_fusion_run_async(self._ctx, C.byref(cp), arg), "thj_fusion_run_async")
            n = C.c_int64()
            rc = self.lib.thj_fusion_finish(self._ctx, C.byref(n))
            if rc != -7 or attempt == 2:
                _check(self.lib, rc, "thj_fusion_finish")
                return int(n.value)
        raise AssertionError("unreachable")

    def fusions(self, runs, ignore_ref_ids=()) -> np.ndarray:
        """reset; thj_fusion_run_async for every (params, batch); finish; download -> FUSION_DTYPE array"""
        n = C.c_int64(self._fusion_pass(runs, ignore_ref_ids))
        out = np.zeros(max(1, n.value), dtype=FUSION_DTYPE)
        _check(self.lib, self.lib.thj_fusion_download(self._ctx, _ptr(out)), "thj_fusion_download")
        return out[:n.value]

    def device_insertions(self) -> Tuple[int, int, int]:
        k, v, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        _check(self.lib, self.lib.thj_segjuncs_device_insertions(self._ctx, C.byref(k), C.byref(v), C.byref(n)),
               "thj_segjuncs_device_insertions")
        return (k.value or 0), (v.value or 0), n.value

    def merge_insertions(self, d_keys: int, d_vals: int, n: int):
        _check(self.lib, self.lib.thj_segjuncs_merge_insertions_async(self._ctx, C.c_void_p(d_keys), C.c_void_p(d_vals), C.c_int64(n)),
               "thj_segjuncs_merge_insertions_async")

    def merge_keys(self, kind: int, d_keys: int, n: int):
        _check(self.lib, self.lib.thj_segjuncs_merge_keys_async(self._ctx, kind, C.c_void_p(d_keys), C.c_int64(n)),
               "thj_segjuncs_merge_keys_async")


# ------------------------------------------------------------------ long_spanning_reads

from .batch import Aln, SpanBatch  # noqa: E402

ALN_DTYPE = np.dtype([
    ("read_idx", "<u4"), ("ref_id", "<u4"), ("left", "<i4"),
    ("flags", "u1"), ("mismatches", "u1"), ("edit_dist", "u1"), ("n_cigar", "u1"),
    ("AS", "<i2"), ("XM", "u1"), ("XO", "u1"), ("XG", "u1"), ("md_len", "u1"), ("order", "<u2"),
    ("cigar", "<u4", (16,)), ("md", "S40"),
])
assert ALN_DTYPE.itemsize == 128


class CSpanBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("nseg", C.c_int32), ("words_per_plane", C.c_int32), ("qual_stride", C.c_int32),
                ("seg_off", C.c_void_p), ("hits", C.c_void_p), ("read_planes", C.c_void_p), ("read_len", C.c_void_p),
                ("quals", C.c_void_p), ("hit_heads", C.c_void_p)]


def encode_ins_seq(seq: str) -> int:
    v = 0
    for k, ch in enumerate(seq):
        v |= (INS_CODE.index(ch) if ch in INS_CODE else 4) << (3 * k)
    return v


def pack_span_batch(b: SpanBatch, lib=None):
    """-> dict of contiguous HOST arrays in the thj_span_batch layout"""
    lib = lib or load_lib()
    n = b.n_reads
    lens_ = np.diff(b.read_off) if n else np.zeros(0, dtype=np.int64)
    mx = int(lens_.max()) if n else 1
    W = words_per_plane(mx)
    planes = np.zeros(n * 3 * W, dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint16)
    bases = np.ascontiguousarray(b.bases, dtype=np.uint8)
    off = np.ascontiguousarray(b.read_off, dtype=np.int64)
    _check(lib, lib.thj_reads_pack(C.c_int64(n), _ptr(off), _ptr(bases), W, _ptr(planes), _ptr(lens)), "thj_reads_pack")
    stride = (mx + 3) // 4 * 4
    quals = np.zeros(n * stride, dtype=np.uint8)
    for r in range(n):
        quals[r * stride:r * stride + lens_[r]] = b.quals[off[r]:off[r + 1]]
    return dict(n_reads=n, nseg=b.nseg, W=W, qual_stride=stride, seg_off=np.ascontiguousarray(b.seg_off, dtype=np.uint32),
                hits=np.ascontiguousarray(b.hits), planes=planes, read_len=lens, quals=quals)


MD_ON_HOST = 255
_RC = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


def md_on_host(seq_of_contig, read_seq: str, antisense: bool, left: int, cigar, lib=None, seq_of_contig2=None) -> str:
    """thj_md_string / thj_md_string2: the MD:Z of an alignment whose string does not fit a device record.  seq_of_contig: the
    contig's bases (str / bytes), seq_of_contig2: the second contig of a fusion alignment; read_seq: the read as sequenced"""
    lib = lib or load_lib()
    ref = seq_of_contig if isinstance(seq_of_contig, bytes) else seq_of_contig.encode()
    s = read_seq.encode()
    if antisense:
        s = s.translate(_RC)[::-1]
    cig = (C.c_uint32 * max(1, len(cigar)))(*cigar)
    buf = C.create_string_buffer(4096)
    if seq_of_contig2 is not None:
        ref2 = seq_of_contig2 if isinstance(seq_of_contig2, bytes) else seq_of_contig2.encode()
        n = lib.thj_md_string2(ref, C.c_int64(len(ref)), ref2, C.c_int64(len(ref2)), s, len(s), int(left), cig, len(cigar), buf, 4096)
    else:
        n = lib.thj_md_string(ref, C.c_int64(len(ref)), s, len(s), int(left), cig, len(cigar), buf, 4096)
    if n < 0:
        raise ThjError("thj_md_string: %s" % lib.thj_last_error().decode())
    return buf.value.decode()


SPAN_FUSION_DTYPE = np.dtype([("ref_id1", "<u4"), ("ref_id2", "<u4"), ("left", "<u4"), ("right", "<u4"), ("dir", "<u4")])   # thj_span_fusion
_FUSION_OPS = (7, 8, 9, 10)


def alns_from_array(a: np.ndarray, md_resolver=None) -> List[Aln]:
    """md_resolver(read_idx, ref_id, antisense, left, cigar) -> MD string, for records flagged THJ_MD_ON_HOST"""
    out = []
    for x in a:
        n = int(x["n_cigar"])
        cig = tuple(int(c) for c in x["cigar"][:n])
        ref_id2 = int(x["cigar"][15]) if any((c >> 28) in _FUSION_OPS for c in cig) else 0     # fusion alignment: ref_id2 in the last slot
        if int(x["md_len"]) == MD_ON_HOST:
            if md_resolver is None:
                raise ThjError("a record's MD string is left to the host (THJ_MD_ON_HOST) and no resolver was given")
            if ref_id2:
                md = md_resolver(int(x["read_idx"]), int(x["ref_id"]), bool(x["flags"] & 1), int(x["left"]), cig, ref_id2)
            else:
                md = md_resolver(int(x["read_idx"]), int(x["ref_id"]), bool(x["flags"] & 1), int(x["left"]), cig)
        else:
            md = x["md"][:int(x["md_len"])].decode()
        out.append(Aln(int(x["read_idx"]), int(x["ref_id"]), int(x["left"]), bool(x["flags"] & 1), bool(x["flags"] & 4),
                       int(x["mismatches"]), int(x["edit_dist"]), cig, int(x["AS"]), int(x["XM"]), int(x["XO"]), int(x["XG"]), md, ref_id2))
    return out


def span_md_resolver(seqs, batches, lib=None):
    """resolver over the host copies of a pass: seqs[ref_id - 1] = contig bases, batches = the SpanBatch objects in run order
    (read_idx counts through them)"""
    starts = [0]
    for b in batches:
        starts.append(starts[-1] + b.n_reads)

    def resolve(read_idx, ref_id, antisense, left, cigar, ref_id2=0):
        k = max(i for i in range(len(batches)) if starts[i] <= read_idx)
        b = batches[k]
        r = read_idx - starts[k]
        return md_on_host(seqs[ref_id - 1], b.bases[b.read_off[r]:b.read_off[r + 1]].tobytes().decode(), antisense, left, cigar, lib,
                          seqs[ref_id2 - 1] if ref_id2 else None)
    return resolve


def _ins_table(insertions) -> np.ndarray:
    t = np.zeros((max(1, len(insertions)), 4), dtype=np.uint32)
    for k, (ref, left, seq) in enumerate(insertions):
        t[k] = (ref, left, len(seq), encode_ins_seq(seq))
    return t


def _span_methods():
    def upload_span_fusions(self, fusions: np.ndarray):
        """--fusion-search: the .fusions list (SPAN_FUSION_DTYPE, Fusion::operator< order) for runs with Params.fusion_search"""
        f = np.ascontiguousarray(fusions, dtype=SPAN_FUSION_DTYPE)
        _check(self.lib, self.lib.thj_span_fusions_upload(self._ctx, _ptr(f) if len(f) else None, C.c_int64(len(f))), "thj_span_fusions_upload")

    def upload_span_sets(self, juncs: np.ndarray, insertions):
        j = np.ascontiguousarray(juncs, dtype=JUNC_DTYPE)
        t = _ins_table(insertions)
        _check(self.lib, self.lib.thj_span_sets_upload(self._ctx, _ptr(j), C.c_int64(len(j)), _ptr(t), C.c_int64(len(insertions))),
               "thj_span_sets_upload")

    def span_sets_from_segjuncs(self):
        _check(self.lib, self.lib.thj_span_sets_from_segjuncs(self._ctx), "thj_span_sets_from_segjuncs")

    def span_fusions_from_segjuncs(self):
        """--fusion-search: the list of the last fusion_search() / fusions() of this context, handed over on the device"""
        _check(self.lib, self.lib.thj_span_fusions_from_segjuncs(self._ctx), "thj_span_fusions_from_segjuncs")

    def fusion_search(self, runs, ignore_ref_ids=()) -> int:
        """fusions() without the download: reset; thj_fusion_run_async for every (params, batch); finish -> number of fusions"""
        return self._fusion_pass(runs, ignore_ref_ids)

    def span_hit_heads(self, d_hits: int, n_hits: int, d_heads: int):
        """the dense 16-byte head array of device-resident hit records (CSpanBatch.hit_heads)"""
        _check(self.lib, self.lib.thj_span_hit_heads_async(self._ctx, C.c_void_p(d_hits), C.c_int64(n_hits), C.c_void_p(d_heads)),
               "thj_span_hit_heads_async")

    def upload_span_batch(self, b: SpanBatch):
        d = pack_span_batch(b, self.lib)
        cb = CSpanBatch()
        cb.n_reads, cb.nseg, cb.words_per_plane, cb.qual_stride = d["n_reads"], d["nseg"], d["W"], d["qual_stride"]
        cb.seg_off, cb.hits, cb.read_planes, cb.read_len, cb.quals = [d[k].ctypes.data for k in ("seg_off", "hits", "planes", "read_len", "quals")]
        out = C.c_void_p()
        _check(self.lib, self.lib.thj_span_batch_upload(self._ctx, C.byref(cb), C.c_int64(len(d["hits"])), C.byref(out)),
               "thj_span_batch_upload")
        self._span_batches = getattr(self, "_span_batches", []) + [out]
        if not hasattr(self, "_span_host"):
            self._span_host = {}
        self._span_host[out.value] = b
        return out

    def span_reset(self):
        _check(self.lib, self.lib.thj_span_reset_async(self._ctx), "thj_span_reset_async")

    def span_run(self, p: Params, batch):
        cp = p.as_ctypes()
        arg = C.byref(batch) if isinstance(batch, CSpanBatch) else batch
        _check(self.lib, self.lib.thj_span_run_async(self._ctx, C.byref(cp), arg), "thj_span_run_async")

    def span_finish(self) -> int:
        n = C.c_int64()
        _check(self.lib, self.lib.thj_span_finish(self._ctx, C.byref(n)), "thj_span_finish")
        return n.value

    def span_download(self, n: int) -> np.ndarray:
        a = np.zeros(max(1, n), dtype=ALN_DTYPE)
        _check(self.lib, self.lib.thj_span_download(self._ctx, _ptr(a)), "thj_span_download")
        return a[:n]

    def spanning(self, p: Params, batches, md_resolver=None) -> List[Aln]:
        """md_resolver: see alns_from_array; built automatically when the genome was uploaded from strings kept by the caller
        (Context.host_seqs) and the batches came from upload_span_batch"""
        for attempt in range(4):
            self.span_reset()
            for b in batches:
                self.span_run(p, b)
            n = C.c_int64()
            rc = self.lib.thj_span_finish(self._ctx, C.byref(n))
            if rc != -7:                       # THJ_ERETRY: the extra-record pool was enlarged, the pass runs again
                break
        _check(self.lib, rc, "thj_span_finish")
        a = self.span_download(n.value)
        if md_resolver is None and getattr(self, "host_seqs", None) is not None:
            hb = [self._span_host.get(getattr(b, "value", None)) for b in batches]
            if all(x is not None for x in hb):
                md_resolver = span_md_resolver(self.host_seqs, hb, self.lib)
        return alns_from_array(a, md_resolver)

    def span_run_pair(self, p: Params, batch0, batch1):
        """two batches (the two sides of a pass) beside each other: thj_span_run_pair_async"""
        cp = p.as_ctypes()
        a0 = C.byref(batch0) if isinstance(batch0, CSpanBatch) else batch0
        a1 = C.byref(batch1) if isinstance(batch1, CSpanBatch) else batch1
        _check(self.lib, self.lib.thj_span_run_pair_async(self._ctx, C.byref(cp), a0, a1), "thj_span_run_pair_async")

    def span_tier0_pair(self, p: Params, batch0, batch1):
        """the pair's tier 0 ahead of span_run_pair on the same batches (needs no junction set): thj_span_tier0_pair_async"""
        cp = p.as_ctypes()
        a0 = C.byref(batch0) if isinstance(batch0, CSpanBatch) else batch0
        a1 = C.byref(batch1) if isinstance(batch1, CSpanBatch) else batch1
        _check(self.lib, self.lib.thj_span_tier0_pair_async(self._ctx, C.byref(cp), a0, a1), "thj_span_tier0_pair_async")

    def span_tier_counts(self):
        """of the batch launched last: reads to the closure kernels, to the multihit kernel, on to the general kernel"""
        c = (C.c_int64 * 5)()
        _check(self.lib, self.lib.thj_span_tier_counts(self._ctx, c), "thj_span_tier_counts")
        return int(c[0]), int(c[1]), int(c[2])

    def span_chain_count(self):
        """of the batch launched last: the reads that travelled as chain entries (tier 0 -> thj_k_join -> thj_k_finish)"""
        c = (C.c_int64 * 5)()
        _check(self.lib, self.lib.thj_span_tier_counts(self._ctx, c), "thj_span_tier_counts")
        return int(c[3])

    def stream_info(self):
        """thj_ctx_stream_info: {"n_side", "independent": [..], "ratio": [..]} of the side streams made so far"""
        n = C.c_int32()
        ind = (C.c_int32 * 3)()
        ratio = (C.c_double * 3)()
        _check(self.lib, self.lib.thj_ctx_stream_info(self._ctx, C.byref(n), ind, ratio), "thj_ctx_stream_info")
        return {"n_side": int(n.value), "independent": [bool(ind[k]) for k in range(n.value)], "ratio": [float(ratio[k]) for k in range(n.value)]}

    def probe_streams(self, need=2):
        _check(self.lib, self.lib.thj_ctx_probe_streams(self._ctx, int(need)), "thj_ctx_probe_streams")
        return self.stream_info()

    def span_chain_groups(self):
        """of the batch launched last: the multihit reads whose chains travelled as chain entries (thj_k_chains)"""
        c = (C.c_int64 * 5)()
        _check(self.lib, self.lib.thj_span_tier_counts(self._ctx, c), "thj_span_tier_counts")
        return int(c[4])

    Context.SPAN_KERNELS = ("thj_k_stitch_contig", "thj_k_chains", "thj_k_join", "thj_k_join_closure", "thj_k_finish", "thj_k_stitch", "thj_k_stitch_pack", "thj_k_stitch_generic")

    def profile_span(self, enable: bool = True):
        """-> ([ms per launch of each entry of SPAN_KERNELS], launches)"""
        ms = (C.c_double * 8)()
        n = C.c_int64()
        _check(self.lib, self.lib.thj_profile_span(self._ctx, 1 if enable else 0, ms, C.byref(n)), "thj_profile_span")
        return list(ms), n.value

    for f in (upload_span_fusions, upload_span_sets, span_sets_from_segjuncs, span_fusions_from_segjuncs, fusion_search, upload_span_batch, span_reset, span_run, span_finish,
              span_download, spanning, profile_span, span_tier_counts, span_hit_heads, span_run_pair, span_tier0_pair, span_chain_count, span_chain_groups, stream_info, probe_streams):
        setattr(Context, f.__name__, f)


_span_methods()

ABI_SYMBOLS += ["thj_bgzf_inflate", "thj_ingest_seg_batch", "thj_ingest_span_hits", "thj_span_batch_attach_reads"]
ABI_SYMBOLS += ["thj_juncbed_configure", "thj_juncbed_reset_async", "thj_juncbed_add_span_async", "thj_juncbed_add_records",
                "thj_juncbed_finish", "thj_juncbed_download"]
ABI_SYMBOLS += ["thj_md_string"]
ABI_SYMBOLS += ["thj_microexon_reset_async", "thj_microexon_collect", "thj_microexon_candidates", "thj_microexon_run"]
ABI_SYMBOLS += ["thj_butterfly_run", "thj_covsearch_add_reads_bam", "thj_covsearch_reserve_reads"]
ABI_SYMBOLS += ["thj_span_sets_upload", "thj_span_sets_from_segjuncs", "thj_span_fusions_from_segjuncs", "thj_span_batch_upload", "thj_span_batch_free",
                "thj_span_reset_async", "thj_span_run_async", "thj_span_run_pair_async", "thj_span_tier0_pair_async", "thj_span_finish", "thj_span_download", "thj_profile_span",
                "thj_span_tier_counts", "thj_span_device_records"]


# ------------------------------------------------------------------ junction consensus (thj_juncbed_*)

JUNCSTAT_DTYPE = np.dtype([("ref_id", "<u4"), ("left", "<u4"), ("right", "<u4"), ("antisense", "<u4"), ("left_extent", "<u4"),
                           ("right_extent", "<u4"), ("support", "<u4"), ("reserved", "<u4")])


def junctions_bed_text(js: np.ndarray, names: Sequence[str]) -> str:
    """print_junctions / print_junction (junctions.cpp:100-120, :330-350)"""
    out = ['track name=junctions description="TopHat junctions"\n']
    for k, j in enumerate(js):
        start = int(j["left"]) + 1 - int(j["left_extent"])
        end = int(j["right"]) + int(j["right_extent"])
        out.append("%s\t%d\t%d\tJUNC%08d\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (
            names[int(j["ref_id"]) - 1], start, end, k + 1, int(j["support"]), "-" if j["antisense"] else "+", start, end,
            int(j["left_extent"]), int(j["right_extent"]), int(j["right"]) - start))
    return "".join(out)


def _juncbed_methods():
    def juncbed_configure(self, capacity: int):
        _check(self.lib, self.lib.thj_juncbed_configure(self._ctx, C.c_int64(capacity)), "thj_juncbed_configure")

    def juncbed_reset(self):
        _check(self.lib, self.lib.thj_juncbed_reset_async(self._ctx), "thj_juncbed_reset_async")

    def juncbed_add_span(self):
        """folds the records of the last spanning pass (still on the device) in"""
        _check(self.lib, self.lib.thj_juncbed_add_span_async(self._ctx), "thj_juncbed_add_span_async")

    def juncbed_add_records(self, recs: np.ndarray):
        """recs: ALN_DTYPE array (ref_id, left, flags & 4 = antisense splice, n_cigar, cigar are read)"""
        a = np.ascontiguousarray(recs, dtype=ALN_DTYPE)
        _check(self.lib, self.lib.thj_juncbed_add_records(self._ctx, _ptr(a) if len(a) else None, C.c_int64(len(a)), 0), "thj_juncbed_add_records")

    def juncbed_finish(self, min_anchor_len: int = 8) -> np.ndarray:
        n = C.c_int64()
        _check(self.lib, self.lib.thj_juncbed_finish(self._ctx, min_anchor_len, C.byref(n)), "thj_juncbed_finish")
        out = np.zeros(max(1, n.value), dtype=JUNCSTAT_DTYPE)
        _check(self.lib, self.lib.thj_juncbed_download(self._ctx, _ptr(out)), "thj_juncbed_download")
        return out[:n.value]

    for f in (juncbed_configure, juncbed_reset, juncbed_add_span, juncbed_add_records, juncbed_finish):
        setattr(Context, f.__name__, f)


_juncbed_methods()


MX_CAND_DTYPE = np.dtype([("ordinal", "<u4"), ("rank", "<u2"), ("side", "u1"), ("len", "u1"), ("ref_id", "<u4"), ("left", "<i4"), ("right", "<i4"),
                          ("reserved", "<u4"), ("str", "<u8")])
MX_WINDOW_DTYPE = np.dtype([("ref_id", "<u4"), ("left", "<i4"), ("right", "<i4"), ("side", "<i4")])
assert MX_CAND_DTYPE.itemsize == 32 and MX_WINDOW_DTYPE.itemsize == 16


def microexon_merge_windows(cands: np.ndarray):
    """add_to_microexon_windows (segment_juncs.cpp:3675-3735) over the candidates in visiting order -- the Python mirror of
    csrc/host/thj_mx_host.h (the executables use that one) -> (windows, strs, str_len, str_window) for Context.microexon_run"""
    import bisect
    order = np.lexsort((cands["rank"], cands["ordinal"], cands["side"]))
    keys, vals = [], []                                   # sorted (ref, left, right) -> (side, [candidate indices])

    def overlap(ll, lr, rl, rr):
        return (rl <= ll < rr) or (rl < lr < rr) or (ll <= rl < lr) or (ll < rr < lr)
    for ci in order:
        c = cands[ci]
        ref, lbd, rbd, side = int(c["ref_id"]), int(c["left"]), int(c["right"]), int(c["side"])
        key = (ref, lbd, rbd)
        lb = bisect.bisect_left(keys, key)
        ub = bisect.bisect_left(keys, (ref, rbd, rbd + 1))
        if lb == len(keys):
            keys.append(key); vals.append((side, [int(ci)]))
            continue
        first, last, new_vec, have = None, ub, [], False
        for k in range(lb, ub):
            if overlap(keys[k][1], keys[k][2], lbd, rbd):
                have = True
                if first is None:
                    first = k
                key = (ref, min(keys[k][1], lbd), max(keys[k][2], rbd))
                new_vec += vals[k][1]
            elif first is not None:
                last = k
        if first is not None:
            del keys[first:last]; del vals[first:last]
        new_vec = new_vec + [int(ci)] if have else [int(ci)]
        at = bisect.bisect_left(keys, key)
        if at < len(keys) and keys[at] == key:
            continue                                       # map::insert leaves an existing key alone
        keys.insert(at, key); vals.insert(at, (side, new_vec))
    windows = np.zeros(len(keys), dtype=MX_WINDOW_DTYPE)
    strs, lens, wins = [], [], []
    for w, (k, (side, idx)) in enumerate(zip(keys, vals)):
        windows[w] = (k[0], k[1], k[2], side)
        for ci in idx:
            strs.append(int(cands[ci]["str"])); lens.append(int(cands[ci]["len"])); wins.append(w)
    return windows, np.array(strs, dtype=np.uint64), np.array(lens, dtype=np.uint8), np.array(wins, dtype=np.uint32)


def aln_array_from_tuples(recs) -> np.ndarray:
    """[(ref_id, left, antisense_splice, [(op, len) ...][, ref_id2])] -> ALN_DTYPE array (the fields the junction consensus reads; a fusion
    alignment has at most 15 ops and its second contig in cigar[15])"""
    a = np.zeros(len(recs), dtype=ALN_DTYPE)
    for k, rec in enumerate(recs):
        ref, left, anti, cig = rec[:4]
        a[k]["ref_id"], a[k]["left"], a[k]["flags"], a[k]["n_cigar"] = ref, left, 4 if anti else 0, len(cig)
        for i, (op, ln) in enumerate(cig):
            a[k]["cigar"][i] = (op << 28) | ln
        if len(rec) > 4 and rec[4]:
            a[k]["cigar"][15] = rec[4]
    return a


# ------------------------------------------------------------------ the BAM writer's device side (thj_bamout.hip)

ABI_SYMBOLS += ["thj_span_bam_encode", "thj_bgzf_deflate", "thj_bam_stream_upload", "thj_bam_stream_download", "thj_span_batch_reads_host"]


def bgzf_plan_cuts(sizes: Sequence[int], block: int = 0x10000) -> List[int]:
    """where BGZF members end in a stream of records that starts a member: bam_write1 calls bgzf_flush_try(4 + block_len) -- a record
    that does not fit what is left of the 64 KiB block starts a new one (bam.c:225, bgzf.c:587-592) -- and bgzf_write flushes a full
    block (bgzf.c:594-623).  The last member is closed at the end of the stream.  The Python mirror of BamWriter::plan_cuts."""
    cuts, off, pos = [], 0, 0
    for s in sizes:
        if off + s > block and off > 0:
            cuts.append(pos); off = 0
        while s > 0:
            c = min(block - off, s)
            off += c; pos += c; s -= c
            if off == block:
                cuts.append(pos); off = 0
    if off:
        cuts.append(pos)
    return cuts


def _bamout_methods():
    def bam_stream_upload(self, data: bytes):
        buf = np.frombuffer(data, dtype=np.uint8)
        _check(self.lib, self.lib.thj_bam_stream_upload(self._ctx, _ptr(buf) if len(data) else None, C.c_int64(len(data))), "thj_bam_stream_upload")
        self._bam_bytes = len(data)

    def bam_stream_download(self, n: int) -> bytes:
        buf = np.zeros(max(1, n), dtype=np.uint8)
        _check(self.lib, self.lib.thj_bam_stream_download(self._ctx, _ptr(buf)), "thj_bam_stream_download")
        return buf[:n].tobytes()

    def bgzf_deflate(self, member_end: Sequence[int]):
        """the context's stream cut at member_end -> ([raw DEFLATE stream per member], [CRC-32 per member])"""
        ends = np.asarray(member_end, dtype=np.int64)
        n = len(ends)
        comp = np.zeros(max(1, n) * 65536, dtype=np.uint8)
        clen = np.zeros(max(1, n), dtype=np.uint32); crc = np.zeros(max(1, n), dtype=np.uint32)
        total = C.c_int64()
        _check(self.lib, self.lib.thj_bgzf_deflate(self._ctx, C.c_int64(n), _ptr(ends), _ptr(comp), C.c_int64(comp.size), _ptr(clen), _ptr(crc), C.byref(total)),
               "thj_bgzf_deflate")
        out, at = [], 0
        for k in range(n):
            out.append(comp[at:at + int(clen[k])].tobytes()); at += int(clen[k])
        assert at == total.value
        return out, [int(x) for x in crc[:n]]

    for f in (bam_stream_upload, bam_stream_download, bgzf_deflate):
        setattr(Context, f.__name__, f)


_bamout_methods()
