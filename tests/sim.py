"""ctypes binding of tests/hostsim (CPU build of the kernel-logic header).  Test-only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from locked_make import locked_make

import numpy as np

from tophat_amd import host
from tophat_amd.batch import Events, JUNC_DTYPE, SegBatch
from tophat_amd.params import Params

HS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")


def lib():
    so = os.path.join(HS_DIR, "libhostsim.so")
    locked_make(HS_DIR)
    l = C.CDLL(so)
    l.thj_last_error.restype = C.c_char_p
    return l


def sort_events(j, d, ins_raw) -> Events:
    def su(a):
        if len(a) == 0:
            return a
        a = np.unique(a)
        return a[np.lexsort((a["antisense"], a["right"], a["left"], a["ref_id"]))]
    best = {}
    for (ref, left, ln, seq, prio) in ins_raw:
        k = (ref, left, ln)
        if k not in best or prio < best[k][0]:
            best[k] = (prio, seq)
    ins = [(k[0], k[1], host.decode_ins_seq(best[k][1], k[2])) for k in sorted(best)]
    return Events(su(j), su(d), ins, {})


def segjuncs(p: Params, seqs, b: SegBatch, ordinal_base: int = 0) -> Events:
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    cb, keep, _, _ = host.host_cbatch(b, ordinal_base, lib=l)
    clen = g.lens.astype(np.int32)
    cp = p.as_ctypes()
    pj, pd, pi = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nj, nd, ni = C.c_int64(), C.c_int64(), C.c_int64()
    stats = (C.c_int64 * 4)()
    rc = l.hostsim_segjuncs(C.byref(cp), C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data),
                            C.c_void_p(clen.ctypes.data), g.n_contigs, C.byref(cb),
                            C.byref(pj), C.byref(nj), C.byref(pd), C.byref(nd), C.byref(pi), C.byref(ni), stats)
    assert rc == 0

    def arr(ptr, n):
        if n == 0:
            return np.zeros(0, dtype=JUNC_DTYPE)
        return np.frombuffer((C.c_char * (n * 16)).from_address(ptr.value), dtype=JUNC_DTYPE).copy()
    j, d = arr(pj, nj.value), arr(pd, nd.value)
    raw = []
    if ni.value:
        a = np.frombuffer((C.c_char * (ni.value * 24)).from_address(pi.value), dtype=np.uint32).reshape(-1, 6)
        raw = [(int(x[0]), int(x[1]), int(x[2]), int(x[3]), int(x[4]) | (int(x[5]) << 32)) for x in a]
    for ptr in (pj, pd, pi):
        l.hostsim_free(ptr)
    ev = sort_events(j, d, raw)
    ev.stats = {"windows": stats[0], "indel_pairs": stats[1], "rescue_pairs": stats[2], "trivial_reads": stats[3]}
    return ev


def microexon_search(p: Params, seqs, batches, min_intron: int = 50, max_juncs: int = 5000000):
    """the microexon kernels' logic (thj_cov_core.h) as host loops around the product's window merge (csrc/host/thj_mx_host.h);
    batches = [(SegBatch, side, ordinal_base)] -> (JUNC_DTYPE array in Junction order, number of windows)"""
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    clen = g.lens.astype(np.int32)
    cbs = [host.host_cbatch(b, base, lib=l) for b, _sd, base in batches]
    arr = (C.POINTER(type(cbs[0][0])) * len(cbs))(*[C.pointer(x[0]) for x in cbs]) if cbs else None
    sides = (C.c_int32 * max(1, len(cbs)))(*[sd for _b, sd, _base in batches])
    cp = p.as_ctypes()
    out = C.c_void_p()
    n, nw = C.c_int64(), C.c_int64()
    rc = l.hostsim_microexon(C.byref(cp), C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data), C.c_void_p(clen.ctypes.data), g.n_contigs,
                             arr, sides, len(cbs), int(min_intron), C.c_int64(max_juncs), C.byref(out), C.byref(n), C.byref(nw))
    assert rc == 0
    a = np.zeros(0, dtype=JUNC_DTYPE)
    if n.value:
        a = np.frombuffer((C.c_char * (n.value * 16)).from_address(out.value), dtype=JUNC_DTYPE).copy()
    l.hostsim_free(out)
    a = np.unique(a) if len(a) else a
    if len(a):
        a = a[np.lexsort((a["antisense"], a["right"], a["left"], a["ref_id"]))]
    return a, nw.value


def spanning(p: Params, seqs, b, juncs, insertions, mode: int = 0):
    """-> (list of Aln, status counts) from the CPU build of thj_span_core.h"""
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    d = host.pack_span_batch(b, lib=l)
    clen = g.lens.astype(np.int32)
    cp = p.as_ctypes()
    j = np.ascontiguousarray(juncs, dtype=JUNC_DTYPE)
    t = host._ins_table(insertions)
    out = C.c_void_p()
    n_out = C.c_int64()
    st = (C.c_int64 * 5)()
    rc = l.hostsim_spanning(C.byref(cp), C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data),
                            C.c_void_p(clen.ctypes.data), g.n_contigs, d["n_reads"], d["nseg"], d["W"],
                            C.c_void_p(d["seg_off"].ctypes.data), C.c_void_p(d["hits"].ctypes.data),
                            C.c_void_p(d["planes"].ctypes.data), C.c_void_p(d["read_len"].ctypes.data),
                            C.c_void_p(d["quals"].ctypes.data), d["qual_stride"],
                            C.c_void_p(j.ctypes.data), C.c_int64(len(j)), C.c_void_p(t.ctypes.data), C.c_int64(len(insertions)), mode,
                            C.byref(out), C.byref(n_out), st)
    assert rc == 0, rc
    a = np.frombuffer((C.c_char * (max(1, n_out.value) * 128)).from_address(out.value), dtype=host.ALN_DTYPE)[:n_out.value].copy()
    l.hostsim_free(out)
    return host.alns_from_array(a, host.span_md_resolver(list(seqs), [b])), list(st)


def spanning_fusion(p: Params, seqs, b, juncs, insertions, fusions, skip_tier0: bool = False):
    """-> (list of Aln, status counts): tier 0 + the fusion tier (thj_span_fusion.h) compiled for the CPU; p.fusion_search decides
    whether the fusion branches are taken"""
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    d = host.pack_span_batch(b, lib=l)
    clen = g.lens.astype(np.int32)
    cp = p.as_ctypes()
    j = np.ascontiguousarray(juncs, dtype=JUNC_DTYPE)
    t = host._ins_table(insertions)
    f = np.ascontiguousarray(fusions, dtype=host.SPAN_FUSION_DTYPE)
    out = C.c_void_p()
    n_out = C.c_int64()
    st = (C.c_int64 * 5)()
    rc = l.hostsim_spanning_fusion(C.byref(cp), C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data),
                                   C.c_void_p(clen.ctypes.data), g.n_contigs, d["n_reads"], d["nseg"], d["W"],
                                   C.c_void_p(d["seg_off"].ctypes.data), C.c_void_p(d["hits"].ctypes.data),
                                   C.c_void_p(d["planes"].ctypes.data), C.c_void_p(d["read_len"].ctypes.data),
                                   C.c_void_p(d["quals"].ctypes.data), d["qual_stride"],
                                   C.c_void_p(j.ctypes.data), C.c_int64(len(j)), C.c_void_p(t.ctypes.data), C.c_int64(len(insertions)),
                                   C.c_void_p(f.ctypes.data), C.c_int64(len(f)), 1 if skip_tier0 else 0,
                                   C.byref(out), C.byref(n_out), st)
    assert rc == 0, rc
    a = np.frombuffer((C.c_char * (max(1, n_out.value) * 128)).from_address(out.value), dtype=host.ALN_DTYPE)[:n_out.value].copy()
    l.hostsim_free(out)
    return host.alns_from_array(a, host.span_md_resolver(list(seqs), [b])), list(st)


def fusions(p: Params, seqs, b: SegBatch, ignore_ref_ids=()):
    """raw fusion events of the kernel logic, reduced like FusionSimpleSet (count, min edit_dist)"""
    import orc
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    cb, keep, _, _ = host.host_cbatch(b, 0, lib=l)
    clen = g.lens.astype(np.int32)
    cp = p.as_ctypes()
    out = C.c_void_p()
    n = C.c_int64()
    ign = np.ascontiguousarray(list(ignore_ref_ids), dtype=np.uint32)
    rc = l.hostsim_fusions(C.byref(cp), C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data),
                           C.c_void_p(clen.ctypes.data), g.n_contigs, C.byref(cb), C.c_void_p(ign.ctypes.data), len(ign),
                           C.byref(out), C.byref(n))
    assert rc == 0
    a = np.zeros(0, dtype=orc.FUSION_DTYPE)
    if n.value:
        a = np.frombuffer((C.c_char * (n.value * 32)).from_address(out.value), dtype=orc.FUSION_DTYPE).copy()
    l.hostsim_free(out)
    return orc.merge_fusions(a, np.zeros(0, dtype=orc.FUSION_DTYPE))


def fusions_block(p: Params, seqs, b: SegBatch, n_blocks: int = 1, ignore_ref_ids=()):
    """-> (the fusion set, pairs evaluated where they were found): thj_k_fusion's workgroup algorithm (thj_fusion_block.h: tiles of 256 reads,
    the queue of candidate pairs, the family reads taken by the whole workgroup) over the fibers of simt.h, `n_blocks` workgroups sharing
    the tiles as the device's do; reduced like FusionSimpleSet"""
    import orc
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    cb, keep, _, _ = host.host_cbatch(b, 0, lib=l)
    clen = g.lens.astype(np.int32)
    cp = p.as_ctypes()
    out = C.c_void_p()
    n = C.c_int64()
    now = C.c_int64()
    ign = np.ascontiguousarray(list(ignore_ref_ids), dtype=np.uint32)
    rc = l.hostsim_fusions_block(C.byref(cp), C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data),
                                 C.c_void_p(clen.ctypes.data), g.n_contigs, C.byref(cb), C.c_void_p(ign.ctypes.data), len(ign), int(n_blocks),
                                 C.byref(out), C.byref(n), C.byref(now))
    assert rc == 0
    a = np.zeros(0, dtype=orc.FUSION_DTYPE)
    if n.value:
        a = np.frombuffer((C.c_char * (n.value * 32)).from_address(out.value), dtype=orc.FUSION_DTYPE).copy()
    l.hostsim_free(out)
    return orc.merge_fusions(a, np.zeros(0, dtype=orc.FUSION_DTYPE)), int(now.value)


def _pack_ium(l, ium_reads):
    n = len(ium_reads)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(r) for r in ium_reads], out=off[1:])
    bases = np.frombuffer("".join(ium_reads).encode(), dtype=np.uint8) if n else np.zeros(0, dtype=np.uint8)
    W = host.words_per_plane(max([len(r) for r in ium_reads] + [1]))
    planes = np.zeros(max(1, n * 3 * W), dtype=np.uint64)
    lens = np.zeros(max(1, n), dtype=np.uint16)
    if n:
        assert l.thj_reads_pack(C.c_int64(n), C.c_void_p(off.ctypes.data), C.c_void_p(bases.ctypes.data), W, C.c_void_p(planes.ctypes.data),
                                C.c_void_p(lens.ctypes.data)) == 0
    return planes, lens, W, n


def coverage_state(seqs, hits, ium_reads):
    """one shard's coverage-search state (coverage words, per-contig sizes, extension keys, values) as numpy arrays"""
    from tophat_amd.batch import HIT_DTYPE
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    clen = g.lens.astype(np.int32)
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    planes, lens, W, n = _pack_ium(l, ium_reads)
    nb = len(g.blocks) // 4
    bits = np.zeros(nb, dtype=np.uint64)
    sizes = np.zeros(g.n_contigs + 1, dtype=np.int32)
    keys = np.zeros(n + 1, dtype=np.uint32)          # one record per read: its length (at most 32) ...
    vals = np.zeros(n + 1, dtype=np.uint64)          # ... and its first 32 bases as a 2-bit string
    rc = l.hostsim_coverage_state(C.c_void_p(g.contig_blk.ctypes.data), C.c_void_p(clen.ctypes.data), g.n_contigs, C.c_int64(nb),
                                  C.c_void_p(h.ctypes.data), C.c_int64(len(h)), C.c_void_p(planes.ctypes.data), C.c_void_p(lens.ctypes.data),
                                  C.c_int64(n), W, C.c_void_p(bits.ctypes.data), C.c_void_p(sizes.ctypes.data), C.c_void_p(keys.ctypes.data),
                                  C.c_void_p(vals.ctypes.data))
    assert rc == 0
    return bits, sizes, keys[:n], vals[:n]


def coverage_run(seqs, states, min_cov_length: int, min_intron: int = 50, max_intron: int = 20000, max_juncs: int = 5000000):
    """states merged by thj_covsearch_merge_async's rule (OR of the coverage words, max of the sizes, concatenated
    entries), then the pass -> set of (ref_id, left, right, antisense)"""
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    clen = g.lens.astype(np.int32)
    bits = np.bitwise_or.reduce([s[0] for s in states])
    sizes = np.maximum.reduce([s[1] for s in states])
    keys = np.ascontiguousarray(np.concatenate([s[2] for s in states]))
    vals = np.ascontiguousarray(np.concatenate([s[3] for s in states]))
    out = C.c_void_p()
    n_out = C.c_int64()
    rc = l.hostsim_coverage_run(C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data), C.c_void_p(clen.ctypes.data), g.n_contigs,
                                C.c_int64(len(g.blocks) // 4), C.c_void_p(bits.ctypes.data), C.c_void_p(sizes.ctypes.data),
                                C.c_void_p(keys.ctypes.data), C.c_void_p(vals.ctypes.data), C.c_int64(len(keys)),
                                min_cov_length, min_intron, max_intron, C.c_int64(max_juncs), C.byref(out), C.byref(n_out))
    assert rc == 0
    a = np.frombuffer((C.c_char * (max(1, n_out.value) * 16)).from_address(out.value), dtype=JUNC_DTYPE)[:n_out.value].copy()
    l.hostsim_free(out)
    return {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in a}


def butterfly_search(seqs, hits, ium_reads, min_intron: int = 50, max_intron: int = 20000, max_juncs: int = 5000000):
    """the butterfly-search kernels (thj_cov_core.h: bf_*) as host loops -> set of (ref_id, left, right, antisense)"""
    from tophat_amd.batch import HIT_DTYPE
    l = lib()
    g = host.pack_genome(seqs, lib=l)
    clen = g.lens.astype(np.int32)
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    planes, lens, W, n = _pack_ium(l, ium_reads)
    out = C.c_void_p()
    n_out = C.c_int64()
    rc = l.hostsim_butterfly_search(C.c_void_p(g.blocks.ctypes.data), C.c_void_p(g.contig_blk.ctypes.data), C.c_void_p(clen.ctypes.data), g.n_contigs,
                                    C.c_int64(len(g.blocks) // 4), C.c_void_p(h.ctypes.data), C.c_int64(len(h)), C.c_void_p(planes.ctypes.data),
                                    C.c_void_p(lens.ctypes.data), C.c_int64(n), W, int(min_intron), int(max_intron), C.c_int64(max_juncs), C.byref(out), C.byref(n_out))
    assert rc == 0
    a = np.frombuffer((C.c_char * (max(1, n_out.value) * 16)).from_address(out.value), dtype=JUNC_DTYPE)[:n_out.value].copy()
    l.hostsim_free(out)
    return {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in a}


def coverage_search(seqs, hits, ium_reads, min_cov_length: int, min_intron: int = 50, max_intron: int = 20000, max_juncs: int = 5000000):
    """the coverage-search kernels (thj_cov_core.h) as host loops -> set of (ref_id, left, right, antisense)"""
    return coverage_run(seqs, [coverage_state(seqs, hits, ium_reads)], min_cov_length, min_intron, max_intron, max_juncs)
