"""CPU, world_size 2, gloo: the N>1 data path.  Each rank runs stage 1 on its contiguous read-id shard
(utils.cpp:80-127 partitioning), the ranks all-gather their event sets, merge them the way the device merge
kernels do (set union; earlier (side, read) insertion wins) and must reproduce the single-rank result;
stage 2 on each shard with the global set must reproduce the single-rank records of that shard."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard(seg_recs, reads, lo, hi):
    return [[h for h in seg if lo <= h[0] < hi] for seg in seg_recs], {k: v for k, v in reads.items() if lo <= k < hi}


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    import sim
    from tophat_amd.batch import build_seg_batch, build_span_batch, events_to_span_inputs, merge_events
    from tophat_amd.params import Params
    from tophat_amd.synth import make_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = make_case(seed=33, paired=True, read_len=100, seg_len=25, n_reads=400, boundary_bias=0.5, indel_frac=0.3)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    ids = sorted(case.reads["left"])
    cut = ids[len(ids) // 2]
    lo, hi = (0, cut) if rank == 0 else (cut, 1 << 31)
    local = None
    for sd, side in (("left", 1), ("right", 2)):
        other = "right" if sd == "left" else "left"
        p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20)
        segs, reads = shard(case.seg_recs[sd], case.reads[sd], lo, hi)
        full = [h for h in case.full_recs[other] if lo <= h[0] < hi]
        last = [h for h in case.seg_recs[other][-1] if lo <= h[0] < hi]
        e = sim.segjuncs(p, seqs, build_seg_batch(segs, reads, full, last))     # the kernel logic, on this rank's shard
        local = (e, None) if local is None else (local[0], e)
    # one exchange step: all-gather (left events, right events) of every rank
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    merged = None
    for side_idx in (0, 1):                 # all left reads (rank order) before all right reads
        for r in range(world):
            e = gathered[r][side_idx]
            merged = e if merged is None else merge_events(merged, e)
    juncs, ins = events_to_span_inputs(merged)
    segs, reads = shard(case.seg_recs["left"], case.reads["left"], lo, hi)
    quals = {k: v for k, v in case.quals["left"].items() if k in reads}
    sb = build_span_batch(segs, reads, quals)
    alns, _ = sim.spanning(Params(), seqs, sb, juncs, ins)
    recs = [a.sam_fields(int(sb.read_id[a.read_idx]), case.names) for a in alns]
    q.put((rank, merged.juncs.tolist(), merged.deletions.tolist(), merged.insertions, recs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_merge_equals_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from tophat_amd.batch import build_seg_batch, build_span_batch, events_to_span_inputs, merge_events
    from tophat_amd.params import Params
    from tophat_amd.synth import make_case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in procs])
    for p_ in procs:
        p_.join(60)
    # single-rank reference through the oracle
    case = make_case(seed=33, paired=True, read_len=100, seg_len=25, n_reads=400, boundary_bias=0.5, indel_frac=0.3)
    g = orc.Genome([orc.fold_genome_char(s) for s in case.seqs])
    ev = None
    for sd, side in (("left", 1), ("right", 2)):
        other = "right" if sd == "left" else "left"
        p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20)
        e = orc.segjuncs(p, g, build_seg_batch(case.seg_recs[sd], case.reads[sd], case.full_recs[other], case.seg_recs[other][-1]))
        ev = e if ev is None else merge_events(ev, e)
    for r in res:
        assert r[1] == ev.juncs.tolist() and r[2] == ev.deletions.tolist() and r[3] == ev.insertions
    juncs, ins = events_to_span_inputs(ev)
    sb = build_span_batch(case.seg_recs["left"], case.reads["left"], case.quals["left"])
    want = [a.sam_fields(int(sb.read_id[a.read_idx]), case.names) for a in orc.spanning(Params(), g, sb, juncs, ins)]
    assert res[0][4] + res[1][4] == want
    assert len(want) > 100 and len(ev.insertions) > 0


def cov_worker(rank, world, port, q):
    """coverage search with the reads sharded: each rank builds the state of its shard (coverage words, sizes, extension
    entries -- the kernel logic), the ranks all-gather the states and every rank runs the pass on the merged state"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    import sim
    from cov_util import load
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = load("pe50_cov")
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    hits, ium = c["hits"], c["ium"]
    my_hits = hits[rank::world]                      # any partition of the hits / reads will do: the merge is a union
    my_ium = ium[rank::world]
    state = sim.coverage_state(seqs, my_hits, my_ium)
    gathered = [None] * world
    dist.all_gather_object(gathered, state)
    got = sim.coverage_run(seqs, gathered, c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"])
    q.put((rank, sorted(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_coverage_search_equals_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from cov_util import load
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=cov_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in procs])
    for p_ in procs:
        p_.join(60)
    c = load("pe50_cov")
    g = orc.Genome([orc.fold_genome_char(s) for s in c["seqs"]])
    want = sorted((int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"]))
                  for j in orc.coverage_search(g, c["hits"], c["ium"], c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"]))
    assert len(want) > 10
    for r in res:
        assert r[1] == want
