"""GPU: the multi-GPU exchange step through the C ABI (thj_comm_*, thj_events_allgather_async, thj_fusion_allgather,
thj_covsearch_allgather; SURVEY section 8e, segment_juncs.cpp:4911-4922).

A box with one GPU cannot hold an RCCL communicator of two ranks, so the two-rank cases run two contexts on device 0
joined by the library's loopback transport (same pack / merge kernels, same finish logic, the all-gather done with
stream-ordered device copies), each rank driven by its own host thread as the C ABI asks.  The RCCL transport itself
is exercised at one rank (ncclCommInitRank + ncclAllGather really run)."""
import copy
import os
import threading

import numpy as np
import pytest

import orc
from tophat_amd import host
from tophat_amd.batch import build_seg_batch
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT
from tophat_amd.synth import make_case
from util import assert_events_equal

pytestmark = pytest.mark.gpu


def _shard(seg_recs, reads, lo, hi):
    return [[h for h in seg if lo <= h[0] < hi] for seg in seg_recs], {k: v for k, v in reads.items() if lo <= k < hi}


def _in_threads(fns):
    """runs every fn on its own thread (ctypes releases the GIL while a rank waits for its peers); re-raises"""
    err = []

    def wrap(f):
        try:
            f()
        except BaseException as e:      # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not any(t.is_alive() for t in th), "a rank hung in the exchange step"
    if err:
        raise err[0]


def _case():
    case = make_case(seed=77, paired=False, read_len=100, seg_len=25, n_reads=3000, boundary_bias=0.5, indel_frac=0.3,
                     contig_lens=(60000, 30000), genes_per_contig=12)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    ids = sorted(case.reads["left"])
    cut = ids[len(ids) // 2]
    whole = build_seg_batch(case.seg_recs["left"], case.reads["left"])
    parts = [build_seg_batch(*_shard(case.seg_recs["left"], case.reads["left"], lo, hi)) for lo, hi in ((0, cut), (cut, 1 << 31))]
    return seqs, whole, parts


def _single(seqs, whole, p):
    with host.Context(0) as one:
        one.upload_genome(host.pack_genome(seqs))
        return one.segjuncs([(p, one.upload_batch(whole))])


def _two_rank_run(seqs, parts, p, configure=None):
    got, info, local = [None, None], [None, None], [None, None]
    with host.Context(0) as a, host.Context(0) as b:
        ctxs = (a, b)
        pg = host.pack_genome(seqs)
        handles, base = [], 0
        for ctx, part in zip(ctxs, parts):
            ctx.upload_genome(pg)
            if configure:
                ctx.configure(*configure)
            handles.append(ctx.upload_batch(part, ordinal_base=base))
            base += part.n_reads
        comms = host.Comm.create_local(ctxs)
        assert comms[0].info()["transport"] == "loopback" and comms[1].info()["n_ranks"] == 2

        def rank(r):
            def go():
                for _ in range(2):                    # twice: the second pass reuses the message sizes the first settled on
                    ctxs[r].reset()
                    ctxs[r].run(p, handles[r])
                    comms[r].events_allgather()
                    got[r] = ctxs[r].download(ctxs[r].finish())
                info[r] = comms[r].info()
            return go
        _in_threads([rank(0), rank(1)])
        for r in (0, 1):                              # what a rank finds alone (no exchange): smaller
            local[r] = ctxs[r].segjuncs([(p, handles[r])])
        for c in comms:
            c.close()
    return got, info, local


def test_two_ranks_through_the_c_collective_equal_one_rank():
    seqs, whole, parts = _case()
    p = Params(read_side=1)
    want = _single(seqs, whole, p)
    got, info, local = _two_rank_run(seqs, parts, p)
    assert len(want.juncs) > 20 and len(want.insertions) > 0 and len(want.deletions) > 0
    assert len(local[0].juncs) < len(want.juncs) or len(local[1].juncs) < len(want.juncs)
    for g in got:
        assert_events_equal(g, want, "two ranks")
    assert info[0]["steps"] == 2 and info[0]["repeats"] == 0


def test_exchange_repeats_itself_when_a_message_section_is_too_small(monkeypatch):
    """THJ_XCHG_CAPS: message sections of 4 / 2 / 2 keys -- the gathered headers say so on every rank, finish grows the
    sections and repeats the step; the second pass then needs no repeat"""
    seqs, whole, parts = _case()
    p = Params(read_side=1)
    want = _single(seqs, whole, p)
    monkeypatch.setenv("THJ_XCHG_CAPS", "4,2,2")
    got, info, _ = _two_rank_run(seqs, parts, p)
    for g in got:
        assert_events_equal(g, want, "two ranks, tiny message")
    assert info[0]["repeats"] >= 1 and info[0]["repeats"] == info[1]["repeats"]
    assert info[0]["steps"] == 2 + info[0]["repeats"] and info[0]["repeats"] <= 4


def test_rccl_transport_at_one_rank():
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather through the C ABI (one rank is all a 1-GPU box can hold)"""
    seqs, whole, _ = _case()
    p = Params(read_side=1)
    want = _single(seqs, whole, p)
    uid = host.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        h = ctx.upload_batch(whole)
        comm = host.Comm.create(ctx, uid, 1, 0)
        assert comm.info()["transport"] == "rccl"
        for _ in range(2):
            ctx.reset()
            ctx.run(p, h)
            comm.events_allgather()
            got = ctx.download(ctx.finish())
            assert_events_equal(got, want, "rccl x1")
        comm.close()


def test_merge_that_overflows_a_small_table_grows_it_and_loses_nothing():
    """each rank's keys fit its 1024-entry table, their union does not: the merge sets the overflow flag, finish moves to
    larger tables and merges the gathered keys again"""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    seqs = ["ACGT" * 20000]
    pg = host.pack_genome(seqs)
    # packed junction keys: [gpos+1 : 34][length : 29][antisense : 1]
    pos = rng.choice(70000, size=1500, replace=False).astype(np.int64)
    keys = ((pos + 1) << 30) | (rng.integers(50, 2000, size=1500).astype(np.int64) << 1) | rng.integers(0, 2, size=1500)
    sets = [keys[:750], keys[700:]]                    # 50 shared
    got = [None, None]
    with host.Context(0) as a, host.Context(0) as b:
        ctxs = (a, b)
        dev = []
        for ctx, ks in zip(ctxs, sets):
            ctx.upload_genome(pg)
            ctx.configure(1024, 1024)
            dev.append(torch.from_numpy(np.ascontiguousarray(ks)).cuda())
        torch.cuda.synchronize()
        comms = host.Comm.create_local(ctxs)

        def rank(r):
            def go():
                ctxs[r].reset()
                ctxs[r].merge_keys(0, dev[r].data_ptr(), int(dev[r].numel()))
                comms[r].events_allgather()
                cnt = ctxs[r].finish()
                got[r] = ctxs[r].download(cnt)
            return go
        _in_threads([rank(0), rank(1)])
        for c in comms:
            c.close()
    want = sorted(set(int(k) for k in keys))
    for g in got:
        assert len(g.juncs) == len(want) == 1500
        back = sorted(((int(j["left"]) + 1) << 30) | ((int(j["right"]) - int(j["left"])) << 1) | int(j["antisense"]) for j in g.juncs)
        assert back == want


def test_two_rank_fusion_sets_merge_like_merge_with():
    from test_hostsim_fusions import FUSION_CASES, fusion_batches
    case, batches = fusion_batches(FUSION_CASES[0], n_reads=500)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    pg = host.pack_genome(seqs)
    with host.Context(0) as one:
        one.upload_genome(pg)
        want = one.fusions([(p, one.upload_batch(b)) for p, b in batches])
    # shards: every batch cut in two by read id (both ranks see reads of every side)
    halves = [[], []]
    for p, b in batches:
        cut = int(b.read_id[len(b.read_id) // 2])
        for r, (lo, hi) in enumerate(((0, cut), (cut, 1 << 31))):
            halves[r].append((p, b.select((b.read_id >= lo) & (b.read_id < hi))))
    got = [None, None]
    with host.Context(0) as a, host.Context(0) as b2:
        ctxs = (a, b2)
        for ctx in ctxs:
            ctx.upload_genome(pg)
        comms = host.Comm.create_local(ctxs)

        def rank(r):
            def go():
                mine = ctxs[r].fusions([(p, ctxs[r].upload_batch(b)) for p, b in halves[r]])     # this rank's FusionSimpleSet
                assert len(mine) <= len(want)
                got[r] = comms[r].fusion_allgather()
            return go
        _in_threads([rank(0), rank(1)])
        for c_ in comms:
            c_.close()
    for g in got:
        assert g.tolist() == want.tolist()
    assert len(want) > 30


def test_two_rank_coverage_search_through_the_collective():
    """thj_covsearch_allgather: the left side's hits and unmapped reads on one rank, the right side's on the other; after the
    exchange both ranks run the same coverage search and -- after the event exchange -- hold the one-rank junction set"""
    from cov_util import load
    c = load("pe50_cov")
    seqs = [orc.fold_genome_char(s) for s in c["seqs"]]
    n_left = sum(1 for _ in open(os.path.join(c["dir"], "left.fq"))) // 4
    ium = [c["ium"][:n_left], c["ium"][n_left:]]
    args = (c["cov"]["min_cov_length"], c["cov"]["min_intron"], c["cov"]["max_intron"])
    pg = host.pack_genome(seqs)
    with host.Context(0) as one:
        one.upload_genome(pg)
        runs, base = [], 0
        for side, sb in c["seg_batches"]:
            p = copy.copy(c["p"]); p.read_side = side
            runs.append((p, one.upload_batch(sb, ordinal_base=base)))
            base += sb.n_reads
        want, want_found = one.segjuncs_with_coverage_search(runs, c["ium"], *args)
    got, found = [None, None], [None, None]
    with host.Context(0) as a, host.Context(0) as b:
        ctxs = (a, b)
        for ctx in ctxs:
            ctx.upload_genome(pg)
        comms = host.Comm.create_local(ctxs)
        bases = [0, c["seg_batches"][0][1].n_reads]

        def rank(r):
            def go():
                side, sb = c["seg_batches"][r]
                p = copy.copy(c["p"]); p.read_side = side
                ctx = ctxs[r]
                ctx.reset()
                ctx.covsearch_reset()
                h = ctx.upload_batch(sb, ordinal_base=bases[r])
                ctx.run(p, h)
                ctx.covsearch_add_hits(h)
                ctx.covsearch_add_reads(ium[r])
                comms[r].covsearch_allgather()
                ctx.covsearch_run(*args)
                found[r] = ctx.covsearch_finish()
                comms[r].events_allgather()
                got[r] = ctx.download(ctx.finish())
            return go
        _in_threads([rank(0), rank(1)])
        for c_ in comms:
            c_.close()
    assert want_found > 0
    for r in (0, 1):
        assert found[r] == want_found
        assert_events_equal(got[r], want, "coverage search, rank %d" % r)


def test_full_task_list_fails_loudly_with_an_exchange_step_pending(monkeypatch):
    """the task list of one rank fills up (THJ_XTASK_CAP = 8): its events are incomplete, so after the exchange step the pass must
    fail on that rank with the task-list message and on its peer with the overflow verdict of the gathered headers -- never a
    silently smaller event set (round 4 looked at that flag only without an exchange step)"""
    from test_gpu_segjuncs import rescue_heavy_batch
    seq, heavy = rescue_heavy_batch(big=((7, 40, 1), (70, 30, 2), (140, 40, 3), (141, 5, 1), (290, 64, 1)))
    _, light = rescue_heavy_batch(n_reads=40, seed=12)
    monkeypatch.setenv("THJ_XTASK_CAP", "8")
    errs = [None, None]
    with host.Context(0) as a, host.Context(0) as b:
        ctxs = (a, b)
        pg = host.pack_genome([seq])
        handles = []
        for ctx, part in zip(ctxs, (heavy, light)):
            ctx.upload_genome(pg)
            handles.append(ctx.upload_batch(part))
        comms = host.Comm.create_local(ctxs)

        def rank(r):
            def go():
                ctxs[r].reset()
                ctxs[r].run(Params(), handles[r])
                comms[r].events_allgather()
                try:
                    ctxs[r].finish()
                except host.ThjError as e:
                    errs[r] = str(e)
            return go
        _in_threads([rank(0), rank(1)])
        for c in comms:
            c.close()
    assert errs[0] is not None and "task list" in errs[0]
    # (its peer: the verdict of the gathered headers -- or, when its own small batch overflows the eight-entry list too, its own message)
    assert errs[1] is not None and ("overflow" in errs[1] or "task list" in errs[1])
