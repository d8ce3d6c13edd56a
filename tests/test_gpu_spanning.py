"""GPU parity: thj_k_stitch through the C ABI against the CPU oracle (exact records, exact order)."""
import os

import pytest

import orc
from tophat_amd import host
from tophat_amd.batch import build_seg_batch
from tophat_amd.params import Params
import numpy as np

from test_hostsim_spanning import SPAN_CASES, repeat_span_batch, span_inputs
from tophat_amd.batch import JUNC_DTYPE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", SPAN_CASES, ids=lambda c: "seed%d_rl%d_L%d" % (c["seed"], c["read_len"], c["seg_len"]))
def test_spanning_matches_oracle(cfg):
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, n_reads=max(800, cfg.get("n_reads", 0)))
    want = orc.spanning(p, g, sb, juncs, ins)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(juncs, ins)
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
    assert len(want) > 100
    assert got == want


def test_pipeline_on_device_sets(tmp_path):
    """segment_juncs tables feed the stitch kernel device-to-device; same records as going through
    the host lists (what the .juncs/.deletions/.insertions files carry)."""
    cfg = SPAN_CASES[3]
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, n_reads=max(800, cfg.get("n_reads", 0)))
    want = orc.spanning(p, g, sb, juncs, ins)
    recs = [[h for h in seg if not any(o == 11 and n > p.max_report_intron for o, n in h[9])] for seg in case.seg_recs["left"]]
    b = build_seg_batch(recs, case.reads["left"])
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ev = ctx.segjuncs([(p, ctx.upload_batch(b))])
        ctx.span_sets_from_segjuncs()
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
        assert got == want
        # and the segment_juncs tables still work afterwards
        ev2 = ctx.segjuncs([(p, ctx.upload_batch(b))])
        assert ev2.juncs.tolist() == ev.juncs.tolist()


def test_two_batches_keep_read_order():
    cfg = SPAN_CASES[0]
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, n_reads=600)
    want = orc.spanning(p, g, sb, juncs, ins)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(juncs, ins)
        h = ctx.upload_span_batch(sb)
        a = ctx.spanning(p, [h])
        b = ctx.spanning(p, [h])        # reset + rerun: identical
    assert a == want and b == want


def test_pair_call_gives_the_records_of_two_runs():
    """thj_span_run_pair_async: two batches beside each other on the context's own streams and scratch sets -- the records of
    thj_span_run_async(batch 0) then thj_span_run_async(batch 1), in the same slots; most one-hit-per-segment reads of the
    100-base case travel as chain entries (tier 0 -> thj_k_join -> thj_k_finish)"""
    ca = SPAN_CASES[0]
    case_a, p, seqs, g, sb_a, juncs, ins = span_inputs(ca, n_reads=900)
    # the second batch: other reads of the same genome (same seed -> same genome; different read count changes the reads drawn)
    case_b, p_b, seqs_b, g_b, sb_b, juncs_b, ins_b = span_inputs(ca, n_reads=500)
    assert seqs_b == seqs
    want_a = orc.spanning(p, g, sb_a, juncs, ins)
    want_b = orc.spanning(p, g, sb_b, juncs, ins)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(juncs, ins)
        ha, hb = ctx.upload_span_batch(sb_a), ctx.upload_span_batch(sb_b)
        two = ctx.spanning(p, [ha, hb])
        ctx.span_reset()
        ctx.span_run_pair(p, ha, hb)
        n = ctx.span_finish()
        assert ctx.span_chain_count() > 50                 # of batch b: reads that went through the join and finish kernels
        from tophat_amd.host import alns_from_array
        pair = alns_from_array(ctx.span_download(n), None)
        ctx.span_reset()
        ctx.span_run_pair(p, hb, ha)
        n2 = ctx.span_finish()
        pair2 = alns_from_array(ctx.span_download(n2), None)
    assert len(two) == len(want_a) + len(want_b)
    key = lambda a: (a.ref_id, a.left, a.antisense, a.antisense_splice, a.cigar, a.MD, a.AS, a.XM, a.mismatches, a.edit_dist)
    assert [key(a) for a in pair] == [key(a) for a in two]
    assert [key(a) for a in pair2[:len(want_b)]] == [key(a) for a in want_b]
    assert [key(a) for a in pair2[len(want_b):]] == [key(a) for a in want_a]


def test_tier0_ahead_of_the_junction_set():
    """thj_span_tier0_pair_async: the pair's tier 0 enqueued before the junction set exists (a caller with both stages resident calls it
    before thj_segjuncs_finish) -- the run that follows does the rest; the same records as the run alone.  Anything else after it is a
    state error"""
    ca = SPAN_CASES[0]
    case_a, p, seqs, g, sb_a, juncs, ins = span_inputs(ca, n_reads=900)
    case_b, p_b, seqs_b, g_b, sb_b, juncs_b, ins_b = span_inputs(ca, n_reads=500)
    from tophat_amd.host import alns_from_array
    key = lambda a: (a.ref_id, a.left, a.antisense, a.antisense_splice, a.cigar, a.MD, a.AS, a.XM, a.mismatches, a.edit_dist)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ha, hb = ctx.upload_span_batch(sb_a), ctx.upload_span_batch(sb_b)
        ctx.upload_span_sets(juncs, ins)
        ctx.span_reset()
        ctx.span_run_pair(p, ha, hb)
        n = ctx.span_finish()
        plain = alns_from_array(ctx.span_download(n), None)
        # the sets of another pass (none at all) are resident while tier 0 goes out; the real ones arrive before the run
        ctx.upload_span_sets(juncs[:0], ins[:0])
        ctx.span_reset()
        ctx.span_tier0_pair(p, ha, hb)
        ctx.upload_span_sets(juncs, ins)
        ctx.span_run_pair(p, ha, hb)
        n2 = ctx.span_finish()
        early = alns_from_array(ctx.span_download(n2), None)
        ctx.span_reset()
        ctx.span_tier0_pair(p, ha, hb)
        with pytest.raises(Exception, match="thj_span_tier0_pair_async is followed"):
            ctx.span_run(p, ha)
        with pytest.raises(Exception, match="thj_span_tier0_pair_async is followed"):
            ctx.span_finish()
        ctx.span_reset()                                   # (forgets the pending tier 0)
        ctx.span_run_pair(p, hb, ha)
        assert ctx.span_finish() == n
    assert n2 == n and [key(a) for a in early] == [key(a) for a in plain]


def test_multihit_reads_as_chain_groups():
    """reads from a tandem repeat of two, three, four and eight copies: every segment has that many hits, every first-segment hit
    starts one chain -- thj_k_chains turns each read into a group of chain entries (rank order known up front), thj_k_join and
    thj_k_finish join and finish them on lanes of their own, the finish numbering the records by ballot; the 30-copy case stays
    with the packed tier.  Records and their order = the oracle's"""
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    # (the copies lie 400 bases apart: with introns of up to 300 a hit chains with its own copy's next hit only -- with the default
    # 500 kb every hit has a choice of successors, dfs_seg_hits searches, and the read stays with the packed tier: the second loop)
    p = Params(max_report_intron=300, max_segment_intron=300)
    with host.Context(0) as ctx:
        for copies, want_groups in ((2, True), (3, True), (4, True), (8, False), (30, False)):
            seq, sb = repeat_span_batch(copies=copies, n_reads=300, seed=20 + copies)
            want = orc.spanning(p, orc.Genome([seq]), sb, nj, [])
            assert len(want) == copies * sb.n_reads
            ctx.upload_genome(host.pack_genome([seq]))
            ctx.upload_span_sets(nj, [])
            got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
            assert got == want, copies
            assert (ctx.span_chain_groups() > 0) == want_groups, copies       # (eight copies of four segments are 32 hits: more than thj_k_chains takes)
        pd = Params()
        for copies in (2, 4):
            seq, sb = repeat_span_batch(copies=copies, n_reads=300, seed=40 + copies)
            want = orc.spanning(pd, orc.Genome([seq]), sb, nj, [])
            ctx.upload_genome(host.pack_genome([seq]))
            ctx.upload_span_sets(nj, [])
            assert ctx.spanning(pd, [ctx.upload_span_batch(sb)]) == want
            assert ctx.span_chain_groups() == 0


def test_full_chain_list_hands_reads_to_the_packed_tier(monkeypatch):
    """THJ_CHAIN_CAP=1024: the dense list of the multihit reads' chains holds two rounds' entries -- the workgroups of thj_k_chains that find it
    full pad what is left of it and leave their reads to thj_k_stitch_pack; the records are the oracle's all the same"""
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    p = Params(max_report_intron=300, max_segment_intron=300)
    seq, sb = repeat_span_batch(copies=3, n_reads=700, seed=77)
    want = orc.spanning(p, orc.Genome([seq]), sb, nj, [])
    monkeypatch.setenv("THJ_CHAIN_CAP", "1024")
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        ctx.upload_span_sets(nj, [])
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
        groups = ctx.span_chain_groups()
    assert got == want and len(want) == 3 * 700
    assert 0 < groups < 700               # some groups fitted, the other reads went on


def test_many_joined_alignments_per_read_gpu():
    """30 joined alignments per read (a 30-copy tandem repeat): multihit tier, overflow pool ordering"""
    import numpy as np
    from test_hostsim_spanning import repeat_span_batch
    from tophat_amd.batch import JUNC_DTYPE
    seq, sb = repeat_span_batch()
    p = Params()
    want = orc.spanning(p, orc.Genome([seq]), sb, np.zeros(0, dtype=JUNC_DTYPE), [])
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        ctx.upload_span_sets(np.zeros(0, dtype=JUNC_DTYPE), [])
        got = ctx.spanning(p, [ctx.upload_span_batch(sb)])
    assert len(want) == 30 * sb.n_reads and got == want


def test_extra_record_pool_grows_and_the_pass_is_rerun():
    """200 reads of a 30-copy tandem repeat give 5800 second-and-later records, more than the pool sized for the pass
    (reads x 1.25 + 4096): thj_span_finish enlarges it and answers THJ_ERETRY, the rerun delivers everything"""
    seq, sb = repeat_span_batch(copies=30, n_reads=200, seed=9)
    p = Params()
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    want = orc.spanning(p, orc.Genome([seq]), sb, nj, [])
    assert len(want) == 30 * 200
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        ctx.upload_span_sets(nj, [])
        h = ctx.upload_span_batch(sb)
        # the plain call sequence sees the retry code once ...
        import ctypes as C
        ctx.span_reset()
        ctx.span_run(p, h)
        n = C.c_int64()
        assert ctx.lib.thj_span_finish(ctx._ctx, C.byref(n)) == -7 and b"run the pass again" in ctx.lib.thj_last_error()
        # ... and the wrapper's loop gets the records
        assert ctx.spanning(p, [h]) == want


def test_long_md_strings_are_rebuilt_on_the_host():
    """an adversarial batch (the fuzz generator, seed 24: N-rich genome, deletions up to 10 bases) whose alignments need MD
    strings of up to 51 characters: beyond the 40 a device record holds, they come back flagged THJ_MD_ON_HOST and are
    rebuilt with thj_md_string -- record for record what the oracle says"""
    from test_gpu_fuzz import span_fuzz_case
    seqs, sb, p, ja = span_fuzz_case(24)
    want = orc.spanning(p, orc.Genome(seqs), sb, ja, [])
    assert sum(1 for a in want if len(a.MD) > 40) > 100
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.upload_span_sets(ja, [])
        assert ctx.spanning(p, [ctx.upload_span_batch(sb)]) == want


def test_reads_with_more_joined_alignments_than_a_thread_keeps():
    """120 joined alignments per read (a 120-copy tandem repeat, --max-seg-multihits raised so that the reads are not dropped):
    more than the 96 a thread of the generic tier keeps.  The first pass reports them, thj_span_finish sets up the big
    workspace and answers THJ_ERETRY, the rerun sends such reads through thj_k_stitch_huge -- every record the oracle gives.
    With fusion search on the fusion tier's smaller array (24) overflows the same way on a 30-copy repeat."""
    import ctypes as C
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    seq, sb = repeat_span_batch(copies=120, n_reads=12, seed=11)
    p = Params(max_seg_multihits=200)
    want = orc.spanning(p, orc.Genome([seq]), sb, nj, [])
    assert len(want) == 120 * 12
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        ctx.upload_span_sets(nj, [])
        h = ctx.upload_span_batch(sb)
        ctx.span_reset()
        ctx.span_run(p, h)
        n = C.c_int64()
        assert ctx.lib.thj_span_finish(ctx._ctx, C.byref(n)) == -7 and b"run the pass again" in ctx.lib.thj_last_error()
        assert ctx.spanning(p, [h]) == want
        assert ctx.spanning(p, [h]) == want                 # and again, the workspace now in place from the start
        # a pair call with the workspace in place: both batches list such reads, from index 0 and with batch-local read
        # numbers, and their thj_k_stitch_huge launches run beside each other -- a list and a workspace per scratch set
        seq_b, sb_b = repeat_span_batch(copies=120, n_reads=9, seed=11)
        assert seq_b == seq
        want_b = orc.spanning(p, orc.Genome([seq]), sb_b, nj, [])
        from tophat_amd.host import alns_from_array
        key = lambda a: (a.ref_id, a.left, a.antisense, a.antisense_splice, a.cigar, a.MD, a.AS, a.XM, a.mismatches, a.edit_dist)
        hb = ctx.upload_span_batch(sb_b)
        for first, second, wa, wb in ((h, hb, want, want_b), (hb, h, want_b, want)):
            ctx.span_reset()
            ctx.span_run_pair(p, first, second)
            n2 = ctx.span_finish()
            got = alns_from_array(ctx.span_download(n2), None)
            assert n2 == len(wa) + len(wb)
            assert [key(a) for a in got[:len(wa)]] == [key(a) for a in wa] and [key(a) for a in got[len(wa):]] == [key(a) for a in wb]
        # fusion search: 30 joined alignments per read are already more than the fusion tier keeps per thread
        seq2, sb2 = repeat_span_batch(copies=30, n_reads=10, seed=12)
        ctx.upload_genome(host.pack_genome([seq2]))
        ctx.upload_span_sets(nj, [])
        ctx.upload_span_fusions(np.zeros(0, dtype=host.SPAN_FUSION_DTYPE))
        pf = Params(fusion_search=1)
        wantf = orc.spanning_fusion(pf, orc.Genome([seq2]), sb2, nj, [], np.zeros(0, dtype=orc.SPAN_FUSION_DTYPE), True)
        assert len(wantf) == 30 * 10
        assert ctx.spanning(pf, [ctx.upload_span_batch(sb2)]) == wantf
        # ... and with the break points between the copies known (what segment_juncs --fusion-search lists for such reads) every first-segment
        # hit joins with every other copy's hits: 3 * k * (k - 1) + k alignments a read, found by the 64 lanes of a wave (fusion_read_wave)
        from test_hostsim_spanning import repeat_fusion_list
        for copies, n_reads in ((14, 40), (40, 70)):
            seq3, sb3 = repeat_span_batch(copies=copies, n_reads=n_reads, seed=80 + copies)
            fl = repeat_fusion_list(sb3)
            pw = Params(fusion_search=1, fusion_min_dist=300, max_report_intron=300)
            wantw = orc.spanning_fusion(pw, orc.Genome([seq3]), sb3, nj, [], fl, True)
            assert len(wantw) == n_reads * (3 * copies * (copies - 1) + copies)
            ctx.upload_genome(host.pack_genome([seq3]))
            ctx.upload_span_sets(nj, [])
            ctx.upload_span_fusions(fl)
            assert ctx.spanning(pw, [ctx.upload_span_batch(sb3)]) == wantw


def test_more_records_per_read_than_the_count_byte_holds():
    """300 records per read (single-segment reads with 300 hits each): the per-read record count the device keeps saturates at 255,
    so the prefix sum thj_span_download compacts with does not add up to the pass's record count -- the kernel notices and the
    download walks the slots on the host instead (span_download_host).  Every record, in order."""
    from tophat_amd.batch import SPAN_HIT_DTYPE, SpanBatch
    rng = np.random.default_rng(13)
    copies, n_reads, rl = 300, 6, 25
    unit = "".join(rng.choice(list("ACGT"), size=400))
    flank = "".join(rng.choice(list("ACGT"), size=3000))
    seq = flank + unit * copies + flank
    hits, seg_off, bases, quals, read_off = [], [0], bytearray(), bytearray(), [0]
    for r in range(n_reads):
        off = int(rng.integers(0, 400 - rl))
        n = copies if r % 2 == 0 else 3                  # every other read is an ordinary one
        for c in range(n):
            hits.append((1, 3000 + c * 400 + off, 2, 0, 0, 1, [(1 << 28) | rl, 0, 0, 0, 0]))
        seg_off.append(len(hits))
        bases += unit[off:off + rl].encode()
        quals += b"I" * rl
        read_off.append(len(bases))
    sb = SpanBatch(1, np.arange(1, n_reads + 1, dtype=np.uint32), np.array(read_off, dtype=np.int64),
                   np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.frombuffer(bytes(quals), dtype=np.uint8).copy(),
                   np.array(seg_off, dtype=np.uint32), np.array(hits, dtype=SPAN_HIT_DTYPE))
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    p = Params(max_seg_multihits=400)
    want = orc.spanning(p, orc.Genome([seq]), sb, nj, [])
    assert len(want) == 3 * copies + 3 * 3
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        ctx.upload_span_sets(nj, [])
        assert ctx.spanning(p, [ctx.upload_span_batch(sb)]) == want


def test_side_streams_are_measured_and_reported():
    """thj_ctx_stream_info: the side streams a pair call runs on are chosen by a measurement, and what it found is
    kept -- with the probe on, both are independent of the context's stream and of each other; forced onto one hardware queue the
    flags go false, a warning is printed per stream, the records stay the same."""
    import subprocess
    import sys
    code = r'''
import os, sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import orc
from tophat_amd import host
from tophat_amd.host import alns_from_array
from test_hostsim_spanning import SPAN_CASES, span_inputs
shift = int(sys.argv[1])
dummies = [torch.cuda.Stream() for _ in range(shift)]
ca = SPAN_CASES[0]
case_a, p, seqs, g, sb_a, juncs, ins = span_inputs(ca, n_reads=700)
case_b, p_b, seqs_b, g_b, sb_b, juncs_b, ins_b = span_inputs(ca, n_reads=400)
want = orc.spanning(p, g, sb_a, juncs, ins) + orc.spanning(p, g, sb_b, juncs, ins)
key = lambda a: (a.ref_id, a.left, a.antisense, a.antisense_splice, a.cigar, a.MD, a.AS, a.XM, a.mismatches, a.edit_dist)
with host.Context(0) as ctx:
    ctx.upload_genome(host.pack_genome(seqs))
    ctx.upload_span_sets(juncs, ins)
    ha, hb = ctx.upload_span_batch(sb_a), ctx.upload_span_batch(sb_b)
    assert ctx.stream_info()["n_side"] == 0
    ctx.span_reset()
    ctx.span_run_pair(p, ha, hb)
    n = ctx.span_finish()
    got = alns_from_array(ctx.span_download(n), None)
    info = ctx.stream_info()
assert [key(a) for a in got] == [key(a) for a in want]
print("INFO " + json.dumps(info))
''' % (ROOT, os.path.join(ROOT, "tests"))

    def run(shift, env_extra):
        r = subprocess.run([sys.executable, "-c", code, str(shift)], capture_output=True, text=True, env=dict(os.environ, **env_extra), timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        import json
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("INFO ")][-1][5:]), r.stderr
    info, err = run(1, {})
    assert info["n_side"] == 2 and info["independent"] == [True, True] and all(0.5 < x < 1.5 for x in info["ratio"]), info
    assert "shares a hardware queue" not in err
    # one hardware queue for the whole process (GPU_MAX_HW_QUEUES=1): no stream can be independent of the context's -- the probe sets four
    # aside, takes the next as it is and says so; with THJ_NO_QUEUE_PROBE=1 the first stream is taken and still measured
    for extra in ({"GPU_MAX_HW_QUEUES": "1"}, {"GPU_MAX_HW_QUEUES": "1", "THJ_NO_QUEUE_PROBE": "1"}):
        info, err = run(0, extra)
        assert info["n_side"] == 2 and info["independent"] == [False, False] and min(info["ratio"]) >= 1.5, info
        assert err.count("shares a hardware queue") == 2
