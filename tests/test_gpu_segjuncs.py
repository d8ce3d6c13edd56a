"""GPU parity: the HIP path through the C ABI against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

import orc
from tophat_amd import host
from tophat_amd.batch import merge_events
from tophat_amd.params import Params
from tophat_amd.synth import make_case
from util import CASES, assert_events_equal, case_batches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return host.load_lib()   # fails loudly when the extension is missing


def _run_case(cfg, n_reads, chunk=None):
    case = make_case(seed=cfg["seed"], paired=cfg["paired"], read_len=cfg["read_len"], seg_len=cfg["seg_len"],
                     n_reads=n_reads, **cfg.get("gen", {}))
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    og = orc.Genome(seqs)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        want = None
        runs = []
        base = 0
        for side, b in case_batches(case, cfg["paired"]):
            p = Params(segment_length=cfg["seg_len"], read_side=side, **cfg["extra"])
            e = orc.segjuncs(p, og, b)
            want = e if want is None else merge_events(want, e)
            runs.append((p, ctx.upload_batch(b, ordinal_base=base)))
            base += b.n_reads
        got = ctx.segjuncs(runs)
        return got, want


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "seed%d_%s_rl%d_L%d" % (c["seed"], "pe" if c["paired"] else "se", c["read_len"], c["seg_len"]))
def test_segjuncs_matches_oracle(lib, cfg):
    got, want = _run_case(cfg, 600)
    assert len(want.juncs) > 5
    assert_events_equal(got, want)
    assert got.stats["windows"] == want.stats["windows"]
    assert got.stats["indel_pairs"] == want.stats["indel_pairs"]
    assert got.stats["rescue_pairs"] == want.stats["rescue_pairs"]


def test_rerun_is_idempotent(lib):
    """reset + run twice gives the same events; running the same batch twice without a
    reset adds nothing (set semantics of the reference's std::set merges)."""
    cfg = CASES[2]
    case = make_case(seed=21, paired=True, read_len=100, seg_len=25, n_reads=400)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs = []
        for side, b in case_batches(case, True):
            runs.append((Params(read_side=side, **cfg["extra"]), ctx.upload_batch(b)))
        a = ctx.segjuncs(runs)
        b2 = ctx.segjuncs(runs + runs)
        assert_events_equal(a, b2)


def test_empty_and_ragged(lib):
    """empty batch, reads with no hits in most segments, a contig missing from the FASTA"""
    from tophat_amd.batch import build_seg_batch
    case = make_case(seed=5, n_reads=200, drop_seg_frac=0.5)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    og = orc.Genome(seqs)
    b = build_seg_batch(case.seg_recs["left"], case.reads["left"])
    p = Params()
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        got = ctx.segjuncs([(p, ctx.upload_batch(b))])
        assert_events_equal(got, orc.segjuncs(p, og, b))
        empty = build_seg_batch([[] for _ in range(4)], {})
        got = ctx.segjuncs([(p, ctx.upload_batch(empty))])
        assert len(got.juncs) == 0 and len(got.deletions) == 0 and not got.insertions
    # @SQ entry without sequence: every window on it is skipped (segment_juncs.cpp:2105-2108)
    seqs2 = [None] + seqs
    og2 = orc.Genome(seqs2)
    recs = [[(h[0], h[1] + 1) + h[2:] for h in seg] for seg in case.seg_recs["left"]]
    recs[0] = recs[0] + [(h[0], 1) + h[2:] for h in case.seg_recs["left"][0][:20]]
    recs = [sorted(seg, key=lambda h: h[0]) for seg in recs]
    b2 = build_seg_batch(recs, case.reads["left"])
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs2))
        got = ctx.segjuncs([(p, ctx.upload_batch(b2))])
        assert_events_equal(got, orc.segjuncs(p, og2, b2))


def test_queue_overflow_fallback(lib):
    """A tile whose tasks exceed the LDS queue falls back to un-queued execution with identical results."""
    from tophat_amd.batch import HIT_DTYPE, SegBatch
    rng = np.random.default_rng(3)
    L, nseg, n_reads, mh = 25, 4, 256, 12
    glen = 400000
    seq = "".join(rng.choice(list("ACGT"), size=glen))
    # every read: mh hits in seg 0 and mh hits in seg 1 at intron distance -> mh*mh windows per read
    hits, seg_off, bases, read_off = [], [0], bytearray(), [0]
    for r in range(n_reads):
        base = 1000 + r * 1200
        for k in range(mh):
            hits.append((1, base + k * 3, base + k * 3 + L, 0, 0, 0, L))
        seg_off.append(len(hits))
        for k in range(mh):
            hits.append((1, base + 300 + k * 5, base + 300 + k * 5 + L, 0, 0, 0, L))
        seg_off.append(len(hits))
        seg_off.append(len(hits))
        seg_off.append(len(hits))
        bases += seq[base:base + 100].encode()
        read_off.append(len(bases))
    b = SegBatch(nseg, np.arange(1, n_reads + 1, dtype=np.uint32), np.array(read_off, dtype=np.int64),
                 np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.array(seg_off, dtype=np.uint32),
                 np.array(hits, dtype=HIT_DTYPE))
    p = Params()
    og = orc.Genome([seq])
    want = orc.segjuncs(p, og, b)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        got = ctx.segjuncs([(p, ctx.upload_batch(b))])
    assert got.stats["overflow_blocks"] >= 1
    assert want.stats["windows"] == n_reads * mh * mh
    assert_events_equal(got, want)


def test_missing_genome_fails_loudly(lib):
    case = make_case(seed=2, n_reads=50)
    from tophat_amd.batch import build_seg_batch
    b = build_seg_batch(case.seg_recs["left"], case.reads["left"])
    with host.Context(0) as ctx:
        h = ctx.upload_batch(b)
        with pytest.raises(host.ThjError):
            ctx.run(Params(), h)          # no genome resident -> error, never a fallback
        with pytest.raises(host.ThjError):
            ctx.upload_genome(host.pack_genome(["ACGT" * 100]))
            ctx.run(Params(segment_length=65), h)   # unsupported parameter -> error


def rescue_heavy_batch(n_reads=300, seed=11, big=()):
    """Paired reads whose last segment (across an intron) is missing, with several (hit, mate hit) pairs
    each: every read takes the mate-anchored rescue, a tile has far more pairs than the kernel's LDS slots."""
    from tophat_amd.batch import HIT_DTYPE, SegBatch
    rng = np.random.default_rng(seed)
    L, nseg = 25, 4
    glen = 900000
    seq = "".join(rng.choice(list("ACGT"), size=glen))
    hits, seg_off, bases, read_off, mate_off, mate_hits = [], [0], bytearray(), [0], [0], []
    for r in range(n_reads):
        base = 2000 + r * 2500
        intron = int(rng.integers(80, 900))
        read = seq[base:base + 75] + seq[base + 75 + intron:base + 100 + intron]
        n_left = 17 if r == 5 else int(rng.integers(1, 4))
        n_mate = 17 if r == 5 else int(rng.integers(2, 6))
        for (rr, nl, nm) in big:                          # (read, left hits, mate hits) of reads with many pairs
            if r == rr:
                n_left, n_mate = nl, nm
        for k in range(n_left):                           # the true hit first, decoys a little upstream
            hits.append((1, base - 7 * k, base - 7 * k + L, 0, 0, 0, L))
        seg_off.append(len(hits))
        for sgi in (1, 2):                                # segments 1 and 2 map next to segment 0, the last one is missing
            hits.append((1, base + sgi * L, base + (sgi + 1) * L, 0, 0, 0, L))
            seg_off.append(len(hits))
        seg_off.append(len(hits))
        mleft = base + 100 + intron + 30
        for k in range(n_mate):                           # antisense mates downstream, the first one the true mate
            mate_hits.append((1, mleft + 11 * k, mleft + 11 * k + 100, 1 | 2, 0, 0, 100))
        mate_off.append(len(mate_hits))
        bases += read.encode()
        read_off.append(len(bases))
    b = SegBatch(nseg, np.arange(1, n_reads + 1, dtype=np.uint32), np.array(read_off, dtype=np.int64),
                 np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.array(seg_off, dtype=np.uint32),
                 np.array(hits, dtype=HIT_DTYPE), np.array(mate_off, dtype=np.uint32), np.array(mate_hits, dtype=HIT_DTYPE))
    return seq, b


def test_rescue_slot_overflow(lib):
    """More rescue pairs in a tile than LDS slots: the surplus reads recompute theirs on the fly, same events."""
    seq, b = rescue_heavy_batch()
    p = Params()
    want = orc.segjuncs(p, orc.Genome([seq]), b)
    assert want.stats["rescue_pairs"] > 1000 and len(want.juncs) > 100 and want.stats["windows"] > 1000
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        got = ctx.segjuncs([(p, ctx.upload_batch(b))])
    assert got.stats["rescue_pairs"] == want.stats["rescue_pairs"]
    assert got.stats["windows"] == want.stats["windows"]
    assert_events_equal(got, want)


def test_reads_with_many_hits_share_a_wave(lib):
    """Reads with more than 12 hits are done by a wave (thj_k_segjuncs_shared, lane = hit; the rescue kernels for those that take the rescue), reads with 5..64 (hit, mate hit) pairs
    keep their rescue outcomes in the HBM pool, more than 64 recompute them: the same events and counters as the oracle's loops.
    Multihits up to max_seg_multihits = 40 a segment; a read with 41 is dropped whole (segment_juncs.cpp:3499-3506)."""
    from tophat_amd.batch import HIT_DTYPE, SegBatch
    rng = np.random.default_rng(17)
    L, nseg, n_reads = 25, 4, 300
    glen = 900000
    seq = "".join(rng.choice(list("ACGT"), size=glen))
    hits, seg_off, bases, read_off = [], [0], bytearray(), [0]
    for r in range(n_reads):
        base = 2000 + r * 2900
        mh = (40, 41, 13, 26, 1, 2)[r % 6]               # hits per mapped segment
        intron = int(rng.integers(80, 700))
        read = seq[base:base + 50] + seq[base + 50 + intron:base + 100 + intron]
        for sgi, at in ((0, base), (1, base + L), (2, base + 50 + intron), (3, base + 75 + intron)):
            if sgi == 1 and r % 4 == 0:                   # a missing segment: the window comes from the segment after it
                seg_off.append(len(hits))
                continue
            for k in range(mh):                           # the true hit and decoys a few bases off (no two at the same place)
                hits.append((1, at + 3 * k, at + 3 * k + L, 2 if sgi == 3 else 0, 0, 0, L))
            seg_off.append(len(hits))
        bases += read.encode()
        read_off.append(len(bases))
    b = SegBatch(nseg, np.arange(1, n_reads + 1, dtype=np.uint32), np.array(read_off, dtype=np.int64),
                 np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.array(seg_off, dtype=np.uint32),
                 np.array(hits, dtype=HIT_DTYPE))
    p = Params()
    want = orc.segjuncs(p, orc.Genome([seq]), b)
    assert want.stats["windows"] > 100000 and len(want.juncs) > 100
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        got = ctx.segjuncs([(p, ctx.upload_batch(b))])
    assert got.stats["windows"] == want.stats["windows"] and got.stats["indel_pairs"] == want.stats["indel_pairs"]
    assert_events_equal(got, want)
    # the mate-anchored rescue with 40 x 1, 30 x 2 (pool) and 40 x 3 (recomputed) pairs among ordinary reads
    seq, b = rescue_heavy_batch(big=((7, 40, 1), (70, 30, 2), (140, 40, 3), (141, 5, 1), (290, 64, 1)))
    want = orc.segjuncs(p, orc.Genome([seq]), b)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        got = ctx.segjuncs([(p, ctx.upload_batch(b))])
    assert got.stats["rescue_pairs"] == want.stats["rescue_pairs"] and got.stats["windows"] == want.stats["windows"]
    assert_events_equal(got, want)


def test_event_tables_grow(lib):
    """Tables configured far too small for the run grow between batches (rehash into 4x tables); same events."""
    from tophat_amd.synth import make_case
    from util import case_batches
    case = make_case(seed=21, paired=False, read_len=100, seg_len=25, n_reads=1200, contig_lens=(400000,), genes_per_contig=120,
                     indel_frac=0.3, boundary_bias=0.5)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    (side, b), = case_batches(case, False)
    p = Params(read_side=side)
    want = orc.segjuncs(p, g, b)
    assert len(want.juncs) > 150 and len(want.deletions) + len(want.insertions) > 40
    # the same reads as 24 consecutive batches of 50
    from tophat_amd.batch import SegBatch
    parts = []
    for r0 in range(0, b.n_reads, 50):
        r1 = min(b.n_reads, r0 + 50)
        so = b.seg_off[r0 * b.nseg:r1 * b.nseg + 1]
        ro = b.read_off[r0:r1 + 1]
        parts.append((r0, SegBatch(b.nseg, b.read_id[r0:r1], ro - ro[0], b.bases[ro[0]:ro[-1]], (so - so[0]).astype(np.uint32),
                                   b.hits[so[0]:so[-1]])))
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.configure(64, 16)
        got = ctx.segjuncs([(p, ctx.upload_batch(sb, ordinal_base=r0)) for r0, sb in parts])
    assert_events_equal(got, want)


def test_run_pair_equals_two_runs(lib):
    """thj_segjuncs_run_pair_async (both sides as one call, their side chains beside each other on two scratch sets) gives the
    events and the counters of two thj_segjuncs_run_async calls -- on a multihit-heavy case, where the side chains have work"""
    cfg = CASES[2]
    case = make_case(seed=33, paired=True, read_len=100, seg_len=25, n_reads=1500, repeat_frac=0.6)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    og = orc.Genome(seqs)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        runs, want, base = [], None, 0
        for side, b in case_batches(case, True):
            p = Params(read_side=side, **cfg["extra"])
            e = orc.segjuncs(p, og, b)
            want = e if want is None else merge_events(want, e)
            runs.append((p, ctx.upload_batch(b, ordinal_base=base)))
            base += b.n_reads
        a = ctx.segjuncs(runs)
        for _ in range(3):
            ctx.reset()
            ctx.run_pair(runs[0][0], runs[0][1], runs[1][0], runs[1][1])
            cnt = ctx.finish()
            b2 = ctx.download(cnt)
            assert_events_equal(b2, a)
            assert (cnt.n_windows, cnt.n_indel_pairs, cnt.n_rescue_pairs) == (a.stats["windows"], a.stats["indel_pairs"], a.stats["rescue_pairs"])
        assert_events_equal(a, want)
        assert a.stats["rescue_pairs"] == want.stats["rescue_pairs"]


def test_full_task_list_fails_loudly(lib, monkeypatch):
    """the kernels that enumerate from lists write their tasks to one list in HBM: when it is too small the pass fails
    (THJ_EOVERFLOW), it does not lose events"""
    seq, b = rescue_heavy_batch(big=((7, 40, 1), (70, 30, 2), (140, 40, 3), (141, 5, 1), (290, 64, 1)))
    monkeypatch.setenv("THJ_XTASK_CAP", "8")
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        with pytest.raises(host.ThjError, match="task list"):
            ctx.segjuncs([(Params(), ctx.upload_batch(b))])
