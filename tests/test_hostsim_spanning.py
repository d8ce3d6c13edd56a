"""CPU: long_spanning_reads kernel logic (thj_span_core.h compiled for the host) against the plain-C oracle."""
import pytest

import orc
import sim
from tophat_amd.batch import build_seg_batch, build_span_batch, events_to_span_inputs
from tophat_amd.params import Params
from tophat_amd.synth import make_case

SPAN_CASES = [
    dict(seed=1, read_len=100, seg_len=25, extra={}, gen=dict(boundary_bias=0.6)),
    dict(seed=2, read_len=76, seg_len=25, extra={}, gen=dict(boundary_bias=0.6, spliced_seg_frac=0.8)),
    dict(seed=3, read_len=150, seg_len=25, extra=dict(read_mismatches=4, read_edit_dist=4, read_gap_length=3),
         gen=dict(boundary_bias=0.5, spliced_seg_frac=0.9, err=0.02, repeat_frac=0.3)),
    dict(seed=4, read_len=100, seg_len=20, extra=dict(min_report_intron=30, max_report_intron=2500),
         gen=dict(boundary_bias=0.7, spliced_seg_frac=0.5, n_frac=0.2, indel_frac=0.25)),
    dict(seed=5, read_len=50, seg_len=25, extra={}, gen=dict(boundary_bias=0.8, indel_frac=0.3)),
    dict(seed=6, read_len=100, seg_len=25, extra={}, gen=dict(boundary_bias=0.6, indel_frac=0.5, n_frac=0.3)),
    # short exons: joined alignments with 4+ junctions (more cigar ops than the lean tier holds in registers)
    dict(seed=7, read_len=200, seg_len=25, extra=dict(min_report_intron=30), n_reads=1500,
         gen=dict(exon_range=(26, 40), intron_range=(40, 300), indel_frac=0.2, err=0.003, spliced_seg_frac=1.0)),
    # more than eight segments / more than 256 bases (2 x 250 bp at --segment-length 25 is ten segments)
    dict(seed=8, read_len=250, seg_len=25, extra={}, gen=dict(exon_range=(120, 700), boundary_bias=0.5, spliced_seg_frac=0.6, repeat_frac=0.2)),
    # (a read of 400 bases with 1 % errors has four: --read-mismatches 2 would drop nearly all of them)
    dict(seed=9, read_len=400, seg_len=25, extra=dict(read_mismatches=8, read_edit_dist=8, read_gap_length=3),
         gen=dict(exon_range=(150, 900), boundary_bias=0.5, indel_frac=0.2, err=0.004)),
    dict(seed=10, read_len=500, seg_len=32, extra=dict(read_mismatches=8, read_edit_dist=8),
         gen=dict(exon_range=(200, 1200), boundary_bias=0.6, spliced_seg_frac=0.5, err=0.003)),
]


def span_inputs(cfg, n_reads=400):
    case = make_case(seed=cfg["seed"], paired=False, read_len=cfg["read_len"], seg_len=cfg["seg_len"], n_reads=n_reads,
                     **cfg.get("gen", {}))
    p = Params(segment_length=cfg["seg_len"], **cfg["extra"])
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    # host parsing rule: spliced records longer than max_report_intron are dropped (bwt_map.cpp:1341-1345)
    recs = [[h for h in seg if not any(o == 11 and n > p.max_report_intron for o, n in h[9])] for seg in case.seg_recs["left"]]
    ev = orc.segjuncs(p, g, build_seg_batch(recs, case.reads["left"]))
    juncs, ins = events_to_span_inputs(ev)
    sb = build_span_batch(recs, case.reads["left"], case.quals["left"])
    return case, p, seqs, g, sb, juncs, ins


@pytest.mark.parametrize("cfg", SPAN_CASES, ids=lambda c: "seed%d_rl%d_L%d" % (c["seed"], c["read_len"], c["seg_len"]))
def test_span_logic_matches_oracle(cfg):
    case, p, seqs, g, sb, juncs, ins = span_inputs(cfg, cfg.get("n_reads", 400))
    want = orc.spanning(p, g, sb, juncs, ins)
    assert len(want) > 50
    if cfg["seed"] == 7:
        assert sum(1 for a in want if sum(1 for c in a.cigar if c) > 8) > 50
    assert any(any((c >> 28) == 11 for c in a.cigar) for a in want), "no spliced alignment in the case"
    for mode in (0, 1, 2, 3):  # the tiers as the kernels run them (chain entries -> join -> finish), the generic path alone, the tiers without the packed multihit one and without chain entries, with the packed tier's limits made tiny
        sim.lib().hostsim_chain_reads()
        sim.lib().hostsim_chain_deferred()                           # (both counters are the library's: read = reset)
        got, status = sim.spanning(p, seqs, sb, juncs, ins, mode)
        chains = sim.lib().hostsim_chain_reads()
        deferred = sim.lib().hostsim_chain_deferred()
        assert deferred > 0 or chains == 0, (chains, deferred)         # (the chains of multihit reads count among the deferred too)
        if cfg["seed"] == 2 and chains:
            assert deferred < chains                                 # both passes of the join see reads
        assert (chains > 0) == (mode in (0, 3) and cfg["read_len"] // cfg["seg_len"] <= 4), (mode, chains)
        assert status[1] == 0 and status[2] == 0
        # records of one read are emitted together; across reads the device orders by read index afterwards
        got.sort(key=lambda a: a.read_idx)
        assert got == want, "mode %d" % mode
        if mode == 1:
            assert 0 < status[3]


def repeat_span_batch(copies=30, n_reads=40, seed=5):
    """Reads from a `copies`-fold tandem repeat: every segment has `copies` hits and the read has `copies` distinct
    joined alignments (plus a few plain reads) -- the multihit tier with many results per read."""
    import numpy as np
    from tophat_amd.batch import SPAN_HIT_DTYPE, SpanBatch
    rng = np.random.default_rng(seed)
    unit = "".join(rng.choice(list("ACGT"), size=400))
    flank = "".join(rng.choice(list("ACGT"), size=3000))
    seq = flank + unit * copies + flank
    L, nseg, rl = 25, 4, 100
    hits, seg_off, bases, quals, read_off = [], [0], bytearray(), bytearray(), [0]
    for r in range(n_reads):
        off = int(rng.integers(0, 400 - rl))
        for s in range(nseg):
            for c in range(copies):
                left = 3000 + c * 400 + off + s * L
                flags = 2 if s == nseg - 1 else 0
                hits.append((1, left, flags, 0, 0, 1, [(1 << 28) | (L if s < nseg - 1 else rl - s * L), 0, 0, 0, 0]))
            seg_off.append(len(hits))
        bases += unit[off:off + rl].encode()
        quals += b"I" * rl
        read_off.append(len(bases))
    sb = SpanBatch(nseg, np.arange(1, n_reads + 1, dtype=np.uint32), np.array(read_off, dtype=np.int64),
                   np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.frombuffer(bytes(quals), dtype=np.uint8).copy(),
                   np.array(seg_off, dtype=np.uint32), np.array(hits, dtype=SPAN_HIT_DTYPE))
    return seq, sb


def test_many_joined_alignments_per_read():
    import numpy as np
    seq, sb = repeat_span_batch()
    p = Params()
    g = orc.Genome([seq])
    from tophat_amd.batch import JUNC_DTYPE
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    want = orc.spanning(p, g, sb, nj, [])
    assert len(want) == 30 * sb.n_reads
    got, status = sim.spanning(p, [seq], sb, nj, [], 0)
    assert status[1] == 0
    got.sort(key=lambda a: a.read_idx)       # stable: the records of one read keep their emission order
    assert got == want


def repeat_fusion_list(sb):
    """what segment_juncs --fusion-search reports for the reads of repeat_span_batch when the copies lie further apart than the longest
    intron: a break point at every segment boundary, between any two copies, in both orders"""
    import numpy as np
    rows = set()
    h = sb.hits
    for r in range(sb.n_reads):
        so = sb.seg_off[r * sb.nseg:(r + 1) * sb.nseg + 1]
        for s_ in range(sb.nseg - 1):
            for a in h[so[s_]:so[s_ + 1]]:
                for b_ in h[so[s_ + 1]:so[s_ + 2]]:
                    if int(b_["left"]) != int(a["left"]) + 25:
                        rows.add((1, 1, int(a["left"]) + 24, int(b_["left"]), 7))
                        rows.add((1, 1, int(b_["left"]), int(a["left"]) + 24, 7))
    return np.array(sorted(rows), dtype=orc.SPAN_FUSION_DTYPE)


def test_fusion_search_of_a_repeat_read_by_the_wave(monkeypatch):
    """thj_k_stitch_huge under --fusion-search puts the 64 lanes of a wave on one read (fusion_read_wave: a lane a first-segment
    hit, the common list put back in the one-thread order, index merge sort, a lane a record).  The same function over the fibers
    of tests/hostsim/simt.h against the oracle, on reads whose every segment hits all copies of a tandem repeat; with a workspace
    too small for a read's list the read is reported (SPAN_TOO_MANY_JOINED) and nothing of it is emitted."""
    import numpy as np
    from tophat_amd.batch import JUNC_DTYPE
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    for copies, n_reads in ((3, 12), (14, 6), (70, 2)):           # (70: more first-segment hits than lanes)
        seq, sb = repeat_span_batch(copies=copies, n_reads=n_reads, seed=60 + copies)
        p = Params(fusion_search=1, fusion_min_dist=300, max_report_intron=300, max_seg_multihits=100)         # (the copies lie 400 apart on one contig)
        nf = repeat_fusion_list(sb)
        want = orc.spanning_fusion(p, orc.Genome([seq]), sb, nj, [], nf, True)
        per_read = {}
        for a in want:
            per_read[a.read_idx] = per_read.get(a.read_idx, 0) + 1
        assert max(per_read.values()) > copies
        monkeypatch.setenv("THJ_HOSTSIM_FUSWAVE", "16384")
        got, status = sim.spanning_fusion(p, [seq], sb, nj, [], nf, True)
        assert status[1] == 0 and status[2] == 0
        assert got == want, copies
        if copies == 14:
            # a workspace of more than 65 536 alignments (a root's number within the list would not fit its 16 bits), like more than 1 024
            # first-segment hits: lane 0 alone, the list sorted in the workspace (fusion_tail's merge sort)
            monkeypatch.setenv("THJ_HOSTSIM_FUSWAVE", "70000")
            got, status = sim.spanning_fusion(p, [seq], sb, nj, [], nf, True)
            assert status[1] == 0 and got == want
        monkeypatch.setenv("THJ_HOSTSIM_FUSWAVE", str(min(per_read.values()) - 1))       # no read's list fits
        got, status = sim.spanning_fusion(p, [seq], sb, nj, [], nf, True)
        assert got == [] and status[1] == sb.n_reads
        monkeypatch.delenv("THJ_HOSTSIM_FUSWAVE")


def test_quick_no_of_the_pair_test_is_never_wrong(monkeypatch):
    """dfs_seg_hits' pair test has a quick "no" for a plain candidate once the chain has its fusion (fus_quick_reject: five words instead of
    the cigar scans).  THJ_HOSTSIM_QR_VERIFY: it only predicts, the whole test runs -- over random hit geometry (the fusion fuzz's batches:
    both strands, several contigs, overlapping and distant hits) and a repeat family it must say no often, and never where the whole test
    accepts the pair.  The records are the oracle's either way."""
    import ctypes as C
    import numpy as np
    from tophat_amd.batch import JUNC_DTYPE
    from test_fuzz_cpu import fusion_set_near_hits, rand_genome, rand_span_batch
    monkeypatch.setenv("THJ_HOSTSIM_QR_VERIFY", "1")
    said = C.c_int64(); wrong = C.c_int64()
    sim.lib().hostsim_qr_counts(C.byref(said), C.byref(wrong))
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    total = 0
    for seed in range(120):
        rng = np.random.default_rng(9100 + seed)
        seqs = rand_genome(rng, int(rng.integers(1, 4)))
        L = int(rng.choice([20, 25, 25, 40]))
        nseg = int(rng.choice([2, 3, 4, 6]))
        if L * (nseg + 1) > 256:
            nseg = 256 // L - 1
        sb = rand_span_batch(rng, seqs, 60, L, nseg)
        p = Params(segment_length=L, fusion_search=1, max_insertion_length=int(rng.choice([1, 3])), max_report_intron=int(rng.choice([300, 5000, 500000])),
                   fusion_min_dist=int(rng.choice([100, 1500, 10000000])))
        fus = fusion_set_near_hits(rng, sb)
        want = orc.spanning_fusion(p, orc.Genome(seqs), sb, nj, [], fus, True)
        got, status = sim.spanning_fusion(p, seqs, sb, nj, [], fus, True)
        got.sort(key=lambda a: a.read_idx)
        assert got == want
        sim.lib().hostsim_qr_counts(C.byref(said), C.byref(wrong))
        assert wrong.value == 0, seed
        total += said.value
    assert total > 1000
    seq, sb = repeat_span_batch(copies=14, n_reads=6, seed=74)
    p = Params(fusion_search=1, fusion_min_dist=300, max_report_intron=300)
    nf = repeat_fusion_list(sb)
    monkeypatch.setenv("THJ_HOSTSIM_FUSWAVE", "8192")            # (560 joined alignments a read: the wave's workspace holds them)
    assert sim.spanning_fusion(p, [seq], sb, nj, [], nf, True)[0] == orc.spanning_fusion(p, orc.Genome([seq]), sb, nj, [], nf, True)
    sim.lib().hostsim_qr_counts(C.byref(said), C.byref(wrong))
    assert wrong.value == 0 and said.value > 10000


def test_reads_with_more_joined_alignments_than_a_thread_keeps_cpu(monkeypatch):
    """120 joined alignments per read (a 120-copy tandem repeat, --max-seg-multihits raised): more than the 96 a thread of the general tier
    keeps.  span_read reports such a read and emits nothing of it; thj_k_stitch_huge does it again with a workspace (the list and the merge
    sort's scratch) -- the same call here (THJ_HOSTSIM_HUGE_CAP), every record the oracle gives; a workspace that is too small reports again."""
    import numpy as np
    from tophat_amd.batch import JUNC_DTYPE
    nj = np.zeros(0, dtype=JUNC_DTYPE)
    seq, sb = repeat_span_batch(copies=120, n_reads=5, seed=11)
    p = Params(max_seg_multihits=200)
    want = orc.spanning(p, orc.Genome([seq]), sb, nj, [])
    assert len(want) == 120 * 5
    got, status = sim.spanning(p, [seq], sb, nj, [], 0)
    assert got == [] and status[1] == 5
    monkeypatch.setenv("THJ_HOSTSIM_HUGE_CAP", "8192")
    got, status = sim.spanning(p, [seq], sb, nj, [], 0)
    got.sort(key=lambda a: a.read_idx)
    assert status[1] == 0 and got == want
    monkeypatch.setenv("THJ_HOSTSIM_HUGE_CAP", "100")
    got, status = sim.spanning(p, [seq], sb, nj, [], 0)
    assert got == [] and status[1] == 5
