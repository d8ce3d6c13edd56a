"""CPU: the torch scale generator (bench.py's workload) is self-consistent: the kernel logic and the oracle
agree on it for both stages, planted junctions are recovered, spliced reads are stitched."""
import numpy as np
import torch

import orc
import sim
from bench import sample_segbatch, sample_spanbatch
from tophat_amd.batch import events_to_span_inputs, merge_events
from tophat_amd.params import Params
from tophat_amd.synth import make_device_workload, make_scale_genome
from util import assert_events_equal


def test_scale_workload_both_stages():
    seqs, genes = make_scale_genome(1, [2_000_000], 1500, intron_max=4000, exon_len=300)
    strs = [s.tobytes().decode() for s in seqs]
    n = 6000
    w = make_device_workload(5, seqs, genes, None, n, "cpu", exon_len=300)
    og = orc.Genome(strs)
    ev = None
    for sd, side in (("left", 1), ("right", 2)):
        p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20)
        sb = sample_segbatch(w[sd], n)
        e = orc.segjuncs(p, og, sb)
        assert_events_equal(sim.segjuncs(p, strs, sb), e)
        ev = e if ev is None else merge_events(ev, e)
    truth = {(1, int(g[2]) - 1, int(g[3])) for g in genes}
    found = {(int(j["ref_id"]), int(j["left"]), int(j["right"])) for j in ev.juncs}
    assert len(found & truth) > 0.5 * len(truth)
    juncs, ins = events_to_span_inputs(ev)
    p = Params()
    n_spliced = 0
    for sd in ("left", "right"):
        spb = sample_spanbatch(w[sd], n)
        want = orc.spanning(p, og, spb, juncs, ins)
        got, status = sim.spanning(p, strs, spb, juncs, ins)
        assert status[1] == 0 and status[2] == 0
        assert got == want
        n_spliced += sum(1 for a in want if any((c >> 28) == 11 for c in a.cigar))
        assert len(want) > 0.5 * n
    assert n_spliced > 0.1 * n


def fusion_list_from_events(f):
    """what long_spanning_reads would load from the .fusions file: (ref1, ref2, left, right, dir) sorted unique"""
    rows = sorted({(int(x["ref_id1"]), int(x["ref_id2"]), int(x["left"]), int(x["right"]), int(x["dir"])) for x in f})
    return np.array(rows, dtype=orc.SPAN_FUSION_DTYPE) if rows else np.zeros(0, dtype=orc.SPAN_FUSION_DTYPE)


def test_scale_workload_with_fusion_pairs():
    """the shape of BASELINE configs[3] (2 x 150 bp, chimeric left reads): segment_juncs --fusion-search finds the planted
    fusions, long_spanning_reads --fusion-search joins the chimeric reads through them -- oracle and kernel logic agree"""
    seqs, genes = make_scale_genome(1, [2_000_000, 1_000_000], 1500, intron_max=4000, exon_len=300)
    strs = [s.tobytes().decode() for s in seqs]
    n = 3000
    w = make_device_workload(7, seqs, genes, None, n, "cpu", exon_len=300, read_len=150, fusion_frac=0.04)
    fz = set(w["left"]["fusion_reads"].tolist())
    assert 60 < len(fz) < 200
    og = orc.Genome(strs)
    ev = fus = None
    for sd, side in (("left", 1), ("right", 2)):
        p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20, fusion_min_dist=100000)
        sb = sample_segbatch(w[sd], n)
        e = orc.segjuncs(p, og, sb)
        ev = e if ev is None else merge_events(ev, e)
        f = orc.fusions(p, og, sb, p.fusion_anchor_length, p.fusion_min_dist)
        f2 = sim.fusions(p, strs, sb)
        assert f.tolist() == f2.tolist()
        fus = f if fus is None else orc.merge_fusions(fus, f)
    assert len(fus) > 0.8 * len(fz)
    juncs, ins = events_to_span_inputs(ev)
    fl = fusion_list_from_events(fus)
    p = Params(fusion_search=1, fusion_min_dist=100000)
    spb = sample_spanbatch(w["left"], n)
    want = orc.spanning_fusion(p, og, spb, juncs, ins, fl, True)
    got, status = sim.spanning_fusion(p, strs, spb, juncs, ins, fl)
    assert status[1] == 0
    assert got == want
    fused = [a for a in want if a.is_fusion()]
    assert len({a.read_idx for a in fused}) > 0.8 * len(fz) and {a.read_idx for a in fused} <= fz
    # the reads that are not chimeric come out as they do without fusion search
    plain = orc.spanning(Params(), og, spb, juncs, ins)
    assert [a for a in want if a.read_idx not in fz] == [a for a in plain if a.read_idx not in fz]


def test_scale_workload_with_deletion_reads():
    """bench.py --indel-frac: left reads with a small deletion, on a segment boundary or inside a segment.  segment_juncs finds the
    planted deletions, long_spanning_reads joins the reads through them (nD in the CIGAR) -- oracle and kernel logic agree."""
    seqs, genes = make_scale_genome(1, [2_000_000], 1500, intron_max=4000, exon_len=300)
    strs = [s.tobytes().decode() for s in seqs]
    n = 4000
    w = make_device_workload(9, seqs, genes, None, n, "cpu", exon_len=300, indel_frac=0.05)
    dz = set(w["left"]["deletion_reads"].tolist())
    truth = {tuple(int(v) for v in row) for row in w["left"]["deletions"].tolist()}
    assert 120 < len(dz) < 300
    og = orc.Genome(strs)
    ev = None
    for sd, side in (("left", 1), ("right", 2)):
        p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20)
        sb = sample_segbatch(w[sd], n)
        e = orc.segjuncs(p, og, sb)
        assert_events_equal(sim.segjuncs(p, strs, sb), e)
        ev = e if ev is None else merge_events(ev, e)
    found = {(int(d["ref_id"]), int(d["left"]), int(d["right"])) for d in ev.deletions}
    # a deletion inside a run of equal bases has several equivalent positions: compare by (contig, length, position within 3)
    hit = sum(1 for (c, l, r) in truth if any((c, l + s, r + s) in found for s in range(-3, 4)))
    assert hit > 0.7 * len(truth), (hit, len(truth), len(found))
    juncs, ins = events_to_span_inputs(ev)
    p = Params()
    spb = sample_spanbatch(w["left"], n)
    want = orc.spanning(p, og, spb, juncs, ins)
    got, status = sim.spanning(p, strs, spb, juncs, ins)
    assert status[1] == 0 and status[2] == 0
    assert got == want
    with_del = {a.read_idx for a in want if any((c >> 28) == 5 for c in a.cigar)}
    assert len(with_del & dz) > 0.35 * len(dz), (len(with_del & dz), len(dz))      # not every planted deletion leaves mismatches to explain


def test_scale_workload_with_a_repeat_family(monkeypatch):
    """bench.py's default mix (multihits from planted repeats up to 41): family pairs have every segment hit at 2..41
    copies, a read with 41 is dropped whole by max_seg_multihits in both stages, deletion reads are found -- oracle and kernel logic
    agree on both stages (the stage-2 tiers included: reads with few hits, many hits, too many joined alignments)"""
    seqs, genes = make_scale_genome(1, [4_000_000], 3000, intron_max=1500, exon_len=300)
    S, copies = 40_000, 41
    for k in range(1, copies):
        seqs[0][k * S:(k + 1) * S] = seqs[0][:S]
    fam = (genes[:, 3] + 300 + 1000 < S)
    uniq = genes[:, 1] >= copies * S + 1000
    assert fam.sum() > 5 and uniq.sum() > 100
    genes = genes[fam | uniq]
    strs = [s.tobytes().decode() for s in seqs]
    n = 4000
    w = make_device_workload(9, seqs, genes, None, n, "cpu", exon_len=300, multi_frac=0.3, dup_shift=S, indel_frac=0.05, max_copies=copies)
    og = orc.Genome(strs)
    # copies per read: the hits of segment 0 of every read with a mapped first segment
    cells = (w["right"]["seg_off"][1:] - w["right"]["seg_off"][:-1]).reshape(n, 4)
    per_read = cells.max(dim=1).values
    assert int((per_read == 2).sum()) > 0.15 * n and int(((per_read >= 3) & (per_read <= 8)).sum()) > 0.01 * n
    assert int((per_read >= 9).sum()) > 5 and int(per_read.max()) == 41
    # the mates' hit groups as the e2e leg's files have them (tools/thj_gen.cpp writes a family read's whole-read map and last
    # segment at every one of its copies): a read's mate group has as many hits as the MATE's segments have copies, each
    # the first one shifted by a multiple of the copy distance -- the rescue's double loop runs over k x k pairs (segment_juncs.cpp:3406-3412)
    for sd, osd in (("left", "right"), ("right", "left")):
        mo = w[sd]["mate_off"].to(torch.int64)
        mcnt = mo[1:] - mo[:-1]
        ocells = (w[osd]["seg_off"][1:] - w[osd]["seg_off"][:-1]).reshape(n, 4)
        ocopies = ocells.max(dim=1).values.to(torch.int64)
        has = mcnt > 0
        assert bool((mcnt[has] == ocopies[has]).all()) and int(mcnt.max()) == 41 and int((mcnt == 2).sum()) > 0.1 * n
        mh = w[sd]["mate_hits"]
        first = mo[:-1][has]
        for r in torch.nonzero(mcnt > 1).reshape(-1)[:200].tolist():
            rows = mh[int(mo[r]):int(mo[r + 1])]
            assert bool((rows[:, 0] == rows[0, 0]).all()) and bool((rows[:, 3] == rows[0, 3]).all())
            assert (rows[:, 1] - rows[0, 1]).tolist() == [k * S for k in range(rows.shape[0])]
            assert bool(((rows[:, 2] - rows[:, 1]) == (rows[0, 2] - rows[0, 1])).all())
    ev = None
    pk = dict(inner_dist_mean=50, inner_dist_std_dev=20, max_segment_intron=20000, max_report_intron=20000)
    for sd, side in (("left", 1), ("right", 2)):
        p = Params(read_side=side, **pk)
        sb = sample_segbatch(w[sd], n)
        e = orc.segjuncs(p, og, sb)
        assert_events_equal(sim.segjuncs(p, strs, sb), e)
        # ... and as the kernels of round 6 take the family's reads: a wave per read with the hits in registers (wave_read_enumerate), the
        # rescue against the pseudo-hit list built a left hit at a time (rescue_pseudo_hits) -- k x k (left hit, mate hit) pairs for a k-copy read
        monkeypatch.setenv("THJ_HOSTSIM_WAVE", "1")
        monkeypatch.setenv("THJ_HOSTSIM_PLIST", "1")
        e2 = sim.segjuncs(p, strs, sb)
        monkeypatch.delenv("THJ_HOSTSIM_WAVE")
        monkeypatch.delenv("THJ_HOSTSIM_PLIST")
        assert_events_equal(e2, e)
        assert (e2.stats["windows"], e2.stats["indel_pairs"], e2.stats["rescue_pairs"]) == (e.stats["windows"], e.stats["indel_pairs"], e.stats["rescue_pairs"])
        ev = e if ev is None else merge_events(ev, e)
    assert len(ev.deletions) > 20
    juncs, ins = events_to_span_inputs(ev)
    p = Params(max_segment_intron=20000, max_report_intron=20000)
    for sd in ("left", "right"):
        spb = sample_spanbatch(w[sd], n)
        want = orc.spanning(p, og, spb, juncs, ins)
        for mode in (0, 2, 3):
            sim.lib().hostsim_wave_reads()
            sim.lib().hostsim_chain_groups()
            got, status = sim.spanning(p, strs, spb, juncs, ins, mode)
            got.sort(key=lambda a: a.read_idx)
            assert status[1] == 0 and status[2] == 0
            assert got == want
            took = sim.lib().hostsim_wave_reads()
            print("mode", mode, "packed tier finished", took, "reads; general arrays", status[3])
            groups = sim.lib().hostsim_chain_groups()
            print("mode", mode, "multihit reads as chain entries", groups)
            if mode == 0:          # most of the family's reads travel as chain entries (thj_k_chains); the packed tier (chains over the lanes of a wave, lanes emulated as fibers) finishes the rest
                assert groups > 0.2 * n and took > 50 and status[3] == 0
            if mode == 2:
                assert groups == 0
            if mode == 3:          # ... and with tiny limits still the reads of few hits, in many rounds
                assert took > 50
        by_read = {}
        for a in want:
            by_read[a.read_idx] = by_read.get(a.read_idx, 0) + 1
        assert max(by_read.values()) >= 20           # a read of a 20+-copy repeat has an alignment in every copy


def test_scale_workload_fusion_search_with_a_repeat_family(monkeypatch):
    """configs[3]'s mode on bench.py's mix in small: 2 x 150 bp, chimeric reads AND a repeat family (nine copies, further apart than
    --fusion-min-dist, so that segment_juncs --fusion-search lists break points between them and long_spanning_reads' search joins a
    family read's segments across copies).  Both stages, oracle against the kernel logic as the device runs it: thj_k_fusion's workgroup
    algorithm, the fusion tier's search one thread a read, and the 64 lanes of a wave on a read (fusion_read_wave) -- with the pair test's
    quick "no" only predicting and checked (THJ_HOSTSIM_QR_VERIFY) in a last pass."""
    import ctypes as C
    seqs, genes = make_scale_genome(1, [3_000_000, 1_000_000], 2500, intron_max=1500, exon_len=300)
    S, copies = 120_000, 9
    for k in range(1, copies):
        seqs[0][k * S:(k + 1) * S] = seqs[0][:S]
    fam = (genes[:, 0] == 0) & (genes[:, 3] + 300 + 1000 < S)
    uniq = (genes[:, 0] != 0) | (genes[:, 1] >= copies * S + 1000)
    assert fam.sum() > 5 and uniq.sum() > 100
    genes = genes[fam | uniq]
    strs = [s.tobytes().decode() for s in seqs]
    n = 1500
    w = make_device_workload(21, seqs, genes, None, n, "cpu", exon_len=300, read_len=150, multi_frac=0.2, dup_shift=S, max_copies=copies, fusion_frac=0.04)
    og = orc.Genome(strs)
    ev = fus = None
    for sd, side in (("left", 1), ("right", 2)):
        p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20, fusion_min_dist=100000)
        sb = sample_segbatch(w[sd], n)
        e = orc.segjuncs(p, og, sb)
        ev = e if ev is None else merge_events(ev, e)
        f = orc.fusions(p, og, sb, p.fusion_anchor_length, p.fusion_min_dist)
        assert sim.fusions(p, strs, sb).tolist() == f.tolist()
        assert sim.fusions_block(p, strs, sb, 2)[0].tolist() == f.tolist()
        fus = f if fus is None else orc.merge_fusions(fus, f)
    juncs, ins = events_to_span_inputs(ev)
    fl = fusion_list_from_events(fus)
    assert any(int(x["ref_id1"]) == 1 and int(x["ref_id2"]) == 1 and abs(int(x["right"]) - int(x["left"])) > S // 2 for x in fus)      # between copies
    p = Params(fusion_search=1, fusion_min_dist=100000)
    for sd in ("left", "right"):
        spb = sample_spanbatch(w[sd], n)
        want = orc.spanning_fusion(p, og, spb, juncs, ins, fl, True)
        per_read = {}
        for a in want:
            per_read[a.read_idx] = per_read.get(a.read_idx, 0) + 1
        assert max(per_read.values()) >= copies
        got, status = sim.spanning_fusion(p, strs, spb, juncs, ins, fl)
        fit = {r for r, k in per_read.items() if k <= 20}            # (a thread alone keeps 24 joined alignments before sort + unique)
        assert [a for a in got if a.read_idx in fit] == [a for a in want if a.read_idx in fit]
        monkeypatch.setenv("THJ_HOSTSIM_FUSWAVE", "8192")
        got, status = sim.spanning_fusion(p, strs, spb, juncs, ins, fl)
        assert status[1] == 0 and got == want
        monkeypatch.setenv("THJ_HOSTSIM_QR_VERIFY", "1")
        said, wrong = C.c_int64(), C.c_int64()
        sim.lib().hostsim_qr_counts(C.byref(said), C.byref(wrong))
        got, status = sim.spanning_fusion(p, strs, spb, juncs, ins, fl)
        assert got == want
        sim.lib().hostsim_qr_counts(C.byref(said), C.byref(wrong))
        assert wrong.value == 0
        print(sd, "records", len(want), "most a read", max(per_read.values()), "quick noes", said.value)
        monkeypatch.delenv("THJ_HOSTSIM_QR_VERIFY")
        monkeypatch.delenv("THJ_HOSTSIM_FUSWAVE")
