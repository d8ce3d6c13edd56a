"""CPU: fusion-search kernel logic (thj_core.h: fusion_read) against the plain-C oracle."""
import pytest

import orc
import sim
from tophat_amd.batch import build_seg_batch
from tophat_amd.params import Params
from tophat_amd.synth import make_case

FUSION_CASES = [
    dict(seed=1, paired=False, read_len=100, extra=dict(fusion_min_dist=2000), gen=dict(contig_lens=(40000, 30000))),
    dict(seed=2, paired=True, read_len=150, extra=dict(fusion_min_dist=1000, inner_dist_mean=50, inner_dist_std_dev=20),
         gen=dict(contig_lens=(40000, 30000, 20000))),
    dict(seed=3, paired=True, read_len=76, extra=dict(fusion_min_dist=500, fusion_anchor_length=15, inner_dist_mean=50, inner_dist_std_dev=20),
         gen=dict(contig_lens=(30000, 30000), n_frac=0.2)),
]


def fusion_batches(cfg, n_reads=300):
    case = make_case(seed=cfg["seed"], paired=cfg["paired"], read_len=cfg["read_len"], seg_len=25, n_reads=n_reads,
                     fusion_reads=80, **cfg["gen"])
    out = []
    for sd, side in (("left", 1), ("right", 2)):
        if sd not in case.reads:
            continue
        other = "right" if sd == "left" else "left"
        if cfg["paired"]:
            b = build_seg_batch(case.seg_recs[sd], case.reads[sd], case.full_recs[other], case.seg_recs[other][-1], include_top0=True)
        else:
            b = build_seg_batch(case.seg_recs[sd], case.reads[sd], include_top0=True)
        out.append((Params(read_side=side, **cfg["extra"]), b))
    return case, out


@pytest.mark.parametrize("cfg", FUSION_CASES, ids=lambda c: "seed%d_%s_rl%d" % (c["seed"], "pe" if c["paired"] else "se", c["read_len"]))
def test_fusion_logic_matches_oracle(cfg):
    case, batches = fusion_batches(cfg)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    total = 0
    for p, b in batches:
        want = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist)
        got = sim.fusions(p, seqs, b)
        assert got.tolist() == want.tolist()
        total += len(want)
    assert total > 30


def test_fusion_ignore_chromosomes():
    """--fusion-ignore-chromosomes (segment_juncs.cpp:3214-3231): no pair touching an ignored contig is examined"""
    cfg = FUSION_CASES[1]
    case, batches = fusion_batches(cfg)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    dropped = 0
    for p, b in batches:
        full = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist)
        want = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist, ignore_ref_ids=[2])
        assert all(int(x["ref_id1"]) != 2 and int(x["ref_id2"]) != 2 for x in want)
        assert want.tolist() == [x for x in full.tolist() if x[0] != 2 and x[1] != 2]
        got = sim.fusions(p, seqs, b, ignore_ref_ids=[2])
        assert got.tolist() == want.tolist()
        dropped += len(full) - len(want)
    assert dropped > 5


@pytest.mark.parametrize("cfg", FUSION_CASES, ids=lambda c: "seed%d_%s_rl%d" % (c["seed"], "pe" if c["paired"] else "se", c["read_len"]))
def test_fusion_workgroup_algorithm_matches_oracle(cfg):
    """thj_k_fusion's body (thj_fusion_block.h: a workgroup over tiles of 256 reads, candidate pairs queued and evaluated a pair a lane, the
    mate-anchored part over a list of reads) compiled for the CPU and run over fibers -- one workgroup on all tiles, and three sharing them"""
    case, batches = fusion_batches(cfg, n_reads=700)
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    for p, b in batches:
        want = orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist)
        for n_blocks in (1, 3):
            got, _ = sim.fusions_block(p, seqs, b, n_blocks)
            assert got.tolist() == want.tolist()
        assert sim.fusions_block(p, seqs, b, 1, ignore_ref_ids=[2])[0].tolist() == \
            orc.fusions(p, g, b, p.fusion_anchor_length, p.fusion_min_dist, ignore_ref_ids=[2]).tolist()


def family_fusion_batches(n=3000):
    """bench.py's mix in small (a 41-copy repeat family further apart than --fusion-min-dist, 2 % chimeric reads), and for every third read only
    the first segment mapped: such a read has no partner among its own hits, so the mate-anchored part runs (find_fusions :3117-3202) -- for
    a family read over k x k (hit, mate hit) pairs and then k first-segment hits per pseudo-hit.  -> (genome strings, [(Params, SegBatch)],
    reads with 64 or more (first, last) hit pairs, reads with 256 or more (hit, mate hit, hit) triples)"""
    import numpy as np
    from bench import sample_segbatch
    from tophat_amd.synth import make_device_workload, make_scale_genome
    seqs, genes = make_scale_genome(1, [4_000_000], 3000, intron_max=1500, exon_len=300)
    S, copies = 40_000, 41
    for k in range(1, copies):
        seqs[0][k * S:(k + 1) * S] = seqs[0][:S]
    fam = (genes[:, 3] + 300 + 1000 < S)
    uniq = genes[:, 1] >= copies * S + 1000
    genes = genes[fam | uniq]
    strs = [s.tobytes().decode() for s in seqs]
    w = make_device_workload(9, seqs, genes, None, n, "cpu", exon_len=300, multi_frac=0.3, dup_shift=S, max_copies=copies, fusion_frac=0.02)
    out, heavy_pairs, heavy_mates = [], 0, 0
    for sd, side in (("left", 1), ("right", 2)):
        p = Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20, fusion_min_dist=30000)
        sb = sample_segbatch(w[sd], n)
        so = sb.seg_off.astype(np.int64)
        keep = np.ones(len(sb.hits), dtype=bool)
        for r in range(0, n, 3):
            keep[so[r * sb.nseg + 1]:so[(r + 1) * sb.nseg]] = False
        cnt = np.array([int(keep[so[k]:so[k + 1]].sum()) for k in range(n * sb.nseg)], dtype=np.int64)
        sb.hits = sb.hits[keep]
        sb.seg_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        cells = cnt.reshape(n, sb.nseg)
        heavy_pairs += int(((cells[:, 0] * cells[:, sb.nseg - 1]) >= 64).sum())
        heavy_mates += int((cells[0::3, 0] ** 2 * np.diff(sb.mate_off.astype(np.int64))[0::3] >= 256).sum())
        out.append((p, sb))
    return strs, out, heavy_pairs, heavy_mates


def test_fusion_workgroup_algorithm_on_a_repeat_family():
    """the reads the workgroup takes together (round 6): k x k pairs of a family read a pair a thread, the k x k x 2 x k triples of its
    mate-anchored part a triple a thread after one flank scan per mate hit -- same FusionSimpleSet as the oracle and as the per-read logic, and
    no pair evaluated where it was found (the queue never runs over: it is emptied between the steps)"""
    strs, batches, heavy_pairs, heavy_mates = family_fusion_batches()
    assert heavy_pairs > 20 and heavy_mates > 20
    g = orc.Genome(strs)
    for p, sb in batches:
        want = orc.fusions(p, g, sb, p.fusion_anchor_length, p.fusion_min_dist)
        assert sim.fusions(p, strs, sb).tolist() == want.tolist()
        for n_blocks in (1, 2):
            got, in_place = sim.fusions_block(p, strs, sb, n_blocks)
            assert got.tolist() == want.tolist()
            print("workgroups", n_blocks, "events", int(sum(int(x["count"]) for x in got)), "pairs evaluated in place", in_place)
