"""Microexon search (segment_juncs.cpp:3737-3941): the oracle's restatement against hand-made cases, and the CPU
build of the kernel logic + the product's window merge (tests/hostsim) against the oracle on seeded cases.  Parity status: no
reference vector exists for this mode (oracle/README.md) -- the oracle is a restatement, checked here for what can be checked by hand."""
import numpy as np
import pytest

import mx_util
import orc
import sim
from tophat_amd.batch import build_seg_batch
from tophat_amd.params import Params


def _both(p, seqs, sides, min_intron=50, max_juncs=5000000):
    og = orc.Genome([orc.fold_genome_char(s) for s in seqs])
    want, nw = orc.microexon_search(p, og, [(b, sd) for b, sd, _ in sides], p.min_anchor_len, min_intron, max_juncs)
    got, nw2 = sim.microexon_search(p, [orc.fold_genome_char(s) for s in seqs], sides, min_intron, max_juncs)
    return want, nw, got, nw2


def test_planted_microexon_is_found():
    """one read, one microexon: the junction from exon A into the microexon is the only one a 25-base first segment can support"""
    import random
    rng = random.Random(1)

    def rnd(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    ex_a, mx, ex_b = rnd(100), rnd(12), rnd(200)
    i1 = "GT" + rnd(296).replace("GT", "GA").replace("AG", "AC") + "AG"
    i2 = "GT" + rnd(396).replace("GT", "GA").replace("AG", "AC") + "AG"
    g = rnd(500) + ex_a + i1 + mx + i2 + ex_b + rnd(500)
    pos_mx = 500 + 100 + 300
    pos_b = pos_mx + 12 + 400
    read = ex_a[-13:] + mx + ex_b[:75]
    seg_recs = [[], [], [], []]
    for s_ in (1, 2, 3):
        seg_recs[s_].append((1, 1, pos_b + (s_ - 1) * 25, pos_b + s_ * 25, False, s_ == 3, 0, 0, 25))
    b = build_seg_batch(seg_recs, {1: read})
    want, nw, got, nw2 = _both(Params(), [g], [(b, 1, 0)])
    assert nw == nw2 == 1
    assert [tuple(int(x) for x in r) for r in want] == [(1, 599, 900, 0)]
    assert got.tolist() == want.tolist()
    # the same read reverse-complemented, its segments mirrored: the same junction (strand of the junction from the motif, not the read)
    seg_rc = [[], [], [], []]
    for s_ in (0, 1, 2):
        f0 = 25 + (2 - s_) * 25                          # segment s_ of the sequenced read = forward bases [75 - 25 s_, 100 - 25 s_)
        seg_rc[s_].append((1, 1, pos_b + f0 - 25, pos_b + f0, True, s_ == 3, 0, 0, 25))
    # for an antisense read the LAST segment in file order holds the read's start in transcript order: the reference's test is on file 0,
    # so this read (file 0 mapped, file 3 empty) is not a microexon candidate at all
    b2 = build_seg_batch(seg_rc, {1: mx_util.rc(read)})
    want2, nw_b, got2, nw_b2 = _both(Params(), [g], [(b2, 1, 0)])
    assert nw_b == nw_b2 == 0 and len(want2) == 0 and len(got2) == 0


@pytest.mark.parametrize("seed", range(12))
def test_kernel_logic_against_the_oracle(seed):
    seqs, genes, reads, seg_recs = mx_util.make_case(seed, seg_len=25 if seed % 3 else 20, nseg=4 if seed % 2 else 3)
    L = 25 if seed % 3 else 20
    p = Params(segment_length=L)
    if seed % 4 == 3:
        p.library_type = 2 + (seed // 4) % 2
    ids = sorted(reads)
    half = ids[len(ids) // 2]
    left = {k: v for k, v in reads.items() if k < half}
    right = {k: v for k, v in reads.items() if k >= half}
    bl = build_seg_batch([[r for r in v if r[0] < half] for v in seg_recs], left)
    br = build_seg_batch([[r for r in v if r[0] >= half] for v in seg_recs], right)
    want, nw, got, nw2 = _both(p, seqs, [(bl, 1, 0), (br, 2, 1 << 28)])
    assert nw == nw2 and nw > 3
    assert got.tolist() == want.tolist()
    assert len(want) > 0


def test_the_cut_at_max_juncs():
    seqs, genes, reads, seg_recs = mx_util.make_case(3, n_genes=20, n_reads=400)
    b = build_seg_batch(seg_recs, reads)
    p = Params()
    full, _, got_full, _ = _both(p, seqs, [(b, 1, 0)])
    assert len(full) > 6
    for cap in (1, 3, len(full) - 1):
        want, _, got, _ = _both(p, seqs, [(b, 1, 0)], max_juncs=cap)
        assert got.tolist() == want.tolist() and 0 < len(want) <= cap


def test_window_merging_order_matters_and_is_kept():
    """add_to_microexon_windows merges a new window only with windows that START inside it (std::map lower_bound range): the same
    candidates in another read order give other windows -- both implementations must follow the visiting order"""
    seqs, genes, reads, seg_recs = mx_util.make_case(5, n_genes=6, n_reads=300)
    b = build_seg_batch(seg_recs, reads)
    p = Params()
    w1, n1, g1, m1 = _both(p, seqs, [(b, 1, 0)])
    # the reads renumbered backwards: visiting order reversed
    top = max(reads) + 1
    reads2 = {top - k: v for k, v in reads.items()}
    seg2 = [sorted([(top - r[0],) + tuple(r[1:]) for r in v], key=lambda r: r[0]) for v in seg_recs]
    b2 = build_seg_batch(seg2, reads2)
    w2, n2, g2, m2 = _both(p, seqs, [(b2, 1, 0)])
    assert n1 == m1 and n2 == m2
    assert g1.tolist() == w1.tolist() and g2.tolist() == w2.tolist()
