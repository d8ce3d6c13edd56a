"""Shared helpers for the parity tests."""
from __future__ import annotations

import numpy as np

from tophat_amd.batch import Events, build_seg_batch
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT
from tophat_amd.synth import SynthCase


def assert_events_equal(a: Events, b: Events, what: str = ""):
    assert a.juncs.tolist() == b.juncs.tolist(), "%s junctions differ: %d vs %d" % (what, len(a.juncs), len(b.juncs))
    assert a.deletions.tolist() == b.deletions.tolist(), "%s deletions differ" % what
    assert a.insertions == b.insertions, "%s insertions differ" % what


def case_batches(case: SynthCase, paired: bool):
    """[(side, SegBatch)] in the order the reference processes the sides (left then right)."""
    out = []
    for sd, side in (("left", READ_LEFT), ("right", READ_RIGHT)):
        if sd not in case.reads:
            continue
        other = "right" if sd == "left" else "left"
        if paired:
            b = build_seg_batch(case.seg_recs[sd], case.reads[sd], case.full_recs[other], case.seg_recs[other][-1])
        else:
            b = build_seg_batch(case.seg_recs[sd], case.reads[sd])
        out.append((side, b))
    return out


CASES = [
    dict(seed=1, paired=False, read_len=100, seg_len=25, extra={}),
    dict(seed=2, paired=False, read_len=76, seg_len=25, extra={}),
    dict(seed=3, paired=True, read_len=100, seg_len=25, extra=dict(inner_dist_mean=50, inner_dist_std_dev=20)),
    dict(seed=4, paired=True, read_len=76, seg_len=25, extra=dict(inner_dist_mean=50, inner_dist_std_dev=20)),
    dict(seed=5, paired=False, read_len=150, seg_len=25, extra=dict(library_type=2), gen=dict(repeat_frac=0.3)),
    dict(seed=6, paired=False, read_len=150, seg_len=25, extra=dict(library_type=3), gen=dict(repeat_frac=0.3)),
    dict(seed=7, paired=True, read_len=100, seg_len=20,
         extra=dict(inner_dist_mean=30, inner_dist_std_dev=40, min_segment_intron=30, max_segment_intron=2000),
         gen=dict(err=0.03, n_frac=0.2)),
    dict(seed=8, paired=True, read_len=50, seg_len=25, extra=dict(inner_dist_mean=50, inner_dist_std_dev=20)),
    dict(seed=9, paired=False, read_len=100, seg_len=32, extra={}, gen=dict(n_frac=0.3, err=0.02)),
    # segment_length > 32: 2L read pieces / L+16 support reads on 128-bit plane words
    dict(seed=10, paired=False, read_len=150, seg_len=50, extra={}, gen=dict(indel_frac=0.3)),
    dict(seed=11, paired=True, read_len=200, seg_len=64, extra=dict(inner_dist_mean=50, inner_dist_std_dev=20), gen=dict(indel_frac=0.3, n_frac=0.1)),
    dict(seed=12, paired=False, read_len=160, seg_len=40, extra={}, gen=dict(indel_frac=0.3, err=0.02)),
    # more than eight segments, more than 256 bases (2 x 250 bp at --segment-length 25: ten segments, tophat.py:3486-3492)
    dict(seed=13, paired=True, read_len=250, seg_len=25, extra=dict(inner_dist_mean=50, inner_dist_std_dev=20), gen=dict(exon_range=(150, 900), repeat_frac=0.2)),
    dict(seed=14, paired=False, read_len=400, seg_len=25, extra={}, gen=dict(exon_range=(200, 1200), indel_frac=0.2)),
    dict(seed=15, paired=True, read_len=500, seg_len=32, extra=dict(inner_dist_mean=80, inner_dist_std_dev=30), gen=dict(exon_range=(200, 1500), err=0.02)),
]
