"""CPU: the torch scale generator (bench.py's workload) is self-consistent: the kernel logic and the oracle
agree on it for both stages, planted junctions are recovered, spliced reads are stitched."""
import numpy as np

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
