"""CPU: the bit-parallel kernel logic (thj_core.h compiled for the host) against
the plain-C oracle on seeded synthetic cases.  No GPU involved."""
import pytest

import orc
import sim
from tophat_amd.batch import merge_events
from tophat_amd.params import Params
from tophat_amd.synth import make_case
from util import CASES, assert_events_equal, case_batches


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "seed%d_%s_rl%d_L%d" % (c["seed"], "pe" if c["paired"] else "se", c["read_len"], c["seg_len"]))
def test_kernel_logic_matches_oracle(cfg):
    case = make_case(seed=cfg["seed"], paired=cfg["paired"], read_len=cfg["read_len"], seg_len=cfg["seg_len"],
                     n_reads=300, **cfg.get("gen", {}))
    seqs = [orc.fold_genome_char(s) for s in case.seqs]
    g = orc.Genome(seqs)
    want = got = None
    for side, b in case_batches(case, cfg["paired"]):
        p = Params(segment_length=cfg["seg_len"], read_side=side, **cfg["extra"])
        e1 = orc.segjuncs(p, g, b)
        e2 = sim.segjuncs(p, seqs, b)
        assert e1.stats["windows"] == e2.stats["windows"]
        assert e1.stats["indel_pairs"] == e2.stats["indel_pairs"]
        want = e1 if want is None else merge_events(want, e1)
        got = e2 if got is None else merge_events(got, e2)
    assert len(want.juncs) > 5
    assert_events_equal(got, want)
