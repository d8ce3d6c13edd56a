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
@pytest.mark.parametrize("variant", ["as_kernel", "general", "lazy_rescue", "mate_scans", "pseudo_hit_list", "wave_registers", "no_trivial_skip"])
def test_kernel_logic_matches_oracle(cfg, variant, monkeypatch):
    # as_kernel: reads with at most one hit per segment take flat_read / flat_rescue (thj_k_sj_flat, thj_k_sj_rescue_flat), the others
    #            the general enumeration with precomputed rescue slots
    # general: the general enumeration for every read that read_is_trivial does not drop (thj_k_sj_general)
    # lazy_rescue: rv_foreach computes rescue_pair on the fly (the kernel's fallback when its LDS slot buffer is full)
    # no_trivial_skip: the general enumeration on every read
    if variant != "as_kernel":
        monkeypatch.setenv("THJ_HOSTSIM_NO_FLAT", "1")
    if variant == "lazy_rescue":
        monkeypatch.setenv("THJ_HOSTSIM_LAZY", "1")
    if variant == "mate_scans":        # thj_k_segjuncs_rescue_shared: one rescue_scan per mate hit, the (hit, mate hit) pairs looked up
        monkeypatch.setenv("THJ_HOSTSIM_MSCAN", "1")
    if variant == "pseudo_hit_list":   # ... since round 6: the pseudo-hit list built once, a left hit at a time (rescue_pseudo_hits), and enumerated against
        monkeypatch.setenv("THJ_HOSTSIM_PLIST", "1")
    if variant == "wave_registers":    # thj_k_segjuncs_shared since round 6: a wave per read, the sweeps on registers (wave_read_enumerate; the wave = 64 fibers)
        monkeypatch.setenv("THJ_HOSTSIM_WAVE", "1")
    if variant == "no_trivial_skip":
        monkeypatch.setenv("THJ_HOSTSIM_NO_SKIP", "1")
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
        assert e1.stats["rescue_pairs"] == e2.stats["rescue_pairs"]
        assert (e2.stats["trivial_reads"] > 0) == (variant != "no_trivial_skip")
        want = e1 if want is None else merge_events(want, e1)
        got = e2 if got is None else merge_events(got, e2)
    assert len(want.juncs) > 5
    assert_events_equal(got, want)
