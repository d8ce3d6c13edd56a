#!/usr/bin/env python3
"""Informal differential check of the coverage-search oracle (covsearch_oracle.c) against the survey-stage scratch
build of the reference (same caveats as diffcheck_survey_build.py: NOT a pin).  Cases: short paired or single reads
in two segments, every read of the case given as --ium-reads; compared: the junction file of
`segment_juncs --no-microexon-search` (segment search + coverage search) against oracle segment search U oracle
coverage search.

    python oracle/diffcheck_covsearch_survey_build.py [--seeds 1-20]
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
from tophat_amd.batch import HIT_DTYPE, build_seg_batch, hit_tuple_to_struct, merge_events  # noqa: E402
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT  # noqa: E402
from tophat_amd.synth import make_case, write_case  # noqa: E402

REFBIN = os.environ.get("REFBIN", "/tmp/refbuild/src")


def run_case(seed, paired, read_len, seg_len, min_ci=50, max_ci=20000, keep=False, **kw):
    case = make_case(seed=seed, paired=paired, read_len=read_len, seg_len=seg_len, **kw)
    d = tempfile.mkdtemp(prefix="thjcov_")
    paths = write_case(case, d)
    sides = [sd for sd in ("left", "right") if sd in case.reads]
    ium = [paths["%s_fq" % sd] for sd in sides]
    out = {k: os.path.join(d, "ref." + k) for k in ("juncs", "ins", "del", "fus")}
    cmd = [os.path.join(REFBIN, "segment_juncs"), "--no-microexon-search", "--segment-length", str(seg_len), "--sam-header", paths["hdr"],
           "--min-coverage-intron", str(min_ci), "--max-coverage-intron", str(max_ci), "--ium-reads", ",".join(ium),
           paths["ref"], out["juncs"], out["ins"], out["del"], out["fus"], paths["left_fq"], paths["left_map"], ",".join(paths["left_segs"])]
    if paired:
        cmd += [paths["right_fq"], paths["right_map"], ",".join(paths["right_segs"])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("reference failed:", r.stderr[-2000:])
        return False
    g = orc.Genome([orc.fold_genome_char(s) for s in case.seqs])
    p = Params(segment_length=seg_len)
    ev = None
    hits = []
    for sd, side in (("left", READ_LEFT), ("right", READ_RIGHT)):
        if sd not in case.reads:
            continue
        p.read_side = side
        other = "right" if sd == "left" else "left"
        b = build_seg_batch(case.seg_recs[sd], case.reads[sd], case.full_recs[other], case.seg_recs[other][-1]) if paired \
            else build_seg_batch(case.seg_recs[sd], case.reads[sd])
        e = orc.segjuncs(p, g, b)
        ev = e if ev is None else merge_events(ev, e)
        for recs in case.seg_recs[sd]:
            hits += [hit_tuple_to_struct(h) for h in recs]
    ium_reads = [case.reads[sd][rid] for sd in sides for rid in sorted(case.reads[sd])]
    cov = orc.coverage_search(g, np.array(hits, dtype=HIT_DTYPE), ium_reads, min(20, seg_len - 2), min_ci, max_ci)
    mine = {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ev.juncs} | \
           {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in cov}
    ids = {n: i + 1 for i, n in enumerate(case.names)}
    ref = set()
    for l in open(out["juncs"]):
        t = l.split("\t")
        ref.add((ids[t[0]], int(t[1]), int(t[2]), 1 if t[3][0] == "-" else 0))
    seg_only = {(int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ev.juncs}
    ok = mine == ref
    print("seed %d paired=%d rl=%d: reference %d junctions (%d beyond the segment search), oracle %d -> %s" % (
        seed, paired, read_len, len(ref), len(ref - seg_only), len(mine), "identical" if ok else "DIFFERENT"))
    if not ok:
        print("   only reference:", sorted(ref - mine)[:8], " only oracle:", sorted(mine - ref)[:8], " dir:", d)
    elif not keep:
        subprocess.run(["rm", "-rf", d])
    return ok


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1-8")
    a = ap.parse_args()
    lo, hi = (int(x) for x in a.seeds.split("-"))
    bad = 0
    for seed in range(lo, hi + 1):
        for paired in (False, True):
            bad += not run_case(seed, paired, 50, 25, n_reads=600, contig_lens=(40000,), genes_per_contig=8, spliced_seg_frac=0.0)
    print("FAILED: %d" % bad if bad else "all identical")
    sys.exit(1 if bad else 0)
