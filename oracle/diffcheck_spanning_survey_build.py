#!/usr/bin/env python3
"""Informal differential check of the long_spanning_reads oracle against the survey-stage
scratch build of the reference (same caveats as diffcheck_survey_build.py: that build used
stand-in headers for Boost / config.h, so this is NOT a formal pin).

    python oracle/diffcheck_spanning_survey_build.py [--seeds 1-10]
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
from tophat_amd.bamio import read_bam  # noqa: E402
from tophat_amd.batch import build_seg_batch, build_span_batch, events_to_span_inputs, write_segment_files  # noqa: E402
from tophat_amd.params import Params  # noqa: E402
from tophat_amd.samtext import parse_sam_hits, read_fastq  # noqa: E402
from tophat_amd.synth import make_case, write_case  # noqa: E402

REFBIN = os.environ.get("REFBIN", "/tmp/refbuild/src")


def run_case(seed, read_len, seg_len, extra, keep=False, **kw):
    case = make_case(seed=seed, paired=False, read_len=read_len, seg_len=seg_len, **kw)
    d = tempfile.mkdtemp(prefix="thjspan_")
    paths = write_case(case, d)
    p = Params(segment_length=seg_len, **extra)
    g = orc.Genome([orc.fold_genome_char(s) for s in case.seqs])
    # candidate events from our segment_juncs oracle (already checked against the reference)
    ref_ids = {n: i + 1 for i, n in enumerate(case.names)}
    seg_recs = [list(parse_sam_hits(f_, ref_ids, p.max_report_intron)) for f_ in paths["left_segs"]]   # host parsing rules
    ev = orc.segjuncs(p, g, build_seg_batch(seg_recs, case.reads["left"]))
    f = {k: os.path.join(d, "seg." + k) for k in ("juncs", "ins", "del", "fus")}
    write_segment_files(ev, case.names, f["juncs"], f["ins"], f["del"], f["fus"])
    out_bam = os.path.join(d, "out.bam")
    cmd = [os.path.join(REFBIN, "long_spanning_reads"), "--segment-length", str(seg_len), "--sam-header", paths["hdr"]]
    optmap = {"max_insertion_length": "--max-insertion-length", "max_deletion_length": "--max-deletion-length",
              "min_report_intron": "--min-report-intron", "max_report_intron": "--max-report-intron",
              "read_mismatches": "--read-mismatches", "read_gap_length": "--read-gap-length",
              "read_edit_dist": "--read-edit-dist", "max_seg_multihits": "--max-seg-multihits"}
    for k, v in extra.items():
        cmd += [optmap[k], str(v)]
    cmd += [paths["ref"], paths["left_fq"], f["juncs"], f["ins"], f["del"], f["fus"], out_bam, ",".join(paths["left_segs"])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("reference failed:", r.stderr[-1500:])
        return False
    _, recs = read_bam(out_bam)
    want = [(q, fl, rn, pos, cig) + tuple(tags) for (q, fl, rn, pos, _mq, cig, _s, _ql, *tags) in recs]
    juncs, ins = events_to_span_inputs(ev)
    sb = build_span_batch(seg_recs, case.reads["left"], case.quals["left"])
    alns = orc.spanning(p, g, sb, juncs, ins)
    got = [a.sam_fields(int(sb.read_id[a.read_idx]), case.names) for a in alns]
    ok = got == want
    nsp = sum(1 for w in want if "N" in w[4])
    nindel = sum(1 for w in want if "I" in w[4] or "D" in w[4])
    print("seed %3d rl=%d L=%d %s: %s  (records=%d spliced=%d indel=%d, oracle records=%d)%s" % (
        seed, read_len, seg_len, extra, "OK" if ok else "MISMATCH", len(want), nsp, nindel, len(got),
        "" if ok else "  dir=" + d))
    if not ok:
        sw, sg = set(want), set(got)
        for x in sorted(sw - sg)[:5]:
            print("   ref only:", x)
        for x in sorted(sg - sw)[:5]:
            print("   orc only:", x)
        if sw == sg:
            print("   same set, different order")
    if ok and not keep:
        subprocess.call(["rm", "-rf", d])
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1-6")
    ap.add_argument("--n", type=int, default=500)
    a = ap.parse_args()
    lo, hi = a.seeds.split("-")
    bad = 0
    for s in range(int(lo), int(hi) + 1):
        bad += not run_case(s, 100, 25, {}, n_reads=a.n, boundary_bias=0.6)
        bad += not run_case(s, 76, 25, {}, n_reads=a.n, boundary_bias=0.6, spliced_seg_frac=0.8)
        bad += not run_case(s, 150, 25, {"read_mismatches": 4, "read_edit_dist": 4, "read_gap_length": 3}, n_reads=a.n,
                            boundary_bias=0.5, spliced_seg_frac=0.9, err=0.02, repeat_frac=0.3)
        bad += not run_case(s, 100, 20, {"min_report_intron": 30, "max_report_intron": 2500}, n_reads=a.n,
                            boundary_bias=0.7, spliced_seg_frac=0.5, n_frac=0.2, indel_frac=0.25)
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
