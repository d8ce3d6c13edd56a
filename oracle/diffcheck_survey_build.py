#!/usr/bin/env python3
"""Informal differential check of the oracle against a scratch build of the reference.

NOT part of the test-suite and NOT a pin in the sense of the build rules: the
binaries under $REFBIN (default /tmp/refbuild/src) were produced by the SURVEY
stage with typedef shims standing in for Boost.Thread / boost::shared_ptr and a
hand-written config.h, which this image lacks.  A build that needs stand-ins
does not count as a reference build, so the oracle's formal status stays "parity
unpinned" (oracle/README.md).  The script is kept because it is how the
restatement was debugged, and it documents exactly what was compared.

    python oracle/diffcheck_survey_build.py [--seeds 1-20] [--paired]
"""
import argparse
import filecmp
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
from tophat_amd.batch import build_seg_batch, merge_events, write_segment_files  # noqa: E402
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT  # noqa: E402
from tophat_amd.synth import make_case, write_case  # noqa: E402

REFBIN = os.environ.get("REFBIN", "/tmp/refbuild/src")


def run_case(seed, paired, read_len, seg_len, extra, keep=False, fusion=None, **kw):
    case = make_case(seed=seed, paired=paired, read_len=read_len, seg_len=seg_len, **kw)
    d = tempfile.mkdtemp(prefix="thjdiff_")
    paths = write_case(case, d)
    p = Params(segment_length=seg_len, **extra)
    out = {k: os.path.join(d, "ref." + k) for k in ("juncs", "ins", "del", "fus")}
    cmd = [os.path.join(REFBIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search",
           "--segment-length", str(seg_len), "--sam-header", paths["hdr"]]
    optmap = {"min_segment_intron": "--min-segment-intron", "max_segment_intron": "--max-segment-intron",
              "max_insertion_length": "--max-insertion-length", "max_deletion_length": "--max-deletion-length",
              "inner_dist_mean": "--inner-dist-mean", "inner_dist_std_dev": "--inner-dist-std-dev",
              "max_seg_multihits": "--max-seg-multihits", "segment_mismatches": "--segment-mismatches"}
    if fusion:
        cmd += ["--fusion-search", "--fusion-anchor-length", str(fusion["anchor"]), "--fusion-min-dist", str(fusion["min_dist"])]
    for k, v in extra.items():
        if k == "library_type":
            cmd += ["--library-type", {1: "fr-unstranded", 2: "fr-firststrand", 3: "fr-secondstrand"}[v]]
        else:
            cmd += [optmap[k], str(v)]
    cmd += [paths["ref"], out["juncs"], out["ins"], out["del"], out["fus"],
            paths["left_fq"], paths["left_map"], ",".join(paths["left_segs"])]
    if paired:
        cmd += [paths["right_fq"], paths["right_map"], ",".join(paths["right_segs"])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("reference failed:", r.stderr[-2000:])
        return False
    g = orc.Genome([orc.fold_genome_char(s) for s in case.seqs])
    ev = None
    for sd, side in (("left", READ_LEFT), ("right", READ_RIGHT)):
        if sd not in case.reads:
            continue
        p.read_side = side
        other = "right" if sd == "left" else "left"
        if paired:
            b = build_seg_batch(case.seg_recs[sd], case.reads[sd], case.full_recs[other], case.seg_recs[other][-1])
        else:
            b = build_seg_batch(case.seg_recs[sd], case.reads[sd])
        e = orc.segjuncs(p, g, b)
        ev = e if ev is None else merge_events(ev, e)
    mine = {k: os.path.join(d, "orc." + k) for k in ("juncs", "ins", "del", "fus")}
    write_segment_files(ev, case.names, mine["juncs"], mine["ins"], mine["del"], mine["fus"])
    keys = ["juncs", "ins", "del"]
    if fusion:
        fus = None
        for sd, side in (("left", READ_LEFT), ("right", READ_RIGHT)):
            if sd not in case.reads:
                continue
            p.read_side = side
            other = "right" if sd == "left" else "left"
            if paired:
                b = build_seg_batch(case.seg_recs[sd], case.reads[sd], case.full_recs[other], case.seg_recs[other][-1], include_top0=True)
            else:
                b = build_seg_batch(case.seg_recs[sd], case.reads[sd], include_top0=True)
            f_ = orc.fusions(p, g, b, fusion["anchor"], fusion["min_dist"])
            fus = f_ if fus is None else orc.merge_fusions(fus, f_)
        fus = orc.fusion_filter(fus, ev.juncs)
        orc.write_fusions(fus, case.names, mine["fus"])
        keys.append("fus")
    ok = all(filecmp.cmp(out[k], mine[k], shallow=False) for k in keys)
    nl = sum(1 for _ in open(out["juncs"]))
    nf = sum(1 for _ in open(out["fus"]))
    print("seed %3d paired=%d rl=%d L=%d %s fus=%d: %s  (juncs=%d del=%d ins=%d windows=%d indel_pairs=%d rescue=%d)%s" % (
        seed, paired, read_len, seg_len, extra, nf, "OK" if ok else "MISMATCH", nl,
        sum(1 for _ in open(out["del"])), sum(1 for _ in open(out["ins"])),
        ev.stats.get("windows", 0), ev.stats.get("indel_pairs", 0), ev.stats.get("rescue_pairs", 0),
        "" if ok else "  dir=" + d))
    if ok and not keep:
        subprocess.call(["rm", "-rf", d])
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1-12")
    ap.add_argument("--n", type=int, default=400)
    args = ap.parse_args()
    a, b = args.seeds.split("-")
    seeds = range(int(a), int(b) + 1)
    bad = 0
    for s in seeds:
        bad += not run_case(s, False, 100, 25, {}, n_reads=args.n)
        bad += not run_case(s, False, 76, 25, {}, n_reads=args.n)
        bad += not run_case(s, True, 100, 25, {"inner_dist_mean": 50, "inner_dist_std_dev": 20}, n_reads=args.n)
        bad += not run_case(s, True, 76, 25, {"inner_dist_mean": 50, "inner_dist_std_dev": 20}, n_reads=args.n)
        bad += not run_case(s, False, 150, 25, {"library_type": 2 + s % 2}, n_reads=args.n, repeat_frac=0.3)
        bad += not run_case(s, False, 100, 25, {}, n_reads=args.n, fusion=dict(anchor=20, min_dist=2000), fusion_reads=80,
                            contig_lens=(40000, 30000))
        bad += not run_case(s, True, 150, 25, {"inner_dist_mean": 50, "inner_dist_std_dev": 20}, n_reads=args.n,
                            fusion=dict(anchor=20, min_dist=1000), fusion_reads=80, contig_lens=(40000, 30000, 20000))
        bad += not run_case(s, True, 100, 20, {"inner_dist_mean": 30, "inner_dist_std_dev": 40,
                                               "min_segment_intron": 30, "max_segment_intron": 2000}, n_reads=args.n,
                            err=0.03, n_frac=0.2)
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
