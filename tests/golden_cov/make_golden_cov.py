#!/usr/bin/env python3
"""Mints the coverage-search fixtures under tests/golden_cov/ (inputs + expected outputs): short reads in two
segments, `segment_juncs` run WITHOUT --no-coverage-search and with every read of the case as --ium-reads.

Expected outputs come from the survey-stage scratch build of the reference ($REFBIN, default /tmp/refbuild/src) -- a
build that needed stand-in headers (see oracle/README.md), so these fixtures are regression data for the oracle, not
a formal pin.  Inputs are produced by tophat_amd.synth (seeded).  expected.juncs = segment search + coverage search;
expected.seg_only.juncs = the same run with --no-coverage-search.

    python tests/golden_cov/make_golden_cov.py
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tophat_amd.bamio import read_bam  # noqa: E402
from tophat_amd.synth import make_case, write_case  # noqa: E402

REFBIN = os.environ.get("REFBIN", "/tmp/refbuild/src")

CASES = {
    # reads of a single segment: no segment search at all, the coverage search alone finds the junctions
    "se30_cov": dict(gen=dict(seed=203, paired=False, read_len=30, seg_len=25, n_reads=900, contig_lens=(30000,), genes_per_contig=10,
                              spliced_seg_frac=0.0), opts=[]),
    "se50_cov": dict(gen=dict(seed=201, paired=False, read_len=50, seg_len=25, n_reads=900, contig_lens=(40000,), genes_per_contig=10,
                              spliced_seg_frac=0.0), opts=[]),
    "pe50_cov": dict(gen=dict(seed=202, paired=True, read_len=50, seg_len=25, n_reads=700, contig_lens=(30000, 20000), genes_per_contig=8,
                              spliced_seg_frac=0.0, n_frac=0.05),
                     opts=["--inner-dist-mean", "50", "--inner-dist-std-dev", "20", "--min-coverage-intron", "60", "--max-coverage-intron", "8000"]),
}


def main():
    only = sys.argv[1:]
    for name, cfg in CASES.items():
        if only and name not in only:
            continue
        d = os.path.join(HERE, name)
        if os.path.exists(d):
            shutil.rmtree(d)
        case = make_case(**cfg["gen"])
        paths = write_case(case, d)
        sides = ("left", "right") if cfg["gen"]["paired"] else ("left",)
        ium = ",".join(paths["%s_fq" % sd] for sd in sides)
        for tag, extra in (("", ["--ium-reads", ium]), ("seg_only.", ["--no-coverage-search"])):
            outs = [os.path.join(d, "expected.%s%s" % (tag, k)) for k in ("juncs", "insertions", "deletions", "fusions")]
            cmd = [os.path.join(REFBIN, "segment_juncs"), "--no-microexon-search", "--segment-length", str(cfg["gen"]["seg_len"]),
                   "--sam-header", paths["hdr"]] + cfg["opts"] + extra + [paths["ref"]] + outs + \
                  [paths["left_fq"], paths["left_map"], ",".join(paths["left_segs"])]
            if cfg["gen"]["paired"]:
                cmd += [paths["right_fq"], paths["right_map"], ",".join(paths["right_segs"])]
            subprocess.run(cmd, check=True, capture_output=True)
            os.remove(outs[3])
            if tag:
                os.remove(outs[1]); os.remove(outs[2])
        # long_spanning_reads on the junctions of the full run (short reads: one or two segments per read)
        for sd in sides:
            bam = os.path.join(d, "span_%s.bam" % sd)
            subprocess.run([os.path.join(REFBIN, "long_spanning_reads"), "--segment-length", str(cfg["gen"]["seg_len"]), "--sam-header", paths["hdr"],
                            paths["ref"], paths["%s_fq" % sd], os.path.join(d, "expected.juncs"), os.path.join(d, "expected.insertions"),
                            os.path.join(d, "expected.deletions"), "/dev/null", bam, ",".join(paths["%s_segs" % sd])], check=True, capture_output=True)
            _, recs = read_bam(bam)
            with open(os.path.join(d, "expected.span_%s.sam" % sd), "w") as f:
                for r in recs:
                    f.write("\t".join(str(x) for x in r) + "\n")
            os.remove(bam)
            if os.path.exists(bam + ".index"):
                os.remove(bam + ".index")
        with open(os.path.join(d, "options.txt"), "w") as f:
            f.write(" ".join(cfg["opts"]) + "\n")
            f.write("segment_length=%d paired=%d\n" % (cfg["gen"]["seg_len"], cfg["gen"]["paired"]))
        nj = sum(1 for _ in open(os.path.join(d, "expected.juncs")))
        ns = sum(1 for _ in open(os.path.join(d, "expected.seg_only.juncs")))
        print(name, "%d junctions, %d from the segment search alone;" % (nj, ns),
              sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d)) // 1024, "KiB")


if __name__ == "__main__":
    main()
