#!/usr/bin/env python3
"""Mints the fixtures under tests/golden/ (inputs + expected outputs).

Expected outputs come from the survey-stage scratch build of the reference binaries ($REFBIN, default
/tmp/refbuild/src) -- a build that needed stand-in headers for Boost / config.h (see oracle/README.md), so these
fixtures are regression data for the oracle, not a formal pin.  Inputs are produced by tophat_amd.synth (seeded).

    python tests/golden/make_golden.py
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tophat_amd.bamio import read_bam, write_bam_from_sam  # noqa: E402
from tophat_amd.synth import make_case, write_case  # noqa: E402

REFBIN = os.environ.get("REFBIN", "/tmp/refbuild/src")

CASES = {
    "se100": dict(gen=dict(seed=101, paired=False, read_len=100, seg_len=25, n_reads=160, boundary_bias=0.5,
                           spliced_seg_frac=0.5, contig_lens=(30000,), genes_per_contig=6, indel_frac=0.15), opts=[]),
    "pe76": dict(gen=dict(seed=102, paired=True, read_len=76, seg_len=25, n_reads=120, boundary_bias=0.4,
                          contig_lens=(24000, 12000), genes_per_contig=4, n_frac=0.1),
                 opts=["--inner-dist-mean", "50", "--inner-dist-std-dev", "20"]),
    "se150_multihit": dict(gen=dict(seed=103, paired=False, read_len=150, seg_len=25, n_reads=120, boundary_bias=0.5,
                                    spliced_seg_frac=0.8, repeat_frac=0.5, contig_lens=(30000,), genes_per_contig=6),
                           opts=["--library-type", "fr-firststrand"]),
    "se100_juncdb": dict(gen=dict(seed=104, paired=False, read_len=100, seg_len=25, n_reads=160, boundary_bias=0.3,
                                  spliced_seg_frac=1.0, juncdb=True, contig_lens=(30000,), genes_per_contig=6, indel_frac=0.05), opts=[]),
    "pe100_fusion": dict(gen=dict(seed=105, paired=True, read_len=100, seg_len=25, n_reads=120, fusion_reads=60,
                                  contig_lens=(24000, 16000), genes_per_contig=4),
                         opts=["--inner-dist-mean", "50", "--inner-dist-std-dev", "20", "--fusion-search", "--fusion-min-dist", "1500"],
                         fusion=True),
    "pe100_fusion_ignore": dict(gen=dict(seed=106, paired=True, read_len=100, seg_len=25, n_reads=120, fusion_reads=80,
                                         contig_lens=(24000, 16000, 12000), genes_per_contig=4),
                                opts=["--inner-dist-mean", "50", "--inner-dist-std-dev", "20", "--fusion-search", "--fusion-min-dist", "1500",
                                      "--fusion-ignore-chromosomes", "chr3"],
                                fusion=True),
}


FUSION_SPAN_CASES = {
    # the whole --fusion-search path incl. long_spanning_reads (tools/fusion_diff.py: segments without a genome hit are placed on
    # the junction database exhaustively, the fusion contigs included)
    "pe100_fusion_span": dict(seed=105, n_reads=120, fusion_reads=60),
    "pe100_fusion_span3": dict(seed=301, n_reads=100, fusion_reads=80, all_segments=True,
                               gen_extra=dict(contig_lens=(20000, 14000, 9000), genes_per_contig=5, indel_frac=0.15, boundary_bias=0.4)),
}


def fusion_span_cases(only):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fusion_diff
    for name, cfg in FUSION_SPAN_CASES.items():
        if only and name not in only:
            continue
        d = os.path.join(HERE, name)
        if os.path.exists(d):
            shutil.rmtree(d)
        fusion_diff.run_case(d, **cfg)
        for f_ in os.listdir(d):
            if f_.endswith(".to_spliced.bam"):
                os.remove(os.path.join(d, f_))
        print(name, sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d)) // 1024, "KiB")


def main():
    only = sys.argv[1:]
    fusion_span_cases(only)
    for name, cfg in CASES.items():
        if only and name not in only:
            continue
        d = os.path.join(HERE, name)
        if os.path.exists(d):
            shutil.rmtree(d)
        case = make_case(**cfg["gen"])
        paths = write_case(case, d)
        seg = [os.path.join(REFBIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length",
               str(cfg["gen"]["seg_len"]), "--sam-header", paths["hdr"]] + cfg["opts"]
        outs = [os.path.join(d, "expected.%s" % k) for k in ("juncs", "insertions", "deletions", "fusions")]
        seg += [paths["ref"]] + outs + [paths["left_fq"], paths["left_map"], ",".join(paths["left_segs"])]
        if cfg["gen"]["paired"]:
            seg += [paths["right_fq"], paths["right_map"], ",".join(paths["right_segs"])]
        subprocess.run(seg, check=True, capture_output=True)
        with open(os.path.join(d, "options.txt"), "w") as f:
            f.write(" ".join(cfg["opts"]) + "\n")
            f.write("segment_length=%d paired=%d\n" % (cfg["gen"]["seg_len"], cfg["gen"]["paired"]))
        # juncs_db on the lists just produced (tophat.py:2574-2586: min_anchor 8, max segment length)
        with open(os.path.join(d, "expected.juncs_db.fa"), "w") as f:
            subprocess.run([os.path.join(REFBIN, "juncs_db"), "8", str(cfg["gen"]["seg_len"]), outs[0], outs[1], outs[2],
                            outs[3] if cfg.get("fusion") else "/dev/null", paths["ref"]], check=True, stdout=f, stderr=subprocess.DEVNULL)
        if cfg.get("fusion"):
            continue          # long_spanning_reads with --fusion-search is not part of the fixtures yet
        os.remove(outs[3])
        for sd in (("left", "right") if cfg["gen"]["paired"] else ("left",)):
            bam = os.path.join(d, "span_%s.bam" % sd)
            lsr = [os.path.join(REFBIN, "long_spanning_reads"), "--segment-length", str(cfg["gen"]["seg_len"]),
                   "--sam-header", paths["hdr"], paths["ref"], paths["%s_fq" % sd], outs[0], outs[1], outs[2], "/dev/null",
                   bam, ",".join(paths["%s_segs" % sd])]
            if "%s_spliced" % sd in paths:        # junction-db maps must be BAM for the reference (samopen "rb")
                sp = []
                for f_ in paths["%s_spliced" % sd]:
                    write_bam_from_sam(f_, f_[:-4] + ".bam")
                    sp.append(f_[:-4] + ".bam")
                lsr.append(",".join(sp))
            subprocess.run(lsr, check=True, capture_output=True)
            for f_ in paths.get("%s_spliced" % sd, []):
                os.remove(f_[:-4] + ".bam")
            _, recs = read_bam(bam)
            with open(os.path.join(d, "expected.span_%s.sam" % sd), "w") as f:
                for r in recs:
                    f.write("\t".join(str(x) for x in r) + "\n")
            os.rename(bam, os.path.join(d, "expected.span_%s.bam" % sd))     # byte-level BAM encoding fixture (B8)
            if os.path.exists(bam + ".index"):
                os.remove(bam + ".index")
        with open(os.path.join(d, "options.txt"), "w") as f:
            f.write(" ".join(cfg["opts"]) + "\n")
            f.write("segment_length=%d paired=%d\n" % (cfg["gen"]["seg_len"], cfg["gen"]["paired"]))
        print(name, sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d)) // 1024, "KiB")


if __name__ == "__main__":
    main()
