#!/usr/bin/env python3
"""End-to-end timing of the drop-in executables (GPU box): generate a seeded paired-end case, write the files
tophat.py would hand over (FASTA, FASTQ, id-sorted segment maps as SAM text and as BAM), run
tophat_amd/bin/segment_juncs and long_spanning_reads on them and report wall-clock reads/s per stage.
Usage: python tools/e2e_bench.py [n_pairs] [--bam] [--short]
--short: 2x50 bp reads in two segments, segment_juncs with the coverage search (every read given as --ium-reads), the
way tophat.py runs reads of fewer than three segments."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tophat_amd.synth import make_case, write_case  # noqa: E402
from tophat_amd.bamio import write_bam_from_sam  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200000
use_bam = "--bam" in sys.argv
short = "--short" in sys.argv
RL, NSEG = (50, 2) if short else (100, 4)
BIN = os.path.join(ROOT, "tophat_amd", "bin")
d = tempfile.mkdtemp(prefix="thj_e2e_")
t = time.time()
case = make_case(seed=7, contig_lens=(8_000_000,), n_reads=n_pairs, paired=True, read_len=RL, seg_len=25,
                 genes_per_contig=1500, spliced_seg_frac=0.0 if short else 0.5)
write_case(case, d)
gen_s = time.time() - t


def f(name):
    p = os.path.join(d, name)
    if use_bam and name.endswith(".sam") and name != "hdr.sam":
        o = p[:-4] + ".bam"
        if not os.path.exists(o):
            write_bam_from_sam(p, o)
        return o
    return p


segs = {sd: ",".join(f("%s_seg%d.sam" % (sd, k + 1)) for k in range(NSEG)) for sd in ("left", "right")}
out = {k: os.path.join(d, "out." + k) for k in ("juncs", "insertions", "deletions", "fusions")}
res = {"n_pairs": n_pairs, "read_len": RL, "inputs": "bam" if use_bam else "sam", "gen_seconds": round(gen_s, 1)}
mode = ["--ium-reads", f("left.fq") + "," + f("right.fq")] if short else ["--no-coverage-search"]
cmd = [os.path.join(BIN, "segment_juncs")] + mode + ["--no-microexon-search", "--segment-length", "25",
       "--sam-header", f("hdr.sam"), "-p", "1", "--inner-dist-mean", "50", "--inner-dist-std-dev", "20",
       f("ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"],
       f("left.fq"), f("left_map.sam"), segs["left"], f("right.fq"), f("right_map.sam"), segs["right"]]
t = time.time()
r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, THJ_TIMING="1"))
dt = time.time() - t
assert r.returncode == 0, r.stderr[-2000:]
res["segment_juncs_s"] = round(dt, 3)
res["segment_juncs_reads_per_s"] = round(2 * n_pairs / dt)
res["junctions"] = sum(1 for _ in open(out["juncs"]))
res["segment_juncs_log_tail"] = r.stderr.strip().splitlines()[-14:]
tot = dt
for sd in ("left", "right"):
    cmd = [os.path.join(BIN, "long_spanning_reads"), "--segment-length", "25", "--sam-header", f("hdr.sam"), f("ref.fa"), f("%s.fq" % sd),
           out["juncs"], out["insertions"], out["deletions"], "/dev/null", os.path.join(d, "span_%s.bam" % sd), segs[sd]]
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, THJ_TIMING="1"))
    dt = time.time() - t
    assert r.returncode == 0, r.stderr[-2000:]
    res["long_spanning_reads_%s_s" % sd] = round(dt, 3)
    res["long_spanning_reads_%s_log_tail" % sd] = r.stderr.strip().splitlines()[-14:]
    tot += dt
res["pairs_per_s_both_stages"] = round(n_pairs / tot)
print(json.dumps(res, indent=1))
