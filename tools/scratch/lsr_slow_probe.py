#!/usr/bin/env python3
"""One-off probe (GPU box): the long_spanning_reads that runs right after another process -- on some boxes 1.2-1.3 s instead of 0.65.
Plain 2x50 bp files (where it showed); order, pauses and switches varied; the slowest run's per-phase lines."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "tophat_amd", "bin")
d = "/dev/shm/thj_probe2"
os.makedirs(d, exist_ok=True)
if not os.path.exists(d + "/ref.fa"):
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", "10000000", "--read-len", "50", "--genome-len", "64444167", "--introns", "20000"], stdout=subprocess.DEVNULL)
f = lambda n: os.path.join(d, n)
segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k + 1)) for k in range(2)) for sd in ("left", "right")}
out = {k: f("out." + k) for k in ("juncs", "insertions", "deletions", "fusions")}
env = dict(os.environ, THJ_TIMING="1")
sj = [os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", "25", "--sam-header", f("hdr.sam"), "--inner-dist-mean", "50", "--inner-dist-std-dev", "20",
      f("ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"], f("left_reads.bam"), f("left_map.bam"), segs["left"], f("right_reads.bam"), f("right_map.bam"), segs["right"]]
def lsr(sd, o=None):
    return [os.path.join(BIN, "long_spanning_reads"), "--segment-length", "25", "--sam-header", f("hdr.sam"), f("ref.fa"), f("%s_reads.bam" % sd), out["juncs"], out["insertions"], out["deletions"], "/dev/null", o or f("span_%s.bam" % sd), segs[sd]]
def run(name, cmd, pause=0.0, extra=None, show=False):
    if pause: time.sleep(pause)
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, **(extra or {})))
    dt = time.time() - t
    print("%-44s wall %.3f rc %d" % (name, dt, r.returncode))
    if show or dt > 1.0:
        for l in r.stderr.splitlines():
            if l.startswith("[timing]") and "unix" not in l or l.startswith("[worker") or "ingest]" in l: print("      " + l)
    sys.stdout.flush()
    return dt
run("segment_juncs", sj)
a = run("lsr left", lsr("left"))
b = run("lsr right", lsr("right"))
run("lsr right again", lsr("right"))
run("lsr left again", lsr("left"))
run("lsr right, 1 s pause", lsr("right"), pause=1.0)
run("lsr right, output to /dev/null", lsr("right", "/dev/null"))
run("lsr right, THJ_INGEST_TIMING", lsr("right"), extra={"THJ_INGEST_TIMING": "1"}, show=True)
run("lsr right, THJ_NO_DRAIN", lsr("right"), extra={"THJ_NO_DRAIN": "1"})
run("lsr right, THJ_HOST_BAM", lsr("right"), extra={"THJ_HOST_BAM": "1"})
run("lsr left", lsr("left"))
run("lsr right", lsr("right"))
for i in range(3):
    run("sj", sj); run("  lsr left", lsr("left")); run("  lsr right", lsr("right"))
