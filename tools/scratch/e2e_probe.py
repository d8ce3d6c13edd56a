#!/usr/bin/env python3
"""One-off probe (GPU box): the three processes of the e2e leg on one set of files, in different orders and with pauses between
them, with every [timing] line: is a process slower because of the one that ran before it?"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from e2e_bench import mix_gen_args
BIN = os.path.join(ROOT, "tophat_amd", "bin")
d = "/dev/shm/thj_probe"
os.makedirs(d, exist_ok=True)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
if not os.path.exists(d + "/ref.fa"):
    subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", str(pairs), "--read-len", "100", "--genome-len", "64444167", "--introns", "20000"] + mix_gen_args(0.05, 41, 0.03), stdout=subprocess.DEVNULL)
f = lambda n: os.path.join(d, n)
segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k + 1)) for k in range(4)) for sd in ("left", "right")}
out = {k: f("out." + k) for k in ("juncs", "insertions", "deletions", "fusions")}
env = dict(os.environ, THJ_TIMING="1")
def sj():
    return [os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", "25", "--sam-header", f("hdr.sam"), "--inner-dist-mean", "50", "--inner-dist-std-dev", "20",
            f("ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"], f("left_reads.bam"), f("left_map.bam"), segs["left"], f("right_reads.bam"), f("right_map.bam"), segs["right"]]
def lsr(sd):
    return [os.path.join(BIN, "long_spanning_reads"), "--segment-length", "25", "--sam-header", f("hdr.sam"), f("ref.fa"), f("%s_reads.bam" % sd), out["juncs"], out["insertions"], out["deletions"], "/dev/null", f("span_%s.bam" % sd), segs[sd]]
def run(name, cmd, pause=0.0, extra=None):
    if pause:
        time.sleep(pause)
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, **(extra or {})))
    dt = time.time() - t
    tl = [l for l in r.stderr.splitlines() if l.startswith("[timing]")]
    st = [l.split()[-2:] for l in tl if "unix time" in l]
    before = after = None
    if st:
        a, b = map(float, st[0]); before, after = a - t, t + dt - b
    print("%-28s wall %.3f  before main %.3f  after report %.3f  rc %d" % (name, dt, before or -1, after or -1, r.returncode))
    for l in tl:
        if "unix time" not in l:
            print("      " + l[9:])
    sys.stdout.flush()
def runp(name, cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, THJ_EXIT_PROBE="1"))
    print(name, [l for l in r.stderr.splitlines() if "exit-probe" in l and "unix" not in l])
    sys.stdout.flush()
for rep in range(2):
    run("segment_juncs", sj())
    run("lsr left", lsr("left"))
    run("lsr right", lsr("right"))
