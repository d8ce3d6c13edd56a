#!/usr/bin/env python3
"""round 5, VERDICT item 5: the 8-byte load of SEQ nibbles in thj_k_read_planes that 'came back wrong now and then'.  One set of
files (80 000 pairs of the mix), segment_juncs once, then long_spanning_reads N times per variant; a run is bad when its spanning
BAM's stream differs from the byte-load reference's.  Usage: r05_load64_repro.py [N]"""
import gzip, hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "tophat_amd", "bin")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d = tempfile.mkdtemp(prefix="thj_l64_", dir="/dev/shm")
subprocess.check_call([os.path.join(ROOT, "tools", "bin", "thj_gen"), "--out", d, "--pairs", "80000", "--genome-len", "52000000", "--introns", "16000",
                       "--multihit-frac", "0.08", "--max-copies", "41", "--indel-frac", "0.03", "--threads", "16"], stdout=subprocess.DEVNULL)
f = lambda n: os.path.join(d, n)
segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k)) for k in (1, 2, 3, 4)) for sd in ("left", "right")}
subprocess.check_call([os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", "25", "--sam-header", f("hdr.sam"),
                       "--inner-dist-mean", "50", "--inner-dist-std-dev", "20", f("ref.fa"), f("o.juncs"), f("o.ins"), f("o.del"), f("o.fus"),
                       f("left_reads.bam"), f("left_map.bam"), segs["left"], f("right_reads.bam"), f("right_map.bam"), segs["right"]], stderr=subprocess.DEVNULL)

def run(env):
    out = f("span.bam")
    r = subprocess.run([os.path.join(BIN, "long_spanning_reads"), "--segment-length", "25", "--sam-header", f("hdr.sam"), f("ref.fa"), f("left_reads.bam"),
                        f("o.juncs"), f("o.ins"), f("o.del"), "/dev/null", out, segs["left"]], env=dict(os.environ, **env), capture_output=True, text=True)
    if r.returncode:
        return "rc%d" % r.returncode
    return hashlib.sha256(gzip.open(out, "rb").read()).hexdigest()[:12]

ref = run({"THJ_SHARDS": "7"})
print("reference (byte loads):", ref, flush=True)
for name, env in (("byte loads", {}), ("8-byte load", {"THJ_PLANES_LOAD64": "1"}), ("8-byte load + device idle before thj_k_read_planes", {"THJ_PLANES_LOAD64": "1", "THJ_INGEST_SYNC": "1"}),
                  ("8-byte load, AMD_SERIALIZE_KERNEL=3", {"THJ_PLANES_LOAD64": "1", "AMD_SERIALIZE_KERNEL": "3"}),
                  ("8-byte load, one context per GPU", {"THJ_PLANES_LOAD64": "1", "THJ_CTX_PER_GPU": "1"}),
                  ("8-byte load, one-lane inflater (THJ_INFLATE=one)", {"THJ_PLANES_LOAD64": "1", "THJ_INFLATE": "one"})):
    hs = [run(dict(env, THJ_SHARDS="7")) for _ in range(N)]
    bad = sum(1 for h in hs if h != ref)
    print("%-60s %d of %d runs differ; distinct outcomes %d" % (name, bad, N, len(set(hs))), flush=True)
