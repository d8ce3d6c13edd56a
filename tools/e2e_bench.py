#!/usr/bin/env python3
"""End-to-end timing of the drop-in executables (GPU box): write the files tophat.py would hand over for a synthetic
paired-end run of BASELINE configs[1]'s shape (tools/bin/thj_gen: FASTA, SAM header, reads as unaligned BAM, whole-read and
per-segment maps as id-sorted BAM, every BAM with its .index), run tophat_amd/bin/segment_juncs and long_spanning_reads
(left, then right) on them and report wall-clock pairs/s for both stages together -- the metric of SURVEY.md section 8d.

    python tools/e2e_bench.py [--pairs N] [--read-len R] [--genome-len L] [--introns K] [--keep DIR] [--env K=V ...]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tophat_amd", "bin")
GEN = os.path.join(ROOT, "tools", "bin", "thj_gen")


GRCH38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
               135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
               46709983, 50818468, 156040895, 57227415, 16569]


def mix_gen_args(multihit_frac, max_copies, indel_frac):
    """thj_gen options of SURVEY 8(d)'s mix"""
    out = []
    if multihit_frac > 0:
        out += ["--multihit-frac", repr(float(multihit_frac)), "--max-copies", str(int(max_copies))]
    if indel_frac > 0:
        out += ["--indel-frac", repr(float(indel_frac))]
    return out


def _prefix(env, stage):
    """THJ_EXEC_PREFIX='rocprofv3 --kernel-trace --stats -d /tmp/p_{stage} -o res --' runs each executable under a profiler"""
    pre = env.get("THJ_EXEC_PREFIX", "")
    return pre.replace("{stage}", stage).split() if pre else []


def _run(cmd, env):
    """a stage with a deadline: a process that hangs (a collective that never completes, say) is an error here, not a hung bench"""
    try:
        return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=float(os.environ.get("THJ_STAGE_TIMEOUT", "600")))
    except subprocess.TimeoutExpired as e:
        raise RuntimeError("%s did not finish within %s s" % (os.path.basename(cmd[0]), e.timeout))


def run_e2e(pairs, read_len=100, genome_len=64444167, introns=20000, workdir=None, env_extra=None, keep=False, coverage_search=False, gen_args=(), bindir=None, fusion_search=False):
    """gen_args: further thj_gen options (SURVEY 8d's mix: --multihit-frac F --max-copies C --indel-frac F)
    bindir: where segment_juncs / long_spanning_reads are taken from (default: the product executables; tools/bin/cpuport =
    the same host sources over the CPU oracle, bench.py's files-to-files CPU figure -- no thj_junctions there, that leg is skipped)
    fusion_search: --fusion-search to both executables (long_spanning_reads then reads the .fusions list segment_juncs wrote)"""
    BIN = bindir or globals()["BIN"]
    nseg = max(1, read_len // 25)
    d = workdir or tempfile.mkdtemp(prefix="thj_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    env = dict(os.environ, THJ_TIMING="1", **(env_extra or {}))
    res = {"pairs": pairs, "read_len": read_len, "genome_len": genome_len, "dir": d}
    t = time.time()
    if not os.path.exists(os.path.join(d, "ref.fa")):
        subprocess.check_call([GEN, "--out", d, "--pairs", str(pairs), "--read-len", str(read_len), "--genome-len", str(genome_len),
                           "--introns", str(introns)] + list(gen_args), stdout=subprocess.DEVNULL)
    res["gen_seconds"] = round(time.time() - t, 2)
    res["input_bytes"] = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".bam"))

    def f(name):
        return os.path.join(d, name)
    segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k + 1)) for k in range(nseg)) for sd in ("left", "right")}
    out = {k: f("out." + k) for k in ("juncs", "insertions", "deletions", "fusions")}
    mode = ["--ium-reads", f("left_reads.bam") + "," + f("right_reads.bam")] if coverage_search else ["--no-coverage-search"]
    fus = ["--fusion-search"] if fusion_search else []
    cmd = _prefix(env, "segment_juncs") + [os.path.join(BIN, "segment_juncs")] + mode + fus + ["--no-microexon-search", "--segment-length", "25",
           "--sam-header", f("hdr.sam"), "--inner-dist-mean", "50", "--inner-dist-std-dev", "20",
           f("ref.fa"), out["juncs"], out["insertions"], out["deletions"], out["fusions"],
           f("left_reads.bam"), f("left_map.bam"), segs["left"], f("right_reads.bam"), f("right_map.bam"), segs["right"]]
    import resource

    def child_cpu():
        ru = resource.getrusage(resource.RUSAGE_CHILDREN)
        return ru.ru_utime + ru.ru_stime
    t = time.time(); c0 = child_cpu()
    r = _run(cmd, env)
    dt = time.time() - t
    if r.returncode != 0:
        raise RuntimeError("segment_juncs failed:\n" + r.stderr[-3000:])
    res["segment_juncs_s"] = round(dt, 3)
    for l in r.stderr.splitlines():                         # time before main() and after the report: loader, HIP start-up of the static objects, process teardown
        if "unix time at start / report" in l:
            a, b = map(float, l.split()[-2:])
            res["segment_juncs_before_main_after_report_s"] = [round(a - t, 3), round(t + dt - b, 3)]
    # CPU seconds (user + system) of the process and what it waited for; with the output hand-off the working child outlives its parent
    # and is not counted -- run with THJ_NO_HANDOFF=1 for this figure
    res["segment_juncs_cpu_s"] = round(child_cpu() - c0, 3)
    res["junctions"] = sum(1 for _ in open(out["juncs"]))
    res["segment_juncs_log_tail"] = r.stderr.strip().splitlines()[-12:]
    res["segment_juncs_log_all"] = [l for l in r.stderr.splitlines() if "declined" in l]
    tot = dt
    for sd in ("left", "right"):
        cmd = _prefix(env, "lsr_" + sd) + [os.path.join(BIN, "long_spanning_reads")] + fus + ["--segment-length", "25", "--sam-header", f("hdr.sam"), f("ref.fa"),
               f("%s_reads.bam" % sd), out["juncs"], out["insertions"], out["deletions"], out["fusions"] if fusion_search else "/dev/null", f("span_%s.bam" % sd), segs[sd]]
        t = time.time(); c0 = child_cpu()
        r = _run(cmd, env)
        dt = time.time() - t
        if r.returncode != 0:
            raise RuntimeError("long_spanning_reads failed:\n" + r.stderr[-3000:])
        res["long_spanning_reads_%s_s" % sd] = round(dt, 3)
        res["long_spanning_reads_%s_cpu_s" % sd] = round(child_cpu() - c0, 3)
        for l in r.stderr.splitlines():                     # time before main() and after the report: loader, process teardown
            if "unix time at start / report" in l:
                a, b = map(float, l.split()[-2:])
                res["long_spanning_reads_%s_before_main_after_report_s" % sd] = [round(a - t, 3), round(t + dt - b, 3)]
        res["long_spanning_reads_%s_log_tail" % sd] = [l for l in r.stderr.strip().splitlines() if not l.startswith("[trace]")][-8:]
        res["long_spanning_reads_%s_log_all" % sd] = [l for l in r.stderr.splitlines() if "declined" in l or l.startswith("[huge timers]")][-40:]
        if env.get("THJ_TRACE"):                     # the per-shard timeline for tools/lsr_trace.py
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "lsr_%s.trace" % sd), "w").write(r.stderr)
        res["span_%s_bytes" % sd] = os.path.getsize(f("span_%s.bam" % sd))
        tot += dt
    res["both_stages_s"] = round(tot, 3)
    import re as _re
    res["host_ingest_fallback_shards"] = sum(int(m.group(1)) for st in ("segment_juncs", "long_spanning_reads_left", "long_spanning_reads_right")
                                             for l in res.get(st + "_log_all", []) for m in [_re.search(r"device-side ingest declined them: (\d+)", l)] if m)
    oms = [res.get(k + "_before_main_after_report_s") for k in ("segment_juncs", "long_spanning_reads_left", "long_spanning_reads_right")]
    if all(oms):
        res["outside_main_s"] = round(sum(a + b for a, b in oms), 3)          # of both_stages_s: what the three processes spend before main() and after their report
    res["pairs_per_s_both_stages"] = round(pairs / tot)
    # the junction consensus (tophat_reports' part of the metric's "junctions.bed"): timed on its own, not part of both_stages_s
    if not os.path.exists(os.path.join(BIN, "thj_junctions")):
        if not keep and workdir is None:
            shutil.rmtree(d, ignore_errors=True)
        return res
    t = time.time()
    r = subprocess.run([os.path.join(BIN, "thj_junctions"), "--sam-header", f("hdr.sam"), f("ref.fa"), f("junctions.bed"),
                        f("span_left.bam") + "," + f("span_right.bam")], capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError("thj_junctions failed:\n" + r.stderr[-3000:])
    res["junctions_bed_s"] = round(time.time() - t, 3)
    res["junctions_bed_lines"] = sum(1 for _ in open(f("junctions.bed"))) - 1
    if not keep and workdir is None:
        shutil.rmtree(d, ignore_errors=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1000000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--genome-len", type=int, default=64444167)
    ap.add_argument("--introns", type=int, default=20000)
    ap.add_argument("--grch38", action="store_true", help="configs[2]'s genome: 25 contigs with the GRCh38 primary-assembly lengths (3.09 Gb; pass --introns 300000)")
    ap.add_argument("--coverage-search", action="store_true")
    ap.add_argument("--fusion-search", action="store_true", help="--fusion-search to both executables (configs[3]'s mode; the generator plants no fusions)")
    ap.add_argument("--keep", default=None)
    ap.add_argument("--multihit-frac", type=float, default=0.05, help="SURVEY 8(d)'s mix (the default, as bench.py's): share of the pairs from the repeat family")
    ap.add_argument("--max-copies", type=int, default=41)
    ap.add_argument("--indel-frac", type=float, default=0.03)
    ap.add_argument("--plain", action="store_true", help="configs[1] without the mix (rounds 1-3)")
    ap.add_argument("--intron-max", type=int, default=0, help="longest planted intron (configs[4]: 499999); 0: the generator's default")
    ap.add_argument("--env", nargs="*", default=[])
    a = ap.parse_args()
    ga = [] if a.plain else mix_gen_args(a.multihit_frac, a.max_copies, a.indel_frac)
    if a.intron_max > 0:
        ga += ["--intron-max", str(a.intron_max)]
    if a.grch38:
        ga += ["--contigs", ",".join(str(x) for x in GRCH38_LENS)]
    res = run_e2e(a.pairs, a.read_len, sum(GRCH38_LENS) if a.grch38 else a.genome_len, a.introns, workdir=a.keep, env_extra=dict(x.split("=", 1) for x in a.env), keep=bool(a.keep),
                  coverage_search=a.coverage_search, fusion_search=a.fusion_search, gen_args=ga)
    print(json.dumps(res, indent=1))
