#!/usr/bin/env python3
"""Developer harness (build container only): run the survey-stage scratch build of the reference ($REFBIN, default
/tmp/refbuild/src -- see oracle/README.md for what that build is) through the whole --fusion-search path on a seeded
synthetic case and leave inputs + outputs in a directory:

    segment_juncs --fusion-search  ->  .juncs/.insertions/.deletions/.fusions
    juncs_db                       ->  junction database (with the fusion contigs)
    segments without a genome hit  ->  placed exhaustively on the junction-db contigs (bowtie's part), *.to_spliced.bam
    long_spanning_reads --fusion-search  ->  expected.span_<side>.sam (records as tuples) + .bam

    python tools/fusion_diff.py OUTDIR [seed] [n_reads] [fusion_reads]

The directory is what tests/golden/make_golden.py turns into a fixture.  Nothing here runs on the GPU box.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tophat_amd.bamio import read_bam, write_bam_from_sam  # noqa: E402
from tophat_amd.synth import make_case, write_case, md_nm  # noqa: E402

REFBIN = os.environ.get("REFBIN", "/tmp/refbuild/src")
_COMP = str.maketrans("ACGTN", "TGCAN")


def placements(target, query, max_mm=2):
    if len(target) < len(query):
        return []
    t = np.frombuffer(target.encode(), dtype=np.uint8)
    w = np.lib.stride_tricks.sliding_window_view(t, len(query))
    mm = (w != np.frombuffer(query.encode(), dtype=np.uint8)).sum(axis=1)
    return [(int(p), int(mm[p])) for p in np.nonzero(mm <= max_mm)[0]]


def read_fasta(fn):
    names, seqs = [], []
    for line in open(fn):
        if line.startswith(">"):
            names.append(line[1:].strip())
            seqs.append([])
        else:
            seqs[-1].append(line.strip())
    return names, ["".join(s) for s in seqs]


def run_case(d, seed=105, n_reads=120, fusion_reads=60, seg_len=25, read_len=100, extra_opts=(), gen_extra=None, all_segments=False):
    gen = dict(seed=seed, paired=True, read_len=read_len, seg_len=seg_len, n_reads=n_reads, fusion_reads=fusion_reads,
               contig_lens=(24000, 16000), genes_per_contig=4)
    gen.update(gen_extra or {})
    case = make_case(**gen)
    paths = write_case(case, d)
    opts = ["--inner-dist-mean", "50", "--inner-dist-std-dev", "20", "--fusion-search", "--fusion-min-dist", "1500"] + list(extra_opts)
    outs = [os.path.join(d, "expected.%s" % k) for k in ("juncs", "insertions", "deletions", "fusions")]
    seg = [os.path.join(REFBIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", str(seg_len),
           "--sam-header", paths["hdr"]] + opts + [paths["ref"]] + outs + [paths["left_fq"], paths["left_map"], ",".join(paths["left_segs"]),
                                                                            paths["right_fq"], paths["right_map"], ",".join(paths["right_segs"])]
    subprocess.run(seg, check=True, capture_output=True)
    dbfa = os.path.join(d, "juncs_db.fa")
    with open(dbfa, "w") as f:
        subprocess.run([os.path.join(REFBIN, "juncs_db"), "8", str(seg_len), outs[0], outs[1], outs[2], outs[3], paths["ref"]],
                       check=True, stdout=f, stderr=subprocess.DEVNULL)
    dbn, dbs = read_fasta(dbfa)
    dbhdr = "@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)) for n, s in zip(dbn, dbs))
    nseg = read_len // seg_len
    res = {}
    for sd in ("left", "right"):
        have = [set() for _ in range(nseg)]
        for k in range(nseg):
            for line in case.seg_sam[sd][k]:
                have[k].add(int(line.split("|", 1)[0]))
        sps = []
        for k in range(nseg):
            lines = []
            for rid in sorted(case.reads[sd]):
                if rid in have[k] and not all_segments:
                    continue
                read = case.reads[sd][rid]
                s0 = k * seg_len
                s1 = read_len if k == nseg - 1 else (k + 1) * seg_len
                piece = read[s0:s1]
                qn = "%d|%d:%d:%d" % (rid, s0, k, nseg)
                hits = []
                for anti in (False, True):
                    q = piece.translate(_COMP)[::-1] if anti else piece
                    for ci, t in enumerate(dbs):
                        for pos, mm in placements(t, q):
                            hits.append((ci, pos, anti, q))
                if len(hits) > 40:
                    continue
                for ci, pos, anti, q in hits:
                    nm_, md = md_nm(dbs[ci][pos:pos + len(q)], q)
                    lines.append("%s\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                        qn, 16 if anti else 0, dbn[ci], pos + 1, len(q), q, "I" * len(q), nm_, md))
            p = os.path.join(d, "%s_seg%d.to_spliced.sam" % (sd, k + 1))
            with open(p, "w") as f:
                f.write(dbhdr)
                f.writelines(lines)
            write_bam_from_sam(p, p[:-4] + ".bam")
            sps.append(p[:-4] + ".bam")
        bam = os.path.join(d, "expected.span_%s.bam" % sd)
        lsr = [os.path.join(REFBIN, "long_spanning_reads"), "--segment-length", str(seg_len), "--sam-header", paths["hdr"]] + opts + [
            paths["ref"], paths["%s_fq" % sd], outs[0], outs[1], outs[2], outs[3], bam, ",".join(paths["%s_segs" % sd]), ",".join(sps)]
        r = subprocess.run(lsr, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("long_spanning_reads: " + r.stderr[-2000:])
        _, recs = read_bam(bam)
        with open(os.path.join(d, "expected.span_%s.sam" % sd), "w") as f:
            for rec in recs:
                f.write("\t".join(str(x) for x in rec) + "\n")
        if os.path.exists(bam + ".index"):
            os.remove(bam + ".index")
        res[sd] = recs
    with open(os.path.join(d, "options.txt"), "w") as f:
        f.write(" ".join(opts) + "\n")
        f.write("segment_length=%d paired=1\n" % seg_len)
    return case, paths, res


if __name__ == "__main__":
    d = sys.argv[1]
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 105
    n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    fr = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    case, paths, res = run_case(d, seed, n_reads, fr)
    for sd, recs in res.items():
        nf = sum(1 for r in recs if any("XF:Z" in str(x) for x in r))
        print(sd, len(recs), "records,", nf, "with XF")
    print(open(os.path.join(d, "expected.fusions")).read()[:600])
