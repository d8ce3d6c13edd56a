"""Loading the coverage-search fixtures under tests/golden_cov/ through the host parsing rules.  Test infrastructure."""
import os
import re

import numpy as np

from tophat_amd.batch import HIT_DTYPE, build_seg_batch, build_span_batch, hit_tuple_to_struct
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT
from tophat_amd.samtext import parse_header, parse_sam_hits, read_fasta, read_fastq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_cov")
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d)))


def load(name):
    d = os.path.join(GOLD, name)
    opts = open(os.path.join(d, "options.txt")).read().split("\n")
    argv = opts[0].split()
    kv = dict(x.split("=") for x in opts[1].split())
    p = Params(segment_length=int(kv["segment_length"]))
    cov = dict(min_intron=50, max_intron=20000)                 # common.cpp:112-113
    for i in range(0, len(argv), 2):
        if argv[i] == "--inner-dist-mean":
            p.inner_dist_mean = int(argv[i + 1])
        elif argv[i] == "--inner-dist-std-dev":
            p.inner_dist_std_dev = int(argv[i + 1])
        elif argv[i] == "--min-coverage-intron":
            cov["min_intron"] = int(argv[i + 1])
        elif argv[i] == "--max-coverage-intron":
            cov["max_intron"] = int(argv[i + 1])
    cov["min_cov_length"] = min(20, p.segment_length - 2)       # segment_juncs.cpp:62, :5350
    names, _ = parse_header(os.path.join(d, "hdr.sam"))
    fa_names, fa_seqs = read_fasta(os.path.join(d, "ref.fa"))
    seqs = [dict(zip(fa_names, fa_seqs)).get(n) for n in names]
    ref_ids = {n: i + 1 for i, n in enumerate(names)}
    paired = kv["paired"] == "1"
    nseg = len([f for f in os.listdir(d) if re.fullmatch(r"left_seg\d+\.sam", f)])
    sides = {}
    for sd in (("left", "right") if paired else ("left",)):
        sides[sd] = dict(reads=read_fastq(os.path.join(d, "%s.fq" % sd)),
                         segs=[list(parse_sam_hits(os.path.join(d, "%s_seg%d.sam" % (sd, k + 1)), ref_ids, p.max_report_intron)) for k in range(nseg)],
                         full=list(parse_sam_hits(os.path.join(d, "%s_map.sam" % sd), ref_ids, p.max_report_intron)))
    seg_batches, hits, ium = [], [], []
    for sd, side in (("left", READ_LEFT), ("right", READ_RIGHT)):
        if sd not in sides:
            continue
        other = "right" if sd == "left" else "left"
        # include_top0: reads mapped in their first segment only are neutral for the segment search, but their hits
        # are part of the coverage map (build_coverage_map walks every record of every segment map)
        b = build_seg_batch(sides[sd]["segs"], sides[sd]["reads"], sides[other]["full"], sides[other]["segs"][-1], include_top0=True) if paired \
            else build_seg_batch(sides[sd]["segs"], sides[sd]["reads"], include_top0=True)
        seg_batches.append((side, b))
        for recs in sides[sd]["segs"]:                          # all_segmap_fnames: left maps then right maps (:4929-4935)
            hits += [hit_tuple_to_struct(h) for h in recs]
        ium += [sides[sd]["reads"][rid] for rid in sorted(sides[sd]["reads"])]      # --ium-reads left.fq[,right.fq]
    span_batches, exp_span = {}, {}
    for sd in sides:                 # long_spanning_reads of the same reads (fixture: the scratch build on expected.juncs etc.)
        quals = {}
        with open(os.path.join(d, "%s.fq" % sd)) as f:
            while True:
                h = f.readline()
                if not h:
                    break
                f.readline(); f.readline()
                quals[int(h[1:].split()[0])] = f.readline().strip()
        span_batches[sd] = build_span_batch(sides[sd]["segs"], sides[sd]["reads"], quals)
        rows = [tuple(l.rstrip("\n").split("\t")) for l in open(os.path.join(d, "expected.span_%s.sam" % sd))]
        exp_span[sd] = [(r[0], int(r[1]), r[2], int(r[3]), r[5]) + r[8:] for r in rows]       # QNAME FLAG RNAME POS CIGAR tags...
    return dict(p=p, cov=cov, names=names, seqs=seqs, seg_batches=seg_batches, span_batches=span_batches, exp_span=exp_span, hits=np.array(hits, dtype=HIT_DTYPE), ium=ium,
                expected=open(os.path.join(d, "expected.juncs")).read(), expected_seg_only=open(os.path.join(d, "expected.seg_only.juncs")).read(),
                dir=d, paired=paired, nseg=nseg)


def juncs_text(juncs, names):
    """segment.juncs lines (segment_juncs.cpp:5035-5060) of a set of (ref_id, left, right, antisense)"""
    return "".join("%s\t%d\t%d\t%s\n" % (names[r - 1], l, rt, "-" if a else "+") for (r, l, rt, a) in sorted(juncs))


def edge_case(seed):
    """hits and unmapped reads made directly (no read simulation): islands at contig starts / ends, coverage runs around the
    length threshold, several contigs, short reads in the unmapped set, N runs -> (seqs, hits, ium reads, (min_cov_length,
    min_intron, max_intron))"""
    rng = np.random.default_rng(seed)
    n_contigs = int(rng.integers(1, 4))
    lens = [int(x) for x in rng.integers(200, 6000, size=n_contigs)]
    seqs = ["".join(rng.choice(list("ACGT"), size=n)) for n in lens]
    if seed % 3 == 0:                        # N runs
        s = list(seqs[0]); a = int(rng.integers(0, max(1, lens[0] - 60))); s[a:a + 50] = "N" * len(s[a:a + 50]); seqs[0] = "".join(s)
    min_cov = int(rng.choice([20, 18, 10, 8]))
    hits = []
    for k, n in enumerate(lens):             # islands: runs of hits of assorted lengths, some abutting, some at the contig ends
        for _ in range(int(rng.integers(2, 12))):
            ln = int(rng.choice([min_cov - 2, min_cov - 1, min_cov, min_cov + 1, 25, 40, 75]))
            left = int(rng.choice([0, 1, 2, max(0, n - ln), max(0, n - ln - 1), int(rng.integers(0, max(1, n - ln)))]))
            right = min(n, left + ln)
            if right > left:
                hits.append((k + 1, left, right, 0, 0, 0, min(255, right - left)))
    # unmapped reads: spliced reads across random pairs of positions (so that some donor/acceptor pairs are extendable)
    ium = []
    for _ in range(int(rng.integers(20, 200))):
        k = int(rng.integers(0, n_contigs)); s = seqs[k]; n = lens[k]
        a = int(rng.integers(0, max(1, n - 40))); b = int(rng.integers(a, n))
        la = int(rng.integers(5, 28))
        r = (s[max(0, a - la):a] + s[b:b + 40])[:int(rng.choice([9, 12, 25, 32, 50]))]
        if r:
            ium.append(r.replace("N", "A") if seed % 2 else r)
    args = (min_cov, int(rng.choice([1, 20, 50])), int(rng.choice([300, 2000, 20000])))
    return seqs, np.array(hits, dtype=HIT_DTYPE), ium, args


def butterfly_case(seed):
    """hits, unmapped reads and a genome made for the butterfly search: introns planted with GT..AG (forward) or CT..AC (reverse strand)
    between covered exon ends, unmapped reads spliced across them in either orientation (so that left and right keys meet), islands at
    contig starts, islands reaching the last 47 bases (dropped windows), islands 90 bases apart (abutting windows), sites within 32 bases
    of a contig end (no mer attached), N runs, a second copy of an exon end (several sites under one key)
    -> (seqs, hits, ium reads, (min_intron, max_intron))"""
    rng = np.random.default_rng(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    n_contigs = int(rng.integers(1, 4))
    seqs, hits, ium = [], [], []
    for k in range(n_contigs):
        n = int(rng.integers(400, 5000))
        s = list(rng.choice(list("ACGT"), size=n))
        n_introns = int(rng.integers(1, 6))
        cuts = sorted(int(x) for x in rng.integers(40, n - 40, size=2 * n_introns))
        introns = []
        for i in range(0, len(cuts) - 1, 2):
            d, a = cuts[i], cuts[i + 1]                 # intron = [d, a): s[d:d+2] = donor motif, s[a-2:a] = acceptor motif
            if a - d < 8:
                continue
            rev = bool(rng.integers(0, 2))
            s[d:d + 2] = "CT" if rev else "GT"
            s[a - 2:a] = "AC" if rev else "AG"
            introns.append((d, a, rev))
        if seed % 4 == 0 and n > 200:                   # an N run somewhere
            a0 = int(rng.integers(0, n - 60)); s[a0:a0 + int(rng.integers(1, 50))] = "N" * len(s[a0:a0 + int(rng.integers(1, 50))])
        if seed % 5 == 0 and introns:                   # a second copy of the bases before a donor, with its own GT: two sites, one key
            d = introns[0][0]
            if d >= 20 and n - 80 > d + 40:
                at = int(rng.integers(d + 40, n - 40)); s[at - 16:at + 2] = s[d - 16:d + 2]
        s = "".join(s)
        seqs.append(s)
        for d, a, rev in introns:                       # islands on both exon ends, of assorted extents and distances from the sites
            for end, lo, hi in ((d, max(0, d - int(rng.integers(10, 120))), d - int(rng.integers(0, 50))), (a, a + int(rng.integers(0, 50)), min(n, a + int(rng.integers(10, 120))))):
                if hi > lo >= 0:
                    hits.append((k + 1, lo, min(n, hi), 0, 0, 0, min(255, hi - lo)))
            for _ in range(int(rng.integers(1, 6))):    # reads spliced across the intron, either orientation, cut to assorted lengths
                la, lb = int(rng.integers(6, 30)), int(rng.integers(6, 30))
                r = s[max(0, d - la):d] + s[a:a + lb]
                if rng.integers(0, 2):
                    r = "".join(comp[c] for c in reversed(r))
                ium.append(r[:int(rng.choice([12, 20, 32, 40, 64]))])
        extra = [(0, int(rng.integers(1, 60))), (max(0, n - int(rng.integers(1, 60))), n), (max(0, n - 47 - int(rng.integers(0, 30))), max(1, n - 47 + int(rng.integers(-2, 3))))]
        if n > 400:
            e0 = int(rng.integers(50, n - 300)); extra += [(e0, e0 + 30), (e0 + 30 + 90, e0 + 150)]        # windows that abut exactly
        for lo, hi in extra:
            if rng.integers(0, 3) and hi > lo:
                hits.append((k + 1, lo, hi, 0, 0, 0, min(255, hi - lo)))
        for _ in range(int(rng.integers(0, 20))):       # reads joining random places
            a0, b0 = int(rng.integers(0, n - 20)), int(rng.integers(0, n - 20))
            ium.append((s[a0:a0 + 16] + s[b0:b0 + 16]).replace("N", "A"))
    args = (int(rng.choice([1, 6, 50])), int(rng.choice([300, 2000, 20000])))
    return seqs, np.array(hits, dtype=HIT_DTYPE), ium, args
