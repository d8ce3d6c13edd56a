"""Loading the fixtures under tests/golden/ through the host parsing rules (tophat_amd.samtext)."""
import os
import re

from tophat_amd.batch import build_seg_batch, build_span_batch
from tophat_amd.params import LIBRARY_TYPES, Params, READ_LEFT, READ_RIGHT
from tophat_amd.samtext import parse_header, parse_sam_hits, parse_spliced_sam_hits, read_fasta, read_fastq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_ALL = sorted(d for d in os.listdir(GOLD) if os.path.isfile(os.path.join(GOLD, d, "options.txt")))      # (tests/golden/ref_samtools holds vectors, not a case)
FUSION_SPAN_CASES = [d for d in _ALL if "fusion_span" in d]     # the whole --fusion-search path incl. long_spanning_reads
CASES = [d for d in _ALL if d not in FUSION_SPAN_CASES]


def read_fastq_quals(path):
    out = {}
    with open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            f.readline()
            f.readline()
            out[int(h[1:].split()[0])] = f.readline().strip()
    return out


def load(name):
    d = os.path.join(GOLD, name)
    opts = open(os.path.join(d, "options.txt")).read().split("\n")
    argv = opts[0].split()
    kv = dict(x.split("=") for x in opts[1].split())
    p = Params(segment_length=int(kv["segment_length"]))
    i = 0
    fusion = False
    ignore_names = []
    while i < len(argv):
        if argv[i] == "--fusion-search":
            fusion = True
            i += 1
            continue
        if argv[i] == "--inner-dist-mean":
            p.inner_dist_mean = int(argv[i + 1])
        elif argv[i] == "--inner-dist-std-dev":
            p.inner_dist_std_dev = int(argv[i + 1])
        elif argv[i] == "--library-type":
            p.library_type = LIBRARY_TYPES[argv[i + 1]]
        elif argv[i] == "--fusion-min-dist":
            p.fusion_min_dist = int(argv[i + 1])
        elif argv[i] == "--fusion-anchor-length":
            p.fusion_anchor_length = int(argv[i + 1])
        elif argv[i] == "--fusion-ignore-chromosomes":
            ignore_names = argv[i + 1].split(",")
        i += 2
    names, _ = parse_header(os.path.join(d, "hdr.sam"))
    fa_names, fa_seqs = read_fasta(os.path.join(d, "ref.fa"))
    seqs = [dict(zip(fa_names, fa_seqs)).get(n) for n in names]
    ref_ids = {n: i + 1 for i, n in enumerate(names)}
    paired = kv["paired"] == "1"
    nseg = len([f for f in os.listdir(d) if re.fullmatch(r"left_seg\d+\.sam", f)])
    sides = {}
    for sd in (("left", "right") if paired else ("left",)):
        sides[sd] = dict(
            reads=read_fastq(os.path.join(d, "%s.fq" % sd)), quals=read_fastq_quals(os.path.join(d, "%s.fq" % sd)),
            segs=[list(parse_sam_hits(os.path.join(d, "%s_seg%d.sam" % (sd, k + 1)), ref_ids, p.max_report_intron)) for k in range(nseg)],
            full=list(parse_sam_hits(os.path.join(d, "%s_map.sam" % sd), ref_ids, p.max_report_intron)),
            spliced=[list(parse_spliced_sam_hits(os.path.join(d, "%s_seg%d.to_spliced.sam" % (sd, k + 1)), ref_ids,
                                                 p.max_report_intron, p.min_anchor_len))
                     if os.path.exists(os.path.join(d, "%s_seg%d.to_spliced.sam" % (sd, k + 1))) else [] for k in range(nseg)])
    seg_batches, span_batches = [], {}
    for sd, side in (("left", READ_LEFT), ("right", READ_RIGHT)):
        if sd not in sides:
            continue
        other = "right" if sd == "left" else "left"
        if paired:
            b = build_seg_batch(sides[sd]["segs"], sides[sd]["reads"], sides[other]["full"], sides[other]["segs"][-1], include_top0=fusion)
        else:
            b = build_seg_batch(sides[sd]["segs"], sides[sd]["reads"], include_top0=fusion)
        seg_batches.append((side, b))
        span_batches[sd] = build_span_batch(sides[sd]["segs"], sides[sd]["reads"], sides[sd]["quals"], sides[sd]["spliced"])
    exp = {k: open(os.path.join(d, "expected.%s" % k)).read() for k in ("juncs", "insertions", "deletions")}
    exp_span = {}
    if fusion:
        exp["fusions"] = open(os.path.join(d, "expected.fusions")).read()
    for sd in sides:
        if not os.path.exists(os.path.join(d, "expected.span_%s.sam" % sd)):
            continue
        rows = [tuple(l.rstrip("\n").split("\t")) for l in open(os.path.join(d, "expected.span_%s.sam" % sd))]
        # (QNAME FLAG RNAME POS CIGAR tags...) -- drop MAPQ/SEQ/QUAL columns
        exp_span[sd] = [(r[0], int(r[1]), r[2], int(r[3]), r[5]) + r[8:] for r in rows]
    exp_span_full = {}
    if name in FUSION_SPAN_CASES:
        # (QNAME FLAG RNAME POS CIGAR SEQ QUAL tags...): fusion alignments are two records with the whole alignment in XF:Z
        for sd in sides:
            rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(d, "expected.span_%s.sam" % sd))]
            exp_span_full[sd] = [(r[0], r[1], r[2], r[3], r[5], r[6], r[7]) + tuple(r[8:]) for r in rows]
    elif fusion:
        span_batches = {}
    return dict(sides=sides, exp_span_full=exp_span_full, ref_ids=ref_ids, dir=d, p=p, names=names, seqs=seqs, seg_batches=seg_batches, span_batches=span_batches, exp=exp, exp_span=exp_span, fusion=fusion,
                fusion_ignore=[ref_ids[n] for n in ignore_names])


def events_text(ev, names, tmp_path):
    from tophat_amd.batch import write_segment_files
    f = {k: str(tmp_path / ("got." + k)) for k in ("juncs", "insertions", "deletions")}
    write_segment_files(ev, names, f["juncs"], f["insertions"], f["deletions"])
    return {k: open(v).read() for k, v in f.items()}


def fusions_text(fus, juncs, names, tmp_path):
    """filter + writer of segment_juncs.cpp:5096-5182 (oracle's restatement) on a reduced fusion set"""
    import orc
    f = orc.fusion_filter(fus, juncs)
    path = str(tmp_path / "got.fusions")
    orc.write_fusions(f, names, path)
    return open(path).read()
