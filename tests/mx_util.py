"""Seeded cases for the microexon search (test infrastructure): a genome with planted microexons (exon A, an intron, a microexon of
8..24 bases, another intron, exon B), reads that start in exon A's last bases, cross the microexon and run into exon B -- their first
segment does not map, the others do (exactly or with mismatches; a share of them on the reverse strand, with several second-segment hits,
near contig ends, with N in the read) -- plus decoy reads with an empty first segment and nothing to find, and reads whose other segments
are incomplete (no window at all)."""
import random

import numpy as np

from tophat_amd.batch import build_seg_batch

_COMP = str.maketrans("ACGTN", "TGCAN")


def rc(s):
    return s.translate(_COMP)[::-1]


def make_case(seed, n_genes=12, n_reads=160, seg_len=25, nseg=4, n_contigs=2, with_n=True):
    rng = random.Random(seed)

    def rnd(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    def intron(n, motif):
        body = rnd(n - 4)
        return motif[0] + body + motif[1]
    seqs, genes = [], []
    for c in range(n_contigs):
        parts = [rnd(rng.randrange(10, 400))]
        pos = len(parts[0])
        for _ in range(n_genes):
            ex_a, mx, ex_b = rnd(rng.randrange(60, 200)), rnd(rng.randrange(8, 25)), rnd(seg_len * nseg + rng.randrange(10, 100))
            motif = rng.choice((("GT", "AG"), ("GT", "AG"), ("CT", "AC")))            # the second pair: the same introns on the other strand
            i1, i2 = intron(rng.randrange(55, 900), motif), intron(rng.randrange(55, 1500), motif)
            a0 = pos
            m0 = a0 + len(ex_a) + len(i1)
            b0 = m0 + len(mx) + len(i2)
            genes.append(dict(contig=c, ex_a=(a0, a0 + len(ex_a)), mx=(m0, m0 + len(mx)), ex_b=(b0, b0 + len(ex_b))))
            piece = ex_a + i1 + mx + i2 + ex_b + rnd(rng.randrange(50, 3000))
            parts.append(piece)
            pos += len(piece)
        tail = rnd(rng.randrange(5, 60))
        parts.append(tail)
        s = "".join(parts)
        if with_n and c == 0:
            k = rng.randrange(len(s) - 40)
            s = s[:k] + "N" * 7 + s[k + 7:]
        seqs.append(s)
    reads, seg_recs = {}, [[] for _ in range(nseg)]
    rl = seg_len * nseg
    for rid in range(1, n_reads + 1):
        gene = rng.choice(genes)
        s = seqs[gene["contig"]]
        kind = rng.random()
        mx_len = gene["mx"][1] - gene["mx"][0]
        k_a = seg_len - mx_len - rng.randrange(0, 4) if kind < 0.7 else rng.randrange(0, seg_len)      # bases of exon A in the read
        k_a = max(1, min(k_a, seg_len - 1))
        k_m = min(mx_len, seg_len - k_a)
        body_b = rl - k_a - k_m
        read = s[gene["ex_a"][1] - k_a:gene["ex_a"][1]] + s[gene["mx"][0]:gene["mx"][0] + k_m] + s[gene["ex_b"][0]:gene["ex_b"][0] + body_b]
        if len(read) != rl or "N" in read and rng.random() < 0.5:
            continue
        anti = rng.random() < 0.4
        # the read as sequenced; segment s of an antisense read maps to the mirrored place
        seq = rc(read) if anti else read
        if rng.random() < 0.1:
            k = rng.randrange(rl)
            seq = seq[:k] + "N" + seq[k + 1:]
        reads[rid] = seq
        b_off = k_a + k_m                                              # where exon B's bases start in the read (forward orientation)
        drop = rng.random()
        for sg in range(nseg):
            f0, f1 = (sg * seg_len, (sg + 1) * seg_len if sg < nseg - 1 else rl)       # the segment of the sequenced read
            g0, g1 = (rl - f1, rl - f0) if anti else (f0, f1)          # its place in the forward read
            if g0 < b_off:
                continue                                               # holds exon A / microexon bases: unmapped (the first segment in transcript order)
            if drop < 0.12 and sg == (1 if not anti else nseg - 2):
                continue                                               # another segment missing as well: no microexon window for this read
            left = gene["ex_b"][0] + g0 - b_off
            mm = rng.choice((0, 0, 0, 1, 2))
            seg_recs[sg].append((rid, gene["contig"] + 1, left, left + (g1 - g0), anti, sg == nseg - 1, mm, mm, g1 - g0))
            if rng.random() < 0.08:                                    # a second placement of the segment somewhere else
                l2 = rng.randrange(0, len(s) - seg_len)
                seg_recs[sg].append((rid, gene["contig"] + 1, l2, l2 + (g1 - g0), rng.random() < 0.5, sg == nseg - 1, 1, 1, g1 - g0))
    for v in seg_recs:
        v.sort(key=lambda r: r[0])
    # only reads with hits in the LAST segment are visited (process_next_hit_group reads the last file)
    visited = {r[0] for r in seg_recs[-1]}
    reads = {k: v for k, v in reads.items() if k in visited}
    seg_recs = [[r for r in v if r[0] in visited] for v in seg_recs]
    return seqs, genes, reads, seg_recs
