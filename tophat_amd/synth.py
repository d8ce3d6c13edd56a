"""Seeded synthetic workloads for the junction-discovery hot path.

Two generators:

* `make_case` (pure Python, small): genome with planted GT-AG / GC-AG / AT-AC
  introns on both strands and N runs, single/paired reads that span 0-2
  junctions or carry a small indel, and *directly synthesised* id-sorted segment
  maps (bowtie is not in this image): a segment is "mapped" where the truth puts
  it when it has <= `segment_mismatches` mismatches against the contiguous
  genome there, including segments that overhang a junction by a few bases.
  Used by the tests, the golden fixtures and the differential checks.

* `make_device_workload` (torch, any device): the BASELINE.json config shapes at
  scale (millions of reads) built with tensor ops so bench.py can synthesise the
  batch directly in HBM.  See bench.py.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from .batch import HitRec
from .samtext import md_nm

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s: str) -> str:
    return "".join(COMP.get(c, "N") for c in reversed(s))


@dataclass
class Gene:
    ref: int                 # 0-based contig index
    exons: List[Tuple[int, int]]   # half-open genome intervals, increasing
    strand: str              # '+' or '-': strand of the transcript (motif orientation)


@dataclass
class SynthCase:
    names: List[str]
    seqs: List[str]
    genes: List[Gene]
    read_len: int
    seg_len: int
    reads: Dict[str, Dict[int, str]] = field(default_factory=dict)          # side -> id -> seq
    quals: Dict[str, Dict[int, str]] = field(default_factory=dict)
    seg_sam: Dict[str, List[List[str]]] = field(default_factory=dict)        # side -> [seg] -> SAM lines
    seg_recs: Dict[str, List[List[HitRec]]] = field(default_factory=dict)    # side -> [seg] -> HitRec
    full_sam: Dict[str, List[str]] = field(default_factory=dict)             # side -> full-read map SAM lines
    full_recs: Dict[str, List[HitRec]] = field(default_factory=dict)
    truth_juncs: set = field(default_factory=set)                            # (ref_id, left, right, strand)
    spliced_sam: Dict[str, List[List[str]]] = field(default_factory=dict)   # side -> [seg] -> junction-db SAM lines
    juncdb: Dict[str, int] = field(default_factory=dict)                    # junction-db contig name -> length

    @property
    def nseg(self) -> int:
        return max(1, self.read_len // self.seg_len)


MOTIFS = [("GT", "AG")] * 18 + [("GC", "AG")] * 1 + [("AT", "AC")] * 1


def _plant(seq: List[str], start: int, end: int, strand: str, rng: random.Random) -> None:
    """Make [start,end) look like an intron of `strand`."""
    d, a = rng.choice(MOTIFS)
    if strand == "+":
        seq[start:start + 2] = list(d)
        seq[end - 2:end] = list(a)
    else:
        seq[start:start + 2] = list(revcomp(a))
        seq[end - 2:end] = list(revcomp(d))


def make_genome(rng: random.Random, contig_lens: Sequence[int], n_genes: int,
                intron_range=(60, 3000), exon_range=(30, 400), n_runs: int = 2):
    names = ["chr%s" % (i + 1) for i in range(len(contig_lens))]
    seqs = [[rng.choice("ACGT") for _ in range(n)] for n in contig_lens]
    genes: List[Gene] = []
    for ci, n in enumerate(contig_lens):
        pos = 300
        while pos < n - 8000 and len([g for g in genes if g.ref == ci]) < n_genes:
            strand = rng.choice("+-")
            nex = rng.randint(2, 5)
            exons = []
            p = pos
            for e in range(nex):
                el = rng.randint(*exon_range) if 0 < e < nex - 1 else rng.randint(150, 500)
                exons.append((p, p + el))
                p += el
                if e < nex - 1:
                    il = int(rng.uniform(intron_range[0] ** 0.5, intron_range[1] ** 0.5) ** 2)
                    _plant(seqs[ci], p, p + il, strand, rng)
                    p += il
            genes.append(Gene(ci, exons, strand))
            pos = p + rng.randint(200, 1500)
        for _ in range(n_runs):
            s = rng.randint(0, max(0, n - 200))
            ln = rng.randint(5, 60)
            seqs[ci][s:s + ln] = ["N"] * ln
    return names, ["".join(s) for s in seqs], genes


def _mutate(rng: random.Random, s: str, rate: float) -> str:
    out = list(s)
    for i in range(len(out)):
        if rng.random() < rate:
            out[i] = rng.choice([c for c in "ACGT" if c != out[i]])
    return "".join(out)


def _tx_to_genome(exons, t):
    """transcript offset -> genome coordinate"""
    for (a, b) in exons:
        if t < b - a:
            return a + t
        t -= b - a
    return exons[-1][1] + t


def _place_piece(gseq: str, exons, t0: int, ln: int, piece: str, max_mm: int, overhang: int):
    """Try to place transcript interval [t0,t0+ln) contiguously on the genome
    the way a short-read mapper would: wholly inside one exon, or overhanging a
    junction by <= `overhang` bases at either end.  Returns (pos, nm) or None."""
    cum = 0
    for (a, b) in exons:
        el = b - a
        if t0 >= cum and t0 + ln <= cum + el:
            pos = a + (t0 - cum)
            break
        # overhang at the right end of this exon
        if t0 >= cum and t0 < cum + el and t0 + ln - (cum + el) <= overhang:
            pos = a + (t0 - cum)
            break
        # overhang at the left end of this exon
        if t0 < cum and cum - t0 <= overhang and t0 + ln <= cum + el and t0 + ln > cum:
            pos = a - (cum - t0)
            break
        cum += el
    else:
        return None
    if pos < 0 or pos + ln > len(gseq):
        return None
    ref = gseq[pos:pos + ln]
    nm = sum(1 for r, q in zip(ref, piece) if r != q)
    if nm > max_mm:
        return None
    return pos, nm


def _place_spliced(gseq: str, exons, t0: int, ln: int, piece: str):
    """Transcript interval [t0,t0+ln) that crosses exactly one exon-exon junction ->
    (pos, a_len, gap, b_len, nm, ref_bases) or None."""
    cum = 0
    for i, (a, b) in enumerate(exons):
        el = b - a
        if cum <= t0 < cum + el:
            a_len = cum + el - t0
            if a_len >= ln or i + 1 >= len(exons):
                return None
            b_len = ln - a_len
            na, nb = exons[i + 1]
            if b_len > nb - na:
                return None
            pos = a + (t0 - cum)
            ref = gseq[pos:pos + a_len] + gseq[na:na + b_len]
            nm = sum(1 for r, q in zip(ref, piece) if r != q)
            if nm > 2:
                return None
            return pos, a_len, na - b, b_len, nm, ref
        cum += el
    return None


def make_case(seed: int = 1, contig_lens: Sequence[int] = (60000,), n_reads: int = 300,
              read_len: int = 100, seg_len: int = 25, paired: bool = False,
              err: float = 0.01, indel_frac: float = 0.08, n_frac: float = 0.02,
              genes_per_contig: int = 12, intron_range=(60, 3000), inner_mean: int = 50,
              inner_sd: int = 20, repeat_frac: float = 0.0, drop_seg_frac: float = 0.03,
              overhang: int = 3, spliced_seg_frac: float = 0.0, boundary_bias: float = 0.0, juncdb: bool = False,
              fusion_reads: int = 0, exon_range=(30, 400)) -> SynthCase:
    rng = random.Random(seed)
    names, seqs, genes = make_genome(rng, contig_lens, genes_per_contig, intron_range, exon_range)
    # optional planted repeats -> multihits
    repeats: List[Tuple[int, int, int]] = []
    if repeat_frac > 0:
        seqs_l = [list(s) for s in seqs]
        for _ in range(max(1, int(len(genes) * repeat_frac))):
            g = rng.choice(genes)
            a, b = g.exons[0]
            unit = seqs[g.ref][a + 20:a + 20 + 60]
            for _k in range(rng.randint(1, 3)):
                ci = rng.randrange(len(seqs))
                p = rng.randint(0, len(seqs[ci]) - 100)
                if any(ex[0] - 100 < p < ex[1] + 100 for gg in genes if gg.ref == ci for ex in gg.exons):
                    continue
                seqs_l[ci][p:p + 60] = list(unit)
                repeats.append((g.ref, a + 20, 60))
                repeats.append((ci, p, 60))
        seqs = ["".join(s) for s in seqs_l]
    case = SynthCase(names, seqs, genes, read_len, seg_len)
    nseg = case.nseg
    for g in genes:
        for (e0, e1) in zip(g.exons[:-1], g.exons[1:]):
            case.truth_juncs.add((g.ref + 1, e0[1] - 1, e1[0], g.strand))

    sides = ["left", "right"] if paired else ["left"]
    for sd in sides:
        case.reads[sd] = {}
        case.quals[sd] = {}
        case.seg_sam[sd] = [[] for _ in range(nseg)]
        case.seg_recs[sd] = [[] for _ in range(nseg)]
        case.full_sam[sd] = []
        case.full_recs[sd] = []
        case.spliced_sam[sd] = [[] for _ in range(nseg)]

    def emit(sd: str, rid: int, frag_exons, t0: int, anti: bool, gref: int, with_indel: bool, gene_strand: str = "+"):
        """fragment = transcript interval [t0, t0+read_len) of the exon chain."""
        gseq = seqs[gref]
        tx = "".join(gseq[a:b] for (a, b) in frag_exons)
        F = tx[t0:t0 + read_len]
        if len(F) < read_len:
            return False
        indel = None
        if with_indel:
            kb = rng.randint(1, max(1, nseg - 1))
            p = max(3, min(read_len - 5, kb * seg_len + rng.randint(-3, 3)))
            if rng.random() < 0.5:
                k = rng.randint(1, 3)      # insertion in the read
                F = (F[:p] + "".join(rng.choice("ACGT") for _ in range(k)) + F[p:])[:read_len]
                indel = ("I", p, k)
            else:
                k = rng.randint(1, 3)      # deletion from the read
                ext = tx[t0 + read_len:t0 + read_len + k]
                if len(ext) == k:
                    F = F[:p] + F[p + k:] + ext
                    indel = ("D", p, k)
        F = _mutate(rng, F, err)
        if rng.random() < n_frac:
            i = rng.randrange(read_len)
            F = F[:i] + "N" + F[i + 1:]
        seq = revcomp(F) if anti else F
        case.reads[sd][rid] = seq
        case.quals[sd][rid] = "I" * read_len
        # segment maps
        for k in range(nseg):
            s0 = k * seg_len
            s1 = read_len if k == nseg - 1 else (k + 1) * seg_len
            if rng.random() < drop_seg_frac:
                continue
            if anti:
                f0, f1 = read_len - s1, read_len - s0
            else:
                f0, f1 = s0, s1
            piece = F[f0:f1]
            # transcript offset of this piece, accounting for the indel; a segment
            # that overlaps the indel by a couple of bases can still map (with
            # mismatches) anchored on either side of it
            offs = [f0]
            if indel:
                kind, p, kk = indel
                sh = -kk if kind == "I" else kk
                if f0 >= p + (kk if kind == "I" else 0):
                    offs = [f0 + sh]
                elif f1 > p:
                    offs = [f0, f0 + sh]
            placed = None
            for off in offs:
                cand = _place_piece(gseq, frag_exons, t0 + off, f1 - f0, piece, 2, overhang)
                if cand is not None and (placed is None or cand[1] < placed[1]):
                    placed = cand
            if placed is None:
                # a segment that straddles a junction: optionally emit the spliced alignment the
                # junction-db mapping (juncs_db + bowtie + SplicedBAMHitFactory) would produce
                if spliced_seg_frac > 0 and not indel and rng.random() < spliced_seg_frac:
                    sp = _place_spliced(gseq, frag_exons, t0 + f0, f1 - f0, piece)
                    if sp is not None:
                        spos, a_len, gap, b_len, snm, sref = sp
                        nm_, md = md_nm(sref, piece)
                        flag = 16 if anti else 0
                        qn = "%d|%d:%d:%d" % (rid, s0, k, nseg)
                        xs = "-" if gene_strand == "-" else "+"
                        if juncdb:
                            # the record bowtie would write against the junction database built by juncs_db
                            # (juncs_db.cpp:109-150): contig name|left_start|l-r|right_end|GTAG|fwd/rev
                            jl, jr = spos + a_len - 1, spos + a_len + gap
                            half = seg_len
                            ls = max(0, jl - half + 1)
                            re_ = min(jr + half, len(gseq))
                            dbname = "%s|%d|%d-%d|%d|GTAG|%s" % (names[gref], ls, jl, jr, re_, "rev" if xs == "-" else "fwd")
                            case.juncdb[dbname] = (ls + half - ls) + (re_ - jr)
                            dbseq = gseq[ls:ls + half] + gseq[jr:re_]
                            dpos = spos - ls
                            nm2, md2 = md_nm(dbseq[dpos:dpos + (f1 - f0)], piece)
                            case.spliced_sam[sd][k].append("%s\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                                qn, flag, dbname, dpos + 1, f1 - f0, piece, "I" * (f1 - f0), nm2, md2))
                            continue
                        case.seg_sam[sd][k].append("%s\t%d\t%s\t%d\t255\t%dM%dN%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\tXS:A:%s\n" % (
                            qn, flag, names[gref], spos + 1, a_len, gap, b_len, piece, "I" * (f1 - f0), nm_, md, xs))
                        case.seg_recs[sd][k].append((rid, gref + 1, spos, spos + a_len + gap + b_len, anti, k == nseg - 1, nm_, nm_,
                                                     f1 - f0, [(1, a_len), (11, gap), (1, b_len)], xs == "-"))
                continue
            pos, nm = placed
            places = [(gref, pos, nm)]
            for (rc_, rp, rl_) in repeats:   # multihits through planted repeats
                for (oc, op, ol) in repeats:
                    if (oc, op) != (rc_, rp) and rc_ == gref and rp <= pos and pos + (f1 - f0) <= rp + rl_ and ol == rl_:
                        q = op + (pos - rp)
                        ref2 = seqs[oc][q:q + (f1 - f0)]
                        nm2 = sum(1 for r_, q_ in zip(ref2, piece) if r_ != q_)
                        if len(ref2) == f1 - f0 and nm2 <= 2 and (oc, q, nm2) not in places:
                            places.append((oc, q, nm2))
            for (pc, pp, pnm) in places:
                refs = seqs[pc][pp:pp + (f1 - f0)]
                nm_, md = md_nm(refs, piece)
                flag = 16 if anti else 0
                qn = "%d|%d:%d:%d" % (rid, s0, k, nseg)
                case.seg_sam[sd][k].append("%s\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                    qn, flag, names[pc], pp + 1, f1 - f0, piece, "I" * (f1 - f0), nm_, md))
                case.seg_recs[sd][k].append((rid, pc + 1, pp, pp + (f1 - f0), anti, k == nseg - 1, nm_, nm_, f1 - f0, [(1, f1 - f0)], False))
        # full-read map: only when the fragment is unspliced and indel-free
        if not indel:
            placed = _place_piece(gseq, frag_exons, t0, read_len, F, 2, 0)
            if placed is not None:
                pos, nm = placed
                nm_, md = md_nm(gseq[pos:pos + read_len], F)
                case.full_sam[sd].append("%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                    rid, 16 if anti else 0, names[gref], pos + 1, read_len, F, "I" * read_len, nm_, md))
                case.full_recs[sd].append((rid, gref + 1, pos, pos + read_len, anti, True, nm_, nm_, read_len, [(1, read_len)], False))
        return True

    rid = 0
    attempts = 0
    while rid < n_reads and attempts < n_reads * 20:
        attempts += 1
        g = rng.choice(genes)
        tlen = sum(b - a for a, b in g.exons)
        frag_len = 2 * read_len + max(0, int(rng.gauss(inner_mean, inner_sd))) if paired else read_len
        if tlen < frag_len + 2:
            continue
        # bias starts so that many reads span a junction
        if rng.random() < 0.6:
            j = rng.randrange(len(g.exons) - 1)
            cum = sum(b - a for a, b in g.exons[:j + 1])
            if rng.random() < boundary_bias:     # junction within +-4 of a segment boundary
                kb = rng.randint(1, max(1, nseg - 1))
                t0 = cum - (kb * seg_len + rng.randint(-4, 4))
            else:
                t0 = cum - rng.randint(3, read_len - 3)
            if paired and rng.random() < 0.5:
                t0 -= frag_len - read_len
        else:
            t0 = rng.randint(0, tlen - frag_len)
        t0 = max(0, min(t0, tlen - frag_len))
        rid += 1
        if paired:
            # FR library: left read sense at the fragment start, right read antisense at the end
            flip = rng.random() < 0.5
            l_anti, r_anti = (True, False) if flip else (False, True)
            l_t0, r_t0 = (t0 + frag_len - read_len, t0) if flip else (t0, t0 + frag_len - read_len)
            emit("left", rid, g.exons, l_t0, l_anti, g.ref, rng.random() < indel_frac, g.strand)
            emit("right", rid, g.exons, r_t0, r_anti, g.ref, rng.random() < indel_frac, g.strand)
        else:
            emit("left", rid, g.exons, t0, rng.random() < 0.5, g.ref, rng.random() < indel_frac, g.strand)
    # chimeric (fusion) reads built the way fusion_test/testcases/generate_fasta does: two pieces taken from
    # unrelated loci / strands, glued; segments that lie wholly inside one piece are mapped where that piece is
    for _ in range(fusion_reads):
        rid += 1
        k = rng.randint(seg_len + 2, read_len - seg_len - 2)
        parts = []
        for ln in (k, read_len - k):
            ci = rng.randrange(len(seqs))
            pp = rng.randint(50, len(seqs[ci]) - read_len - 50)
            st = rng.choice("+-")
            g_ = seqs[ci][pp:pp + ln]
            parts.append((ci, pp, st, ln, g_ if st == "+" else revcomp(g_)))
        seq = _mutate(rng, parts[0][4] + parts[1][4], err / 2)
        sd = "left"
        case.reads[sd][rid] = seq
        case.quals[sd][rid] = "I" * read_len
        if paired:
            case.reads["right"][rid] = "".join(rng.choice("ACGT") for _ in range(read_len))
            case.quals["right"][rid] = "I" * read_len
        for kk in range(nseg):
            s0 = kk * seg_len
            s1 = read_len if kk == nseg - 1 else (kk + 1) * seg_len
            if s1 <= k:
                ci, pp, st, ln, _ = parts[0]
                o0, o1 = s0, s1
            elif s0 >= k:
                ci, pp, st, ln, _ = parts[1]
                o0, o1 = s0 - k, s1 - k
            else:
                continue
            gp = pp + o0 if st == "+" else pp + ln - o1
            piece = seqs[ci][gp:gp + (s1 - s0)]
            readpiece = seq[s0:s1] if st == "+" else revcomp(seq[s0:s1])
            nm_, md = md_nm(piece, readpiece)
            if nm_ > 2 or "N" in piece:
                continue
            anti = st == "-"
            qn = "%d|%d:%d:%d" % (rid, s0, kk, nseg)
            case.seg_sam[sd][kk].append("%s\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (
                qn, 16 if anti else 0, names[ci], gp + 1, s1 - s0, readpiece, "I" * (s1 - s0), nm_, md))
            case.seg_recs[sd][kk].append((rid, ci + 1, gp, gp + (s1 - s0), anti, kk == nseg - 1, nm_, nm_, s1 - s0, [(1, s1 - s0)], False))
    return case


def write_case(case: SynthCase, d: str) -> Dict[str, object]:
    """Write the files the reference binaries take (text-SAM twins, FASTQ, FASTA, header)."""
    import os
    from .samtext import sam_header
    os.makedirs(d, exist_ok=True)
    hdr = sam_header(case.names, [len(s) for s in case.seqs])
    paths: Dict[str, object] = {}
    with open(os.path.join(d, "hdr.sam"), "w") as f:
        f.write(hdr)
    paths["hdr"] = os.path.join(d, "hdr.sam")
    with open(os.path.join(d, "ref.fa"), "w") as f:
        for n, s in zip(case.names, case.seqs):
            f.write(">%s\n" % n)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    paths["ref"] = os.path.join(d, "ref.fa")
    for sd in case.reads:
        fq = os.path.join(d, "%s.fq" % sd)
        with open(fq, "w") as f:
            for rid in sorted(case.reads[sd]):
                f.write("@%d\n%s\n+\n%s\n" % (rid, case.reads[sd][rid], case.quals[sd][rid]))
        paths["%s_fq" % sd] = fq
        segs = []
        for k, lines in enumerate(case.seg_sam[sd]):
            p = os.path.join(d, "%s_seg%d.sam" % (sd, k + 1))
            with open(p, "w") as f:
                f.write(hdr)
                f.writelines(lines)
            segs.append(p)
        paths["%s_segs" % sd] = segs
        if any(case.spliced_sam.get(sd, [])):
            dbhdr = "@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in sorted(case.juncdb.items()))
            sps = []
            for k, lines in enumerate(case.spliced_sam[sd]):
                p = os.path.join(d, "%s_seg%d.to_spliced.sam" % (sd, k + 1))
                with open(p, "w") as f:
                    f.write(dbhdr)
                    f.writelines(lines)
                sps.append(p)
            paths["%s_spliced" % sd] = sps
        p = os.path.join(d, "%s_map.sam" % sd)
        with open(p, "w") as f:
            f.write(hdr)
            f.writelines(case.full_sam[sd])
        paths["%s_map" % sd] = p
    return paths


# ---------------------------------------------------------------------------
# Scale generator (torch; runs on any device).  BASELINE.json config shapes.
# ---------------------------------------------------------------------------

def make_scale_genome(seed: int, contig_lens: Sequence[int], n_introns: int, intron_min: int = 70,
                      intron_max: int = 200000, exon_len: int = 600):
    """Random ACGT contigs with as many planted two-exon genes (exon-intron-exon) as fit, up to
    n_introns: intron lengths log-uniform in [intron_min, intron_max], motifs GT-AG 90 % / GC-AG 7 % /
    AT-AC 3 %, both strands.  Returns (list of uint8 ASCII numpy arrays, gene table int64[n,5] =
    (contig, exon1_start, intron_start, intron_end, minus_strand))."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    genes = []
    per = [int(round(n_introns * l / float(sum(contig_lens)))) for l in contig_lens]
    comp = {ord("A"): ord("T"), ord("C"): ord("G"), ord("G"): ord("C"), ord("T"): ord("A")}
    for ci, n in enumerate(contig_lens):
        s = lut[rng.integers(0, 4, size=n, dtype=np.uint8)]
        want = max(1, per[ci])
        lens = np.exp(rng.uniform(np.log(intron_min), np.log(intron_max), size=want)).astype(np.int64)
        pos = 1000
        for il in lens:
            if pos + 2 * exon_len + il + 1000 > n:
                break
            d0 = pos + exon_len            # first intron base
            a1 = d0 + int(il)              # first base after the intron
            u = rng.random()
            don, acc = (b"GT", b"AG") if u < 0.90 else ((b"GC", b"AG") if u < 0.97 else (b"AT", b"AC"))
            minus = rng.random() >= 0.5
            if not minus:
                s[d0:d0 + 2] = np.frombuffer(don, dtype=np.uint8)
                s[a1 - 2:a1] = np.frombuffer(acc, dtype=np.uint8)
            else:   # minus-strand gene: rc(acceptor) .. rc(donor)
                s[d0:d0 + 2] = np.array([comp[acc[1]], comp[acc[0]]], dtype=np.uint8)
                s[a1 - 2:a1] = np.array([comp[don[1]], comp[don[0]]], dtype=np.uint8)
            genes.append((ci, pos, d0, a1, int(minus)))
            pos = a1 + exon_len + 200
        # a few N runs
        for _ in range(5):
            st = int(rng.integers(0, max(1, n - 2000)))
            s[st:st + 1000] = ord("N")
        seqs.append(s)
    return seqs, np.array(genes, dtype=np.int64).reshape(-1, 5)


def make_device_workload(seed: int, seqs_ascii, genes, contig_blk, n_pairs: int, device, read_len: int = 100,
                         seg_len: int = 25, inner_mean: float = 50.0, inner_sd: float = 20.0, err: float = 0.01,
                         exon_len: int = 600, chunk: int = 1 << 20, multi_frac: float = 0.0, dup_shift: int = 0,
                         fusion_frac: float = 0.0, indel_frac: float = 0.0, max_copies: int = 2):
    """Synthesises both sides of `n_pairs` paired reads directly as device-resident thj_seg_batch arrays
    (torch tensors).  Model: a fragment of 2*read_len + max(0, N(inner_mean, inner_sd)) bases drawn
    uniformly from a gene's two-exon transcript; left read = its first read_len bases (sense), right
    read = reverse complement of its last read_len bases; half of the pairs are flipped.  A segment is
    mapped (one hit) where the truth puts it when it lies wholly in one exon and has <= 2 substitution
    errors; segments that straddle the junction are unmapped.  The mate group of a read is the mate's
    full-read hit when the mate is unspliced with <= 2 errors, else the mate's last-segment hit.
    multi_frac > 0 (with a genome whose [dup_shift, 2*dup_shift) is a copy of [0, dup_shift) and genes in the first copy
    only): that fraction of the reads gets every segment hit reported twice, at the locus and at locus + dup_shift --
    a two-copy repeat, which is what sends reads to the multihit tier of the stitch kernels.
    max_copies > 2 (planted repeat family, SURVEY 8d: "multihits from planted repeats up to 41"): the genome holds max_copies copies of
    [0, dup_shift) back to back; the genes inside the first copy are the family, the genes behind the last copy are unique, and
    multi_frac is the share of PAIRS drawn from family genes.  Both reads of such a pair get every segment hit reported at the
    first c copies, c = 2 for 85 % of them, 3..8 for 12 %, 9..40 for 2.7 %, 41 for 0.3 % (those reads are dropped whole by
    max_seg_multihits = 40, segment_juncs.cpp:3499-3506, long_spanning_reads.cpp:2625-2632), capped at max_copies.
    fusion_frac > 0: that fraction of the pairs gets a chimeric LEFT read (the shape of BASELINE configs[3]): its first kb
    segments are the start of one gene's first exon read forward, the rest comes from another gene's first exon (any contig),
    forward or reverse-complemented, the break exactly on a segment boundary -- every segment maps where its part lies, no
    full-read hit; the pair's right read is left as it was.  out["left"]["fusion_reads"] = their row numbers.
    indel_frac > 0: that fraction of the pairs gets a LEFT read with a small deletion (1..3 reference bases missing from the read,
    inside a gene's first exon), one to three bases before the end of a segment: that segment is placed ungapped in the frame
    before the deletion with the mismatches its last bases then show (unmapped with more than two), the following segments
    d bases further on -- the picture find_insertions_and_deletions works from (it records a deletion only where the split
    alignment explains mismatches of the two hits, segment_juncs.cpp:2589-2627).  No full-read hit.
    out["left"]["deletion_reads"] = their row numbers, out["left"]["deletions"] = (contig, left, right) rows.
    Returns {side: dict of tensors} with the field names of thj_seg_batch."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nseg = max(1, read_len // seg_len)
    W = (read_len + 63) // 64
    # genome codes on device, concatenated, for base lookup
    import numpy as np
    code_lut = np.full(256, 4, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code_lut[c] = i
    offs = np.zeros(len(seqs_ascii) + 1, dtype=np.int64)
    for i, s in enumerate(seqs_ascii):
        offs[i + 1] = offs[i] + len(s)
    gcodes = torch.empty(int(offs[-1]), dtype=torch.uint8, device=device)
    for i, s in enumerate(seqs_ascii):
        gcodes[int(offs[i]):int(offs[i + 1])] = torch.from_numpy(code_lut[s]).to(device)
    coff = torch.from_numpy(offs).to(device)
    genes_t = torch.from_numpy(genes).to(device)
    E = exon_len
    out = {}
    sides = ("left", "right")
    bufs = {sd: dict(planes=torch.zeros((n_pairs, 3 * W), dtype=torch.int64, device=device),
                     read_len=torch.full((n_pairs,), read_len, dtype=torch.int16, device=device),
                     seg_mapped=torch.zeros((n_pairs, nseg), dtype=torch.bool, device=device),
                     seg_hits=torch.zeros((n_pairs, nseg, 4), dtype=torch.int32, device=device),
                     span_mapped=torch.zeros((n_pairs, nseg), dtype=torch.bool, device=device),
                     span_hits=torch.zeros((n_pairs, nseg, 8), dtype=torch.int32, device=device),
                     full_ok=torch.zeros((n_pairs,), dtype=torch.bool, device=device),
                     full_hit=torch.zeros((n_pairs, 4), dtype=torch.int32, device=device)) for sd in sides}
    ar = torch.arange(read_len, device=device)
    bitw = (torch.ones(64, dtype=torch.int64, device=device) << torch.arange(64, device=device))
    family = max_copies > 2 and multi_frac > 0 and dup_shift > 0
    ncopy = None
    if family:
        in_fam = (genes_t[:, 0] == 0) & (genes_t[:, 3] + E + 1000 < dup_shift)
        in_uniq = (genes_t[:, 0] != 0) | (genes_t[:, 1] >= max_copies * dup_shift + 1000)
        fam_idx, uniq_idx = torch.nonzero(in_fam).reshape(-1), torch.nonzero(in_uniq).reshape(-1)
        if fam_idx.numel() == 0 or uniq_idx.numel() == 0:
            raise ValueError("the repeat family needs genes inside the first copy and behind the last one")
        ncopy = torch.ones(n_pairs, dtype=torch.int32, device=device)
    for c0 in range(0, n_pairs, chunk):
        n = min(chunk, n_pairs - c0)
        if family:
            is_multi = torch.rand(n, generator=g, device=device) < multi_frac
            gi = torch.where(is_multi, fam_idx[torch.randint(0, fam_idx.numel(), (n,), generator=g, device=device)],
                             uniq_idx[torch.randint(0, uniq_idx.numel(), (n,), generator=g, device=device)])
            u = torch.rand(n, generator=g, device=device)
            c = torch.where(u < 0.85, torch.full((n,), 2, device=device),
                            torch.where(u < 0.97, 3 + torch.randint(0, 6, (n,), generator=g, device=device),
                                        torch.where(u < 0.997, 9 + torch.randint(0, 32, (n,), generator=g, device=device), torch.full((n,), 41, device=device))))
            ncopy[c0:c0 + n] = torch.where(is_multi, c.clamp(max=max_copies), torch.ones_like(c)).to(torch.int32)
        else:
            gi = torch.randint(0, genes_t.shape[0], (n,), generator=g, device=device)
        gene = genes_t[gi]                                   # contig, exon1_start, intron_start, intron_end
        inner = torch.clamp(torch.randn(n, generator=g, device=device) * inner_sd + inner_mean, min=0).to(torch.int64)
        frag = 2 * read_len + inner
        t0 = (torch.rand(n, generator=g, device=device) * (2 * E - frag).clamp(min=1).to(torch.float32)).to(torch.int64)
        flip = torch.rand(n, generator=g, device=device) < 0.5
        ex1, d0, a1, ctg, gminus = gene[:, 1], gene[:, 2], gene[:, 3], gene[:, 0], gene[:, 4]
        for sd in sides:
            first = (sd == "left")
            # which end of the fragment this read takes, and its strand
            at_start = torch.where(flip, torch.tensor(not first, device=device), torch.tensor(first, device=device))
            anti = ~at_start
            ft0 = torch.where(at_start, t0, t0 + frag - read_len)      # transcript offset of the forward piece F
            tpos = ft0[:, None] + ar[None, :]                           # (n, rl) transcript coords of F
            gpos = torch.where(tpos < E, ex1[:, None] + tpos, a1[:, None] + (tpos - E))
            F = gcodes[(coff[ctg][:, None] + gpos)]
            # substitution errors
            errm = torch.rand((n, read_len), generator=g, device=device) < err
            sub = torch.randint(1, 4, (n, read_len), generator=g, device=device, dtype=torch.uint8)
            isn = F == 4
            Fm = torch.where(errm & ~isn, (F + sub) & 3, F)
            mism = (errm & ~isn) | isn            # an N in the genome mismatches any read base we emit (A)
            Fm = torch.where(isn, torch.zeros_like(Fm), Fm)
            # read as sequenced
            seq = torch.where(anti[:, None], (3 - Fm).flip(1), Fm)
            pad = torch.zeros((n, W * 64), dtype=torch.int64, device=device)
            pad[:, :read_len] = seq.to(torch.int64)
            pw = pad.view(n, W, 64)
            lo = ((pw & 1) * bitw).sum(-1)
            hi = (((pw >> 1) & 1) * bitw).sum(-1)
            b = bufs[sd]
            b["planes"][c0:c0 + n, 0:W] = lo
            b["planes"][c0:c0 + n, W:2 * W] = hi
            # segments: segment k of the read covers F[f0:f1]
            cm = torch.cumsum(mism.to(torch.int32), 1)
            cm = torch.cat([torch.zeros((n, 1), dtype=torch.int32, device=device), cm], 1)
            for k in range(nseg):
                s0, s1 = k * seg_len, (read_len if k == nseg - 1 else (k + 1) * seg_len)
                f0 = torch.where(anti, torch.tensor(read_len - s1, device=device), torch.tensor(s0, device=device))
                ln = s1 - s0
                ts = ft0 + f0
                inside = (ts + ln <= E) | (ts >= E)
                nm = cm.gather(1, (f0 + ln)[:, None]).squeeze(1) - cm.gather(1, f0[:, None]).squeeze(1)
                ok = inside & (nm <= 2)
                left = torch.where(ts < E, ex1 + ts, a1 + (ts - E))
                meta = anti.to(torch.int32) | (2 if k == nseg - 1 else 0) | (nm << 8) | (nm << 16) | (ln << 24)
                b["seg_mapped"][c0:c0 + n, k] = ok
                b["seg_hits"][c0:c0 + n, k, 0] = (ctg + 1).to(torch.int32)
                b["seg_hits"][c0:c0 + n, k, 1] = left.to(torch.int32)
                b["seg_hits"][c0:c0 + n, k, 2] = (left + ln).to(torch.int32)
                b["seg_hits"][c0:c0 + n, k, 3] = meta.to(torch.int32)
                # long_spanning_reads input: the contig hit, or -- for a segment that straddles the
                # junction -- the spliced hit the junction-db mapping step would deliver (aM gN bM, XS)
                strad = ~inside
                a_len = (E - ts).clamp(min=1, max=ln - 1)
                sp_ok = strad & (nm <= 2)
                flags = anti.to(torch.int32) | (2 if k == nseg - 1 else 0)
                flags = torch.where(strad, flags | (gminus.to(torch.int32) << 2), flags)
                ncig = torch.where(strad, torch.tensor(3, device=device), torch.tensor(1, device=device)).to(torch.int32)
                smeta = flags | (nm << 8) | (nm << 16) | (ncig << 24)
                sl = ex1 + ts
                c0_ = torch.where(strad, (1 << 28) | a_len, torch.tensor((1 << 28) | ln, device=device))
                c1_ = torch.where(strad, (11 << 28) | (a1 - d0), torch.zeros_like(ts))
                c2_ = torch.where(strad, (1 << 28) | (ln - a_len), torch.zeros_like(ts))
                b["span_mapped"][c0:c0 + n, k] = ok | sp_ok
                sh = b["span_hits"]
                sh[c0:c0 + n, k, 0] = (ctg + 1).to(torch.int32)
                sh[c0:c0 + n, k, 1] = torch.where(strad, sl, left).to(torch.int32)
                sh[c0:c0 + n, k, 2] = smeta.to(torch.int32)
                sh[c0:c0 + n, k, 3] = c0_.to(torch.int64).to(torch.int32)
                sh[c0:c0 + n, k, 4] = c1_.to(torch.int64).to(torch.int32)
                sh[c0:c0 + n, k, 5] = c2_.to(torch.int64).to(torch.int32)
            nm_all = cm[:, read_len]
            unspliced = (ft0 + read_len <= E) | (ft0 >= E)
            b["full_ok"][c0:c0 + n] = unspliced & (nm_all <= 2)
            fl = torch.where(ft0 < E, ex1 + ft0, a1 + (ft0 - E))
            b["full_hit"][c0:c0 + n, 0] = (ctg + 1).to(torch.int32)
            b["full_hit"][c0:c0 + n, 1] = fl.to(torch.int32)
            b["full_hit"][c0:c0 + n, 2] = (fl + read_len).to(torch.int32)
            b["full_hit"][c0:c0 + n, 3] = (anti.to(torch.int32) | 2 | (nm_all << 8) | (nm_all << 16) | (read_len << 24)).to(torch.int32)
    fusion_rows = None
    if fusion_frac > 0 and nseg >= 2:
        fz = torch.nonzero(torch.rand(n_pairs, generator=g, device=device) < fusion_frac).squeeze(1)
        nf = int(fz.shape[0])
        if nf:
            b = bufs["left"]
            ga = genes_t[torch.randint(0, genes_t.shape[0], (nf,), generator=g, device=device)]
            gb = genes_t[torch.randint(0, genes_t.shape[0], (nf,), generator=g, device=device)]
            kb = torch.randint(1, nseg, (nf,), generator=g, device=device)              # segments taken from locus A
            len1 = kb * seg_len
            len2 = read_len - len1
            pa = ga[:, 1] + torch.randint(0, max(1, E - read_len), (nf,), generator=g, device=device)
            pb = gb[:, 1] + torch.randint(0, max(1, E - read_len), (nf,), generator=g, device=device)
            rcb = torch.rand(nf, generator=g, device=device) < 0.5                      # part B reverse-complemented
            in_a = ar[None, :] < len1[:, None]
            ib = ar[None, :] - len1[:, None]                                            # index into part B
            gp_a = coff[ga[:, 0]][:, None] + pa[:, None] + ar[None, :]
            gp_b = coff[gb[:, 0]][:, None] + pb[:, None] + torch.where(rcb[:, None], len2[:, None] - 1 - ib, ib)
            codes = gcodes[torch.where(in_a, gp_a, gp_b).clamp(min=0, max=gcodes.shape[0] - 1)]
            codes = torch.where(~in_a & rcb[:, None] & (codes < 4), 3 - codes, codes)
            codes = torch.where(codes > 3, torch.zeros_like(codes), codes)              # (first exons hold no N: make_scale_genome)
            pad = torch.zeros((nf, W * 64), dtype=torch.int64, device=device)
            pad[:, :read_len] = codes.to(torch.int64)
            pw = pad.view(nf, W, 64)
            b["planes"][fz, 0:W] = ((pw & 1) * bitw).sum(-1)
            b["planes"][fz, W:2 * W] = (((pw >> 1) & 1) * bitw).sum(-1)
            b["planes"][fz, 2 * W:3 * W] = 0
            b["full_ok"][fz] = False
            for k in range(nseg):
                s0, s1 = k * seg_len, (read_len if k == nseg - 1 else (k + 1) * seg_len)
                ln = s1 - s0
                a_side = kb > k
                o0 = s0 - len1                                                           # offset of the segment inside part B
                left_b = torch.where(rcb, pb + len2 - (o0 + ln), pb + o0)
                ctgk = torch.where(a_side, ga[:, 0], gb[:, 0])
                leftk = torch.where(a_side, pa + s0, left_b)
                antik = (~a_side & rcb).to(torch.int32)
                meta = antik | (2 if k == nseg - 1 else 0) | (ln << 24)
                b["seg_mapped"][fz, k] = True
                b["seg_hits"][fz, k, 0] = (ctgk + 1).to(torch.int32)
                b["seg_hits"][fz, k, 1] = leftk.to(torch.int32)
                b["seg_hits"][fz, k, 2] = (leftk + ln).to(torch.int32)
                b["seg_hits"][fz, k, 3] = meta.to(torch.int32)
                b["span_mapped"][fz, k] = True
                sh = b["span_hits"]
                sh[fz, k, 0] = (ctgk + 1).to(torch.int32)
                sh[fz, k, 1] = leftk.to(torch.int32)
                sh[fz, k, 2] = (antik | (2 if k == nseg - 1 else 0) | (1 << 24)).to(torch.int32)
                sh[fz, k, 3] = (1 << 28) | ln
                sh[fz, k, 4] = 0
                sh[fz, k, 5] = 0
            fusion_rows = fz
    deletion_rows = deletion_truth = None
    if indel_frac > 0 and nseg >= 3:
        pick = torch.rand(n_pairs, generator=g, device=device) < indel_frac
        if fusion_rows is not None:
            pick[fusion_rows] = False
        dz = torch.nonzero(pick).squeeze(1)
        nd = int(dz.shape[0])
        if nd:
            b = bufs["left"]
            ga = genes_t[torch.randint(0, genes_t.shape[0], (nd,), generator=g, device=device)]
            pa = ga[:, 1] + torch.randint(0, max(1, E - read_len - 4), (nd,), generator=g, device=device)
            dl = torch.randint(1, 4, (nd,), generator=g, device=device)                 # deleted reference bases
            kb = torch.randint(1, max(2, nseg - 1), (nd,), generator=g, device=device)  # the deletion sits at the end of segment kb - 1 ...
            m = torch.randint(1, 4, (nd,), generator=g, device=device)                  # ... m bases before its end
            x = kb * seg_len - m
            shift = torch.where(ar[None, :] >= x[:, None], dl[:, None], torch.zeros_like(dl)[:, None])
            base0 = coff[ga[:, 0]][:, None] + pa[:, None] + ar[None, :]
            codes = gcodes[(base0 + shift).clamp(min=0, max=gcodes.shape[0] - 1)]       # the read
            refc = gcodes[base0.clamp(min=0, max=gcodes.shape[0] - 1)]                  # the reference in the frame before the deletion
            codes = torch.where(codes > 3, torch.zeros_like(codes), codes)
            # the segment that holds the deletion is placed ungapped in that frame: its last m bases are read against shifted ones
            in_tail = (ar[None, :] >= x[:, None]) & (ar[None, :] < (kb * seg_len)[:, None])
            nm_tail = (in_tail & (codes != refc)).sum(1).to(torch.int32)
            pad = torch.zeros((nd, W * 64), dtype=torch.int64, device=device)
            pad[:, :read_len] = codes.to(torch.int64)
            pw = pad.view(nd, W, 64)
            b["planes"][dz, 0:W] = ((pw & 1) * bitw).sum(-1)
            b["planes"][dz, W:2 * W] = (((pw >> 1) & 1) * bitw).sum(-1)
            b["planes"][dz, 2 * W:3 * W] = 0
            b["full_ok"][dz] = False
            for k in range(nseg):
                s0, s1 = k * seg_len, (read_len if k == nseg - 1 else (k + 1) * seg_len)
                ln = s1 - s0
                holds = kb - 1 == k
                nmk = torch.where(holds, nm_tail, torch.zeros_like(nm_tail))
                ok = nmk <= 2                                                            # an ungapped mapper with two mismatches allowed
                leftk = torch.where(kb <= k, pa + s0 + dl, pa + s0)
                endf = 2 if k == nseg - 1 else 0
                b["seg_mapped"][dz, k] = ok
                b["seg_hits"][dz, k, 0] = (ga[:, 0] + 1).to(torch.int32)
                b["seg_hits"][dz, k, 1] = leftk.to(torch.int32)
                b["seg_hits"][dz, k, 2] = (leftk + ln).to(torch.int32)
                b["seg_hits"][dz, k, 3] = (endf | (nmk << 8) | (nmk << 16) | (ln << 24)).to(torch.int32)
                b["span_mapped"][dz, k] = ok
                sh = b["span_hits"]
                sh[dz, k, 0] = (ga[:, 0] + 1).to(torch.int32)
                sh[dz, k, 1] = leftk.to(torch.int32)
                sh[dz, k, 2] = (endf | (nmk << 8) | (nmk << 16) | (1 << 24)).to(torch.int32)
                sh[dz, k, 3] = (1 << 28) | ln
                sh[dz, k, 4] = 0
                sh[dz, k, 5] = 0
            deletion_rows = dz
            deletion_truth = torch.stack([ga[:, 0] + 1, pa + x - 1, pa + x + dl], 1)
    del gcodes

    def take(rows, mask):
        # torch's masked / indexed row selection returns wrong rows on this ROCm build once the result passes 2^29 elements
        # (tools/gen40_probe2.py): select in pieces of 2^24 rows
        P = 1 << 24
        if rows.shape[0] <= P:
            return rows[mask].contiguous()
        return torch.cat([rows[i:i + P][mask[i:i + P]] for i in range(0, rows.shape[0], P)]).contiguous()

    def csr(mapped, rows, multi, left_cols):
        """CSR offsets + rows of the mapped (read, segment) cells; cells of `multi` reads appear twice, the copy
        shifted by dup_shift in the columns `left_cols`"""
        cnt = mapped.to(torch.int32)
        if multi is not None:
            cnt = cnt * (multi if multi.dtype == torch.int32 else 1 + multi.to(torch.int32)).repeat_interleave(nseg)
        off = torch.zeros(n_pairs * nseg + 1, dtype=torch.int32, device=device)
        off[1:] = torch.cumsum(cnt, 0)
        PIECE = 1 << 24
        if multi is None:
            return off, take(rows, mapped)
        cell = torch.arange(n_pairs * nseg, device=device).repeat_interleave(cnt.to(torch.int64))
        copy = torch.arange(cell.shape[0], device=device, dtype=torch.int64) - off[:-1].to(torch.int64)[cell]
        out_rows = torch.cat([rows[cell[i:i + PIECE]] for i in range(0, cell.shape[0], PIECE)]) if cell.shape[0] > PIECE else rows[cell].clone()
        for col in left_cols:
            out_rows[:, col] += (copy * dup_shift).to(out_rows.dtype)
        return off, out_rows.contiguous()

    def side_multi(sd):
        multi = None
        if family:
            multi = ncopy.clone()                            # copies per pair (int32): both reads of a family pair are multihits
            if sd == "left":                                 # ... but not a left read that was replaced by a chimeric / deletion read
                if fusion_rows is not None:
                    multi[fusion_rows] = 1
                if deletion_rows is not None:
                    multi[deletion_rows] = 1
        elif multi_frac > 0 and dup_shift > 0:
            multi = torch.rand(n_pairs, generator=g, device=device) < multi_frac
        return multi

    multis = {sd: side_multi(sd) for sd in sides}
    for sd in sides:
        b = bufs[sd]
        osd = "right" if sd == "left" else "left"
        other = bufs[osd]
        multi = multis[sd]
        mapped = b["seg_mapped"].reshape(-1)
        seg_off, hits = csr(mapped, b["seg_hits"].reshape(-1, 4), multi, (1, 2))
        # the mate's hit group (find_gaps, segment_juncs.cpp:3321-3348): its whole-read map when it has one, else its last segment's
        # hits -- and a family read's mate maps at every one of ITS copies, as tools/thj_gen.cpp writes the map files (the rescue is a
        # double loop over (left-segment hit, mate hit), :3406-3412).  With a single-sided workload the mate keeps one hit.
        m_has = other["full_ok"] | other["seg_mapped"][:, nseg - 1]
        m_rows = torch.where(other["full_ok"][:, None], other["full_hit"], other["seg_hits"][:, nseg - 1, :])
        m_multi = multis.get(osd)
        if m_multi is not None and m_multi.dtype != torch.int32:
            m_multi = 1 + m_multi.to(torch.int32)
        m_cnt = m_has.to(torch.int32) * (m_multi if m_multi is not None else 1)
        mate_off = torch.zeros(n_pairs + 1, dtype=torch.int32, device=device)
        mate_off[1:] = torch.cumsum(m_cnt, 0)
        if m_multi is None:
            mh = take(m_rows, m_has)
        else:
            cell = torch.arange(n_pairs, device=device).repeat_interleave(m_cnt.to(torch.int64))
            copy = torch.arange(cell.shape[0], device=device, dtype=torch.int64) - mate_off[:-1].to(torch.int64)[cell]
            mh = m_rows[cell].clone()
            for col in (1, 2):
                mh[:, col] += (copy * dup_shift).to(mh.dtype)
            mh = mh.contiguous()
        smapped = b["span_mapped"].reshape(-1)
        span_off, span_hits = csr(smapped, b["span_hits"].reshape(-1, 8), multi, (1,))
        quals = torch.full((n_pairs * read_len,), ord("I"), dtype=torch.uint8, device=device)
        out[sd] = dict(n_reads=n_pairs, nseg=nseg, W=W, seg_off=seg_off, hits=hits,
                       span_off=span_off, span_hits=span_hits, quals=quals, qual_stride=read_len,
                       planes=b["planes"].reshape(-1).contiguous(), read_len=b["read_len"],
                       mate_off=mate_off, mate_hits=mh)
    if fusion_rows is not None:
        out["left"]["fusion_reads"] = fusion_rows
    if deletion_rows is not None:
        out["left"]["deletion_reads"] = deletion_rows
        out["left"]["deletions"] = deletion_truth
    return out
