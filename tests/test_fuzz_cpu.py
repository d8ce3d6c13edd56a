"""CPU fuzz: adversarial, not biologically plausible inputs -- hits at contig ends, tiny contigs, N runs, random
strands, multihits, random mates, odd parameters -- through the kernel logic (tests/hostsim) against the plain-C
oracle.  Every window / split / flank scan of the bit-plane code touches genome words near contig boundaries and
guard blocks here, which the generator-shaped cases never do."""
import os

import numpy as np
import pytest

import orc
import sim
from tophat_amd.batch import HIT_DTYPE, JUNC_DTYPE, SPAN_HIT_DTYPE, SegBatch, SpanBatch, events_to_span_inputs
from tophat_amd.params import Params
from util import assert_events_equal

# THJ_FUZZ_SEEDS=<n> widens the sweep (e.g. a few thousand seeds before a release)
N_SEEDS = int(os.environ.get("THJ_FUZZ_SEEDS", "60"))


def rand_genome(rng, n_contigs):
    seqs = []
    for _ in range(n_contigs):
        n = int(rng.choice([70, 130, 400, 2500, 9000]))
        s = rng.choice(list("ACGT"), size=n)
        # splice motifs everywhere so that windows actually fire, and a few N runs
        for _k in range(n // 40):
            p = int(rng.integers(0, max(1, n - 2)))
            s[p:p + 2] = list(rng.choice(["GT", "AG", "GC", "AT", "AC", "CT"]))
        for _k in range(int(rng.integers(0, 3))):
            p = int(rng.integers(0, n))
            s[p:p + int(rng.integers(1, 30))] = "N"
        seqs.append("".join(s))
    return seqs


def rand_hit(rng, seqs, L, near=None):
    ref = int(rng.integers(1, len(seqs) + 1)) if near is None or rng.random() < 0.15 else near[0]
    n = len(seqs[ref - 1])
    if near is not None and ref == near[0] and rng.random() < 0.8:
        left = near[1] + int(rng.choice([L, L, L + 1, L - 1, L + 60, L + 300, -L, -L - 80, 3, 0]))
    else:
        left = int(rng.choice([0, 1, max(0, n - L), max(0, n - L - 1), int(rng.integers(0, max(1, n - L + 1)))]))
    left = min(max(left, 0), max(0, n - L))
    return ref, left


def rand_seg_batch(rng, seqs, n_reads, L, nseg, paired):
    rl = L * nseg + int(rng.integers(0, L))
    hits, seg_off, bases, read_off, mate_off, mate_hits = [], [0], bytearray(), [0], [0], []
    for _r in range(n_reads):
        anchor = rand_hit(rng, seqs, L)
        anti = int(rng.random() < 0.5)
        for s in range(nseg):
            k = int(rng.choice([0, 1, 1, 1, 2, 3]))
            for _ in range(k):
                ref, left = rand_hit(rng, seqs, L, (anchor[0], anchor[1] + (s * L if not anti else -s * L)))
                a = anti if rng.random() < 0.85 else 1 - anti
                ln = L if s < nseg - 1 else rl - s * L
                ln = min(ln, len(seqs[ref - 1]) - left)
                mm = int(rng.integers(0, 3))
                hits.append((ref, left, left + ln, a | (2 if s == nseg - 1 else 0), mm, mm, ln))
            seg_off.append(len(hits))
        # read: mostly genome-derived around the anchor so that scans find things, sometimes random / with N
        ref, left = anchor
        g = seqs[ref - 1]
        piece = (g[left:left + rl] + "".join(rng.choice(list("ACGT"), size=rl)))[:rl]
        if rng.random() < 0.3:
            cut = int(rng.integers(5, rl - 5))
            p2 = int(rng.integers(0, max(1, len(g) - rl)))
            piece = (piece[:cut] + g[p2:p2 + rl] + "A" * rl)[:rl]
        piece = "".join(c if rng.random() > 0.02 else "N" for c in piece)
        bases += piece.encode()
        read_off.append(len(bases))
        if paired:
            for _ in range(int(rng.choice([0, 1, 1, 2]))):
                mref, mleft = rand_hit(rng, seqs, L, anchor)
                ml = min(rl, len(seqs[mref - 1]) - mleft)
                mate_hits.append((mref, mleft, mleft + ml, int(rng.random() < 0.5) | 2, 0, 0, min(ml, 255)))
            mate_off.append(len(mate_hits))
    args = [nseg, np.arange(1, n_reads + 1, dtype=np.uint32), np.array(read_off, dtype=np.int64),
            np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.array(seg_off, dtype=np.uint32),
            np.array(hits, dtype=HIT_DTYPE) if hits else np.zeros(0, dtype=HIT_DTYPE)]
    if paired:
        args += [np.array(mate_off, dtype=np.uint32), np.array(mate_hits, dtype=HIT_DTYPE) if mate_hits else np.zeros(0, dtype=HIT_DTYPE)]
    return SegBatch(*args)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_segment_juncs(seed):
    rng = np.random.default_rng(1000 + seed)
    seqs = rand_genome(rng, int(rng.integers(1, 4)))
    L = int(rng.choice([20, 25, 25, 32, 33, 47, 50, 64]))
    nseg = int(rng.choice([2, 3, 4, 6])) if L <= 32 else int(rng.choice([2, 3]))
    paired = bool(seed % 2)
    b = rand_seg_batch(rng, seqs, 60, L, nseg, paired)
    p = Params(segment_length=L, read_side=1 + seed % 2, library_type=int(rng.choice([0, 0, 1, 2, 3])),
               min_segment_intron=int(rng.choice([10, 50])), max_segment_intron=int(rng.choice([400, 5000, 500000])),
               max_insertion_length=int(rng.choice([1, 3, 6])), max_deletion_length=int(rng.choice([1, 3, 10])),
               inner_dist_mean=int(rng.choice([0, 30, 50])), inner_dist_std_dev=int(rng.choice([5, 20, 60])),
               segment_mismatches=int(rng.choice([0, 2, 3])))
    g = orc.Genome(seqs)
    want = orc.segjuncs(p, g, b)
    got = sim.segjuncs(p, seqs, b)
    assert_events_equal(got, want, "seed %d" % seed)
    assert got.stats["windows"] == want.stats["windows"] and got.stats["indel_pairs"] == want.stats["indel_pairs"]
    assert got.stats["rescue_pairs"] == want.stats["rescue_pairs"]
    # fusion search on the same adversarial batch
    pf = Params(**{**p.__dict__, "fusion_min_dist": int(rng.choice([50, 1000])), "fusion_anchor_length": int(rng.choice([10, 20]))})
    wf = orc.fusions(pf, g, b, pf.fusion_anchor_length, pf.fusion_min_dist)
    assert sim.fusions(pf, seqs, b).tolist() == wf.tolist()


def rand_span_batch(rng, seqs, n_reads, L, nseg):
    rl = L * nseg + int(rng.integers(0, L))
    hits, seg_off, bases, quals, read_off = [], [0], bytearray(), bytearray(), [0]
    for _r in range(n_reads):
        anchor = rand_hit(rng, seqs, L)
        anti = int(rng.random() < 0.5)
        for s in range(nseg):
            k = int(rng.choice([0, 1, 1, 1, 1, 2, 3])) if s else int(rng.choice([1, 1, 2]))
            for _ in range(k):
                ln = L if s < nseg - 1 else rl - s * L
                step = s * L if not anti else (nseg - 1 - s) * L
                ref, left = rand_hit(rng, seqs, L, (anchor[0], anchor[1] + step - L))
                a = anti if rng.random() < 0.9 else 1 - anti
                mm = int(rng.integers(0, 3))
                flags = a | (2 if s == nseg - 1 else 0)
                shape = rng.random()
                room = len(seqs[ref - 1]) - left - ln          # hits never run off their contig (bowtie cannot report that)
                if room < 0:
                    ln += room
                if ln < 1:
                    continue
                if shape < 0.7 or ln < 12 or room < 900:
                    cig, ed = [(1 << 28) | ln, 0, 0, 0, 0], mm
                elif shape < 0.85:      # spliced segment hit aM gN bM
                    a_ = int(rng.integers(3, ln - 3))
                    cig, ed = [(1 << 28) | a_, (11 << 28) | int(rng.integers(20, 900)), (1 << 28) | (ln - a_), 0, 0], mm
                    flags |= 4 if rng.random() < 0.5 else 0
                elif shape < 0.93:      # deletion inside the segment
                    a_ = int(rng.integers(3, ln - 3)); d_ = int(rng.integers(1, 3))
                    cig, ed = [(1 << 28) | a_, (5 << 28) | d_, (1 << 28) | (ln - a_), 0, 0], mm + d_
                else:                   # insertion inside the segment
                    a_ = int(rng.integers(3, ln - 5)); i_ = int(rng.integers(1, 3))
                    cig, ed = [(1 << 28) | a_, (3 << 28) | i_, (1 << 28) | (ln - a_ - i_), 0, 0], mm + i_
                hits.append((ref, left, flags, mm, ed & 0xFF, sum(1 for c in cig if c), cig))
            seg_off.append(len(hits))
        ref, left = anchor
        g = seqs[ref - 1]
        piece = (g[left:left + rl] + "".join(rng.choice(list("ACGT"), size=rl)))[:rl]
        piece = "".join(c if rng.random() > 0.03 else rng.choice(list("ACGTN")) for c in piece)
        bases += piece.encode()
        quals += bytes(int(x) for x in rng.integers(33, 75, size=rl))
        read_off.append(len(bases))
    return SpanBatch(nseg, np.arange(1, n_reads + 1, dtype=np.uint32), np.array(read_off, dtype=np.int64),
                     np.frombuffer(bytes(bases), dtype=np.uint8).copy(), np.frombuffer(bytes(quals), dtype=np.uint8).copy(),
                     np.array(seg_off, dtype=np.uint32), np.array(hits, dtype=SPAN_HIT_DTYPE))


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_long_spanning_reads(seed):
    rng = np.random.default_rng(5000 + seed)
    seqs = rand_genome(rng, int(rng.integers(1, 3)))
    L = int(rng.choice([20, 25, 25, 40]))
    nseg = int(rng.choice([1, 2, 3, 4, 6]))
    if L * (nseg + 1) > 256:          # reads stay within the 256 bp the device path supports
        nseg = 256 // L - 1
    sb = rand_span_batch(rng, seqs, 70, L, nseg)
    p = Params(segment_length=L, max_insertion_length=int(rng.choice([1, 3])), max_deletion_length=int(rng.choice([1, 3, 10])),
               min_report_intron=int(rng.choice([10, 50])), max_report_intron=int(rng.choice([300, 5000, 500000])),
               read_mismatches=int(rng.choice([2, 4])), read_edit_dist=int(rng.choice([2, 5])), read_gap_length=int(rng.choice([2, 3])))
    # a junction / insertion set dense around the hits: every plausible gap between two hits of a read, plus noise
    juncs, ins = set(), {}
    h = sb.hits
    for k in range(0, len(h) - 1):
        a, b_ = h[k], h[k + 1]
        if a["ref_id"] != b_["ref_id"]:
            continue
        ra = int(a["left"]) + sum(int(c & 0x0FFFFFFF) for c in a["cigar"] if (c >> 28) in (1, 5, 11))
        for d in (-2, 0, 1):
            l_, r_ = ra - 1 + d, int(b_["left"]) + d
            if r_ > l_ + 1 and l_ >= 0:
                juncs.add((int(a["ref_id"]), l_, r_, int(rng.integers(0, 2))))
        if 0 < ra - int(b_["left"]) <= 3:
            ins[(int(a["ref_id"]), int(b_["left"]) + int(rng.integers(-1, 2)), ra - int(b_["left"]))] = "".join(rng.choice(list("ACGT"), size=ra - int(b_["left"])))
    jl = sorted(juncs)
    ja = np.array(jl, dtype=JUNC_DTYPE) if jl else np.zeros(0, dtype=JUNC_DTYPE)
    il = [(k[0], k[1], v) for k, v in sorted(ins.items()) if k[1] >= 0]
    g = orc.Genome(seqs)
    want = orc.spanning(p, g, sb, ja, il)
    # the second restatement (spanning_fusion_oracle.c, the one with the fusion branches) agrees when fusion search is off
    assert orc.spanning_fusion(p, g, sb, ja, il, np.zeros(0, dtype=orc.SPAN_FUSION_DTYPE), False) == want
    for mode in (0, 1, 2, 3):
        got, status = sim.spanning(p, seqs, sb, ja, il, mode)
        assert status[1] == 0
        got.sort(key=lambda a: a.read_idx)
        # MD strings over 40 characters do not fit a device record: the kernels flag them (THJ_MD_ON_HOST) and the host
        # rebuilds them (thj_md_string) -- the records must come out all the same
        assert status[2] == 0
        assert got == want, "seed %d mode %d" % (seed, mode)


def fusion_set_near_hits(rng, sb):
    """fusions whose break points sit where two hits of neighbouring segments of a read end / start, in every direction and in
    both contig orders, a few bases off as well -- what segment_juncs --fusion-search would have reported, and noise"""
    rows = set()
    h = sb.hits
    for k in range(0, len(h) - 1):
        a, b_ = h[k], h[k + 1]
        ra = int(a["left"]) + sum(int(c & 0x0FFFFFFF) for c in a["cigar"][:a["n_cigar"]] if (c >> 28) in (1, 5, 11))
        for d in (-2, 0, 1):
            for (x, y) in ((ra - 1 + d, int(b_["left"]) + d), (int(a["left"]) + d, int(b_["left"]) + d), (ra - 1 + d, int(b_["left"]) + 24 + d)):
                if x < 0 or y < 0:
                    continue
                dr = int(rng.integers(7, 11))
                r1, r2 = int(a["ref_id"]), int(b_["ref_id"])
                rows.add((r1, r2, x, y, dr))
                if rng.random() < 0.5:
                    rows.add((r2, r1, y, x, int(rng.integers(7, 11))))
    return np.array(sorted(rows), dtype=orc.SPAN_FUSION_DTYPE) if rows else np.zeros(0, dtype=orc.SPAN_FUSION_DTYPE)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_fusion_tier(seed):
    """the fusion tier's logic (thj_span_fusion.h compiled for the CPU) against the oracle on adversarial batches: with fusion
    search off it is a third implementation of the plain path, with it on the fusion branches run on random hit geometry"""
    rng = np.random.default_rng(7100 + seed)
    seqs = rand_genome(rng, int(rng.integers(1, 4)))
    L = int(rng.choice([20, 25, 25, 40]))
    nseg = int(rng.choice([1, 2, 3, 4, 6]))
    if L * (nseg + 1) > 256:
        nseg = 256 // L - 1
    sb = rand_span_batch(rng, seqs, 60, L, nseg)
    p = Params(segment_length=L, max_insertion_length=int(rng.choice([1, 3])), max_deletion_length=int(rng.choice([1, 3, 10])),
               min_report_intron=int(rng.choice([10, 50])), max_report_intron=int(rng.choice([300, 5000, 500000])),
               read_mismatches=int(rng.choice([2, 4])), read_edit_dist=int(rng.choice([2, 5])), read_gap_length=int(rng.choice([2, 3])))
    p.fusion_min_dist = int(rng.choice([100, 1500, 10000000]))
    juncs = set()
    h = sb.hits
    for k in range(0, len(h) - 1):
        a, b_ = h[k], h[k + 1]
        if a["ref_id"] != b_["ref_id"]:
            continue
        ra = int(a["left"]) + sum(int(c & 0x0FFFFFFF) for c in a["cigar"] if (c >> 28) in (1, 5, 11))
        for d in (-2, 0, 1):
            l_, r_ = ra - 1 + d, int(b_["left"]) + d
            if r_ > l_ + 1 and l_ >= 0:
                juncs.add((int(a["ref_id"]), l_, r_, int(rng.integers(0, 2))))
    jl = sorted(juncs)
    ja = np.array(jl, dtype=JUNC_DTYPE) if jl else np.zeros(0, dtype=JUNC_DTYPE)
    g = orc.Genome(seqs)
    fus = fusion_set_near_hits(rng, sb)
    for fs in (0, 1):
        p.fusion_search = fs
        want = orc.spanning_fusion(p, g, sb, ja, [], fus, bool(fs))
        if fs == 0:
            assert want == orc.spanning(p, g, sb, ja, [])
        for skip0 in (False, True):
            got, status = sim.spanning_fusion(p, seqs, sb, ja, [], fus, skip0)
            assert status[1] == 0
            got.sort(key=lambda a: a.read_idx)
            assert got == want, "seed %d fusion_search %d skip_tier0 %s" % (seed, fs, skip0)
