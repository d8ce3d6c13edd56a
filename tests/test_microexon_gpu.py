"""GPU: the microexon search (segment_juncs.cpp:3737-3941) through the C ABI -- thj_microexon_collect / _candidates / _run -- and through
the segment_juncs executable (no --no-microexon-search on its command line), against the oracle on the seeded cases of
tests/mx_util.py.  No reference vector exists for this mode: see oracle/README.md."""
import os
import subprocess

import numpy as np
import pytest

import mx_util
import orc
from tophat_amd import host
from tophat_amd.batch import build_seg_batch, merge_events, write_segment_files
from tophat_amd.params import Params

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _sides(seed):
    L = 25 if seed % 3 else 20
    seqs, genes, reads, seg_recs = mx_util.make_case(seed, seg_len=L, nseg=4 if seed % 2 else 3, n_reads=240)
    p = Params(segment_length=L)
    if seed % 4 == 3:
        p.library_type = 2 + (seed // 4) % 2
    ids = sorted(reads)
    half = ids[len(ids) // 2]
    left = {k: v for k, v in reads.items() if k < half}
    right = {k: v for k, v in reads.items() if k >= half}
    bl = build_seg_batch([[r for r in v if r[0] < half] for v in seg_recs], left)
    br = build_seg_batch([[r for r in v if r[0] >= half] for v in seg_recs], right)
    return p, [orc.fold_genome_char(s) for s in seqs], bl, br


@pytest.mark.parametrize("seed", range(10))
def test_c_abi_against_the_oracle(seed):
    p, seqs, bl, br = _sides(seed)
    og = orc.Genome(seqs)
    want, nw = orc.microexon_search(p, og, [(bl, 1), (br, 2)], p.min_anchor_len, 50, 5000000)
    want_seg = merge_events(orc.segjuncs(Params(**{**p.__dict__, "read_side": 1}), og, bl), orc.segjuncs(Params(**{**p.__dict__, "read_side": 2}), og, br))
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.reset()
        ctx.microexon_reset()
        dl, dr = ctx.upload_batch(bl, ordinal_base=0), ctx.upload_batch(br, ordinal_base=1 << 28)
        # in two calls per side order does not matter: the right side is collected first here, the merge sorts
        ctx.microexon_collect(p, dr, 2)
        ctx.microexon_collect(p, dl, 1)
        cands = ctx.microexon_candidates()
        windows, strs, lens, wins = host.microexon_merge_windows(cands)
        assert len(windows) == nw
        found = ctx.microexon_run(windows, strs, lens, wins, 50, p.library_type)
        ev = ctx.download(ctx.finish())
        assert found >= len(want)                          # counted with repeats of a pair that overlapping windows share
        assert ev.juncs.tolist() == want.tolist()
        # on top of the segment search: the union, as segment_juncs.cpp:5026-5028 inserts them into one set
        ctx.reset()
        for sd, b in ((1, dl), (2, dr)):
            ctx.run(Params(**{**p.__dict__, "read_side": sd}), b)
        ctx.microexon_run(windows, strs, lens, wins, 50, p.library_type)
        ev2 = ctx.download(ctx.finish())
    both = np.unique(np.concatenate([want, want_seg.juncs])) if len(want) + len(want_seg.juncs) else want
    both = both[np.lexsort((both["antisense"], both["right"], both["left"], both["ref_id"]))] if len(both) else both
    assert ev2.juncs.tolist() == both.tolist()
    assert len(want) > 0


def test_the_cut_on_the_device():
    p, seqs, bl, br = _sides(1)
    og = orc.Genome(seqs)
    full, _ = orc.microexon_search(p, og, [(bl, 1), (br, 2)], p.min_anchor_len, 50, 5000000)
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome(seqs))
        ctx.microexon_reset()
        ctx.microexon_collect(p, ctx.upload_batch(bl, ordinal_base=0), 1)
        ctx.microexon_collect(p, ctx.upload_batch(br, ordinal_base=1 << 28), 2)
        parts = host.microexon_merge_windows(ctx.microexon_candidates())
        for cap in (1, 2, max(1, len(full) - 1)):
            want, _ = orc.microexon_search(p, og, [(bl, 1), (br, 2)], p.min_anchor_len, 50, cap)
            ctx.reset()
            ctx.microexon_run(*parts, 50, p.library_type, cap)
            got = ctx.download(ctx.finish())
            assert got.juncs.tolist() == want.tolist(), cap


def test_segment_juncs_executable_with_the_microexon_search(tmp_path):
    """the drop-in program without --no-microexon-search (what `tophat --microexon-search` runs): its junction file = the segment
    search's junctions and the microexon search's, as the oracle finds them"""
    seed = 7
    L = 25
    seqs, genes, reads, seg_recs = mx_util.make_case(seed, seg_len=L, nseg=4, n_reads=300, with_n=False)
    p = Params(segment_length=L)
    names = ["chr%d" % (k + 1) for k in range(len(seqs))]
    open(tmp_path / "ref.fa", "w").write("".join(">%s\n%s\n" % (n_, s_) for n_, s_ in zip(names, seqs)))
    hdr = "@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n_, len(s_)) for n_, s_ in zip(names, seqs))
    open(tmp_path / "hdr.sam", "w").write(hdr)
    open(tmp_path / "left.fq", "w").write("".join("@%d\n%s\n+\n%s\n" % (rid, r, "I" * len(r)) for rid, r in sorted(reads.items())))
    from tophat_amd.samtext import md_nm
    segf = []
    for s_, recs in enumerate(seg_recs):
        path = str(tmp_path / ("left_seg%d.sam" % (s_ + 1)))
        with open(path, "w") as f:
            f.write(hdr)
            for (rid, ref, left, right, anti, _end, _mm, _ed, rl) in recs:
                rd = reads[rid]
                piece = rd[s_ * L:(s_ + 1) * L] if s_ < len(seg_recs) - 1 else rd[s_ * L:]
                q = mx_util.rc(piece) if anti else piece
                nm_, md = md_nm(seqs[ref - 1][left:right], q)
                f.write("%d|%d:%d:%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\tMD:Z:%s\n" % (rid, s_ * L, s_, len(seg_recs), 16 if anti else 0, names[ref - 1], left + 1, rl, q,
                                                                                                  "I" * rl, nm_, md))
        segf.append(path)
    outs = {k: str(tmp_path / ("out." + k)) for k in ("juncs", "insertions", "deletions", "fusions")}
    exe = os.path.join(ROOT, "tophat_amd", "bin", "segment_juncs")
    r = subprocess.run([exe, "--no-coverage-search", "--segment-length", str(L), "--sam-header", str(tmp_path / "hdr.sam"), str(tmp_path / "ref.fa"),
                        outs["juncs"], outs["insertions"], outs["deletions"], outs["fusions"], str(tmp_path / "left.fq"), "/dev/null", ",".join(segf)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Performing microexon-search" in r.stderr
    # the oracle on what the program parsed: hit mismatches come from NM, so rebuild the batch from the files' own numbers
    from tophat_amd.samtext import parse_sam_hits
    ref_ids = {n_: i + 1 for i, n_ in enumerate(names)}
    b = build_seg_batch([list(parse_sam_hits(f, ref_ids, 500000)) for f in segf], reads)
    og = orc.Genome([orc.fold_genome_char(s) for s in seqs])
    pl = Params(segment_length=L, read_side=1)
    seg = orc.segjuncs(pl, og, b)
    mx, nw = orc.microexon_search(pl, og, [(b, 1)], pl.min_anchor_len, 50, 5000000)
    assert len(mx) > 0 and nw > 3
    both = np.unique(np.concatenate([mx, seg.juncs]))
    both = both[np.lexsort((both["antisense"], both["right"], both["left"], both["ref_id"]))]
    seg.juncs = both
    write_segment_files(seg, names, str(tmp_path / "want.juncs"), str(tmp_path / "want.ins"), str(tmp_path / "want.del"))
    assert open(outs["juncs"]).read() == open(tmp_path / "want.juncs").read()
    assert ("microexon segments in %d windows" % nw) in r.stderr
