"""GPU: the HIP path through the C ABI -- and the drop-in executables -- against the reference's own regression cases, all
nine (tests/golden_ref/, see tests/ref_regression.py), and record for record against the oracle on the same inputs.
Reads of 24 bases in two 12-base or three 8-base segments: the smallest shapes the kernels take."""
import copy
import os
import subprocess

import pytest

import orc
import ref_regression as rr
from test_ref_regression_cpu import RECORDED, side_params
from tophat_amd import host
from tophat_amd.bamio import read_bam
from tophat_amd.batch import merge_events

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tophat_amd", "bin")


def _exe_juncs_db(case):
    fa = os.path.join(rr.GOLD, case, "genome.fa")

    def run(names, jf, inf, df, read_len, min_anchor):
        r = subprocess.run([os.path.join(BIN, "juncs_db"), str(min_anchor), str(read_len), jf, inf, df, "/dev/null", fa], capture_output=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.decode()
    return run


@pytest.mark.parametrize("case", rr.CASES)
def test_hip_path_reproduces_the_recorded_results(case, tmp_path):
    c = rr.load(case, tmp_path, _exe_juncs_db(case))
    seq = orc.fold_genome_char(c["genome"])
    og = orc.Genome([seq])
    with host.Context(0) as ctx:
        ctx.upload_genome(host.pack_genome([seq]))
        runs, want_ev, base = [], None, 0
        for sd, p in side_params(c):
            b = c["sides"][sd]["seg_batch"]
            runs.append((p, ctx.upload_batch(b, ordinal_base=base)))
            base += b.n_reads
            e = orc.segjuncs(p, og, b)
            want_ev = e if want_ev is None else merge_events(want_ev, e)
        ev = ctx.segjuncs(runs)
        assert [tuple(j) for j in ev.juncs] == [tuple(j) for j in want_ev.juncs]
        assert [tuple(j) for j in ev.deletions] == [tuple(j) for j in want_ev.deletions] and ev.insertions == want_ev.insertions
        got = sorted((int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ev.juncs)
        if case in rr.SPLICE_CASES:
            assert rr.SPLICE_CASES[case] in got
        if case in ("test_SimpleSplicing", "test_ReverseComplementSplicing"):
            assert got == c["recorded_juncs"] == [(1, 63, 138, 0)]
        ctx.upload_span_sets(c["span_juncs"], c["span_ins"])
        for sd, _p in side_params(c):
            sb = c["sides"][sd]["span_batch"]
            alns = ctx.spanning(c["p"], [ctx.upload_span_batch(sb)])
            assert rr.check_recorded_alignments(c, alns, sd) == RECORDED[case][sd]
            assert alns == orc.spanning(c["p"], og, sb, c["span_juncs"], c["span_ins"])


@pytest.mark.parametrize("case", rr.CASES)
def test_dropin_executables_reproduce_the_recorded_results(case, tmp_path):
    """the same through the three executables on files, the way tophat.py chains them: segment_juncs -> (recorded event
    lists) -> juncs_db -> long_spanning_reads with the junction-database maps"""
    c = rr.load(case, tmp_path, _exe_juncs_db(case))
    f = rr.write_program_inputs(c, tmp_path)
    p = c["p"]
    out = {k: str(tmp_path / ("out." + k)) for k in ("juncs", "insertions", "deletions", "fusions")}
    cmd = [os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", str(c["L"]),
           "--sam-header", f["hdr.sam"], "-p", "1"]
    if c["paired"]:
        cmd += ["--inner-dist-mean", str(p.inner_dist_mean), "--inner-dist-std-dev", str(p.inner_dist_std_dev)]
    cmd += [f["ref.fa"], out["juncs"], out["insertions"], out["deletions"], out["fusions"], f["left"]["reads"], f["left"]["map"], ",".join(f["left"]["segs"])]
    if c["paired"]:
        cmd += [f["right"]["reads"], f["right"]["map"], ",".join(f["right"]["segs"])]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    if case in ("test_SimpleSplicing", "test_ReverseComplementSplicing"):
        assert open(out["juncs"]).read() == open(c["files"]["juncs"]).read() == "fake\t63\t138\t+\n"
    if case in rr.SPLICE_CASES:
        j = rr.SPLICE_CASES[case]
        assert "fake\t%d\t%d\t+\n" % (j[1], j[2]) in open(out["juncs"]).read()
    # the text outputs equal the oracle's on the same inputs
    og = orc.Genome([orc.fold_genome_char(c["genome"])])
    want = None
    for sd, ps in side_params(c):
        e = orc.segjuncs(ps, og, c["sides"][sd]["seg_batch"])
        want = e if want is None else merge_events(want, e)
    from golden_util import events_text
    wt = events_text(want, c["names"], tmp_path)
    for k in ("juncs", "insertions", "deletions"):
        assert open(out[k]).read() == wt[k], k
    for sd, _ps in side_params(c):
        bam = str(tmp_path / ("span_%s.bam" % sd))
        r = subprocess.run([os.path.join(BIN, "long_spanning_reads"), "--segment-length", str(c["L"]), "--read-mismatches", str(p.read_mismatches),
                            "--read-gap-length", str(p.read_gap_length), "--read-edit-dist", str(p.read_edit_dist), "--sam-header", f["hdr.sam"],
                            f["ref.fa"], f[sd]["reads"], c["files"]["juncs"], c["files"]["insertions"], c["files"]["deletions"], "/dev/null", bam,
                            ",".join(f[sd]["segs"]), ",".join(c["sides"][sd]["spliced_sam"])], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        _, recs = read_bam(bam)
        ours = {}
        for rec in recs:                                  # (QNAME FLAG RNAME POS MAPQ CIGAR ... tags)
            nm = [int(str(x)[5:]) for x in rec[6:] if str(x).startswith("NM:i:")]
            ours.setdefault(int(rec[0]), set()).add((int(rec[1]) & 16, int(rec[3]), rec[5], nm[0]))
        exp = c["sides"][sd]["expected"]
        missing = [e for e in exp if e[1:] not in ours.get(e[0], set())]
        assert not missing, "%s: %d recorded alignments not in the BAM, e.g. %s" % (sd, len(missing), missing[0])
