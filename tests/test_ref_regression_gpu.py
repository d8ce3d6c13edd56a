"""GPU: the HIP path through the C ABI -- and the drop-in juncs_db executable for the junction database -- against
the reference's own regression cases (tests/golden_ref/, see tests/ref_regression.py), and record for record against
the oracle on the same inputs.  Reads of 24 bases in two 12-base segments: the smallest shapes the kernels take."""
import os
import subprocess

import pytest

import orc
import ref_regression as rr
from tophat_amd import host
from tophat_amd.bamio import read_bam

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
        ev = ctx.segjuncs([(c["p"], ctx.upload_batch(c["seg_batch"]))])
        want_ev = orc.segjuncs(c["p"], og, c["seg_batch"])
        assert [tuple(j) for j in ev.juncs] == [tuple(j) for j in want_ev.juncs]
        assert [tuple(j) for j in ev.deletions] == [tuple(j) for j in want_ev.deletions] and ev.insertions == want_ev.insertions
        if case == "test_SimpleSplicing":
            got = sorted((int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) for j in ev.juncs)
            assert got == c["recorded_juncs"] == [(1, 63, 138, 0)]
        ctx.upload_span_sets(c["span_juncs"], c["span_ins"])
        alns = ctx.spanning(c["p"], [ctx.upload_span_batch(c["span_batch"])])
    n, gapped = rr.check_recorded_alignments(c, alns)
    assert gapped == {"test_SimpleSplicing": 64, "test_SimpleIndel": 117, "test_IndelWithErrors": 227}[case]
    assert alns == orc.spanning(c["p"], og, c["span_batch"], c["span_juncs"], c["span_ins"])


@pytest.mark.parametrize("case", rr.CASES)
def test_dropin_executables_reproduce_the_recorded_results(case, tmp_path):
    """the same through the three executables on files, the way tophat.py chains them: segment_juncs -> (recorded event
    lists) -> juncs_db -> long_spanning_reads with the junction-database maps"""
    c = rr.load(case, tmp_path, _exe_juncs_db(case))
    f = rr.write_program_inputs(c, tmp_path)
    p = c["p"]
    out = {k: str(tmp_path / ("out." + k)) for k in ("juncs", "insertions", "deletions", "fusions")}
    r = subprocess.run([os.path.join(BIN, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", str(rr.SEG_LEN),
                        "--sam-header", f["hdr.sam"], "-p", "1", f["ref.fa"], out["juncs"], out["insertions"], out["deletions"], out["fusions"],
                        f["reads.fq"], f["left_map.sam"], ",".join(f["segs"])], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    if case == "test_SimpleSplicing":
        assert open(out["juncs"]).read() == open(c["files"]["juncs"]).read() == "fake\t63\t138\t+\n"
    bam = str(tmp_path / "span.bam")
    r = subprocess.run([os.path.join(BIN, "long_spanning_reads"), "--segment-length", str(rr.SEG_LEN), "--read-mismatches", str(p.read_mismatches),
                        "--read-gap-length", str(p.read_gap_length), "--read-edit-dist", str(p.read_edit_dist), "--sam-header", f["hdr.sam"],
                        f["ref.fa"], f["reads.fq"], c["files"]["juncs"], c["files"]["insertions"], c["files"]["deletions"], "/dev/null", bam,
                        ",".join(f["segs"]), ",".join(c["spliced_sam"])], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    _, recs = read_bam(bam)
    ours = {}
    for rec in recs:                                  # (QNAME FLAG RNAME POS MAPQ CIGAR ... tags)
        nm = [int(str(x)[5:]) for x in rec[6:] if str(x).startswith("NM:i:")]
        ours.setdefault(int(rec[0]), set()).add((int(rec[1]) & 16, int(rec[3]), rec[5], nm[0]))
    missing = [e for e in c["expected"] if e[1:] not in ours.get(e[0], set())]
    assert not missing, "%d recorded alignments not in the BAM, e.g. %s" % (len(missing), missing[0])
