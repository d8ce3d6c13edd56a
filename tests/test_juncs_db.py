"""juncs_db: the junction-database FASTA.  CPU: the oracle's restatement against the fixtures
(outputs of the survey-stage scratch build of the reference, see oracle/README.md).  GPU: the drop-in executable
(device gather from the bit-plane genome) against the same files, byte for byte."""
import os
import subprocess

import pytest

import orc
from golden_util import CASES, GOLD, parse_header, read_fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tophat_amd", "bin")


def _files(d):
    fus = os.path.join(d, "expected.fusions")
    return (os.path.join(d, "expected.juncs"), os.path.join(d, "expected.insertions"), os.path.join(d, "expected.deletions"),
            fus if os.path.exists(fus) else "/dev/null")


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_juncs_db_fixture(name):
    d = os.path.join(GOLD, name)
    names, seqs = read_fasta(os.path.join(d, "ref.fa"))          # juncs_db takes its ids from the FASTA order
    g = orc.Genome([orc.fold_genome_char(s) for s in seqs])
    seglen = int(dict(x.split("=") for x in open(os.path.join(d, "options.txt")).read().split("\n")[1].split())["segment_length"])
    j, i, dl, f = _files(d)
    got = orc.juncs_db_text(names, g, j, i, dl, f, seglen, 8)
    want = open(os.path.join(d, "expected.juncs_db.fa")).read()
    assert want.count(">") > 10
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_juncs_db_executable_reproduces_fixture(name, tmp_path):
    d = os.path.join(GOLD, name)
    seglen = dict(x.split("=") for x in open(os.path.join(d, "options.txt")).read().split("\n")[1].split())["segment_length"]
    j, i, dl, f = _files(d)
    r = subprocess.run([os.path.join(BIN, "juncs_db"), "8", seglen, j, i, dl, f, os.path.join(d, "ref.fa")], capture_output=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == open(os.path.join(d, "expected.juncs_db.fa"), "rb").read()


@pytest.mark.gpu
def test_juncs_db_edge_records(tmp_path):
    """junctions / insertions at contig starts and ends (clamped windows, records the reference drops), unknown contigs,
    several list files, an insertion with an ambiguity code; executable == oracle"""
    d = os.path.join(GOLD, "se100")
    names, seqs = read_fasta(os.path.join(d, "ref.fa"))
    n = len(seqs[0])
    c = names[0]
    jf, jf2, inf, df, ff = (str(tmp_path / x) for x in ("a.juncs", "b.juncs", "x.ins", "x.dels", "x.fus"))
    open(jf, "w").write("%s\t3\t200\t+\n%s\t%d\t%d\t-\n%s\t100\t%d\t+\nnope\t5\t9\t+\n" % (c, c, n - 30, n - 2, c, n + 5))
    open(jf2, "w").write("%s\t3\t200\t+\n%s\t3\t200\t-\n%s\t50\t%d\t+\n" % (c, c, c, n))
    open(inf, "w").write("%s\t2\t2\tAC\n%s\t2\t2\tGT\n%s\t%d\t%d\tTTT\n%s\t40\t40\tANA\n" % (c, c, c, n - 3, n - 3, c))
    open(df, "w").write("%s\t11\t14\n%s\t%d\t%d\n" % (c, c, n - 5, n - 1))
    open(ff, "w").write("%s\t5\t%s\t%d\tff\n%s\t5\t%s\t%d\trr\n%s\t%d\t%s\t7\tfr\n%s\t%d\t%s\t7\trf\n" % (
        c, c, n - 3, c, c, n - 3, c, n - 2, c, c, n - 2, c))
    g = orc.Genome([orc.fold_genome_char(s) for s in seqs])
    for f1 in (jf, jf2, inf, df, ff):
        txt = open(f1).read().replace("\\\t", "\t").replace("\\\n", "\n")
        open(f1, "w").write(txt)
    want = None
    # the oracle binding reads one file per kind: concatenate the two junction lists the way the executable's sets merge them
    both = str(tmp_path / "both.juncs")
    open(both, "w").write(open(jf).read() + open(jf2).read())
    names2 = names + ["nope"]
    g2 = orc.Genome([orc.fold_genome_char(s) for s in seqs] + [None])
    want = orc.juncs_db_text(names2, g2, both, inf, df, ff, 25, 8)
    r = subprocess.run([os.path.join(BIN, "juncs_db"), "8", "25", jf + "," + jf2, inf, df, ff, os.path.join(d, "ref.fa")], capture_output=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.decode() == want and want.count(">") >= 8
