#!/usr/bin/env python3
"""Mint the fixtures under tests/golden_ref/ from the regression test cases the reference itself ships
(/root/reference/tests/regression_tests/test_cases: test_SimpleSplicing, test_SimpleIndel, test_IndelWithErrors --
a 495-base genome, 24-base reads, `tophat --segment-length 12`, and the tophat_out/ the reference's authors recorded).
Only DATA is taken over, trimmed to the columns the tests use:

  genome.fa          the case's genome (common_genomes/fake.fa), verbatim
  reads.tsv          read name <tab> bases of input/*.fq, in file order (prep_reads numbers them 1..N in this order)
  junctions.bed / insertions.bed / deletions.bed     the recorded tophat_out files, verbatim
  accepted_hits.tsv  QNAME FLAG POS CIGAR NM of every record of the recorded accepted_hits.sam
  command.txt        the recorded command line

Run in the build container (the reference tree does not exist on the GPU box):  python tests/golden_ref/make_ref_regression.py
"""
import os
import shutil

SRC = "/root/reference/tests/regression_tests/test_cases"
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"test_SimpleSplicing": "fakeReads.fq", "test_SimpleIndel": "fakeReads.fq", "test_IndelWithErrors": "fakeReads_errors.fq"}

for case, fq in CASES.items():
    d = os.path.join(HERE, case)
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(os.path.join(SRC, "common_genomes", "fake.fa"), os.path.join(d, "genome.fa"))
    shutil.copyfile(os.path.join(SRC, case, "command.txt"), os.path.join(d, "command.txt"))
    for bed in ("junctions.bed", "insertions.bed", "deletions.bed"):
        shutil.copyfile(os.path.join(SRC, case, "tophat_out", bed), os.path.join(d, bed))
    lines = open(os.path.join(SRC, case, "input", fq)).read().split("\n")
    with open(os.path.join(d, "reads.tsv"), "w") as f:
        for i in range(0, len(lines) - 3, 4):
            assert lines[i].startswith("@") and lines[i + 2].startswith("+")
            f.write("%s\t%s\n" % (lines[i][1:].split()[0], lines[i + 1].strip()))
    with open(os.path.join(d, "accepted_hits.tsv"), "w") as f:
        for l in open(os.path.join(SRC, case, "tophat_out", "accepted_hits.sam")):
            if l.startswith("@"):
                continue
            t = l.rstrip("\n").split("\t")
            nm = [x[5:] for x in t[11:] if x.startswith("NM:i:")]
            f.write("\t".join([t[0], t[1], t[3], t[5], nm[0] if nm else "."]) + "\n")
    print(case, "->", d)
