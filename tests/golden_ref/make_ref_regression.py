#!/usr/bin/env python3
"""Mint the fixtures under tests/golden_ref/ from the regression test cases the reference itself ships
(/root/reference/tests/regression_tests/test_cases, all nine of them: a 495-base genome, 24-base reads -- single-end,
paired-end, reverse-complemented, lower-case -- `tophat --segment-length 12` (or 8: test_3Segment), and the tophat_out/ the
reference's authors recorded).
Only DATA is taken over, trimmed to the columns the tests use:

  genome.fa          the case's genome (common_genomes/fake.fa), verbatim
  reads.tsv          read name <tab> bases of input/*.fq, in file order (prep_reads numbers them 1..N in this order);
                     reads_right.tsv = the second mates of a paired-end case
  input_insertions.bed   the insertions file test_Indel_1 passes with --insertions
  junctions.bed / insertions.bed / deletions.bed     the recorded tophat_out files, verbatim
  accepted_hits.tsv  QNAME FLAG POS CIGAR NM XS of every record of the recorded accepted_hits.sam
  command.txt        the recorded command line

Run in the build container (the reference tree does not exist on the GPU box):  python tests/golden_ref/make_ref_regression.py
"""
import os
import shutil

SRC = "/root/reference/tests/regression_tests/test_cases"
HERE = os.path.dirname(os.path.abspath(__file__))
# case -> read files (one: single-end; two: the mates of a paired-end run, ids shared by file order)
CASES = {"test_SimpleSplicing": ["fakeReads.fq"], "test_SimpleIndel": ["fakeReads.fq"], "test_IndelWithErrors": ["fakeReads_errors.fq"],
         "test_Paired": ["readOne.fq", "readTwo.fq"], "test_3Segment": ["readOne.fq", "readTwo.fq"],
         "test_ReverseComplementSplicing": ["fakeReads_rc.fq"], "test_ReverseComplementIndel": ["fakeReads_rc.fq"],
         "test_IndelLowerCase": ["fakeReads.fq"], "test_Indel_1": ["fakeReads_read28.fq"]}

for case, fqs in CASES.items():
    d = os.path.join(HERE, case)
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(os.path.join(SRC, "common_genomes", "fake.fa"), os.path.join(d, "genome.fa"))
    shutil.copyfile(os.path.join(SRC, case, "command.txt"), os.path.join(d, "command.txt"))
    for bed in ("junctions.bed", "insertions.bed", "deletions.bed"):
        shutil.copyfile(os.path.join(SRC, case, "tophat_out", bed), os.path.join(d, bed))
    for k, fq in enumerate(fqs):
        lines = open(os.path.join(SRC, case, "input", fq)).read().split("\n")
        with open(os.path.join(d, "reads.tsv" if k == 0 else "reads_right.tsv"), "w") as f:
            for i in range(0, len(lines) - 3, 4):
                assert lines[i].startswith("@") and lines[i + 2].startswith("+")
                f.write("%s\t%s\n" % (lines[i][1:].split()[0], lines[i + 1].strip()))
    if os.path.exists(os.path.join(SRC, case, "input", "insertions.bed")):      # tophat --insertions <file> (test_Indel_1)
        shutil.copyfile(os.path.join(SRC, case, "input", "insertions.bed"), os.path.join(d, "input_insertions.bed"))
    with open(os.path.join(d, "accepted_hits.tsv"), "w") as f:
        for l in open(os.path.join(SRC, case, "tophat_out", "accepted_hits.sam")):
            if l.startswith("@"):
                continue
            t = l.rstrip("\n").split("\t")
            nm = [x[5:] for x in t[11:] if x.startswith("NM:i:")]
            xs = [x[5:] for x in t[11:] if x.startswith("XS:A:")]
            f.write("\t".join([t[0], t[1], t[3], t[5], nm[0] if nm else ".", xs[0] if xs else "."]) + "\n")
    print(case, "->", d)
