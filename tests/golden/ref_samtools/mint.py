#!/usr/bin/env python3
"""Mints tests/golden/ref_samtools/* with REFERENCE code: the reference's vendored samtools 0.1.18 (plain C + zlib), compiled where it
lies under /root/reference by oracle/ref_samtools.mk into oracle/_ref/samref (build container only; the tool never travels and no
test needs it).  For every spanning-output fixture under tests/golden/ (2x100, 2x76, 1x100, 1x150 bp; indels, many junctions,
multihits, fusions):

  <case>.span_<side>.samtools.bam   the records imported by sam_read1 and written by bam_write1 through bgzf.c at zlib's default level:
                                    the BAM byte stream row B8 (GBamRecord / GBamWriter) must equal, and BGZF members as tophat.py's
                                    callers hand them over -- fixed inputs for the inflate kernels
  <case>.span_<side>.calmd.tsv      QNAME FLAG RNAME POS MD NM of every record as bam_fillmd1_core (bam_md.c:23-131) recomputes them
                                    from the record's CIGAR / bases and the case's ref.fa -- what the reference's own regression
                                    harness checks its outputs with (`samtools calmd` must have nothing to correct,
                                    tests/regression_tests/regression_test.py:96-110): pins MD (row B6) and NM (row B5)

and MANIFEST.json (what was made from what, and two self-checks done here with the tool: `samref view` of samtools' BAM gives the
records back, and `samref view` of the BAM this build's BamWriter writes from the same records gives the same text).
Fusion alignments (two-record XF form) are left out of the calmd vectors: their second contig is not in the record.

    python tests/golden/ref_samtools/mint.py        (needs /root/reference; rewrites the directory)"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
GOLD = os.path.dirname(HERE)
TOOL = os.path.join(ROOT, "oracle", "_ref", "samref")
HOSTIO = os.path.join(ROOT, "tests", "hostio", "hostio_check")


def main():
    subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "oracle", "ref_samtools.mk")], cwd=ROOT)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostio")])
    if not os.path.exists(TOOL):
        sys.exit("oracle/_ref/samref was not built (is /root/reference present?)")
    manifest = {"made_by": "tests/golden/ref_samtools/mint.py with oracle/_ref/samref (reference samtools 0.1.18 sources, oracle/ref_samtools.mk)", "cases": {}}
    tmp = tempfile.mkdtemp(prefix="thj_mint_")
    try:
        for case in sorted(os.listdir(GOLD)):
            cdir = os.path.join(GOLD, case)
            for side in ("left", "right"):
                sam = os.path.join(cdir, "expected.span_%s.sam" % side)
                if not os.path.isfile(sam):
                    continue
                key = "%s.span_%s" % (case, side)
                hdr = open(os.path.join(cdir, "hdr.sam")).read()
                recs = [l.rstrip("\n").split("\t") for l in open(sam) if l.strip() and not l.startswith("@")]
                full = os.path.join(tmp, key + ".sam")
                with open(full, "w") as f:
                    f.write(hdr)
                    for c in recs:                 # the trimmed golden form lacks RNEXT PNEXT TLEN: "*", 0, 0 as print_bamhit writes them (bwt_map.cpp:1932-1934)
                        f.write("\t".join(c[:6] + ["*", "0", "0"] + c[6:]) + "\n")
                bam = os.path.join(HERE, key + ".samtools.bam")
                subprocess.check_call([TOOL, "sam2bam", full, bam])
                view = subprocess.check_output([TOOL, "view", bam]).decode()
                want_view = "".join("\t".join(c[:6] + ["*", "0", "0"] + c[6:]) + "\n" for c in recs)
                assert view == want_view, key
                fa = os.path.join(tmp, case + ".fa")
                if not os.path.exists(fa):
                    shutil.copy(os.path.join(cdir, "ref.fa"), fa)
                calmd = subprocess.run([TOOL, "calmd", bam, fa], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
                rows = []
                for c, l in zip(recs, calmd.splitlines()):
                    q, fl, rn, pos, md, nm, _omd, _onm = l.split("\t")
                    assert (q, fl, rn, pos) == (c[0], c[1], c[2], c[3])
                    if any(t.startswith("XF:Z:") for t in c[8:]):
                        continue
                    rows.append("\t".join((q, fl, rn, pos, md, nm)))
                open(os.path.join(HERE, key + ".calmd.tsv"), "w").write("\n".join(rows) + "\n")
                # this build's writer on the same records, read back by the reference's reader
                ours = os.path.join(tmp, key + ".ours.bam")
                subprocess.check_call([HOSTIO, "sam2bam", os.path.join(cdir, "hdr.sam"), sam, ours])
                ours_view = subprocess.check_output([TOOL, "view", ours]).decode()
                manifest["cases"][key] = {"records": len(recs), "calmd_rows": len(rows), "samtools_bam_sha256": hashlib.sha256(open(bam, "rb").read()).hexdigest(),
                                          "samtools_reads_this_builds_bam_to_the_same_text": ours_view == want_view}
                assert ours_view == want_view, key
        # members of full size for the inflate kernels: a generated whole-read map (tools/bin/thj_gen --text) through the same writer
        gen = os.path.join(ROOT, "tools", "bin", "thj_gen")
        gdir = os.path.join(tmp, "gen")
        subprocess.check_call([gen, "--out", gdir, "--pairs", "2500", "--genome-len", "2000000", "--introns", "300", "--text", "--threads", "2"], stdout=subprocess.DEVNULL)
        bam = os.path.join(HERE, "generated_left_map.samtools.bam")
        subprocess.check_call([TOOL, "sam2bam", os.path.join(gdir, "left_map.sam"), bam])
        manifest["bgzf_members"] = {"file": "generated_left_map.samtools.bam", "bytes": os.path.getsize(bam), "sha256": hashlib.sha256(open(bam, "rb").read()).hexdigest(),
                                    "from": "tools/bin/thj_gen --pairs 2500 --genome-len 2000000 --introns 300 --text: left_map.sam"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    json.dump(manifest, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print("minted %d fixtures" % len(manifest["cases"]))


if __name__ == "__main__":
    main()
