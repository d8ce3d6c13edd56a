"""The files-to-files CPU figure of bench.py's cpu_baseline leg (tools/cpuport): the executables' own host sources over the CPU
oracle.  Test infrastructure checking bench infrastructure: on a small generated case its event files and every spanning record
must be what the oracle gives through the Python host mirror, whatever the number of oracle threads."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_BIN = os.path.join(ROOT, "tools", "bin", "cpuport")


@pytest.fixture(scope="module")
def cpuport():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from locked_make import locked_make
    locked_make(os.path.join(ROOT, "oracle"))
    gen = os.path.join(ROOT, "tools", "bin", "thj_gen")
    if not os.path.exists(gen):
        hostdir = os.path.join(ROOT, "tophat_amd", "csrc", "host")
        os.makedirs(os.path.dirname(gen), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "thj_gen.cpp"),
                               os.path.join(ROOT, "tophat_amd", "csrc", "thj_pack.cpp"), "-o", gen, "-lz", "-I" + hostdir])
    locked_make(os.path.join(ROOT, "tools", "cpuport"))
    return CPU_BIN


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


@pytest.mark.parametrize("threads", [1, 3])
def test_cpu_port_files_equal_the_oracle(cpuport, tmp_path, threads):
    sys.path.insert(0, ROOT)
    import bench
    env = {"THJ_HOST_INGEST": "1", "THJ_CTX_PER_GPU": "1", "THJ_NO_HANDOFF": "1", "THJ_CPUPORT_THREADS": str(threads)}
    res = bench.e2e_sample_check(str(tmp_path), None, pairs=3000, bind=cpuport, env=env)
    assert res["junctions"] > 0 and res["spanning_records"] > 3000
    assert res["event_files_identical_to_oracle"]
    assert res["spanning_records_identical_to_oracle"]


def test_cpu_port_declines_what_the_figure_does_not_cover(cpuport, tmp_path):
    """--fusion-search is outside the figure: the port says so instead of writing something"""
    gen = os.path.join(ROOT, "tools", "bin", "thj_gen")
    d = str(tmp_path)
    subprocess.check_call([gen, "--out", d, "--pairs", "200", "--genome-len", "4000000", "--introns", "300"], stdout=subprocess.DEVNULL)
    f = lambda n: os.path.join(d, n)      # noqa: E731
    segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k)) for k in (1, 2, 3, 4)) for sd in ("left", "right")}
    env = dict(os.environ, THJ_HOST_INGEST="1", THJ_CTX_PER_GPU="1", THJ_NO_HANDOFF="1")
    r = subprocess.run([os.path.join(cpuport, "segment_juncs"), "--fusion-search", "--no-coverage-search", "--no-microexon-search", "--segment-length", "25",
                        "--sam-header", f("hdr.sam"), f("ref.fa"), f("o.j"), f("o.i"), f("o.d"), f("o.f"), f("left_reads.bam"), f("left_map.bam"),
                        segs["left"], f("right_reads.bam"), f("right_map.bam"), segs["right"]], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "cpu port" in r.stderr
