"""CPU: the C-ABI library loads and exports every symbol include/thj.h declares; host packers work."""
import re
import os

import numpy as np

import orc
from tophat_amd import host
from tophat_amd.params import Params, CParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = host.load_lib()
    hdr = open(os.path.join(ROOT, "include", "thj.h")).read()
    declared = set(re.findall(r"\b(thj_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libthj_hip.so does not export %s" % sym
    assert set(host.ABI_SYMBOLS) <= declared


def test_params_default_match_reference_defaults():
    lib = host.load_lib()
    c = CParams()
    lib.thj_params_default(host.C.byref(c))
    d = Params().as_ctypes()
    for name, _ in CParams._fields_:
        assert getattr(c, name) == getattr(d, name), name


def test_genome_and_read_packing_roundtrip():
    lib = host.load_lib()
    rng = np.random.default_rng(1)
    seqs = ["".join(rng.choice(list("ACGTN"), size=n, p=[.24, .24, .24, .24, .04])) for n in (1, 63, 64, 65, 1000)]
    seqs.insert(2, None)
    g = host.pack_genome(seqs, lib=lib)
    blk = g.blocks.reshape(-1, 4)
    for ci, s in enumerate(seqs):
        if s is None:
            assert g.lens[ci] == 0
            continue
        b0 = int(g.contig_blk[ci])
        for i, ch in enumerate(s):
            w = blk[b0 + i // 64]
            lo, hi, nm = (int(w[0]) >> (i % 64)) & 1, (int(w[1]) >> (i % 64)) & 1, (int(w[2]) >> (i % 64)) & 1
            got = "N" if nm else "ACGT"[lo | (hi << 1)]
            assert got == ch
        # guard block after the contig is zero
        assert not blk[b0 + (len(s) + 63) // 64].any()


def test_no_device_gives_loud_error():
    import pytest
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    with pytest.raises(host.ThjError):
        host.Context(0)


def test_genome_layout_refuses_what_the_packed_keys_cannot_hold():
    """junction / insertion keys keep the global base coordinate in 34 bits: a layout beyond 2^34 bases is refused, not aliased"""
    import ctypes as C
    import numpy as np
    from tophat_amd import host
    lib = host.load_lib()
    n = 9
    lens = np.full(n, 0x7fffffff, dtype=np.int64)          # 9 x 2^31 > 2^34
    blk = np.zeros(n + 1, dtype=np.uint32)
    nb = C.c_int64()
    assert lib.thj_genome_layout(n, lens.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p), C.byref(nb)) == -1      # THJ_EINVAL
    assert b"2^34" in lib.thj_last_error()
    lens = np.full(7, 0x7fffffff, dtype=np.int64)          # 7 x 2^31 < 2^34: fine
    blk = np.zeros(8, dtype=np.uint32)
    assert lib.thj_genome_layout(7, lens.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0


def test_executables_pass_errors_through_the_output_handoff():
    """The executables run their work in a child process and return when it reports its outputs complete (run_with_handoff); a
    child that ends without reporting -- usage errors, die() -- is waited for, its exit code and messages passed on."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    bindir = os.path.join(ROOT, "tophat_amd", "bin")
    for exe in ("segment_juncs", "long_spanning_reads", "thj_junctions"):
        for env in (dict(os.environ), dict(os.environ, THJ_HANDOFF="1")):
            r = subprocess.run([os.path.join(bindir, exe)], capture_output=True, text=True, env=env, timeout=60)
            assert r.returncode == 1 and "sage" in r.stderr, (exe, r.returncode, r.stderr[-200:])
    # a die() after option parsing: inputs that do not exist
    args = ["--no-coverage-search", "--no-microexon-search", "/nonexistent/ref.fa", "/tmp/j", "/tmp/i", "/tmp/d", "/tmp/f",
            "/nonexistent/reads.bam", "/nonexistent/map.bam", "/nonexistent/seg1.bam,/nonexistent/seg2.bam"]
    r = subprocess.run([os.path.join(bindir, "segment_juncs")] + args, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rror" in r.stderr, (r.returncode, r.stderr[-300:])
