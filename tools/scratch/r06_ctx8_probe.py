"""Developer probe (GPU box): segment_juncs with 8 contexts on one GPU against 1, its [timing] lines.  python tools/scratch/r06_ctx8_probe.py [pairs]"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from e2e_bench import run_e2e, mix_gen_args
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
d = tempfile.mkdtemp(prefix="thj_c8_", dir="/dev/shm")
try:
    for per in (1, 8, 4, 2):
        r = run_e2e(pairs, 100, 64444167, 20000, workdir=d, keep=True, env_extra={"THJ_CTX_PER_GPU": str(per), "THJ_TRACE_XCHG": "1"}, gen_args=mix_gen_args(0.05, 41, 0.03))
        print("=== contexts", per, "segment_juncs", r["segment_juncs_s"], "lsr", r["long_spanning_reads_left_s"], r["long_spanning_reads_right_s"])
        for l in r["segment_juncs_log_tail"]:
            print("   ", l)
finally:
    shutil.rmtree(d, ignore_errors=True)
