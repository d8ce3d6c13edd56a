#!/usr/bin/env python3
"""Developer probe (GPU box): what a process pays AFTER its last instruction -- the driver taking its GPU memory apart -- by kind
and size of what it held.  Each case is a child that initialises HIP through libthj_hip.so's context, allocates, prints a stamp and
leaves with _exit; the parent measures stamp -> waitpid."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    dev_mb, pin_mb, n_dev = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    p = C.c_void_p()
    assert hip.hipSetDevice(0) == 0
    assert hip.hipMalloc(C.byref(p), 256) == 0
    keep = []
    for _ in range(n_dev):
        q = C.c_void_p()
        assert hip.hipMalloc(C.byref(q), C.c_size_t(dev_mb * (1 << 20) // max(1, n_dev))) == 0
        hip.hipMemset(q, 0, C.c_size_t(dev_mb * (1 << 20) // max(1, n_dev)))
        keep.append(q)
    if pin_mb:
        h = C.c_void_p()
        assert hip.hipHostMalloc(C.byref(h), C.c_size_t(pin_mb << 20), 0) == 0
        C.memset(h, 1, pin_mb << 20)
    hip.hipDeviceSynchronize()
    sys.stdout.write("%.6f\n" % time.time())
    sys.stdout.flush()
    os._exit(0)
for dev_mb, pin_mb, n_dev in ((0, 0, 0), (1024, 0, 1), (8192, 0, 1), (8192, 0, 64), (32768, 0, 4), (0, 1024, 0), (0, 4096, 0), (4096, 2048, 16)):
    best = None
    for _ in range(3):
        t0 = time.time()
        pr = subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(dev_mb), str(pin_mb), str(n_dev)], stdout=subprocess.PIPE, text=True)
        stamp = float(pr.stdout.readline())
        pr.wait()
        t1 = time.time()
        v = (stamp - t0, t1 - stamp)
        best = v if best is None or v[1] < best[1] else best
    print("device %6d MB in %2d blocks, pinned %5d MB: start -> stamp %.3f s, stamp -> gone %.3f s" % (dev_mb, n_dev, pin_mb, best[0], best[1]), flush=True)
