#!/usr/bin/env python3
"""GPU box: what a process pays before and after its work -- HIP library load, device count, first allocation, stream, a 1 GiB allocation
and its release, and how long the process takes to exit afterwards (measured by a parent).  Several rounds: boxes differ a lot."""
import ctypes, os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    t0 = time.time(); h = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so"); t1 = time.time()
    n = ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); t2 = time.time()
    p = ctypes.c_void_p(); h.hipMalloc(ctypes.byref(p), 1 << 20); t3 = time.time()
    s = ctypes.c_void_p(); h.hipStreamCreate(ctypes.byref(s)); t4 = time.time()
    q = ctypes.c_void_p(); h.hipMalloc(ctypes.byref(q), ctypes.c_size_t(int(sys.argv[2]) << 20)); h.hipMemset(q, 0, ctypes.c_size_t(int(sys.argv[2]) << 20)); h.hipDeviceSynchronize(); t5 = time.time()
    r = ctypes.c_void_p(); h.hipHostMalloc(ctypes.byref(r), ctypes.c_size_t(int(sys.argv[3]) << 20), 0); t6 = time.time()
    print("load %.3f count %.3f first-malloc %.3f stream %.3f dev-alloc+memset(%s MiB) %.3f pinned(%s MiB) %.3f | now %.6f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, sys.argv[2], t5 - t4, sys.argv[3], t6 - t5, time.time()), flush=True)
    os._exit(0)
for dev_mb, pin_mb in ((1, 1), (2048, 1), (1, 512), (4096, 512)):
    for i in range(3):
        t = time.time()
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(dev_mb), str(pin_mb)], capture_output=True, text=True).stdout.strip()
        te = time.time()
        now = float(out.rsplit("now", 1)[1]) if "now" in out else te
        print("dev %5d MiB pin %4d MiB: total %.3f  exit %.3f  | %s" % (dev_mb, pin_mb, te - t, te - now, out.rsplit("|", 1)[0]))
