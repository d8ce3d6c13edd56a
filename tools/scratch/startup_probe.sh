cd tophat_amd/bin
for i in 1 2 3; do /usr/bin/time -f "usage-exit %e s" ./segment_juncs > /dev/null 2>/tmp/o; tail -1 /tmp/o; done
LD_DEBUG=statistics ./segment_juncs 2>&1 | grep -E "total startup|relocation|load" | head -8
ldd ./segment_juncs | wc -l
python3 - <<'PY'
import ctypes, time
t=time.time(); h=ctypes.CDLL("/opt/rocm/lib/libamdhip64.so"); t1=time.time()
n=ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); t2=time.time()
p=ctypes.c_void_p(); h.hipMalloc(ctypes.byref(p), 1<<20); t3=time.time()
print("dlopen hip %.3f  hipGetDeviceCount %.3f  first hipMalloc %.3f"%(t1-t,t2-t1,t3-t2))
PY
