"""Measured HBM copy rate of the box, to quote beside the 8 TB/s spec peak.  Device-to-device copies of a buffer
far larger than the 256 MB L3; bytes counted = read + written."""
import json
import sys

import torch


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
    dst = torch.empty_like(src)
    out = {}
    for name, fn in (("copy", lambda: dst.copy_(src)), ("read_only_sum_i64", lambda: src.view(torch.int64).sum())):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        moved = n * (2 if name == "copy" else 1)
        out[name] = {"ms": ms, "GB/s": moved / ms / 1e6}
    print(json.dumps({"buffer_bytes": n, "device": torch.cuda.get_device_name(0), **out}))


if __name__ == "__main__":
    main()
