#!/usr/bin/env python3
"""Throughput of thj_k_inflate on the BGZF members of a BAM file (device-resident input and output; HIP events around the
launch through the C ABI's on_device mode).  python tools/inflate_bench.py <file.bam> [repeat]"""
import ctypes as C
import os
import struct
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tophat_amd import host  # noqa: E402

if os.environ.get("THJ_LIB"):                         # developer A/B: another build of the library
    host.LIB_PATH = os.environ["THJ_LIB"]

path = sys.argv[1]
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
data = open(path, "rb").read()
offs, lens, isz, off = [], [], [], 0
while off < len(data):
    bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
    offs.append(off + 18); lens.append(bsize - 26); isz.append(struct.unpack_from("<I", data, off + bsize - 4)[0])
    off += bsize
if len(sys.argv) > 3:                                  # only the first N members (how the rate depends on the size of a launch)
    k = int(sys.argv[3]); offs, lens, isz = offs[:k], lens[:k], isz[:k]
n = len(offs)
blk = np.zeros(n, dtype=[("in_off", "<u8"), ("in_len", "<u4"), ("r", "<u4")])
blk["in_off"], blk["in_len"] = offs, lens
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
ctx = host.Context(0, stream=stream.cuda_stream)
d_comp = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
d_blk = torch.from_numpy(blk.view(np.uint8).copy()).to(dev)
d_out = torch.empty(n << 16, dtype=torch.uint8, device=dev)
d_len = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def run():
    rc = ctx.lib.thj_bgzf_inflate(ctx._ctx, C.c_void_p(d_comp.data_ptr()), C.c_int64(len(data)), C.c_void_p(d_blk.data_ptr()), C.c_int64(n),
                                  C.c_void_p(d_out.data_ptr()), C.c_void_p(d_len.data_ptr()), 1)
    assert rc == 0


run(); ctx.sync()
t = time.time()
for _ in range(rep):
    run()
ctx.sync()
dt = (time.time() - t) / rep
got = d_len.cpu().numpy().astype(np.uint32)
ok = bool((got == np.array(isz, dtype=np.uint32)).all())
print({"file": os.path.basename(path), "members": n, "compressed_MB": round(len(data) / 1e6, 1), "inflated_MB": round(sum(isz) / 1e6, 1),
       "ms": round(dt * 1e3, 2), "inflated_GBs": round(sum(isz) / dt / 1e9, 2), "lengths_ok": ok})
