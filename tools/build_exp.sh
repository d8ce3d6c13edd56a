#!/bin/bash
# Developer-only: the library with THJ_EXP switches compiled in (never loaded by the product path).
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DTHJ_EXP -Wno-unused-value -Wno-unused-result \
  -o tophat_amd/csrc/libthj_exp.so tophat_amd/csrc/thj_segjuncs.hip tophat_amd/csrc/thj_span.hip tophat_amd/csrc/thj_ingest.hip tophat_amd/csrc/thj_bamout.hip tophat_amd/csrc/thj_streams.hip tophat_amd/csrc/thj_pack.cpp
