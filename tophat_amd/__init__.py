"""tophat_amd -- MI355X-native splice-junction discovery hot path of TopHat.

Only what the path needs lives here:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (libthj_hip.so)
  host.py    ctypes binding of include/thj.h (device contexts, batches, events)
  batch.py   host-side batch model: hits_for_read groups in visiting order
  params.py  the option surface of common.cpp that changes hot-path results
  samtext.py minimal SAM-text reader/writer used by tests and fixtures
  synth.py   seeded synthetic genome / reads / segment-hit generator

The product path fails loudly when libthj_hip.so is missing; nothing here
falls back to the CPU oracle under oracle/ (that is test infrastructure).
"""
from .params import Params  # noqa: F401
