#!/usr/bin/env python3
"""bench.py -- read-pairs/s through the MI355X junction-discovery hot path.

    python bench.py --gpus N --steps K --warmup W [--pairs P]

A "step" is one pass of the hot path over one batch of synthetic paired reads
whose inputs (packed genome, hit records, read planes) are already resident in
HBM: reset the event tables, run the left-side batch, run the right-side
batch, compact + sort the event sets (= one `segment_juncs` invocation on a
batch).  N>1: one process per GPU (torch.distributed, backend nccl = RCCL), the
genome replicated, read pairs sharded (weak scaling: every rank gets `--pairs`
pairs), and one all-gather of the per-rank junction/deletion key sets over
xGMI, merged into every rank's table (segment_juncs.cpp:4911-4916 across GPUs).

Prints ONE JSON line (rank 0).  `value` = pairs processed by all ranks / time.
`roofline` = algorithmic bytes of the dominant kernel (thj_k_segjuncs) per
launch / its average duration from HIP events recorded on the launch stream,
against the 8 TB/s HBM peak.  `cpu_baseline` = the plain-C oracle (a port, 1
core) timed on a bounded sample of the same workload on this box's host.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tophat_amd import host  # noqa: E402
from tophat_amd.params import Params, READ_LEFT, READ_RIGHT  # noqa: E402
from tophat_amd.synth import make_device_workload, make_scale_genome  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
CHR20_LEN = 64_444_167         # GRCh38 chr20 length (BASELINE.json configs[1])
# GRCh38 primary assembly, chr1..22, X, Y, M (sum 3 088 286 401 bp)
GRCH38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
               135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
               46709983, 50818468, 156040895, 57227415, 16569]


def cbatch_from_tensors(w, ordinal_base=0) -> host.CSegBatch:
    cb = host.CSegBatch()
    cb.n_reads, cb.nseg, cb.words_per_plane = w["n_reads"], w["nseg"], w["W"]
    cb.seg_off = w["seg_off"].data_ptr()
    cb.hits = w["hits"].data_ptr()
    cb.read_planes = w["planes"].data_ptr()
    cb.read_len = w["read_len"].data_ptr()
    cb.mate_off = w["mate_off"].data_ptr()
    cb.mate_hits = w["mate_hits"].data_ptr()
    cb.ordinal_base = ordinal_base
    return cb


def sample_segbatch(w, m, start=0):
    """Reads [start, start + m) of a device workload as a host SegBatch (ASCII reads) for the oracle."""
    from tophat_amd.batch import HIT_DTYPE, SegBatch
    nseg, W = w["nseg"], w["W"]
    start = min(start, w["n_reads"])
    m = min(m, w["n_reads"] - start)
    so = w["seg_off"][start * nseg:(start + m) * nseg + 1].cpu().numpy().astype(np.int64)
    hits = w["hits"][int(so[0]):int(so[-1])].cpu().numpy().astype(np.int32)
    so = (so - so[0]).astype(np.uint32)
    mo = w["mate_off"][start:start + m + 1].cpu().numpy().astype(np.int64)
    mh = w["mate_hits"][int(mo[0]):int(mo[-1])].cpu().numpy().astype(np.int32)
    mo = (mo - mo[0]).astype(np.uint32)
    pl = w["planes"][start * 3 * W:(start + m) * 3 * W].cpu().numpy().view(np.uint64).reshape(m, 3, W)
    rl = w["read_len"][start:start + m].cpu().numpy().astype(np.int64)
    L = int(rl.max()) if m else 0
    bits = np.arange(64, dtype=np.uint64)
    lo = ((pl[:, 0, :, None] >> bits) & 1).reshape(m, -1)[:, :L]
    hi = ((pl[:, 1, :, None] >> bits) & 1).reshape(m, -1)[:, :L]
    nm = ((pl[:, 2, :, None] >> bits) & 1).reshape(m, -1)[:, :L]
    codes = (lo + 2 * hi).astype(np.uint8)
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    asc[nm == 1] = ord("N")
    read_off = np.zeros(m + 1, dtype=np.int64)
    read_off[1:] = np.cumsum(rl)
    if m and (rl == L).all():
        bases = np.ascontiguousarray(asc).reshape(-1)
    else:
        bases = np.concatenate([asc[i, :rl[i]] for i in range(m)]) if m else np.zeros(0, dtype=np.uint8)
    return SegBatch(nseg, np.arange(1, m + 1, dtype=np.uint32), read_off, bases, so,
                    np.ascontiguousarray(hits).view(HIT_DTYPE).reshape(-1),
                    mo, np.ascontiguousarray(mh).view(HIT_DTYPE).reshape(-1))


def sample_spanbatch(w, m, start=0):
    """Reads [start, start + m) of a device workload as a host SpanBatch for the oracle."""
    from tophat_amd.batch import SPAN_HIT_DTYPE, SpanBatch
    sb = sample_segbatch(w, m, start)
    nseg = w["nseg"]
    start = min(start, w["n_reads"])
    m = sb.n_reads
    so = w["span_off"][start * nseg:(start + m) * nseg + 1].cpu().numpy().astype(np.int64)
    hits = w["span_hits"][int(so[0]):int(so[-1])].cpu().numpy().astype(np.int32)
    so = (so - so[0]).astype(np.uint32)
    st = w["qual_stride"]
    q = w["quals"][start * st:(start + m) * st].cpu().numpy().reshape(m, st)
    rl = np.diff(sb.read_off)
    L = int(rl.max()) if m else 0
    quals = np.ascontiguousarray(q[:, :L]).reshape(-1) if m and (rl == L).all() else \
        np.concatenate([q[i, :rl[i]] for i in range(m)])
    return SpanBatch(nseg, sb.read_id, sb.read_off, sb.bases, quals, so,
                     np.ascontiguousarray(hits).view(SPAN_HIT_DTYPE).reshape(-1))


def span_cbatch_from_tensors(w, ctx=None) -> host.CSpanBatch:
    """ctx given: also builds the batch's optional dense hit-head array (thj_span_batch.hit_heads), kept in w["span_heads"]"""
    cb = host.CSpanBatch()
    cb.n_reads, cb.nseg, cb.words_per_plane, cb.qual_stride = w["n_reads"], w["nseg"], w["W"], w["qual_stride"]
    cb.seg_off = w["span_off"].data_ptr()
    cb.hits = w["span_hits"].data_ptr()
    cb.read_planes = w["planes"].data_ptr()
    cb.read_len = w["read_len"].data_ptr()
    cb.quals = w["quals"].data_ptr()
    if ctx is not None:
        n_hits = int(w["span_off"][-1])
        w["span_heads"] = torch.empty(max(1, n_hits) * 2, dtype=torch.int64, device=w["span_hits"].device)
        torch.cuda.synchronize()
        ctx.span_hit_heads(cb.hits, n_hits, w["span_heads"].data_ptr())
        ctx.sync()
        cb.hit_heads = w["span_heads"].data_ptr()
    return cb


class ProcControl:
    """control plane of a one-process-per-GPU run: torch.distributed over gloo (TCP on 127.0.0.1) carries the RCCL unique id,
    the barriers around the timed region and the max-over-ranks of the elapsed time.  The DATA path -- the exchange step --
    is ncclAllGather called inside libthj_hip.so on each rank's own stream (tophat_amd/csrc/thj_exchange_impl.h)."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("THJ_BENCH_JOIN_TIMEOUT", "60"))))
            dist.barrier()
        except Exception as e:      # noqa: BLE001  (a peer that never came up: say who is waiting, and where)
            print("[bench] rank %d of %d could not join the job within %s s at %s:%s (%s)" % (
                rank, world, os.environ.get("THJ_BENCH_JOIN_TIMEOUT", "60"), os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], repr(e)[:200]), file=sys.stderr, flush=True)
            raise SystemExit(4)

    def bcast_bytes(self, b, n):
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == 0:
            t = torch.tensor(list(b), dtype=torch.uint8)
        self.dist.broadcast(t, 0)
        return bytes(t.tolist())

    def barrier(self):
        self.dist.barrier()

    def max(self, x):
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather_ints(self, vals):
        t = torch.tensor(vals, dtype=torch.int64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.tolist() for o in out]

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


class ThreadControl:
    """the same, for N ranks living in ONE process as threads (functional test of the N > 1 code path on a box with one GPU:
    the ranks share the device and the exchange step uses the library's loopback transport -- not a measurement)"""

    def __init__(self, world):
        import threading
        self.world = world
        self._bar = threading.Barrier(world)
        self._slots = [None] * world

    def view(self, rank):
        v = ThreadControl.__new__(ThreadControl)
        v.__dict__ = dict(self.__dict__, rank=rank)
        return v

    def barrier(self):
        self._bar.wait()

    def max(self, x):
        return max(self._exchange(x))

    def allgather_ints(self, vals):
        return self._exchange(list(vals))

    def _exchange(self, x):
        self._slots[self.rank] = x
        self._bar.wait()
        out = list(self._slots)
        self._bar.wait()
        return out

    def close(self):
        pass


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node to use, one rank (process) each.  Started plainly with N > 1 "
                    "this script spawns its N ranks itself; under torchrun (WORLD_SIZE set) it is one of them")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=[2, 3], default=2,
                    help="2 (default): BASELINE configs[1] -- 10 M pairs per GPU against the chr20-sized genome, weak scaling.  "
                         "3: BASELINE configs[2] -- 100 M pairs in all, sharded over the job's GPUs (12.5 M per GPU at --gpus 8), against "
                         "the 25-contig GRCh38-sized genome: strong scaling over a fixed total.  --pairs overrides the per-GPU count")
    ap.add_argument("--pairs", type=int, default=None, help="read pairs per GPU (default: 10 M for --config 2, 100 M / GPUs for --config 3)")
    ap.add_argument("--genome-len", type=int, default=CHR20_LEN)
    ap.add_argument("--genome", choices=["chr20", "grch38"], default="chr20",
                    help="chr20: one contig of --genome-len bases (configs[1]); grch38: 25 contigs with the GRCh38 primary-assembly "
                         "lengths, 3.09 Gb (the genome of configs[2]; pass --introns 300000)")
    ap.add_argument("--introns", type=int, default=20000)
    ap.add_argument("--intron-max", type=int, default=200000, help="longest planted intron (configs[4]: 499999 with --max-intron 500000)")
    ap.add_argument("--exon-len", type=int, default=300)
    ap.add_argument("--read-len", type=int, default=100, help="read length (BASELINE configs[1]: 100; 150 / 50 are the shapes of configs[3] / [4])")
    ap.add_argument("--coverage-search", type=float, default=0.0, metavar="FRAC",
                    help="run segment_juncs' coverage search too (what tophat does for reads of fewer than three segments, "
                         "e.g. --read-len 50), with the first FRAC of each side's reads playing the initially unmapped reads")
    ap.add_argument("--plain", action="store_true",
                    help="the workload of rounds 1-2: no multihits, no indel reads (--multihit-frac 0 --indel-frac 0)")
    ap.add_argument("--multihit-frac", type=float, default=0.05,
                    help="fraction of the pairs drawn from a planted repeat family (SURVEY 8d: multihits from planted repeats up to 41): every "
                         "segment hit of such a read is reported at 2..41 copies (see --max-copies); 0 = none")
    ap.add_argument("--max-copies", type=int, default=41,
                    help="copies of the repeat family in the genome (41: 85 %% of the family's reads have 2 hits per segment, 12 %% 3..8, 2.7 %% "
                         "9..40, 0.3 %% 41 -- dropped whole by max_seg_multihits); 2: the two-copy genome of round 2 (second half = first half)")
    ap.add_argument("--fusion-search", action="store_true",
                    help="run long_spanning_reads' stage with fusion search on (the shape of configs[3]): reads tiers 0 / 1 cannot join go "
                         "through the fusion branches (thj_k_stitch_fusion) against an empty fusion list")
    ap.add_argument("--fusion-frac", type=float, default=0.0, metavar="FRAC",
                    help="with --fusion-search: this fraction of the pairs gets a chimeric left read (configs[3]: 0.02); the step then "
                         "also runs segment_juncs' fusion kernel and hands its fusions to the spanning stage")
    ap.add_argument("--indel-frac", type=float, default=0.03, metavar="FRAC",
                    help="this fraction of the pairs gets a left read with a 1..3-base deletion near the end of a segment (found by the indel "
                         "search of stage 1, closed with a D op in stage 2)")
    ap.add_argument("--no-hit-heads", action="store_true",
                    help="hand stage 2 the 32-byte hit records only, without the dense 16-byte head array every batch of the library carries "
                         "(thj_span_batch.hit_heads: derived once when a batch is made -- thj_span_batch_upload, the device-side ingest -- so "
                         "part of the resident layout, not of a step)")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="reads per side timed through the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run two steps under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for roofline.traffic "
                    "(the committed table under profiles/ is used instead when the configuration is the one it was taken on)")
    ap.add_argument("--e2e-grch38-pairs", type=int, default=12_500_000, metavar="N",
                    help="run the e2e leg's executables once more on configs[2]'s genome (25 contigs, 3.09 Gb) with N pairs -- its per-GPU shard at "
                         "8 GPUs -- twice: the first run packs the reference and leaves the packed-genome cache, the second maps it "
                         "(`e2e.grch38`, timing only; 0: skip)")
    ap.add_argument("--e2e-config3-pairs", type=int, default=0, metavar="N",
                    help="also run configs[2] at full size through the executables (N = 100000000: 31 GB of BAM against the GRCh38-sized genome; eight contexts "
                         "on this GPU with the exchange step, then one context, outputs compared; the first 20 000 pairs against the oracle): `e2e.config3_full`.  "
                         "Takes ~10 minutes and 45 GB of /dev/shm; off by default, its record is profiles/r06_config3_full.json")
    ap.add_argument("--e2e-pairs-large", type=int, default=40_000_000, metavar="N",
                    help="run the e2e leg a second time on N pairs (timing only; e.g. 40000000) so that the fixed cost of the three processes "
                         "and their rate separate: `e2e.large`, `e2e.fixed_s` and `e2e.rate_pairs_per_s` from the two points")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), metavar="PATH",
                    help="where the whole result goes (per-kernel table, e2e stage timings, checks); stdout carries the contract's line only")
    ap.add_argument("--e2e-pairs", type=int, default=10_000_000,
                    help="also run the drop-in executables end to end (files in, files out: the metric as SURVEY 8d words it) on "
                         "generated files of this many pairs; 0 skips the leg.  Reported as the `e2e` object, never as `value`")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", a.gpus))
    a.pairs_given = a.pairs is not None
    if a.config == 3:
        a.genome = "grch38"
        a.introns = max(a.introns, 300000)
        if a.pairs is None:
            a.pairs = 100_000_000 // max(1, world)
    elif a.pairs is None:
        a.pairs = 10_000_000
    if a.plain:
        a.multihit_frac = 0.0
        a.indel_frac = 0.0
    return a


def rank_commands(n, argv, port):
    """(command line, environment additions) of each of the n ranks `python bench.py --gpus n` starts: one process per GPU, the
    variables a launcher would set, rendezvous on 127.0.0.1"""
    return [([sys.executable, os.path.abspath(__file__)] + list(argv),
             {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}) for r in range(n)]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU), rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r, (cmd, env) in enumerate(rank_commands(args.gpus, sys.argv[1:], port)):
        procs.append(subprocess.Popen(cmd, env=dict(os.environ, **env), stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for r, pr in enumerate(procs):
        pr.wait()
        if pr.returncode:
            print("[bench] rank %d of %d exited with code %d" % (r, args.gpus, pr.returncode), file=sys.stderr, flush=True)
        rc = rc or pr.returncode
    sys.exit(rc)


def pmc_traffic_in_run(args):
    """HBM traffic of the kernels, measured in THIS run: the same workload for two steps under `rocprofv3 --kernel-trace --pmc X`,
    one pass per counter (FETCH_SIZE, WRITE_SIZE; separate passes carrying nothing but --kernel-trace, as MI355X_MICROARCH.md
    prescribes), per-dispatch averages per kernel in KB.  None when rocprofv3 is not there or a pass fails (the caller falls back
    to the table committed under profiles/)."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if args.no_pmc or os.environ.get("THJ_BENCH_PMC_CHILD") or not shutil.which("rocprofv3"):
        return None
    if any(k.startswith("ROCP") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None                               # this process is being profiled itself: no profiler inside a profiler
    fwd = ["--pairs", str(args.pairs), "--genome", args.genome, "--genome-len", str(args.genome_len), "--introns", str(args.introns),
           "--intron-max", str(args.intron_max), "--exon-len", str(args.exon_len), "--read-len", str(args.read_len),
           "--multihit-frac", str(args.multihit_frac), "--max-copies", str(args.max_copies), "--indel-frac", str(args.indel_frac),
           "--fusion-frac", str(args.fusion_frac), "--coverage-search", str(args.coverage_search)]
    if args.fusion_search:
        fwd.append("--fusion-search")
    if args.no_hit_heads:
        fwd.append("--no-hit-heads")
    out = {}
    t0 = time.time()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="thj_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "res", "--", sys.executable, os.path.abspath(__file__)] + fwd + \
                  ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--e2e-pairs", "0", "--no-pmc", "--detail", os.path.join(d, "detail.json")]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", THJ_BENCH_PMC_CHILD="1"), capture_output=True, text=True, timeout=420)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            rows = sqlite3.connect(dbs[0]).execute("select name, count(*), sum(counter_value) from pmc_events where counter_name = ? group by name", (ctr,)).fetchall()
            for name, n, sm in rows:
                m = re.search(r"(thj_k_\w+)(<[^>]*>)?", name)
                if m and n:      # under the name with its template arguments, and -- the instances together: each is launched once per side -- without
                    for key in {m.group(1), m.group(1) + (m.group(2) or "")}:
                        out.setdefault(key, {}).setdefault(ctr, 0.0)
                        out[key][ctr] += sm / n
        except (OSError, subprocess.SubprocessError, sqlite3.Error):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"kernels": out, "seconds": time.time() - t0,
            "source": "this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- python bench.py <this workload> --steps 2 --warmup 1"} if out else None


_E2E_EARLY = None


def main():
    global _E2E_EARLY
    args = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if "WORLD_SIZE" in os.environ:                    # one rank of a launched job (torchrun or spawn_ranks)
        world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        control = ProcControl(rank, world) if world > 1 else None
        if world > 1 and args.e2e_pairs > 0 and args.read_len == 100 and args.genome == "chr20":
            # the executables over all the job's GPUs, before any rank holds device memory; the other ranks wait.  An error is
            # reported in the line (e2e.error) and does not stop the scaling measurement.
            if rank == 0:
                try:
                    _E2E_EARLY = e2e_leg(args, n_gpus=world)
                except Exception as e:      # noqa: BLE001
                    _E2E_EARLY = {"error": repr(e)[:2000], "ok": False, "n_gpus": world}
            control.barrier()
        result = run_rank(args, rank, world, local_rank, control, None)
        finish_stdout(result if rank == 0 else None, args.detail)
        return
    if args.gpus > 1 and os.environ.get("THJ_BENCH_INPROC") == "1":
        # functional test on a one-GPU box: N ranks as threads on device 0, loopback exchange
        import threading
        tc = ThreadControl(args.gpus)
        shared = {"ctxs": [None] * args.gpus, "comms": None, "lock": threading.Lock()}
        results, errs = [None] * args.gpus, []

        def go(r):
            try:
                results[r] = run_rank(args, r, args.gpus, int(os.environ.get("THJ_BENCH_DEVICE", "0")), tc.view(r), shared)
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
                tc._bar.abort()
        th = [threading.Thread(target=go, args=(r,)) for r in range(args.gpus)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        finish_stdout(results[0], args.detail)
        return
    if args.gpus > 1:
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d asked for, %d visible" % (args.gpus, torch.cuda.device_count()))
        spawn_ranks(args)
    # the files-in -> files-out leg first, while this process holds nothing: run after the resident-data bench (a context with
    # tens of GB of device memory, the workload's host arrays and the oracle's records still allocated) the same executables
    # took ~30 % longer
    e2e_failed = False
    if args.e2e_pairs > 0 and args.read_len == 100 and args.genome == "chr20":
        try:
            _E2E_EARLY = e2e_leg(args)
        except Exception as e:      # noqa: BLE001 -- the kernel measurement still runs and the line is still printed; the exit code says it failed
            _E2E_EARLY = {"error": repr(e)[:2000], "ok": False}
            e2e_failed = True
    result = run_rank(args, 0, 1, 0, None, None)
    finish_stdout(result, args.detail)
    if e2e_failed:
        sys.stderr.write("bench.py: the files-in -> files-out leg failed: %s\n" % _E2E_EARLY["error"])
        sys.exit(3)


LINE_LIMIT = 6144          # the driver parses the final stdout line; 16 KB parsed in round 4, 22.5 KB did not


def _r(x, nd=4):
    """numbers of the line rounded to nd significant digits (the detail file keeps them whole)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _pick(d, keys, nd=4):
    return None if not isinstance(d, dict) else {k: _r(d[k], nd) for k in keys if k in d and not isinstance(d[k], dict)}


def compact_line(result):
    """The ONE JSON line of the bench contract: the contract's keys, `roofline`, `cpu_baseline`, `metric_e2e`, numbers only --
    under LINE_LIMIT bytes whatever the run produced.  Everything else (the per-kernel table, stage timings, SHA-256s, the prose
    about samples) is in the detail file (`--detail`, default bench_detail.json beside bench.py and under gpurun_out/) and on
    stderr.  tests/test_bench_line_cpu.py builds it from a canned result."""
    r = result
    cfg = dict(r.get("config") or {})
    wl = cfg.get("workload", "")
    if len(wl) > 200:
        wl = wl[:197] + "..."
    cfg["workload"] = wl
    cfg.pop("workload_detail", None)
    out = {k: _r(r.get(k), 7) for k in ("metric", "value", "value_is", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                         "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {k: _r(v) for k, v in cfg.items()}
    out["roofline"] = _pick(r.get("roofline"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_alone", "frac_step", "traffic",
                                                "algorithmic_bytes_per_launch", "avg_kernel_ms", "avg_kernel_ms_alone", "launches",
                                                "traffic_measured_in_run", "share_of_kernel_time", "measured_copy_GBs", "streams_independent"), 5)
    cpu = r.get("cpu_baseline")
    if cpu is not None:
        c = _pick(cpu, ("value", "unit", "cores", "kind"))
        smp = cpu.get("sample", "")
        c["sample"] = smp if len(smp) <= 160 else smp[:157] + "..."
        for sub in ("all_cores", "files_to_files", "reference_calibration"):
            if isinstance(cpu.get(sub), dict):
                c[sub] = _pick(cpu[sub], ("value", "cores", "outputs_identical_to_the_gpu_executables"))
        out["cpu_baseline"] = c
    else:
        out["cpu_baseline"] = None
    e = r.get("e2e")
    if e:
        m = _pick(r.get("metric_e2e"), ("value", "unit", "checked_against_oracle")) or {}
        if e.get("error"):
            m["error"] = str(e["error"])[:300]
        m.update(_pick(e, ("pairs", "both_stages_s", "marginal_pairs_per_s", "fixed_s", "n_gpus")))
        if isinstance(e.get("large"), dict):
            m["large"] = _pick(e["large"], ("pairs", "value", "both_stages_s"))
        g = e.get("grch38")
        if isinstance(g, dict):
            m["grch38"] = {"pairs": g.get("pairs"),
                           "cold": _r((g.get("first_run_no_cache") or {}).get("value")), "warm": _r((g.get("second_run_cache") or {}).get("value"))}
        c3 = e.get("config3_full")
        if isinstance(c3, dict):
            m["config3_full"] = _pick(c3, ("pairs", "seconds", "ranks", "outputs_identical", "value", "error"))
        else:
            # configs[2] at full size takes ten minutes and is not part of a default run (--e2e-config3-pairs): the line points at the committed
            # record of the last such run, labelled as what it is
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "r06_config3_full_after_probe_cap.json")))
                m["config3_full_recorded"] = dict(_pick(rec, ("pairs", "seconds", "ranks", "outputs_identical", "value")), not_measured_in_this_run="profiles/r06_config3_full_after_probe_cap.json")
            except (OSError, ValueError):
                pass
        f2f = e.get("cpu_files_to_files")
        if isinstance(f2f, dict):
            m["cpu_files_identical"] = f2f.get("outputs_identical_to_the_gpu_executables")
        out["metric_e2e"] = m
    else:
        out["metric_e2e"] = None
    out["exchange"] = _pick(r.get("exchange"), ("transport", "ranks", "n_ranks", "calls", "bytes_per_rank", "gathered_bytes_per_step", "us_per_step", "redo"))
    out["per_rank_ms_per_step"] = _r(r.get("per_rank_ms_per_step"), 5)
    out["events"] = _pick(r.get("events"), ("junctions", "deletions", "insertions", "spanning_records_per_step"))
    out["detail"] = r.get("detail_file")
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:             # cannot happen with the keys above; if it ever does, the contract's keys win
        for k in ("events", "exchange", "per_rank_ms_per_step", "metric_e2e"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    return line


def write_detail(result, path):
    """The whole result (per-kernel table, e2e stage timings, checks) beside the line: the file named by --detail and, when the
    repo has a gpurun_out/ directory, a copy there (what travels back from a GPU box)."""
    txt = json.dumps(result, indent=1)
    written = []
    copy_too = os.path.isdir(os.path.join(ROOT, "gpurun_out")) and os.path.dirname(os.path.abspath(path)) == ROOT
    for p in [path] + ([os.path.join(ROOT, "gpurun_out", os.path.basename(path))] if copy_too else []):
        try:
            with open(p, "w") as f:
                f.write(txt + "\n")
            written.append(p)
        except OSError:
            pass
    return written


def finish_stdout(result, detail_path=None):
    # The JSON line must be the last thing on stdout: RCCL writes a version banner through C stdio, which sits in libc's
    # buffer until exit when stdout is a pipe.  Flush that first, print the line, then point fd 1 at /dev/null so that
    # nothing written later (exit handlers, library destructors) can follow it.  The process still exits normally --
    # profilers attached to it (rocprofv3) finalise in their exit handlers.
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if result is not None:
        if detail_path:
            w = write_detail(result, detail_path)
            result["detail_file"] = os.path.relpath(w[0], ROOT) if w else None
        sys.stderr.write("[bench] full result:\n" + json.dumps(result) + "\n")
        sys.stderr.flush()
        print(compact_line(result), flush=True)
    sys.stdout.flush()
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)


def e2e_leg(args, n_gpus=1):
    """The metric as BASELINE words it: wall clock of the drop-in executables, files in -> files out (segment_juncs, then
    long_spanning_reads on each side), on generated configs[1]-shaped files (tools/bin/thj_gen: BAM inputs with .index, the
    reads as unaligned BAM).  What is checked is what was timed: SHA-256 of the five outputs of the timed run, and the oracle on
    the first pairs of the SAME files (thj_gen writes pair i as a pure function of (seed, i), so the text twin of the first N
    pairs is the first N pairs of the big case) against the same read ids of the timed run's outputs.  A separate small case
    goes through the oracle whole, down to junctions.bed.  An error here fails the bench (main() exits non-zero)."""
    import hashlib
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from e2e_bench import mix_gen_args, run_e2e
    gen_args = mix_gen_args(args.multihit_frac, args.max_copies, args.indel_frac)      # the same mix as the resident-data line
    d = tempfile.mkdtemp(prefix="thj_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    if n_gpus > 1:
        os.environ.setdefault("THJ_STAGE_TIMEOUT", "300")       # (an executable that hangs on its first multi-GPU collective is an error after five minutes, not ten)
    try:
        # the executables drive every GPU they can see (read-id shards round-robin over the contexts): the first n_gpus devices here
        gpu_env = {"HIP_VISIBLE_DEVICES": ",".join(str(k) for k in range(n_gpus))}
        res = run_e2e(args.e2e_pairs, args.read_len, args.genome_len, args.introns, workdir=d, keep=True, env_extra=gpu_env, gen_args=gen_args)
        keep = ("pairs", "input_bytes", "gen_seconds", "segment_juncs_s", "long_spanning_reads_left_s", "long_spanning_reads_right_s",
                "both_stages_s", "junctions", "junctions_bed_s", "junctions_bed_lines", "outside_main_s", "span_left_bytes", "span_right_bytes", "host_ingest_fallback_shards")
        out = {k: res[k] for k in keep if k in res}
        out["workload"] = ("configs[1] with SURVEY 8(d)'s mix -- the same as the resident-data line: %g %% of the pairs from a %d-copy repeat family, %g %% deletion reads"
                           % (100 * args.multihit_frac, args.max_copies, 100 * args.indel_frac)) if gen_args else "configs[1] without the mix (--plain)"
        out["stage_timing"] = {st: [l for l in res.get(st + "_log_tail", []) if l.startswith("[timing]") or l.startswith("[worker-seconds]") or "made on the device" in l]
                               for st in ("segment_juncs", "long_spanning_reads_left", "long_spanning_reads_right")}
        out["value"] = res["pairs"] / res["both_stages_s"]
        out["n_gpus"] = n_gpus
        out["unit"] = "read-pairs/s (wall clock of both executables, files in -> files out, %d GPU%s, host CPUs: %s)" % (n_gpus, "" if n_gpus == 1 else "s", _cpu_quota())
        outs = ("out.juncs", "out.insertions", "out.deletions", "span_left.bam", "span_right.bam", "junctions.bed")

        def sha(name):
            h = hashlib.sha256()
            with open(os.path.join(d, name), "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            return h.hexdigest()
        out["sha256"] = {n: sha(n) for n in outs}
        out["timed_run_check"] = e2e_timed_run_check(d, args)
        # every process of the timed run was ONE process that waited for its own exit (the default since round 3: the output hand-off
        # to a child of rounds 1-2 is opt-in now, THJ_HANDOFF=1), so the figure above is also the figure without hand-off
        out["value_no_handoff"] = out["value"]
        out["no_handoff_seconds"] = [res["segment_juncs_s"], res["long_spanning_reads_left_s"], res["long_spanning_reads_right_s"]]
        # the host's share: CPU seconds (user + system) of the three processes against wall clock x CPUs available.  With the BAM records
        # and BGZF members made on the device the host parses options and FASTA, plans shards, stages compressed input and writes
        # members; the busy fraction says how far that is from being the limit when GPUs are added
        cpu_s = [res["segment_juncs_cpu_s"], res["long_spanning_reads_left_cpu_s"], res["long_spanning_reads_right_cpu_s"]]
        ncpu = _cpu_count()
        out["host"] = {"cpu_seconds": cpu_s, "cpus": ncpu, "busy_fraction": sum(cpu_s) / max(1e-9, res["both_stages_s"] * ncpu),
                       "cpu_seconds_per_million_pairs": sum(cpu_s) / (res["pairs"] / 1e6)}

        def inflated_sha(name):          # SHA-256 of the BAM stream inside the BGZF members (gzip reads them)
            import subprocess
            h = hashlib.sha256()
            with subprocess.Popen(["zcat", os.path.join(d, name)], stdout=subprocess.PIPE) as pr:
                for blk in iter(lambda: pr.stdout.read(1 << 24), b""):
                    h.update(blk)
            if pr.returncode != 0:
                raise RuntimeError("zcat could not read %s" % name)
            return h.hexdigest()
        # the same files again through the HOST's record encoder and compressor (THJ_HOST_BAM=1): at full size the BAM streams inside the
        # two writers' files must be the same bytes (members are cut and deflated differently), the event files the same files
        dev_stream = {n: inflated_sha(n) for n in ("span_left.bam", "span_right.bam")}
        r2 = run_e2e(args.e2e_pairs, args.read_len, args.genome_len, args.introns, workdir=d, keep=True, env_extra=dict(gpu_env, THJ_HOST_BAM="1"), gen_args=gen_args)
        out["value_host_writer"] = r2["pairs"] / r2["both_stages_s"]
        out["host_writer_seconds"] = [r2["segment_juncs_s"], r2["long_spanning_reads_left_s"], r2["long_spanning_reads_right_s"]]
        out["bam_stream_sha256"] = dev_stream
        out["device_writer_stream_equals_host_writer"] = all(inflated_sha(n) == dev_stream[n] for n in dev_stream)
        out["outputs_identical_without_handoff"] = all(sha(n) == out["sha256"][n] for n in outs[:3]) and out["device_writer_stream_equals_host_writer"]
        out["inflate"] = e2e_inflate_roofline(os.path.join(d, "left_seg1.bam"))
        if not getattr(args, "no_cpu_baseline", False) and n_gpus == 1:
            out["cpu_files_to_files"] = e2e_cpu_port(d, args, gen_args, run_e2e, sha, inflated_sha, out["sha256"], dev_stream)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if getattr(args, "e2e_pairs_large", 0) > args.e2e_pairs:
        # a second point: t(n) = fixed + n / rate through (e2e_pairs, both_stages_s) and (e2e_pairs_large, ...).  Timing only (the
        # outputs of the first point are the ones checked); a box without the room for its files keeps the first point and says so
        try:
            big = run_e2e(args.e2e_pairs_large, args.read_len, args.genome_len, args.introns, env_extra=gpu_env, gen_args=gen_args)
            n0, t0, n1, t1 = float(out["pairs"]), float(out["both_stages_s"]), float(big["pairs"]), float(big["both_stages_s"])
            out["large"] = {k: big[k] for k in ("pairs", "input_bytes", "gen_seconds", "segment_juncs_s", "long_spanning_reads_left_s", "long_spanning_reads_right_s", "both_stages_s", "outside_main_s") if k in big}
            out["large"]["value"] = n1 / t1
            if t1 > t0:
                out["rate_pairs_per_s"] = out["marginal_pairs_per_s"] = (n1 - n0) / (t1 - t0)
                out["fixed_s"] = t0 - n0 / out["rate_pairs_per_s"]
        except (RuntimeError, OSError, subprocess.CalledProcessError) as e:
            out["large"] = {"error": str(e)[-300:]}
    if getattr(args, "e2e_grch38_pairs", 0) > 0 and n_gpus == 1:      # (a multi-GPU job keeps its files leg to the two points above: the other ranks wait for it)
        # configs[2]'s genome: where the reference itself costs seconds per process.  Run 1 finds no packed-genome cache (segment_juncs
        # parses the FASTA, packs it and leaves the cache beside its outputs; the two long_spanning_reads map it), run 2 finds it.
        from e2e_bench import GRCH38_LENS
        dg = tempfile.mkdtemp(prefix="thj_e2e_g38_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            ga = gen_args + ["--contigs", ",".join(str(x) for x in GRCH38_LENS)]
            runs = []
            for k in range(2):
                r = run_e2e(args.e2e_grch38_pairs, args.read_len, sum(GRCH38_LENS), 300000, workdir=dg, keep=True, env_extra=gpu_env, gen_args=ga)
                runs.append({key: r[key] for key in ("segment_juncs_s", "long_spanning_reads_left_s", "long_spanning_reads_right_s", "both_stages_s", "outside_main_s") if key in r})
                runs[-1]["value"] = r["pairs"] / r["both_stages_s"]
                runs[-1]["reference_timing"] = {st: [l for l in r.get(st + "_log_tail", []) if "reference" in l]
                                                for st in ("segment_juncs", "long_spanning_reads_left", "long_spanning_reads_right")}
                if k == 0:
                    gen_s, in_bytes = r["gen_seconds"], r["input_bytes"]
            out["grch38"] = {"pairs": args.e2e_grch38_pairs, "genome": "25 contigs with the GRCh38 primary-assembly lengths, 3 088 286 401 bp, 300 000 introns, the same mix",
                             "gen_seconds": gen_s, "input_bytes": in_bytes, "first_run_no_cache": runs[0], "second_run_cache": runs[1]}
        except (RuntimeError, OSError, subprocess.CalledProcessError) as e:
            out["grch38"] = {"error": str(e)[-300:]}
        finally:
            shutil.rmtree(dg, ignore_errors=True)
    if getattr(args, "e2e_config3_pairs", 0) > 0:
        try:
            out["config3_full"] = e2e_config3_full(args, args.e2e_config3_pairs)
        except (RuntimeError, OSError, subprocess.CalledProcessError) as e:
            out["config3_full"] = {"error": str(e)[-600:], "ok": False}
    # a small case through the oracle whole (event files byte for byte, every spanning record, junctions.bed)
    d = tempfile.mkdtemp(prefix="thj_e2e_chk_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        out["sample_check"] = e2e_sample_check(d, args)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    ok = out["timed_run_check"]
    sc = out["sample_check"]
    out["ok"] = bool(ok["events_of_the_sample_found_in_the_timed_outputs"] and ok["spanning_records_identical_to_oracle"] and out["outputs_identical_without_handoff"]
                     and sc["event_files_identical_to_oracle"] and sc["spanning_records_identical_to_oracle"] and sc["junctions_bed_identical_to_oracle"])
    if not out["ok"]:
        raise RuntimeError("e2e leg: outputs differ from the oracle: %s" % json.dumps(out)[:1500])
    return out


def e2e_config3_full(args, pairs=100_000_000, ranks=8, check_pairs=20000):
    """BASELINE configs[2] at FULL size through the executables on the GPUs this box has: `pairs` pairs of
    2x100 bp against the 25-contig GRCh38-sized genome (31 GB of BAM in, ~8 GB out: byte offsets past 2^32 inside one input file, 8x the
    shard count of the per-GPU case).  On one GPU the eight ranks of the 8-GPU run are eight contexts on the device (THJ_CTX_PER_GPU=8:
    the same shard dealing, the same exchange step with eight sections over the loopback transport -- segment_juncs_main.cpp, what the
    hardware allows as a rehearsal).  Run a second time with ONE context: the event files must be the same files, the BAM streams inside
    the spanning files the same bytes.  The first `check_pairs` pairs of the same files go through the oracle as in e2e.timed_run_check."""
    import hashlib
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from e2e_bench import GRCH38_LENS, mix_gen_args, run_e2e
    ga = mix_gen_args(args.multihit_frac, args.max_copies, args.indel_frac) + ["--contigs", ",".join(str(x) for x in GRCH38_LENS)]
    d = tempfile.mkdtemp(prefix="thj_e2e_c3_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    outs = ("out.juncs", "out.insertions", "out.deletions")

    def sha(name):
        h = hashlib.sha256()
        with open(os.path.join(d, name), "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    def inflated_sha(name):
        h = hashlib.sha256()
        n = 0
        with subprocess.Popen(["zcat", os.path.join(d, name)], stdout=subprocess.PIPE) as pr:
            for blk in iter(lambda: pr.stdout.read(1 << 24), b""):
                h.update(blk)
                n += len(blk)
        if pr.returncode != 0:
            raise RuntimeError("zcat could not read %s" % name)
        return h.hexdigest(), n
    os.environ.setdefault("THJ_STAGE_TIMEOUT", "1800")
    try:
        runs = []
        for k, per in enumerate((1, ranks)):       # (the one-context run first: it packs the reference and leaves the packed-genome cache, as a tophat run's first process does)
            r = run_e2e(pairs, args.read_len, sum(GRCH38_LENS), 300000, workdir=d, keep=True, env_extra={"THJ_CTX_PER_GPU": str(per), "HIP_VISIBLE_DEVICES": "0"}, gen_args=ga)
            one = {key: r[key] for key in ("segment_juncs_s", "long_spanning_reads_left_s", "long_spanning_reads_right_s", "both_stages_s", "junctions", "span_left_bytes", "span_right_bytes",
                                           "host_ingest_fallback_shards") if key in r}
            one["contexts"] = per
            one["value"] = r["pairs"] / r["both_stages_s"]
            one["sha256"] = {n: sha(n) for n in outs}
            one["bam_stream"] = {n: inflated_sha(n) for n in ("span_left.bam", "span_right.bam")}
            one["exchange"] = [l for l in r.get("segment_juncs_log_tail", []) if "exchange" in l or "all-gather" in l][:4]
            if k == 0:
                gen_s, in_bytes = r["gen_seconds"], r["input_bytes"]
                largest = max(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".bam") and not f.startswith("span_"))
                largest_out = max(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".bam") and f.startswith("span_"))
                chk = e2e_timed_run_check(d, args, pairs=check_pairs, genome_args=["--contigs", ",".join(str(x) for x in GRCH38_LENS)], introns=300000)
            runs.append(one)
        same = runs[0]["sha256"] == runs[1]["sha256"] and runs[0]["bam_stream"] == runs[1]["bam_stream"]
        return {"pairs": pairs, "ranks": ranks, "seconds": runs[1]["both_stages_s"], "value": runs[1]["value"], "outputs_identical": bool(same),
                "genome": "25 contigs with the GRCh38 primary-assembly lengths, 3 088 286 401 bp, 300 000 introns, the line's mix", "gen_seconds": gen_s,
                "input_bytes": in_bytes, "largest_input_file_bytes": largest, "largest_output_file_bytes": largest_out,
                "offsets_past_2_32": {"in_an_input_file": bool(largest > (1 << 32)), "in_an_output_file": bool(largest_out > (1 << 32))},
                "eight_contexts": runs[1], "one_context": runs[0], "first_pairs_against_the_oracle": chk,
                "ok": bool(same and chk["events_of_the_sample_found_in_the_timed_outputs"] and chk["spanning_records_identical_to_oracle"])}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def e2e_cpu_port(d, args, gen_args, run_e2e, sha, inflated_sha, gpu_sha, gpu_stream):
    """A CPU figure that is like for like with `e2e`: the SAME files through tools/bin/cpuport/{segment_juncs,long_spanning_reads} --
    the executables' own host sources (BGZF inflate, BAM parse, batching, BAM encode, BGZF deflate: tophat_amd/csrc/host/) linked
    against tools/cpuport/thj_cpuport.cpp, which answers the C ABI's compute calls with oracle/liborc.so on host threads (the
    plain-C restatement of the reference; kind "port").  Its outputs on the timed files must be the GPU executables' outputs (event
    files byte for byte, the BAM streams inside the BGZF members): the oracle against the device at full size, every record.
    One thread of the oracle is timed on the files' first pairs cut to a bounded sample."""
    import shutil
    import tempfile
    cpu_bin = os.path.join(ROOT, "tools", "bin", "cpuport")
    if not os.path.exists(os.path.join(cpu_bin, "segment_juncs")):
        return {"error": "tools/bin/cpuport not built (__graft_entry__.build())"}
    C = max(1, min(64, _cpu_count()))
    env = {"THJ_HOST_INGEST": "1", "THJ_CTX_PER_GPU": "1", "THJ_NO_HANDOFF": "1", "THJ_CPUPORT_THREADS": str(C)}
    r = run_e2e(args.e2e_pairs, args.read_len, args.genome_len, args.introns, workdir=d, keep=True, env_extra=env, gen_args=gen_args, bindir=cpu_bin)
    same = all(sha(n) == gpu_sha[n] for n in ("out.juncs", "out.insertions", "out.deletions")) and all(inflated_sha(n) == gpu_stream[n] for n in gpu_stream)
    out = {"value": r["pairs"] / r["both_stages_s"], "unit": "read-pairs/s, files in -> files out", "cores": C, "kind": "port, files to files",
           "sample": "the e2e leg's own %d-pair files through the executables' host code over oracle/liborc.so, %d oracle threads: segment_juncs %.1f s + "
                     "long_spanning_reads %.1f + %.1f s" % (r["pairs"], C, r["segment_juncs_s"], r["long_spanning_reads_left_s"], r["long_spanning_reads_right_s"]),
           "seconds": [r["segment_juncs_s"], r["long_spanning_reads_left_s"], r["long_spanning_reads_right_s"]],
           "outputs_identical_to_the_gpu_executables": bool(same)}
    if not same:
        raise RuntimeError("e2e leg: the CPU port's outputs on the timed files differ from the GPU executables'")
    # one oracle thread, on files of their own (a tenth of the pairs, at most a million)
    m = max(1000, min(1000000, args.e2e_pairs // 10))
    d1 = tempfile.mkdtemp(prefix="thj_e2e_cpu1_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        r1 = run_e2e(m, args.read_len, args.genome_len, args.introns, workdir=d1, keep=True, env_extra=dict(env, THJ_CPUPORT_THREADS="1"), gen_args=gen_args, bindir=cpu_bin)
        out["one_oracle_thread"] = {"value": r1["pairs"] / r1["both_stages_s"], "unit": "read-pairs/s, files in -> files out", "cores": 1,
                                    "sample": "%d pairs generated the same way, one oracle thread (the host readers and writers keep their threads): %.1f + %.1f + %.1f s"
                                              % (m, r1["segment_juncs_s"], r1["long_spanning_reads_left_s"], r1["long_spanning_reads_right_s"])}
    finally:
        shutil.rmtree(d1, ignore_errors=True)
    return out


def e2e_inflate_roofline(bam_path):
    """The kernels the files-in -> files-out number spends most of its device time in (BGZF inflate: thj_k_huffp + thj_k_lz) on one of
    the timed run's input files, device-resident, HIP events around the launches: compressed bytes in + inflated bytes out per
    second against the HBM peak (what the job needs to move; the token stream between the two kernels is extra traffic)."""
    import ctypes as C
    import struct
    data = open(bam_path, "rb").read()
    offs, lens, isz, off = [], [], [], 0
    while off < len(data):
        bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
        offs.append(off + 18); lens.append(bsize - 26); isz.append(struct.unpack_from("<I", data, off + bsize - 4)[0])
        off += bsize
    n = len(offs)
    blk = np.zeros(n, dtype=[("in_off", "<u8"), ("in_len", "<u4"), ("r", "<u4")])
    blk["in_off"], blk["in_len"] = offs, lens
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with host.Context(0, stream=stream.cuda_stream) as ctx:
        d_comp = torch.from_numpy(np.frombuffer(data + bytes(64), dtype=np.uint8).copy()).to(dev)
        d_blk = torch.from_numpy(blk.view(np.uint8).copy()).to(dev)
        d_out = torch.empty(n << 16, dtype=torch.uint8, device=dev)
        d_len = torch.empty(n, dtype=torch.int32, device=dev)

        def run():
            rc = ctx.lib.thj_bgzf_inflate(ctx._ctx, C.c_void_p(d_comp.data_ptr()), C.c_int64(len(data)), C.c_void_p(d_blk.data_ptr()), C.c_int64(n),
                                          C.c_void_p(d_out.data_ptr()), C.c_void_p(d_len.data_ptr()), 1)
            assert rc == 0
        run(); ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rep = 5
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(rep):
                run()
            e1.record(stream)
        ctx.sync()
        ms = e0.elapsed_time(e1) / rep
        ok = bool((d_len.cpu().numpy().astype(np.uint32) == np.array(isz, dtype=np.uint32)).all())
    comp_b, infl_b = float(sum(lens)), float(sum(isz))
    return {"file": os.path.basename(bam_path), "members": n, "compressed_bytes": comp_b, "inflated_bytes": infl_b, "ms_per_launch": ms,
            "inflated_GBs": infl_b / ms / 1e6, "achieved": (comp_b + infl_b) / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (comp_b + infl_b) / ms / 1e6 / HBM_PEAK_GBS, "bound": "instruction issue (serial entropy decoding), not HBM: see DESIGN.md",
            "lengths_ok": ok}


def _bam_records_below(path, id_limit):
    """the records of a spanning BAM whose read id is below id_limit (the file is in read-id order): BGZF members are read one by one
    and the walk stops at the first record beyond the limit"""
    import struct
    import zlib
    from tophat_amd.bamio import parse_bam_record
    recs, buf, hdr_done, names = [], b"", False, []
    with open(path, "rb") as f:
        while True:
            head = f.read(18)
            if len(head) < 18:
                break
            bsize = struct.unpack_from("<H", head, 16)[0] + 1
            body = f.read(bsize - 18)
            buf += zlib.decompress(body[:-8], -15)
            if not hdr_done:
                if len(buf) < 12:
                    continue
                l_text, = struct.unpack_from("<i", buf, 4)
                if len(buf) < 12 + l_text:
                    continue
                off = 8 + l_text
                n_ref, = struct.unpack_from("<i", buf, off)
                off += 4
                okh, names = True, []
                for _ in range(n_ref):
                    if len(buf) < off + 4:
                        okh = False
                        break
                    l_name, = struct.unpack_from("<i", buf, off)
                    names.append(buf[off + 4:off + 4 + l_name - 1].decode())
                    off += 4 + l_name + 4
                if not okh or len(buf) < off:
                    continue
                buf = buf[off:]
                hdr_done = True
            off = 0
            while off + 4 <= len(buf):
                bs, = struct.unpack_from("<i", buf, off)
                if off + 4 + bs > len(buf):
                    break
                r = parse_bam_record(buf[off + 4:off + 4 + bs], names)
                off += 4 + bs
                if int(r[0]) >= id_limit:
                    return recs
                recs.append(r)
            buf = buf[off:]
    return recs


def e2e_timed_run_check(d, args, pairs=20000, genome_args=None, introns=None):
    """The oracle on the first `pairs` pairs of the timed run's own files.  Stage 1: every junction / deletion / insertion the oracle
    finds in those pairs must be in the timed run's event files (set union: a sample's events are a subset; an insertion's bases
    may come from a later read of the left side, so insertions are matched by position and length).  Stage 2: the oracle's
    spanning records of those reads, computed with the timed run's event files as its junction / indel sets, must equal the
    records of the same read ids in the timed run's BAMs, field for field and in order."""
    import subprocess
    import tempfile
    import shutil
    import orc
    from tophat_amd.batch import JUNC_DTYPE, build_seg_batch, build_span_batch, merge_events
    from tophat_amd.samtext import parse_header, parse_sam_hits, read_fasta, read_fastq
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from e2e_bench import mix_gen_args
    gen = os.path.join(ROOT, "tools", "bin", "thj_gen")
    t = tempfile.mkdtemp(prefix="thj_e2e_twin_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        subprocess.check_call([gen, "--out", t, "--pairs", str(pairs), "--read-len", str(args.read_len), "--genome-len", str(args.genome_len),
                               "--introns", str(introns if introns is not None else args.introns), "--text"] + mix_gen_args(args.multihit_frac, args.max_copies, args.indel_frac)
                              + list(genome_args or []), stdout=subprocess.DEVNULL)
        f = lambda n: os.path.join(t, n)      # noqa: E731
        names, _ = parse_header(os.path.join(d, "hdr.sam"))
        same_genome = open(f("ref.fa"), "rb").read(1 << 20) == open(os.path.join(d, "ref.fa"), "rb").read(1 << 20)
        fa_names, fa_seqs = read_fasta(f("ref.fa"))
        seqs = [orc.fold_genome_char(s_) for s_ in fa_seqs]
        ref_ids = {n: i + 1 for i, n in enumerate(names)}
        og = orc.Genome(seqs)
        nseg = max(1, args.read_len // 25)
        sides = {}
        for sd in ("left", "right"):
            sides[sd] = dict(reads=read_fastq(f("%s.fq" % sd)),
                             segs=[list(parse_sam_hits(f("%s_seg%d.sam" % (sd, k + 1)), ref_ids, 500000)) for k in range(nseg)],
                             full=list(parse_sam_hits(f("%s_map.sam" % sd), ref_ids, 500000)))
        want = None
        for sd, side, other in (("left", READ_LEFT, "right"), ("right", READ_RIGHT, "left")):
            b = build_seg_batch(sides[sd]["segs"], sides[sd]["reads"], sides[other]["full"], sides[other]["segs"][-1])
            e = orc.segjuncs(Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20), og, b)
            want = e if want is None else merge_events(want, e)
        # the timed run's event files
        J, D, I = set(), set(), {}
        for l in open(os.path.join(d, "out.juncs")):
            c, a, b_, st = l.split()
            J.add((ref_ids[c], int(a), int(b_), 1 if st == "-" else 0))
        for l in open(os.path.join(d, "out.deletions")):
            c, a, b_ = l.split()[:3]
            D.add((ref_ids[c], int(a) - 1, int(b_)))
        for l in open(os.path.join(d, "out.insertions")):
            c, a, _, sq = l.split()[:4]
            I[(ref_ids[c], int(a), len(sq))] = sq
        miss = sum((int(j["ref_id"]), int(j["left"]), int(j["right"]), int(j["antisense"])) not in J for j in want.juncs)
        miss += sum((int(j["ref_id"]), int(j["left"]), int(j["right"])) not in D for j in want.deletions)
        miss += sum((r, l_, len(sq)) not in I for (r, l_, sq) in want.insertions)
        jj = np.array(sorted(set(J) | set((r, a, b_, 0) for (r, a, b_) in D)), dtype=JUNC_DTYPE) if (J or D) else np.zeros(0, dtype=JUNC_DTYPE)
        ii = sorted((r, a, sq) for (r, a, _n), sq in I.items())
        n_rec, same = 0, True
        for sd in ("left", "right"):
            quals = {k: "I" * len(v) for k, v in sides[sd]["reads"].items()}
            sb = build_span_batch(sides[sd]["segs"], sides[sd]["reads"], quals)
            alns = orc.spanning(Params(), og, sb, jj, ii)
            wrecs = [tuple(str(x) for x in a.sam_fields(int(sb.read_id[a.read_idx]), names)) for a in alns]
            lim = max(sides[sd]["reads"]) + 1
            recs = _bam_records_below(os.path.join(d, "span_%s.bam" % sd), lim)
            grecs = [tuple(str(x) for x in (r[0], r[1], r[2], r[3], r[5]) + tuple(r[8:])) for r in recs]
            n_rec += len(grecs)
            same = same and grecs == wrecs
        return {"pairs": pairs, "same_genome_as_the_timed_run": bool(same_genome), "oracle_junctions": int(len(want.juncs)), "oracle_deletions": int(len(want.deletions)),
                "oracle_insertions": int(len(want.insertions)), "events_of_the_sample_found_in_the_timed_outputs": bool(miss == 0 and same_genome),
                "events_missing": int(miss), "spanning_records": n_rec, "spanning_records_identical_to_oracle": bool(same and n_rec > 0)}
    finally:
        shutil.rmtree(t, ignore_errors=True)


def _cpu_count():
    """CPUs this process may use: the cgroup quota when there is one, else the visible hardware threads"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def _cpu_quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return "%d hardware threads, cgroup quota %s" % (os.cpu_count(), "none" if q == "max" else "%.0f CPUs" % (int(q) / int(p)))
    except (OSError, ValueError):
        return "%d hardware threads" % os.cpu_count()


def e2e_sample_check(d, args, pairs=20000, bind=None, env=None):
    """executables on a small text+BAM twin case vs the oracle: the three event files byte for byte, and every spanning record"""
    import subprocess
    import orc
    from golden_util import events_text
    from tophat_amd.bamio import read_bam
    from tophat_amd.batch import build_seg_batch, build_span_batch, events_to_span_inputs, merge_events
    from tophat_amd.samtext import parse_header, parse_sam_hits, read_fasta, read_fastq
    gen = os.path.join(ROOT, "tools", "bin", "thj_gen")
    # bind / env: other builds of the two executables (tools/bin/cpuport: the CPU port of the cpu_baseline leg; no thj_junctions there)
    bind = bind or os.path.join(ROOT, "tophat_amd", "bin")
    env = dict(os.environ, **(env or {}))
    subprocess.check_call([gen, "--out", d, "--pairs", str(pairs), "--genome-len", "4000000", "--introns", "1500", "--text"], stdout=subprocess.DEVNULL)
    f = lambda n: os.path.join(d, n)      # noqa: E731
    segs = {sd: ",".join(f("%s_seg%d.bam" % (sd, k)) for k in (1, 2, 3, 4)) for sd in ("left", "right")}
    out = {k: f("out." + k) for k in ("juncs", "insertions", "deletions", "fusions")}
    subprocess.check_call([os.path.join(bind, "segment_juncs"), "--no-coverage-search", "--no-microexon-search", "--segment-length", "25", "--sam-header",
                           f("hdr.sam"), "--inner-dist-mean", "50", "--inner-dist-std-dev", "20", f("ref.fa"), out["juncs"], out["insertions"],
                           out["deletions"], out["fusions"], f("left_reads.bam"), f("left_map.bam"), segs["left"], f("right_reads.bam"),
                           f("right_map.bam"), segs["right"]], stderr=subprocess.DEVNULL, env=env)
    names, _ = parse_header(f("hdr.sam"))
    fa_names, fa_seqs = read_fasta(f("ref.fa"))
    seqs = [orc.fold_genome_char(s) for s in fa_seqs]
    ref_ids = {n: i + 1 for i, n in enumerate(names)}
    og = orc.Genome(seqs)
    sides = {}
    for sd in ("left", "right"):
        sides[sd] = dict(reads=read_fastq(f("%s.fq" % sd)),
                         segs=[list(parse_sam_hits(f("%s_seg%d.sam" % (sd, k)), ref_ids, 500000)) for k in (1, 2, 3, 4)],
                         full=list(parse_sam_hits(f("%s_map.sam" % sd), ref_ids, 500000)))
    want = None
    for sd, side, other in (("left", READ_LEFT, "right"), ("right", READ_RIGHT, "left")):
        b = build_seg_batch(sides[sd]["segs"], sides[sd]["reads"], sides[other]["full"], sides[other]["segs"][-1])
        e = orc.segjuncs(Params(read_side=side, inner_dist_mean=50, inner_dist_std_dev=20), og, b)
        want = e if want is None else merge_events(want, e)
    import pathlib
    wt = events_text(want, names, pathlib.Path(d))
    same_events = all(open(out[k]).read() == wt[k] for k in ("juncs", "insertions", "deletions"))
    jj, ii = events_to_span_inputs(want)
    n_rec, same_recs = 0, True
    all_alns = []
    for sd in ("left", "right"):
        bam = f("span_%s.bam" % sd)
        subprocess.check_call([os.path.join(bind, "long_spanning_reads"), "--segment-length", "25", "--sam-header", f("hdr.sam"), f("ref.fa"),
                               f("%s_reads.bam" % sd), out["juncs"], out["insertions"], out["deletions"], "/dev/null", bam, segs[sd]], stderr=subprocess.DEVNULL, env=env)
        quals = {k: "I" * len(v) for k, v in sides[sd]["reads"].items()}
        sb = build_span_batch(sides[sd]["segs"], sides[sd]["reads"], quals)
        alns = orc.spanning(Params(), og, sb, jj, ii)
        all_alns += alns
        wrecs = [tuple(str(x) for x in a.sam_fields(int(sb.read_id[a.read_idx]), names)) for a in alns]
        _, recs = read_bam(bam)
        grecs = [tuple(str(x) for x in (r[0], r[1], r[2], r[3], r[5]) + tuple(r[8:])) for r in recs]
        n_rec += len(grecs)
        same_recs = same_recs and grecs == wrecs
    # junctions.bed: the drop-in consensus program on the two spanning BAMs vs the oracle's consensus of the oracle's records
    want_bed = orc.junctions_bed(orc.junction_consensus(orc.jrecs_from_alns(all_alns)), names)
    same_bed = None
    if os.path.exists(os.path.join(bind, "thj_junctions")):
        subprocess.check_call([os.path.join(bind, "thj_junctions"), "--sam-header", f("hdr.sam"), f("ref.fa"), f("junctions.bed"),
                               f("span_left.bam") + "," + f("span_right.bam")], stderr=subprocess.DEVNULL, env=env)
        same_bed = open(f("junctions.bed")).read() == want_bed
    return {"pairs": pairs, "junctions": len(want.juncs), "event_files_identical_to_oracle": bool(same_events),
            "spanning_records": n_rec, "spanning_records_identical_to_oracle": bool(same_recs),
            "junctions_bed_lines": want_bed.count("\n") - 1, "junctions_bed_identical_to_oracle": None if same_bed is None else bool(same_bed)}


def run_rank(args, rank, world, local_rank, control, shared):
    """one rank = one GPU: workload of --pairs pairs, W + K steps, the exchange step through the C ABI when world > 1"""
    # THJ_FORCE_COLLECTIVE=1 runs the exchange step through RCCL even at world size 1 (single-GPU boxes).
    use_comm = world > 1 or os.environ.get("THJ_FORCE_COLLECTIVE") == "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- workload: BASELINE.json configs[1] shape (per GPU) --------------------------------
    t_gen = time.time()
    contig_lens = [args.genome_len] if args.genome == "chr20" else GRCH38_LENS
    genome_len = int(sum(contig_lens))
    seqs, genes = make_scale_genome(1, contig_lens, args.introns, exon_len=args.exon_len, intron_max=args.intron_max)
    dup_shift = 0
    if args.multihit_frac > 0 and args.max_copies > 2:
        # a planted repeat family: the first 1/96 of contig 0, max_copies copies back to back; its genes are the family,
        # the genes behind the last copy stay unique, the ones in between are overwritten and dropped
        dup_shift = len(seqs[0]) // 96
        if dup_shift <= max(500000, args.intron_max + 1) + 2 * args.exon_len:
            raise SystemExit("--max-copies: the genome is too small for a repeat family whose copies lie further apart than the longest intron")
        for k in range(1, args.max_copies):
            seqs[0][k * dup_shift:(k + 1) * dup_shift] = seqs[0][:dup_shift]
        fam = (genes[:, 0] == 0) & (genes[:, 3] + args.exon_len + 1000 < dup_shift)
        uniq = (genes[:, 0] != 0) | (genes[:, 1] >= args.max_copies * dup_shift + 1000)
        genes = genes[fam | uniq]
    elif args.multihit_frac > 0:                     # two-copy genome: [H, 2H) := [0, H), genes of the first copy only
        dup_shift = len(seqs[0]) // 2
        seqs[0][dup_shift:2 * dup_shift] = seqs[0][:dup_shift]
        genes = genes[(genes[:, 0] == 0) & (genes[:, 3] + args.exon_len + 1000 < dup_shift)]
    lib = host.load_lib()
    strs = [s.tobytes().decode() for s in seqs]
    pg = host.pack_genome(strs, lib=lib)
    stream = torch.cuda.Stream(device=dev)
    ctx = host.Context(local_rank, stream=stream.cuda_stream)
    ctx.upload_genome(pg)
    w = make_device_workload(100 + rank, seqs, genes, None, args.pairs, dev, read_len=args.read_len, seg_len=25,
                             inner_mean=50.0, inner_sd=20.0, exon_len=args.exon_len, multi_frac=args.multihit_frac, dup_shift=dup_shift,
                             fusion_frac=args.fusion_frac if args.fusion_search else 0.0, indel_frac=args.indel_frac, max_copies=args.max_copies)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    max_intron = max(500000, args.intron_max + 1)
    pk = dict(inner_dist_mean=50, inner_dist_std_dev=20, max_segment_intron=max_intron, max_report_intron=max_intron)
    if args.fusion_search:
        pk["fusion_min_dist"] = 100000
    p_left = Params(read_side=READ_LEFT, **pk)
    p_right = Params(read_side=READ_RIGHT, **pk)
    p_span = Params(max_segment_intron=max_intron, max_report_intron=max_intron, fusion_search=1 if args.fusion_search else 0,
                    fusion_min_dist=100000 if args.fusion_search else 10000000)
    n_fusions = [0]
    if args.fusion_search:
        ctx.upload_span_fusions(np.zeros(0, dtype=host.SPAN_FUSION_DTYPE))
    # first-inserted-wins priority of std::set<Insertion>: all left reads (rank order) before all right reads
    cb_left = cbatch_from_tensors(w["left"], rank * args.pairs)
    cb_right = cbatch_from_tensors(w["right"], world * args.pairs + rank * args.pairs)
    ctx.configure(1 << 22, 1 << 20)
    use_heads = not args.no_hit_heads
    sp_left = span_cbatch_from_tensors(w["left"], ctx if use_heads else None)
    sp_right = span_cbatch_from_tensors(w["right"], ctx if use_heads else None)

    # ---- the communicator: one RCCL rank per GPU (or the loopback transport in the in-process functional mode)
    comm = None
    if use_comm:
        if shared is not None:                        # ranks are threads of this process
            shared["ctxs"][rank] = ctx
            control.barrier()
            if rank == 0:
                shared["comms"] = host.Comm.create_local(shared["ctxs"])
            control.barrier()
            comm = shared["comms"][rank]
        else:
            uid = host.comm_unique_id() if rank == 0 else None
            if control is not None:
                uid = control.bcast_bytes(uid, host.COMM_ID_BYTES)
            comm = host.Comm.create(ctx, uid, world, rank)

    n_ium = int(args.coverage_search * args.pairs)
    cov_found = [0]
    local_juncs = [0]
    xchg_events = None         # (start, end) events around the exchange step of the timed steps

    host_t = {} if os.environ.get("THJ_BENCH_HOST_TIMING") == "1" else None      # developer switch: host seconds inside each call of a step, on stderr

    def timed(name, f, *a):
        if host_t is None:
            return f(*a)
        t = time.perf_counter()
        r = f(*a)
        host_t[name] = host_t.get(name, 0.0) + time.perf_counter() - t
        return r

    t0_first = os.environ.get("THJ_BENCH_T0_FIRST", "0") == "1" and os.environ.get("THJ_BENCH_NO_EARLY_T0") != "1" and os.environ.get("THJ_BENCH_NO_PAIR") != "1" and not args.fusion_search

    def step():
        # ---- segment_juncs stage
        ctx.reset()
        if n_ium:
            ctx.covsearch_reset()
        if t0_first:
            # stage 2's tier 0 needs nothing of stage 1: it goes out FIRST, on the side streams, and streams its 2 x 2.5 GB beside stage 1's
            # kernels (most of which wait on dependent gathers and leave the HBM idle)
            timed("span_reset", ctx.span_reset)
            timed("span_tier0_pair", ctx.span_tier0_pair, p_span, sp_left, sp_right)
        if os.environ.get("THJ_BENCH_NO_PAIR") == "1":
            ctx.run(p_left, cb_left)
            ctx.run(p_right, cb_right)
        else:
            timed("run_pair", ctx.run_pair, p_left, cb_left, p_right, cb_right)      # both sides as one call: their side chains run beside each other
        if n_ium:                                     # coverage search: coverage map of all hits, extension table, island pairing
            ctx.covsearch_add_hits(cb_left)
            ctx.covsearch_add_hits(cb_right)
            for sd in ("left", "right"):
                ctx.covsearch_add_reads_device(n_ium, w[sd]["W"], w[sd]["planes"].data_ptr(), w[sd]["read_len"].data_ptr())
            if comm is not None:
                comm.covsearch_allgather()            # OR of the ranks' coverage maps, concatenation of their extension tables
            ctx.covsearch_run(min(20, 25 - 2), 50, 20000)
            cov_found[0] = ctx.covsearch_finish()
        if comm is not None:
            if xchg_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            comm.events_allgather()                   # ONE ncclAllGather on the context stream, merge kernels behind it
            if xchg_events is not None:
                e1.record(stream)
                xchg_events.append((e0, e1))
        early_t0 = os.environ.get("THJ_BENCH_NO_EARLY_T0") != "1" and os.environ.get("THJ_BENCH_NO_PAIR") != "1" and not args.fusion_search
        if early_t0 and not t0_first:
            # stage 2's tier 0 (the reads whose hits abut end to end) needs no junction set: it goes out before stage 1's counts are
            # asked for, and the GPU has it while the host waits and the event lists are sorted (thj_span_tier0_pair_async)
            timed("span_reset", ctx.span_reset)
            timed("span_tier0_pair", ctx.span_tier0_pair, p_span, sp_left, sp_right)
        cnt = timed("finish", ctx.finish)             # the only host round trip of the stage
        if args.fusion_search and args.fusion_frac > 0:
            # segment_juncs --fusion-search: find_fusions over both sides; the (small) list goes to the spanning stage the way the
            # .fusions file would carry it
            n_fusions[0] = ctx.fusion_search([(p_left, cb_left), (p_right, cb_right)])
            ctx.span_fusions_from_segjuncs()                             # device to device, like the junction set below
        # ---- long_spanning_reads stage, fed device-to-device with the (global) junction set
        timed("span_sets_from_segjuncs", ctx.span_sets_from_segjuncs)
        if not early_t0:
            timed("span_reset", ctx.span_reset)
        if os.environ.get("THJ_BENCH_NO_PAIR") == "1":
            ctx.span_run(p_span, sp_left)
            ctx.span_run(p_span, sp_right)
        else:
            timed("span_run_pair", ctx.span_run_pair, p_span, sp_left, sp_right)          # both sides as one call: the two sides' kernels run beside each other
        try:
            n_alns = timed("span_finish", ctx.span_finish)
        except host.ThjError as e:
            # THJ_ERETRY: a pool or the workspace for reads with many joined alignments was set up by this call (a repeat family under
            # --fusion-search): the pass is run again, as the executables and host.Context.spanning do; it happens in the first warm-up step only
            if "run the pass again" not in str(e):
                raise
            ctx.span_reset()
            ctx.span_run_pair(p_span, sp_left, sp_right)
            n_alns = ctx.span_finish()
        return cnt, n_alns

    def barrier():
        if control is not None:
            control.barrier()
        torch.cuda.synchronize()

    if use_comm and os.environ.get("THJ_BENCH_VERIFY") == "1":
        ctx.reset()
        ctx.run(p_left, cb_left)
        ctx.run(p_right, cb_right)
        local_juncs[0] = ctx.finish().n_juncs          # what this rank finds alone
    for _ in range(args.warmup):
        cnt, n_alns = step()
    ctx.profile(True)
    ctx.profile_span(True)
    xchg_events = [] if comm is not None else None
    if host_t is not None:
        host_t.clear()
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        cnt, n_alns = step()
    barrier()
    elapsed = time.time() - t0
    if host_t is not None:
        print("[host] ms per step inside: " + ", ".join("%s %.3f" % (k, 1e3 * v / args.steps) for k, v in host_t.items()), file=sys.stderr)
    kern_ms, launches, sj_stats = ctx.profile(False)
    span_ms, span_launches = ctx.profile_span(False)
    torch.cuda.synchronize()
    # outside the timed region: the same step replayed with every kernel on one stream, for each kernel's OWN duration.  In the timed
    # region the two sides' kernels (and stage 1's side chains) run beside each other and share the GPU: a latency-bound kernel's
    # events then bracket two or three times its own time, which says how the step overlaps, not how good the kernel is
    kern_ms_alone = span_ms_alone = None
    if not use_comm and not n_ium and not os.environ.get("THJ_BENCH_NO_REPLAY"):
        ctx.profile_serial(True)
        step()
        ctx.profile(True)
        ctx.profile_span(True)
        for _ in range(max(2, min(args.steps, 5))):
            step()
        torch.cuda.synchronize()
        kern_ms_alone, _, _ = ctx.profile(False)
        span_ms_alone, _ = ctx.profile_span(False)
        ctx.profile_serial(False)
        torch.cuda.synchronize()
    stream_info = ctx.stream_info()              # what the queue probe measured for the side streams the steps ran on (thj_ctx_stream_info)
    comm_info = comm.info() if comm is not None else None
    if comm_info is not None and xchg_events:
        comm_info["us_per_step"] = 1e3 * sum(a.elapsed_time(b) for a, b in xchg_events) / len(xchg_events)      # pack + ncclAllGather + merge kernels, on the context stream
        comm_info["gathered_bytes_per_step"] = comm_info["bytes_per_rank"] * comm_info["n_ranks"]
    xchg_events = None
    if use_comm and os.environ.get("THJ_BENCH_VERIFY") == "1":
        # functional check of the exchange step: every rank must hold the same merged sets, no smaller than its own
        ev = ctx.download(cnt)
        import zlib
        h = [len(ev.juncs), zlib.crc32(ev.juncs.tobytes()), len(ev.deletions), zlib.crc32(ev.deletions.tobytes()),
             len(ev.insertions), zlib.crc32(repr(ev.insertions).encode())]
        allh = control.allgather_ints(h) if control is not None else [h]
        same = all(x == allh[0] for x in allh) and h[0] >= local_juncs[0]
        print("[verify] rank %d: %d junctions after the exchange step (%d alone), sets %s across %d ranks, transport %s" % (
            rank, h[0], local_juncs[0], "identical" if same else "DIFFER", world, comm_info["transport"]), file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(3)
    per_rank_ms = [elapsed * 1e3 / args.steps]
    if control is not None:
        per_rank_ms = [x[0] / 1e3 / args.steps for x in control.allgather_ints([int(elapsed * 1e6)])]
        elapsed = control.max(elapsed)

    # ---- roofline, per launch (one launch = one side's batch of `pairs` reads); DESIGN.md "Roofline" ---------
    rl_bytes = (args.read_len + 3) // 4 + (args.read_len + 7) // 8            # packed read: 2-bit bases + N mask
    n_launch = 2
    nseg = w["left"]["nseg"]
    # stage 1.  The classifying kernels (thj_k_sj_flat and, for the reads that are not flat, thj_k_sj_general / thj_k_segjuncs_shared)
    # stream 16 B per hit record + 4 B per (read, segment) CSR offset; thj_k_sj_tasks reads per RefSeg window two 64-B genome
    # lines and the read, per indel pair one genome line and the read, and writes 8 B per distinct event emitted
    # the reads with several hits in a segment (stage 1's second group takes them): their count and hit records, per launch
    sj_multi_reads = sj_multi_hits = 0
    sj_class = [[0.0, 0.0], [0.0, 0.0], [0.0, 0.0]]          # [reads, hits] per launch of the reads with <= 12, <= 32, more hits among them
    for sd in ("left", "right"):
        cells1 = (w[sd]["seg_off"][1:] - w[sd]["seg_off"][:-1]).reshape(args.pairs, nseg)
        mr1 = (cells1.max(dim=1).values > 1)
        if w[sd].get("mate_off") is not None:                # a read with more than two mate hits goes the general way too
            mo = w[sd]["mate_off"]
            mr1 = mr1 | ((mo[1:] - mo[:-1]) > 2)
        sj_multi_reads += int(mr1.sum()) / n_launch
        sj_multi_hits += int(cells1[mr1].sum()) / n_launch
        nh1 = cells1[mr1].sum(dim=1)
        for ci, sel in enumerate((nh1 <= 12, (nh1 > 12) & (nh1 <= 32), nh1 > 32)):
            sj_class[ci][0] += int(sel.sum()) / n_launch
            sj_class[ci][1] += int(nh1[sel].sum()) / n_launch
    cls_alg = 16.0 * (cnt.n_hits_read / n_launch - sj_multi_hits) + 4.0 * (args.pairs * nseg + 1)
    task_alg = (cnt.n_windows / n_launch) * (128 + rl_bytes) + (cnt.n_indel_pairs / n_launch) * (64 + rl_bytes) \
        + 8.0 * (cnt.n_juncs + cnt.n_deletions + cnt.n_insertions) / n_launch
    # stage 2, four kernels per launch.  Per finished read: 38 B of read planes, two 64-B genome lines (consistency
    # check + MD pass share them) and the 64-B lead line of its record; tier 0 streams every read's CSR row and 32-B hit records;
    # tiers 1/2/3 re-read CSR + hits of their worklist reads (4-B list entry each) and one 64-B line of junction keys
    # per closure.  Counters come from the kernels (tier sizes of the last launch, record count of the step).
    n_lean, n_multi, n_gen = ctx.span_tier_counts()
    n_chain = ctx.span_chain_count()                 # those of n_lean that travel as chain entries (tier 0 -> thj_k_join -> thj_k_finish)
    hits_per_read = float(int(w["left"]["span_off"][-1]) + int(w["right"]["span_off"][-1])) / (2.0 * args.pairs)
    # reads with a segment that has several hits carry most of the hit records of a mixed workload: the multihit tier's share of
    # the hit bytes is counted from them, not from the average read
    multi_reads = multi_hits = 0
    for sd in ("left", "right"):
        cells = (w[sd]["span_off"][1:] - w[sd]["span_off"][:-1]).reshape(args.pairs, nseg)
        mr = (cells.max(dim=1).values > 1)
        multi_reads += int(mr.sum())
        multi_hits += int(cells[mr].sum())
    hits_single = (hits_per_read * 2.0 * args.pairs - multi_hits) / max(1.0, 2.0 * args.pairs - multi_reads)      # per read without multihits
    hits_multi = multi_hits / max(1.0, float(multi_reads))                                                       # per read with
    rec_per_read = n_alns / (2.0 * args.pairs)
    # the record: one 64-B lead line (thj_aln_slot); the tail line is written only for > 4 cigar ops or an MD string of > 24
    # characters, which this workload's reads do not have
    per_read_done = rl_bytes + rec_per_read * (128 + 64)
    n_t0 = args.pairs - n_lean - n_multi
    hit_b = 16.0 if use_heads else 32.0              # tier 0 streams the dense 16-byte heads when the batch has them
    t0_alg = 4.0 * (args.pairs * nseg + 1) + hit_b * hits_per_read * args.pairs + n_t0 * per_read_done + 4.0 * (n_lean + n_multi)
    t1_alg = (n_lean - n_chain) * (4 + 4.0 * (nseg + 1) + 32.0 * hits_single + per_read_done + 64)
    # a chain entry: 32 B written by tier 0 and read by the join, the chain's hit records (32 B each), one 64-B line of junction keys, the
    # joined hit (32 B) written and read again
    join_alg = n_chain * (32.0 + 32.0 * hits_single + 64 + 32)
    fin_alg = n_chain * (32.0 + per_read_done)
    t2_alg = n_multi * (4 + 4.0 * (nseg + 1) + 32.0 * (hits_multi if multi_reads else hits_per_read) + per_read_done + 64) + 4.0 * n_gen
    t3_alg = n_gen * (4 + 4.0 * (nseg + 1) + 32.0 * hits_per_read + per_read_done + 64)
    # thj_k_segjuncs_rescue: per (hit, mate hit) pair the read, its CSR row + hits, the mate hit and ~3 genome lines of flank
    resc_alg = (cnt.n_rescue_pairs / n_launch) * (4 + 4.0 * (nseg + 1) + 16.0 * cnt.n_hits_read / (2.0 * args.pairs) + 16 + rl_bytes + 192)
    # stage 1's kernels one by one (HIP events around each, on the stream it runs on).  Bytes: the flat kernel streams the CSR and the
    # hit records of its reads; the kernels of the reads with several hits a segment read the list entry, CSR row and hit records of
    # theirs (classes by hit count, as thj_k_sj_flat lists them: <= 12, <= 32, more); the rescue pairs and the tasks are split between
    # the flat reads' kernels and the others' by the counts the kernels keep (flat rescue pairs, tasks in the list)
    tasks_per_launch = max(1.0, (cnt.n_windows + cnt.n_indel_pairs) / n_launch)
    list_share = min(1.0, sj_stats["list_tasks"] / tasks_per_launch)
    resc_pairs = max(1.0, cnt.n_rescue_pairs / n_launch)
    flat_resc_share = min(1.0, sj_stats["flat_rescue_pairs"] / resc_pairs)
    per_multi = 4 + 4.0 * (nseg + 1)

    def sj_entries(task_b, resc_b):
        return [cls_alg, sj_class[0][0] * per_multi + 16.0 * sj_class[0][1], sj_class[1][0] * per_multi + 16.0 * sj_class[1][1],
                sj_class[2][0] * per_multi + 16.0 * sj_class[2][1], (1.0 - flat_resc_share) * resc_b, list_share * task_b,
                flat_resc_share * resc_b, (1.0 - list_share) * task_b]
    kernels = [{"kernel": nm, "avg_kernel_ms": kern_ms[i], "launches": launches, "algorithmic_bytes_per_launch": v}
               for i, (nm, v) in enumerate(zip(host.Context.SJ_KERNELS, sj_entries(task_alg, resc_alg)))]
    for i in (1, 2, 3, 4, 5):
        kernels[i]["stream"] = "side stream: runs beside the flat reads' kernels (the last three entries)"
    # the multihit reads: with the chains on (reads of up to four segments, no fusion search) thj_k_chains looks at every one of them
    # (CSR row + 16-B heads) and turns those whose chains need no search into chain entries -- their join and finish are thj_k_join's
    # and thj_k_finish's work from then on, the packed tier keeps the rest.  The split of the multihit reads' bytes between the two
    # follows the reads (span_ms[6] still brackets the packed tier alone)
    chains_on = n_chain > 0
    n_grp = ctx.span_chain_groups() if chains_on else 0          # multihit reads that travel as chain entries
    grp_share = min(1.0, n_grp / max(1.0, float(n_multi)))
    ch_alg = n_multi * (4 + 4.0 * (nseg + 1) + 16.0 * (hits_multi if multi_reads else hits_per_read)) + grp_share * n_multi * 32.0 * (hits_multi / max(1.0, nseg) if multi_reads else 1.0)
    kernels += [
        {"kernel": "thj_k_stitch_contig", "avg_kernel_ms": span_ms[0], "launches": span_launches, "algorithmic_bytes_per_launch": t0_alg + 32.0 * n_chain},
        {"kernel": "thj_k_chains", "avg_kernel_ms": span_ms[1], "launches": span_launches, "algorithmic_bytes_per_launch": ch_alg if chains_on else 0.0},
        {"kernel": "thj_k_join", "avg_kernel_ms": span_ms[2], "launches": span_launches, "algorithmic_bytes_per_launch": join_alg + grp_share * t2_alg * 0.4},
        {"kernel": "thj_k_join_closure", "avg_kernel_ms": span_ms[3], "launches": span_launches, "algorithmic_bytes_per_launch": 0.15 * join_alg},
        {"kernel": "thj_k_finish", "avg_kernel_ms": span_ms[4], "launches": span_launches, "algorithmic_bytes_per_launch": fin_alg + grp_share * t2_alg * 0.6},
        {"kernel": "thj_k_stitch", "avg_kernel_ms": span_ms[5], "launches": span_launches, "algorithmic_bytes_per_launch": t1_alg},
        {"kernel": "thj_k_stitch_pack", "avg_kernel_ms": span_ms[6], "launches": span_launches, "algorithmic_bytes_per_launch": (1.0 - grp_share) * t2_alg},
        {"kernel": "thj_k_stitch_fusion" if args.fusion_search else "thj_k_stitch_generic", "avg_kernel_ms": span_ms[7], "launches": span_launches, "algorithmic_bytes_per_launch": t3_alg},
    ]
    for i in range(len(kernels) - 6, len(kernels) - 3):
        kernels[i]["stream"] = "side stream: runs beside the kernels of the reads that do not travel as chain entries, and beside the other side's"
    # The same kernels in SURVEY 8(d)'s byte terms -- what the ALGORITHM has to move, whatever layout a build chose: 16 B per hit
    # record (this build's stage-2 record is 32 B), the packed read, <= 128 B of genome per window / per joined hit, one 64-B
    # line of junction keys per closure, 16 B per candidate event, 32 + 8 x ncigar B per joined alignment (this build writes a
    # 64-B line, two for a long record).  `frac` below is computed from THESE; the layout-byte figure stays next to it as frac_layout.
    cig_per_rec = 1.0 + 2.0 * (n_lean + n_multi) / max(1.0, float(args.pairs))         # contiguous: 1 op; one closure: 3
    out_rec = 32.0 + 8.0 * cig_per_rec
    done_8d = rl_bytes + rec_per_read * (128.0 + out_rec)
    cls_8d = cls_alg
    task_8d = (cnt.n_windows / n_launch) * (128 + rl_bytes) + (cnt.n_indel_pairs / n_launch) * (64 + rl_bytes) \
        + 16.0 * (cnt.n_juncs + cnt.n_deletions + cnt.n_insertions) / n_launch
    t0_8d = 4.0 * (args.pairs * nseg + 1) + 16.0 * hits_per_read * args.pairs + n_t0 * done_8d + 4.0 * (n_lean + n_multi)
    t1_8d = (n_lean - n_chain) * (4 + 4.0 * (nseg + 1) + 16.0 * hits_single + done_8d + 64)
    # the chain reads in 8(d)'s terms: the join needs the hits (16 B each; the entry carries them) and one line of junction keys, the
    # finish the read, <= 128 B of genome and the alignment it writes; the joined hit handed from one to the other is layout
    join_8d = n_chain * (16.0 * hits_single + 64)
    fin_8d = n_chain * done_8d
    t2_8d = n_multi * (4 + 4.0 * (nseg + 1) + 16.0 * (hits_multi if multi_reads else hits_per_read) + done_8d + 64) + 4.0 * n_gen
    t3_8d = n_gen * (4 + 4.0 * (nseg + 1) + 16.0 * hits_per_read + done_8d + 64)
    resc_8d = (cnt.n_rescue_pairs / n_launch) * (4 + 4.0 * (nseg + 1) + 16.0 * cnt.n_hits_read / (2.0 * args.pairs) + 16 + rl_bytes + 128)
    ch_8d = n_multi * (4.0 * (nseg + 1) + 16.0 * (hits_multi if multi_reads else hits_per_read)) if chains_on else 0.0
    for k, b8 in zip(kernels, sj_entries(task_8d, resc_8d) + [t0_8d, ch_8d, join_8d + grp_share * t2_8d * 0.4, 0.15 * join_8d, fin_8d + grp_share * t2_8d * 0.6, t1_8d, (1.0 - grp_share) * t2_8d, t3_8d]):
        k["algorithmic_bytes_8d_per_launch"] = b8
    for k in kernels:
        k["achieved_layout"] = k["algorithmic_bytes_per_launch"] / (k["avg_kernel_ms"] * 1e-3) / 1e9 if k["avg_kernel_ms"] > 0 else 0.0
        k["frac_layout"] = k["achieved_layout"] / HBM_PEAK_GBS
        k["achieved"] = k["algorithmic_bytes_8d_per_launch"] / (k["avg_kernel_ms"] * 1e-3) / 1e9 if k["avg_kernel_ms"] > 0 else 0.0
        k["frac"] = k["achieved"] / HBM_PEAK_GBS
    # HBM traffic from the PMC counters cannot be sampled inside this process: when rocprofv3 is on PATH the same workload is
    # run again for two steps under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, pmc_traffic_in_run);
    # otherwise the table committed under profiles/ is used, and only when the configuration is the one it was taken on.  Per MI355X_MICROARCH.md
    # FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950: traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB,
    # traffic_low = (FETCH_SIZE + WRITE_SIZE) KB (exact for narrow accesses; see the calibration note in the profile).
    traffic_in_run = False
    try:
        pm = pmc_traffic_in_run(args) if rank == 0 and world == 1 else None
        traffic_in_run = pm is not None
        if pm is None:
            pm = json.load(open(os.path.join(ROOT, "profiles", os.environ.get("THJ_PMC_FILE", "r04_plain_pmc_traffic.json" if args.multihit_frac == 0 and args.indel_frac == 0 else "r06_pmc_traffic.json"))))
        want_cfg = {"pairs_per_gpu": args.pairs, "genome_len": genome_len, "exon_len": args.exon_len}
        if args.multihit_frac > 0 or args.indel_frac > 0:
            want_cfg.update(multihit_frac=args.multihit_frac, indel_frac=args.indel_frac, max_copies=args.max_copies)
        if traffic_in_run or (args.read_len == 100 and args.genome == "chr20" and use_heads and not args.fusion_search and not n_ium and pm["config"] == want_cfg):
            for k in kernels:
                def _lookup(nm):
                    # the profiler prints every template argument (thj_k_sj_general<8, 256, true, 9, 4>); the line's names stop at the ones that tell the instances apart
                    v = pm["kernels"].get(nm)
                    if v is None and nm.endswith(">"):
                        cands = [key for key in pm["kernels"] if key.startswith(nm[:-1] + ",")]
                        v = pm["kernels"][cands[0]] if len(cands) == 1 else None
                    return v
                parts = [_lookup(nm) for nm in k["kernel"].split(" + ")]
                c = None if not any(parts) else {key: sum(x.get(key, 0.0) for x in parts if x) for key in ("FETCH_SIZE", "WRITE_SIZE")}
                if c:
                    k["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                    k["traffic_low"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                    k["traffic_source"] = pm["source"]
    except (OSError, ValueError, KeyError):
        pass
    # (a tier nothing was routed to still launches, on an empty work-list: it is listed with "no_work" and cannot be the dominant kernel)
    for k in kernels:
        if k["algorithmic_bytes_8d_per_launch"] <= 0:
            k["no_work"] = True
    # the dominant kernel: the one with the largest share of a step's kernel time (launches x average duration) -- a kernel that gets
    # faster can only lose the title to one that now takes longer than it (round 4 picked the longest single launch, and the headline
    # fraction fell from 0.43 to 0.18 when the kernel that had it got faster); `frac_step` beside it is the whole step's figure: the
    # algorithmic bytes of every launch over the step's wall time
    alone = (list(kern_ms_alone) + list(span_ms_alone)) if kern_ms_alone is not None else None
    for i, k in enumerate(kernels):
        k["avg_kernel_ms_alone"] = alone[i] if alone is not None else None
        t_own = alone[i] if alone is not None and alone[i] > 0 else k["avg_kernel_ms"]
        k["frac_alone"] = k["algorithmic_bytes_8d_per_launch"] / (t_own * 1e-3) / 1e9 / HBM_PEAK_GBS if t_own > 0 else 0.0
    dom = max((k for k in kernels if not k.get("no_work")), key=lambda k: (k["avg_kernel_ms_alone"] if k["avg_kernel_ms_alone"] else k["avg_kernel_ms"]) * k["launches"])
    step_bytes = sum(k["algorithmic_bytes_8d_per_launch"] * k["launches"] for k in kernels) / max(1, args.steps)
    own = lambda k: k["avg_kernel_ms_alone"] if k["avg_kernel_ms_alone"] else k["avg_kernel_ms"]
    kernel_ms_per_step = sum(own(k) * k["launches"] for k in kernels) / max(1, args.steps)
    for k in kernels:
        k["share_of_kernel_time"] = own(k) * k["launches"] / max(1, args.steps) / max(1e-9, kernel_ms_per_step)

    workload_text = ("%s: %d x 2x%d bp PE synthetic vs %d bp %s genome per GPU, inputs resident in HBM; both stages on device: "
                     "segment_juncs (main + rescue kernels, event dedup+sort%s) then long_spanning_reads (four stitch tiers fed "
                     "device-to-device with the junction set; records land in BAM order)"
                     % (("configs[1]" + ("" if args.multihit_frac == 0 and args.indel_frac == 0 else " with SURVEY 8(d)'s mix (%g %% of the pairs from a %d-copy repeat family -- 2 hits a segment for 85 %% of them, 3..8 for 12 %%, 9..40 for 2.7 %%, 41 for 0.3 %% --, "
                                          "%g %% deletion reads)" % (100 * args.multihit_frac, args.max_copies, 100 * args.indel_frac)))
                        if args.read_len == 100 and args.genome == "chr20" and not n_ium and not args.fusion_search else
                        ("configs[2]'s shard (100 M pairs over %d GPUs)" % world if args.config == 3 and not args.pairs_given else
                         "configs[2]'s per-GPU shard at 8 GPUs" if args.config == 3 and args.pairs == 12_500_000 else "shape of another config"), args.pairs,
                        args.read_len, genome_len, "chr20-sized" if args.genome == "chr20" else "GRCh38-sized (25 contigs)",
                        ", one RCCL all-gather of the event sets inside the C ABI" if use_comm else ""))
    if not use_heads:
        workload_text += "; stage 2 batches WITHOUT the dense hit-head array (32-byte records only)"
    if n_ium:
        workload_text += "; with the coverage search (first %d reads of each side as --ium-reads, %d coverage junctions)" % (n_ium, cov_found[0])
    # the line's own name of the workload (<= 200 characters; the long text above is config.workload_detail in the detail file)
    workload_short = "%s: %d x 2x%d bp PE synthetic vs %s genome (%d bp) per GPU, resident in HBM; segment_juncs then long_spanning_reads on device" % (
        ("configs[1], SURVEY 8(d) mix" if args.multihit_frac > 0 or args.indel_frac > 0 else "configs[1]") if args.read_len == 100 and args.genome == "chr20" and not n_ium and not args.fusion_search
        else ("configs[2] shard" if args.config == 3 else "other shape"), args.pairs, args.read_len, "chr20-sized" if args.genome == "chr20" else "GRCh38-sized", genome_len)
    if n_ium:
        workload_short += "; coverage search (%d coverage junctions)" % cov_found[0]
    if args.fusion_search:
        workload_short += "; --fusion-search"
    result = None
    if rank == 0:
        # what this box's HBM sustains on a plain copy (quote it beside the 8 TB/s spec peak): 2 GiB device-to-device
        # with torch's copy kernel, bytes = read + written; outside the timed region
        hbm_copy = None
        try:
            src = torch.empty(2 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            c0.record()
            for _ in range(10):
                dst.copy_(src)
            c1.record()
            torch.cuda.synchronize()
            hbm_copy = 2.0 * src.numel() * 10 / c0.elapsed_time(c1) / 1e6
            del src, dst
        except RuntimeError:
            pass
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N=1 only
            import orc
            from tophat_amd.batch import events_to_span_inputs, merge_events
            m = min(args.cpu_sample, args.pairs)
            og = orc.Genome(strs)
            sb_l, sb_r = sample_segbatch(w["left"], m), sample_segbatch(w["right"], m)
            sp_l, sp_r = sample_spanbatch(w["left"], m), sample_spanbatch(w["right"], m)
            t1 = time.time()
            e_l = orc.segjuncs(p_left, og, sb_l)
            e_r = orc.segjuncs(p_right, og, sb_r)
            jj, ii = events_to_span_inputs(merge_events(e_l, e_r))
            t2 = time.time()
            lib_o = orc._lib()
            n_rec = orc.spanning_count(p_span, og, sp_l, jj, ii) + orc.spanning_count(p_span, og, sp_r, jj, ii)
            t3 = time.time()
            dt = t3 - t1
            cpu = {"value": m / dt, "unit": "read-pairs/s", "cores": 1, "kind": "port",
                   # the port is not the reference: it leaves out the reference's O(window) copies (52 % of segment_juncs' profile) and its
                   # FASTQ / SAM text handling.  What the reference itself did where it could be built:
                   "reference_calibration": {"value": 11000.0, "unit": "read-pairs/s per core",
                                             "what": "the reference's own segment_juncs (41 k reads/s) + long_spanning_reads (52 k reads/s), g++ 11.4 -O2, one thread of "
                                                     "an 8-vCPU container, 200 000 x 100 bp reads on a 1 Mb genome, text-SAM maps (survey-stage scratch build with stand-in "
                                                     "Boost headers: a calibration, not a pin)",
                                             "port_over_reference": None},
                   "sample": "first %d pairs of the same synthetic batch through both stages with oracle/liborc.so "
                             "(plain-C restatement, 1 thread): segment_juncs %.1f s + long_spanning_reads %.1f s, %d records" % (
                                 m, t2 - t1, t3 - t2, n_rec)}
            cpu["reference_calibration"]["port_over_reference"] = cpu["value"] / 11000.0
            del sb_l, sb_r, sp_l, sp_r
            # the same sample on all the cores the container gives us: the sample cut into one chunk per core, the chunks' event sets
            # merged between the stages (what a multi-threaded host run of the reference does, segment_juncs.cpp:4776-4922)
            from concurrent.futures import ThreadPoolExecutor
            C = min(64, _cpu_count())
            if C > 1:
                per = (m + C - 1) // C
                chunks = [(k * per, min(per, m - k * per)) for k in range(C) if k * per < m]
                pre = [(sample_segbatch(w["left"], n_, s_), sample_segbatch(w["right"], n_, s_), sample_spanbatch(w["left"], n_, s_),
                        sample_spanbatch(w["right"], n_, s_)) for s_, n_ in chunks]
                with ThreadPoolExecutor(C) as ex:        # ctypes releases the GIL inside the oracle
                    t4 = time.time()
                    evs = list(ex.map(lambda c_: merge_events(orc.segjuncs(p_left, og, c_[0]), orc.segjuncs(p_right, og, c_[1])), pre))
                    ev_all = evs[0]
                    for e_ in evs[1:]:
                        ev_all = merge_events(ev_all, e_)
                    jj2, ii2 = events_to_span_inputs(ev_all)
                    t5 = time.time()
                    n_rec2 = sum(ex.map(lambda c_: orc.spanning_count(p_span, og, c_[2], jj2, ii2) + orc.spanning_count(p_span, og, c_[3], jj2, ii2), pre))
                    t6 = time.time()
                cpu["all_cores"] = {"value": m / (t6 - t4), "unit": "read-pairs/s", "cores": C, "kind": "port",
                                    "sample": "the same %d pairs in %d chunks on %d threads: segment_juncs %.1f s + long_spanning_reads %.1f s, %d records" % (
                                        m, len(chunks), C, t5 - t4, t6 - t5, n_rec2)}
                del pre
        e2e = _E2E_EARLY
        if e2e is None and args.e2e_pairs > 0 and world == 1 and args.read_len == 100 and args.genome == "chr20":
            e2e = e2e_leg(args)
        if cpu is not None and e2e and e2e.get("cpu_files_to_files"):
            # beside `e2e`: the same files through the same host code with the oracle in the device's place
            cpu["files_to_files"] = e2e["cpu_files_to_files"]
        result = {
            "metric": "paired reads/sec through segment_juncs+long_spanning_reads; junctions.bed diff=0",
            "value": args.pairs * world * args.steps / elapsed,
            "value_is": "resident-data kernel rate (both stages, batches in HBM); files in -> files out is metric_e2e",
            "unit": "read-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.config == 3 and not args.pairs_given else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload_short, "workload_detail": workload_text,
                       "pairs_per_gpu": args.pairs, "segment_length": 25, "genes": int(genes.shape[0]),
                       "fusion_search": bool(args.fusion_search), "fusion_frac": args.fusion_frac, "indel_frac": args.indel_frac, "multihit_frac": args.multihit_frac, "max_copies": args.max_copies if args.multihit_frac > 0 else 1, "fusions_found": n_fusions[0], "reads_to_tiers_1_2or_fusion_3": [int(n_lean), int(n_multi), int(n_gen)],
                       "parallelism": "reads sharded x%d, genome replicated" % world},
            "roofline": {"bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom["achieved"] / HBM_PEAK_GBS,
                         "frac_step": step_bytes / max(1e-9, elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "picked_by": "largest share of a step's kernel time, each kernel at its own duration (avg_kernel_ms_alone: the step replayed on one stream after the "
                                      "timed region); `frac` / `avg_kernel_ms` are from the timed region, where the two sides' kernels share the GPU",
                         "share_of_kernel_time": dom["share_of_kernel_time"], "avg_kernel_ms_alone": dom["avg_kernel_ms_alone"], "frac_alone": dom["frac_alone"],
                         "traffic": dom.get("traffic"), "traffic_low": dom.get("traffic_low"),
                         "traffic_source": dom.get("traffic_source"), "traffic_measured_in_run": bool(traffic_in_run and dom.get("traffic") is not None), "kernel": dom["kernel"],
                         "avg_kernel_ms": dom["avg_kernel_ms"], "launches": dom["launches"],
                         "byte_terms": "SURVEY 8(d): 16 B/hit, packed read, <=128 B genome per window or joined hit, 32+8*ncigar B per alignment",
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_8d_per_launch"],
                         "achieved_layout_bytes": dom["achieved_layout"], "frac_layout_bytes": dom["frac_layout"],
                         "layout_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                         "measured_copy_GBs": hbm_copy, "frac_of_measured_copy": dom["achieved"] / hbm_copy if hbm_copy else None,
                         # the two sides of a step ran on side streams measured to have hardware queues of their own (false: same results, the
                         # sides partly one after the other -- the line is then not the build's best)
                         "streams_independent": bool(stream_info["n_side"] >= 2 and all(stream_info["independent"][:2])), "stream_overlap_ratio": stream_info["ratio"]},
            # all kernels of a step together: algorithmic bytes of every launch / time spent in them
            "roofline_all_kernels": {"achieved": sum(k["algorithmic_bytes_8d_per_launch"] * k["launches"] for k in kernels)
                                     / max(1e-9, sum(k["avg_kernel_ms"] * k["launches"] for k in kernels)) / 1e6,
                                     "achieved_layout_bytes": sum(k["algorithmic_bytes_per_launch"] * k["launches"] for k in kernels)
                                     / max(1e-9, sum(k["avg_kernel_ms"] * k["launches"] for k in kernels)) / 1e6,
                                     # two of stage 1's groups run side by side, so the sum above counts some time twice; the same bytes over the step's wall time:
                                     "achieved_over_step_time": sum(k["algorithmic_bytes_8d_per_launch"] * k["launches"] for k in kernels) / max(1, args.steps) / max(1e-9, elapsed / args.steps) / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s"},
            "kernels": kernels,
            # the exchange step as it ran: ranks, transport ("rccl" across processes / GPUs, "loopback" for contexts on one device), calls and
            # bytes; per-rank step times (the line's ms_per_step is their maximum)
            "exchange": None if comm_info is None else dict(comm_info, ranks=comm_info["n_ranks"]),
            "per_rank_ms_per_step": per_rank_ms,
            # the metric as BASELINE words it (wall clock of the executables, files in -> files out); "value" above is the rate of the
            # kernels on data resident in HBM, as the bench contract defines it
            "metric_e2e": None if not e2e else {"value": e2e.get("value"), "unit": "read-pairs/s, files in -> files out, both executables, %d GPU%s" % (e2e.get("n_gpus", 1), "" if e2e.get("n_gpus", 1) == 1 else "s"),
                                                "value_no_handoff": e2e.get("value_no_handoff"), "checked_against_oracle": e2e.get("ok")},
            "e2e": e2e,
            "cpu_baseline": cpu,
            "events": {"junctions": cnt.n_juncs, "deletions": cnt.n_deletions, "insertions": cnt.n_insertions,
                       "windows_per_step": cnt.n_windows, "rescue_pairs_per_step": cnt.n_rescue_pairs,
                       "overflow_blocks": cnt.n_overflow_blocks, "spanning_records_per_step": n_alns},
            "gen_seconds": t_gen,
        }
    if comm is not None and shared is None:
        comm.close()
    if shared is not None:
        control.barrier()
        if rank == 0:
            for c_ in shared["comms"]:
                c_.close()
        control.barrier()
    ctx.close()
    if control is not None:
        control.close()
    return result


if __name__ == "__main__":
    main()
