#!/usr/bin/env python3
"""bench.py -- throughput of the sdrx hot path on MI355X, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|decim64|chan32|chan128|cfg4|fi64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no torchrun around it: bench.py itself starts N worker processes (one per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE anything touches HIP, and rank 0 prints the line.
On a box with fewer GPUs than ranks the workers rendezvous over gloo and share the cards (a rehearsal, flagged).

A "step" of the default (headline) workload is ONE pass of the metric's path over one batch of synthetic int16
I/Q already resident in HBM:   decimate64_cen (BASELINE.json configs[1], Decimators<qint32,qint16,16,12>)   AND
the 32-channel DownChannelizer bank (configs[2], 61.44 MS/s-shaped stream) over the same batch.  `value` =
samples per second that went through BOTH (batch x steps / elapsed).  Every rank owns one GPU and one independent
stream (streams shard, no collective on the data path); per-GPU work is fixed -> "weak" scaling.
`also.chan128` is BASELINE configs[4]'s per-GPU share (128 channels per stream), measured with the same protocol.

roofline : per kernel, ALGORITHMIC bytes (4 B read + 4/2^n B written per input sample and channel)
           / its average duration from HIP events on the launch stream (sdrx_*_get_timing) against 8 TB/s HBM3E.
           `frac` of the headline is the LOWER of the two kernels' fractions; `bound` says what the SQ counters
           show (VALU issue, profiles/), `valu_ceiling_frac` is the HBM fraction at which integer-MAC issue alone
           (v_dot2c_i32_i16: 2 taps per 4 SIMD cycles, profiles/r01_valu_issue_rates.txt) caps the kernel.
cpu_baseline : the reference's own code (oracle/_ref/libsdrref_fast.so: the reference's classes compiled from
           /root/reference with the reference's flags -O3 -ffast-math -ftree-vectorize -msse4.1, built in the
           build container; kind "reference") or, if that .so did not travel, the oracle port
           (oracle/libsdro_fast.so; kind "port"), on this box's host cores, bounded sample.  The oracle is only
           the checker/baseline: nothing on the measured GPU path touches it.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet (MI355X_MICROARCH.md); ~6300 achievable
# integer MAC issue: 256 CUs x 4 SIMDs x 64 lanes / 4 cycles per v_dot2c (2 taps) at the 2.4 GHz maximum clock
DOT2_PER_S = 256 * 4 * 16 * 2.4e9
METRIC = "MS/s complex int16 IQ through decim-64 + DownChannelizer, 1/2/4/8 GPU; % HBM roofline"


# ------------------------------------------------------------------------------------------------ host side
def host_threads():
    """threads the CPU baseline may use: the cores this process is allowed on (cpuset / cgroup quota aware)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _ref_lib():
    """(lib, kind): the reference compiled with its own flags, else the strict reference build, else None"""
    for name in ("libsdrref_fast.so", "libsdrref.so"):
        p = os.path.join(ROOT, "oracle", "_ref", name)
        if os.path.exists(p):
            try:
                L = C.CDLL(p)
                L.ref_decim_new.restype = C.c_void_p; L.ref_decim_new.argtypes = [C.c_int]
                L.ref_decim_process.restype = C.c_int
                L.ref_decim_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32, C.c_void_p]
                L.ref_chain_new.restype = C.c_void_p; L.ref_chain_new.argtypes = [C.c_int, C.c_void_p]
                L.ref_chain_feed.restype = C.c_int64; L.ref_chain_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
                return L, "reference", name
            except (OSError, AttributeError):
                continue
    return None, "port", "libsdro_fast.so"


def _port_lib():
    from tests import oracle_py as orc
    fast = os.path.join(ROOT, "oracle", "libsdro_fast.so")
    if not os.path.exists(fast):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsdro_fast.so"])
    return orc.lib(fast=True)


def _run_threads(n_threads, work):
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0


def cpu_decim(log2: int, sample_cplx: int, reps: int, n_thr: int, seed: int = 1234):
    """decimateK_cen <16,12> (K = 2^log2): one thread as sdrangelbench runs it (sdrbench/mainbench.cpp:83-104: the same
    buffer, `reps` repetitions, state carried), then one independent stream per thread on n_thr threads."""
    import numpy as np
    from tests import oracle_py as orc
    x = orc.synth_iq(sample_cplx, seed=seed, amp=2047)
    L, kind, so = _ref_lib()
    if L is not None:
        mk = lambda: L.ref_decim_new(12)
        if hasattr(L, "ref_decim_run"):
            # the call as sdrangelbench times it: the output vector is allocated once and the result stays in it
            L.ref_decim_run.restype = C.c_int; L.ref_decim_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32]
            run = lambda h, buf, out: L.ref_decim_run(h, log2, 2, buf.ctypes.data, buf.size)
        else:
            run = lambda h, buf, out: L.ref_decim_process(h, log2, 2, buf.ctypes.data, buf.size, out.ctypes.data)
    else:
        P = _port_lib()
        mk = lambda: P.sdro_decim_new(log2, 2, 12)
        run = lambda h, buf, out: P.sdro_decim_process(h, buf.ctypes.data, buf.size, out.ctypes.data)

    def timed(nt):
        hs = [mk() for _ in range(nt)]
        bufs = [x.copy() for _ in range(nt)]
        outs = [np.empty(sample_cplx // (1 << log2) * 2 + 64, np.int16) for _ in range(nt)]
        for i in range(nt):
            run(hs[i], bufs[i][: 2 * 65536], outs[i])          # touch / warm
        def work(i):
            for _ in range(reps):
                run(hs[i], bufs[i], outs[i])
        return nt * reps * sample_cplx / _run_threads(nt, work) / 1e6

    one = timed(1)
    allc = timed(n_thr) if n_thr > 1 else one
    return {"value": round(allc, 2), "unit": "MS/s", "cores": n_thr, "kind": kind, "lib": so, "single_thread_MSps": round(one, 2),
            "sample": f"decimate{1 << log2}_cen <16,12> on {sample_cplx} synthetic int16 I/Q samples x {reps} repetitions per thread, "
                      f"one independent stream per thread ({n_thr} threads); single_thread_MSps = 1 thread, as sdrangelbench runs it"}


def cpu_chan(fcs, sample_cplx: int, n_thr: int):
    """Reference DownChannelizer stage chains (IntHalfbandFilterEO<qint32,qint32,48> objects driven by the loop of
    DownChannelizer::feed), every channel re-filtering the full-rate stream on its own thread like
    ThreadedBasebandSampleSink does (threadedbasebandsamplesink.cpp:74-78); value = input MS/s the host sustains
    for the WHOLE bank."""
    import numpy as np
    from tests import oracle_py as orc
    n_thr = max(1, min(n_thr, len(fcs)))
    x = orc.synth_iq(sample_cplx, seed=77, amp=2047)
    plans = [orc.chan_plan(61_440_000, 48000, int(fc))[0] for fc in fcs]
    L, kind, so = _ref_lib()
    if L is not None:
        hs = [L.ref_chain_new(len(m), np.ascontiguousarray(m).ctypes.data) for m in plans]
        feed = lambda h, out: L.ref_chain_feed(h, x.ctypes.data, sample_cplx, out.ctypes.data)
    else:
        P = _port_lib()
        hs = [P.sdro_chain_new(len(m), np.ascontiguousarray(m).ctypes.data) for m in plans]
        feed = lambda h, out: P.sdro_chain_feed(h, x.ctypes.data, sample_cplx, out.ctypes.data)
    outs = [np.empty(sample_cplx // 64 + 64, np.int16) for _ in hs]
    def work(t):
        for c in range(t, len(hs), n_thr):
            feed(hs[c], outs[c])
    dt = _run_threads(n_thr, work)
    return {"value": round(sample_cplx / dt / 1e6, 3), "unit": "MS/s", "cores": n_thr, "kind": kind, "lib": so,
            "sample": f"{len(fcs)} DownChannelizer chains (one per channel, each over the full-rate stream) on {sample_cplx} synthetic samples, "
                      f"{n_thr} host threads, channels dealt round-robin; value = input rate sustained for the whole bank"}


def cpu_fi(sample_cplx: int, reps: int, n_thr: int):
    """Reference (or port) DecimatorsFI::decimate64_cen, one stream per thread, on the host cores."""
    import numpy as np
    from tests import oracle_py as orc
    x = (orc.synth_iq(sample_cplx, seed=99, amp=2047).astype(np.float32) / np.float32(4096.0))
    L, kind, so = _ref_lib()
    if L is not None:
        try:
            L.ref_fdecim_new.restype = C.c_void_p; L.ref_fdecim_new.argtypes = [C.c_int] * 3
            L.ref_fdecim_process.restype = C.c_int; L.ref_fdecim_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32, C.c_void_p]
        except AttributeError:
            L = None
    if L is not None:
        mk = lambda: L.ref_fdecim_new(0, 0, 16)
        run = lambda h, out: L.ref_fdecim_process(h, 6, 2, x.ctypes.data, x.size, out.ctypes.data)
    else:
        O = orc.lib(); orc._sig_fdecim(O); kind, so = "port", "libsdro.so"
        mk = lambda: O.sdro_fdecim_new(6, 2, 0, 0, 16)
        run = lambda h, out: O.sdro_fdecim_process(h, x.ctypes.data, x.size, out.ctypes.data)

    def timed(nt):
        hs = [mk() for _ in range(nt)]
        outs = [np.empty(x.size + 64, np.int16) for _ in range(nt)]
        def work(i):
            for _ in range(reps):
                run(hs[i], outs[i])
        return nt * reps * sample_cplx / _run_threads(nt, work) / 1e6

    one = timed(1)
    allc = timed(n_thr) if n_thr > 1 else one
    return {"value": round(allc, 2), "unit": "MS/s", "cores": n_thr, "kind": kind, "lib": so, "single_thread_MSps": round(one, 2),
            "sample": f"DecimatorsFI::decimate64_cen on {sample_cplx} synthetic float I/Q samples x {reps} passes per thread, one stream per thread ({n_thr} threads)"}


def load_traffic(kernel: str, batch: int, workload: str):
    """(HBM bytes per step, source file) from the newest committed rocprofv3 --pmc summary (profiles/*traffic*.json) that
    holds this kernel at this batch; (None, None) otherwise.  PMC passes cannot run inside the timed bench (separate
    rocprofv3 runs, MI355X_MICROARCH.md), so the figure is labelled with the file it comes from."""
    import glob
    best = (None, None)
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        for e in d.get("kernels", []):
            if e.get("workload") == workload and e.get("batch_cplx") == batch and e.get("kernel", "").startswith(kernel.split("<")[0]):
                best = (e.get("hbm_bytes_per_step", e.get("hbm_bytes_per_launch")), "profiles/" + os.path.basename(p))
    return best


# ------------------------------------------------------------------------------------------------ N-rank launcher
def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without torchrun: N worker processes, one per GPU.  Called before this process has
    loaded libsdrx.so / torch or made any HIP call; the workers are plain children (no exec of a GPU-initialised process)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        for p in procs:
            r = p.wait()
            rc = rc or r
            if r:                                           # one rank failed: the others would wait at the rendezvous
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


# ------------------------------------------------------------------------------------------------ main
def chan_centres(n_ch: int, workload: str):
    import numpy as np
    k = np.arange(n_ch, dtype=np.float64)
    if workload == "cfg4":
        return (-25_000_000 + k * (50_000_000 / 255)).astype(np.int64).tolist()
    return (-15_000_000 + k * (30_000_000 / (n_ch - 1)) + 137 * k).astype(np.int64).tolist()     # SURVEY.md 8d, cfg 3


def bank_figures(bank, n_ch):
    """algorithmic bytes per input sample (4 + sum_c 4/2^n_c) and dot2 issue per input sample of the stage trie"""
    depth = 0.0
    nodes = {}
    for c in range(n_ch):
        modes = tuple(int(m) for m in bank.info(c)[0])
        depth += 4.0 / (1 << len(modes))
        for l in range(1, len(modes) + 1):
            nodes.setdefault(l, set()).add(modes[:l])
    # order-48 stage: 24 odd-arm taps + centre = 13 dot2 per output and component (12.5 + 1 on average); a node at level l
    # produces 2^-l outputs per input sample.  Lower/upper siblings share the odd-arm sum (tree_kernel.hpp): 12.5 + 2.
    d2 = 0.0
    for l, ns in nodes.items():
        fused = set()
        for m in ns:
            if m[-1] in (1, 2) and (m[:-1] + (3 - m[-1],)) in ns:
                fused.add(m[:-1])
        single = len(ns) - 2 * len(fused)
        d2 += (single * 13.5 + len(fused) * 14.5) * 2 / (1 << l)
    return 4.0 + depth, d2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=["headline", "decim64", "chan32", "chan128", "cfg4", "fi64", "launchcheck"])
    ap.add_argument("--batch", type=int, default=None,
                    help="complex samples per step per GPU; default per workload: headline/decim64/chan32 1 Gi (4 GiB of int16 I/Q), "
                         "chan128/cfg4 256 Mi, fi64 512 Mi (4 GiB of float I/Q)")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="decimator part: split the batch into this many independent device streams served by ONE batched launch "
                         "(sdrx_decim_process_dev_batch); the bank part keeps one stream")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))       # nothing above touched HIP

    from sdrangel_amd import shard
    rank, local, world = shard.world_from_env()
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.workload == "launchcheck":
        # the N-rank protocol alone (launcher, rendezvous, barrier + max-over-ranks timing, rank-0 line) with a host-only
        # step and no GPU work: what tests/test_shard_gloo.py drives on the CPU.  Not a measurement.
        import numpy as np
        dist = shard.init_process_group("gloo", rank, world) if world > 1 else None
        buf = np.arange(1 << 16, dtype=np.int64) + rank
        acc = []
        el = shard.timed_region(lambda: acc.append(int(buf.sum())), args.steps, args.warmup, lambda: None, dist=dist)
        if rank == 0:
            print(json.dumps({"metric": "launcher self-test (host-only step, no GPU work)", "value": round(world * args.steps * buf.size / el / 1e6, 1),
                              "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "selftest",
                              "config": {"workload": "launchcheck", "streams": world}}), flush=True)
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    import sdrangel_amd as sa
    sa.lib()                                    # before torch: one HIP runtime (see sdrangel_amd/__init__.py)
    import torch
    n_dev = max(torch.cuda.device_count(), 1)
    backend = os.environ.get("SDRX_BENCH_BACKEND", "nccl")
    rehearsal = world > n_dev                   # fewer GPUs than ranks: ranks share the cards, rendezvous over gloo
    if rehearsal:
        backend = "gloo"
    dev = torch.device("cuda", (local % n_dev) if world > 1 else 0)
    torch.cuda.set_device(dev)
    dist = shard.init_process_group(backend, rank, world, device=dev) if world > 1 else None   # "nccl" == RCCL on ROCm
    n_gpus = world
    my_streams = shard.streams_of_rank(n_gpus, rank, world)       # one stream per GPU: stream s -> GPU s
    assert my_streams == [rank]
    red_dev = dev if (dist is not None and backend != "gloo") else None

    wl = args.workload
    B = args.batch if args.batch else {"headline": 1 << 30, "decim64": 1 << 30, "chan32": 1 << 30, "chan128": 1 << 28, "cfg4": 1 << 28, "fi64": 1 << 29}[wl]
    if rehearsal and not args.batch:
        B = min(B, 1 << 28)
    def make_input(n):
        # sdrbench-shaped data: uniform 12-bit noise, I/Q interleaved (mainbench.cpp:76-79) + an in-band tone
        g = torch.Generator(device=dev); g.manual_seed(5489 + rank)
        v = torch.randint(-2048, 2048, (2 * n,), generator=g, device=dev, dtype=torch.int32)
        t = torch.arange(n, device=dev, dtype=torch.int32).remainder_(10000).to(torch.float32)    # 0.0011 * 10000 = 11 whole cycles: exact at any n
        v[0::2] += (600 * torch.cos(2 * torch.pi * 0.0011 * t)).to(torch.int32)
        v[1::2] += (600 * torch.sin(2 * torch.pi * 0.0011 * t)).to(torch.int32)
        return v.clamp_(-32768, 32767).to(torch.int16)
    x = None
    while x is None:
        try:
            x = make_input(B)
            ok = 1
        except torch.cuda.OutOfMemoryError:                  # a device with less free memory than an MI355X: same workload, smaller step
            if args.batch:
                raise
            torch.cuda.empty_cache()
            ok = 0
        if dist is not None:                                 # every rank runs the same batch: agree on the smallest
            t = torch.tensor([B if ok else B // 4], dtype=torch.int64, device=red_dev if red_dev is not None else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) != B:
                B = int(t.item()); x = None; torch.cuda.empty_cache()
        elif not ok:
            B //= 4
    # ONE explicit stream for every handle of the step (the decimator and the bank run back to back on it, so a kernel trace of the
    # step adds up to ms_per_step).  Not torch's current stream: that is the null stream, whose handle 0 the library reads as "use the
    # handle's own stream" -- in round 2 the two halves therefore ran on two unordered streams and overlapped in the profile.
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    torch.cuda.synchronize(dev)                 # inputs were produced on torch's default stream

    S = max(1, args.streams_per_gpu)
    timed_handles = {}                          # name -> (handle, algorithmic bytes per sample, samples per step)

    def make_decim(n):
        """decimate64_cen over n samples as S independent streams, one (batched) launch"""
        if S == 1:
            h = sa.Decimators(6, sa.FC_CEN, 12, device=dev.index)
            out = torch.empty(2 * (n >> 6) + 64, dtype=torch.int16, device=dev)
            h.set_stream(stream)
            return h, (lambda: h.decimate_dev(x.data_ptr(), 2 * n, out.data_ptr())), [h, out]
        per = (n // S) & ~1023
        hs = [sa.Decimators(6, sa.FC_CEN, 12, device=dev.index) for _ in range(S)]
        hs[0].set_stream(stream)
        outs = [torch.empty(2 * (per >> 6) + 64, dtype=torch.int16, device=dev) for _ in range(S)]
        ins = [x.data_ptr() + 4 * per * i for i in range(S)]
        optr = [o.data_ptr() for o in outs]
        cnt = [2 * per] * S
        return hs[0], (lambda: sa.decimate_dev_batch(hs, ins, cnt, optr)), [hs, outs]

    def make_bank(n_ch, kind):
        fcs = chan_centres(n_ch, kind)
        b = sa.ChannelizerBank(61_440_000, [48000] * n_ch, fcs, device=dev.index)
        b.set_stream(stream)
        return b, fcs

    be = None
    fcs = None
    if wl == "fi64":
        # DecimatorsFI::decimate64_cen (AirspyHF thread), float I/Q in, int16 Samples out
        xf = (x[: 2 * B].to(torch.float32) / 4096.0).contiguous()
        del x
        x = xf
        torch.cuda.synchronize(dev)
        h = sa.FloatDecimators("fi", 6, sa.FC_CEN, device=dev.index)
        out = torch.empty(2 * (B >> 6) + 64, dtype=torch.int16, device=dev)
        h.set_stream(stream)
        step = lambda: h.decimate_dev(x.data_ptr(), 2 * B, out.data_ptr())
        timed_handles["fdecim"] = (h, 8.0 + 4.0 / 64, None)
        workload = f"8f.4: DecimatorsFI::decimate64_cen (IntHalfbandFilterEOF<64> x 6), one stream per GPU, {B} complex float32 samples per step, device-resident"
    elif wl == "decim64":
        h, step, _keep = make_decim(B)
        timed_handles["decim64"] = (h, 4.0 + 4.0 / 64, 32.5)
        workload = f"cfg2: decimate64_cen Decimators<qint32,qint16,16,12>, {S} stream(s) per GPU in one launch, {B} complex int16 samples per step, device-resident"
    elif wl == "headline":
        h, dstep, _keep = make_decim(B)
        bank, fcs = make_bank(32, "chan32")
        def step():
            dstep()
            bank.feed_dev(x.data_ptr(), B)
            for c in range(32):               # consumer side: drop the queued outputs (host bookkeeping only)
                bank.skip(c)
        bps_b, d2_b = bank_figures(bank, 32)
        timed_handles["decim64"] = (h, 4.0 + 4.0 / 64, 32.5)
        timed_handles["chan32"] = (bank, bps_b, d2_b)
        workload = (f"cfg2 + cfg3 over the same resident batch: decimate64_cen Decimators<qint32,qint16,16,12> ({S} stream(s), one launch) and the 32-channel "
                    f"DownChannelizer bank (48 kS/s channels from a 61.44 MS/s-shaped stream), {B} complex int16 samples per step and GPU")
    else:
        # cfg 3 = 32 channels, cfg 5 = 128 channels per stream/GPU, cfg 4 = 256 channels + demod front
        n_ch = {"chan32": 32, "chan128": 128, "cfg4": 256}[wl]
        bank, fcs = make_bank(n_ch, wl)
        if wl == "cfg4":
            cfgs = []
            for c in range(n_ch):
                _m, out_rate, ofs = bank.info(c)
                cfgs.append(sa.BackendCfg(in_rate=out_rate, nco_freq=-ofs, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5,
                                          filt_mode=2, f1=300 / 48000, f2=5000 / 48000, discri=1, fm_scaling=48000 / 2000))
            be = sa.BackendBank(cfgs, device=dev.index)
        def step():
            bank.feed_dev(x.data_ptr(), B)
            if be is not None:
                be.feed_bank(bank)               # device-ordered hand-over; the back-end runs on its own stream
                if os.environ.get("SDRX_BENCH_CFG4_SERIAL"):
                    be.sync()                    # default since round 3: the back-end of step k overlaps the channelizer of step k + 1 (6.29 vs 6.54 ms/step)
            for c in range(n_ch):
                bank.skip(c)
        bps_b, d2_b = bank_figures(bank, n_ch)
        if be is not None:
            bps_b += n_ch * 4.0 * 48000 / 61_440_000
        timed_handles[wl] = (bank, bps_b, d2_b)
        workload = (f"{wl}: DownChannelizer bank, {n_ch} channels (48 kS/s each) from one 61.44 MS/s-shaped stream per GPU"
                    + (" + NCO/Interpolator/fftfilt-SSB/NFM-discriminator front per channel" if be is not None else "")
                    + f", {B} samples per step")

    for _ in range(args.warmup):
        step()
    sync = (lambda: (be.sync(), torch.cuda.synchronize(dev))) if be is not None else (lambda: torch.cuda.synchronize(dev))
    sync()
    elapsed = shard.timed_region(step, args.steps, 0, sync, dist=dist, device=red_dev)
    # median of >= 10 individually timed steps (each bracketed by a device synchronisation) next to the mean above
    singles = []
    for _ in range(max(10, args.steps)):
        t0 = time.perf_counter(); step(); sync(); singles.append(time.perf_counter() - t0)
    singles.sort()
    ms_median = 1e3 * (singles[len(singles) // 2] if len(singles) % 2 else 0.5 * (singles[len(singles) // 2 - 1] + singles[len(singles) // 2]))

    # per-kernel durations: HIP events recorded by the library on the launch stream (sdrx_*_get_timing).  With two
    # different kernels queued back to back the start event of the second is stamped when the command processor reaches
    # it, not when the first kernel has drained, so each part is timed in a pass of its own (same buffers, same sizes).
    solo = {"decim64": (lambda: dstep()) if wl == "headline" else step, "fdecim": step}
    if wl == "headline":
        def bank_only():
            bank.feed_dev(x.data_ptr(), B)
            for c in range(32):
                bank.skip(c)
        solo["chan32"] = bank_only
    per_kernel = {}
    n_solo = max(3, min(args.steps, 10))
    for name, (hh, bps, d2) in timed_handles.items():
        fn = solo.get(name, step)
        fn(); sync()
        hh.set_timing(True)
        for _ in range(n_solo):
            fn()
        sync()
        k_ms, k_n = hh.get_timing()
        hh.set_timing(False)
        ll = hh.last_launch()
        per_feed = k_ms / n_solo                             # a bank feed is several tree_kernel launches: per step
        ach = bps * B / (per_feed * 1e-3) / 1e9
        e = {"kernel": ll["kernel"] + (" (all passes of a feed)" if "tree" in ll["kernel"] else ""), "kernel_ms": round(per_feed, 4),
             "launches": k_n, "algorithmic_bytes_per_sample": round(bps, 4), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
             "grid": ll["grid"], "block": ll["block"], "lds_bytes": ll["lds_bytes"]}
        eng = os.environ.get("SDRX_CHAN_ENGINE" if "tree" in ll["kernel"] else "SDRX_DECIM_ENGINE", "mfma")
        if d2 and eng == "valu":                              # the dot2 issue ceiling only binds the VALU engine
            e["dot2_per_sample"] = round(d2, 2)
            e["valu_ceiling_frac"] = round(DOT2_PER_S / d2 * bps / 1e9 / HBM_PEAK_GBS, 4)
        tr, src = load_traffic(ll["kernel"], B, name)
        e["traffic"] = tr; e["traffic_source"] = src
        per_kernel[name] = e

    # cfg 5's per-GPU share, every rank, same protocol (also at N = 1 so the driver's per-N lines give its curve too)
    also = {}
    if wl == "headline" and not args.no_also:
        nb = min(B, 1 << 28)
        b128, _f = make_bank(128, "chan128")
        def s128():
            b128.feed_dev(x.data_ptr(), nb)
            for c in range(128):
                b128.skip(c)
        for _ in range(2):
            s128()
        torch.cuda.synchronize(dev)
        b128.set_timing(True)
        el128 = shard.timed_region(s128, 5, 0, lambda: torch.cuda.synchronize(dev), dist=dist, device=red_dev)
        ms128, _n = b128.get_timing(); b128.set_timing(False)
        bps128, d2_128 = bank_figures(b128, 128)
        also["chan128"] = {"workload": f"cfg5 per GPU: DownChannelizer bank, 128 channels x 48 kS/s from one 61.44 MS/s-shaped stream per GPU, {nb} samples per feed",
                           "value": round(n_gpus * 5 * nb / el128 / 1e6, 1), "unit": "MS/s", "n_gpus": n_gpus, "kernel": b128.last_launch()["kernel"] + " (all passes of a feed)",
                           "kernel_ms": round(ms128 / 5, 4), "algorithmic_bytes_per_sample": round(bps128, 4),
                           "roofline_frac": round(bps128 * nb / (ms128 / 5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del b128

    if wl == "headline" and not args.no_also and n_gpus == 1:
        # BASELINE cfg 2 and cfg 3 at the configs' OWN sizes: a 10 M-sample block of the 10 MS/s stream through decimate64_cen,
        # and one second (61.44 M samples) of the LimeSDR-shaped stream through the 32-channel bank; the headline runs both at 1 Gi samples.
        def sized(name, make, n, bps, reps=20):
            hh, fn = make(n)
            for _ in range(3):
                fn()
            sync()
            hh.set_timing(True)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t0)
            k_ms, _n = hh.get_timing(); hh.set_timing(False)
            ts.sort()
            also[name] = {"samples_per_step": n, "kernel_ms": round(k_ms / reps, 4), "kernel": hh.last_launch()["kernel"],
                          "value": round(n / (k_ms / reps * 1e-3) / 1e6, 1), "unit": "MS/s (device-resident, kernel time)",
                          "step_ms_median_host_clock": round(1e3 * ts[len(ts) // 2], 4),
                          "frac": round(bps * n / (k_ms / reps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        def mk_d(n):
            hh = sa.Decimators(6, sa.FC_CEN, 12, device=dev.index); hh.set_stream(stream)
            oo = torch.empty(2 * (n >> 6) + 64, dtype=torch.int16, device=dev)
            return hh, (lambda: hh.decimate_dev(x.data_ptr(), 2 * (n & ~63), oo.data_ptr()))
        def mk_b(n):
            bb, _f = make_bank(32, "chan32")
            def fn():
                bb.feed_dev(x.data_ptr(), n)
                for c in range(32):
                    bb.skip(c)
            return bb, fn
        if B >= 61_440_000:
            sized("cfg2_10M", mk_d, 10_000_000, 4.0 + 4.0 / 64)
            sized("cfg3_61M44", mk_b, 61_440_000, per_kernel["chan32"]["algorithmic_bytes_per_sample"], reps=10)
        # end to end INCLUDING the host link: 4 Mi-sample blocks written into the decimator's pinned ring, H2D + kernel + D2H
        # overlapped (sdrx_decim_ring_*); PCIe-bound, reported for completeness, never `value`
        try:
            import numpy as np
            nblk, slots = 1 << 22, 6
            hr = sa.Decimators(6, sa.FC_CEN, 12, device=dev.index)
            hr.ring_create(2 * nblk, slots, 1)
            def run(k):
                inflight = 0
                for _ in range(k):
                    if inflight == slots - 1:
                        hr.ring_retire(); inflight -= 1
                    hr.ring_acquire(); hr.ring_submit(2 * nblk); inflight += 1
                while inflight:
                    hr.ring_retire(); inflight -= 1
            run(2 * slots)
            t0 = time.perf_counter(); run(48); dt = (time.perf_counter() - t0) / 48
            also["host_ring"] = {"workload": "decimate64_cen from pinned host blocks (sdrx_decim_ring_*): H2D + kernel + D2H per 4 Mi-sample block, 6 slots",
                                 "value": round(nblk / dt / 1e6, 1), "unit": "MS/s end to end", "GBps_over_host_link": round(4 * nblk / dt / 1e9, 2)}
            del hr
        except Exception as ex:                                    # the ring is optional equipment of the line
            also["host_ring"] = {"error": str(ex)}

    if rank == 0:
        value = n_gpus * args.steps * B / elapsed / 1e6
        ms_step = elapsed / args.steps * 1e3
        dom = max(per_kernel.values(), key=lambda e: e["kernel_ms"])
        low = min(per_kernel.values(), key=lambda e: e["frac"])
        engine = {"decim": os.environ.get("SDRX_DECIM_ENGINE", "mfma"), "chan": os.environ.get("SDRX_CHAN_ENGINE", "mfma")}
        roof = {"bound": "valu" if wl != "fi64" else "lds+valu", "engine": engine, "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": low["frac"], "traffic": dom["traffic"], "traffic_source": dom["traffic_source"],
                "kernel": dom["kernel"], "kernel_ms": dom["kernel_ms"], "launches": dom["launches"],
                "algorithmic_bytes_per_sample": dom["algorithmic_bytes_per_sample"],
                "grid": dom["grid"], "block": dom["block"], "lds_bytes": dom["lds_bytes"],
                "bound_evidence": "profiles/r03_*: the half-band FIRs run on the matrix cores (i8 MFMA, 16-20 % busy); what is left is the VALU "
                                  "epilogue (limb recombination, centre tap, packing), LDS traffic and wave latency -- not HBM",
                "per_kernel": per_kernel}
        if "valu_ceiling_frac" in dom:
            roof["valu_ceiling_frac"] = dom["valu_ceiling_frac"]
        if wl == "headline":
            # `frac` is the LOWER of the two kernels' fractions (what the review asked for); the step as a whole, for whoever
            # divides the step's algorithmic bytes by ms_per_step: both kernels read the batch, each writes its outputs
            bps_step = sum(e["algorithmic_bytes_per_sample"] for e in per_kernel.values())
            whole = bps_step * B / (ms_step * 1e-3) / 1e9
            roof["whole_step"] = {"algorithmic_bytes_per_sample": round(bps_step, 4), "achieved": round(whole, 1), "frac": round(whole / HBM_PEAK_GBS, 4),
                                  "ms_per_step": round(ms_step, 4)}
        if wl == "cfg4":
            # the step is tree passes + schedule/mix/FIR/FFT/finish of the back-end: the fraction is over the WHOLE step
            whole = per_kernel[wl]["algorithmic_bytes_per_sample"] * B / (ms_step * 1e-3) / 1e9
            roof["tree_ms"] = dom["kernel_ms"]; roof["backend_ms"] = round(ms_step - dom["kernel_ms"], 4)
            roof["achieved"] = round(whole, 1); roof["frac"] = round(whole / HBM_PEAK_GBS, 4)
            roof["kernel"] = "whole step: tree_kernel passes + be_schedule/mix/fir/fft/finish"; roof["kernel_ms"] = round(ms_step, 4)
        line = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "MS/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "ms_per_step_median": round(ms_median, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if wl == "fi64" else "int32", "data": "synthetic",
            "config": {"workload": workload, "streams": n_gpus * (S if wl in ("headline", "decim64") else 1), "streams_per_gpu": S if wl in ("headline", "decim64") else 1,
                       "parallelism": f"{n_gpus} GPU(s), one independent 61.44 MS/s-shaped stream set per GPU, no collective"
                                      + (" [REHEARSAL: ranks share GPUs, gloo rendezvous]" if rehearsal else "")},
            "roofline": roof,
        }
        if also:
            line["also"] = also
        if n_gpus == 1:
            # the denominator twice -- datasheet (peak/frac above) and a read-only streaming kernel on this box
            del x
            torch.cuda.empty_cache()
            meas = sa.measure_hbm_read(dev.index, 4 << 30, 5)
            roof["peak_measured"] = round(meas, 1)
            roof["frac_of_measured"] = round(roof["frac"] * HBM_PEAK_GBS / meas, 4)
        if n_gpus == 1 and not args.no_cpu:
            nt = host_threads()
            host = {"cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "threads_used": nt}
            if wl == "fi64":
                cb = cpu_fi(4 * 1024 * 1024, 8, nt)
            elif wl == "decim64":
                cb = cpu_decim(6, 8 * 1024 * 1024, 8, nt)
            elif wl == "headline":
                d = cpu_decim(6, 8 * 1024 * 1024, 6, nt)
                c = cpu_chan(fcs, 12 * 1024 * 1024, nt)
                both = 1.0 / (1.0 / d["value"] + 1.0 / c["value"])
                cb = {"value": round(both, 3), "unit": "MS/s", "cores": nt, "kind": d["kind"], "lib": d["lib"],
                      "sample": "the step's two halves on the host cores: decim64 (" + d["sample"] + ") and chan32 (" + c["sample"]
                                + "); value = samples/s through both = 1 / (1/decim64 + 1/chan32)",
                      "decim64": d, "chan32": c}
            else:
                cb = cpu_chan(fcs, (4 if wl == "chan32" else 1) * 1024 * 1024, nt)
            if wl in ("headline", "decim64"):
                # BASELINE.json configs[0]: sdrangelbench -t decimateii -l 4 -n 10000000 -r 10 (sdrbench/mainbench.cpp:69-110)
                cb["sdrbench_decim16"] = cpu_decim(4, 10_000_000, 10, nt, seed=5489)
            cb["host"] = host
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
