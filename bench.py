#!/usr/bin/env python3
"""bench.py -- throughput of the sdrx hot path on MI355X, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload decim64|chan32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch of synthetic int16 I/Q that is already resident
in HBM: `sdrx_decim_process_dev` (decimate64_cen, Decimators<qint32,qint16,16,12>, BASELINE.json
configs[1]) over a batch of 256 Mi complex samples (1 GiB) of one stream.  With N > 1 every rank owns one
GPU and one independent stream (streams shard, no collective on the data path);
per-GPU work is fixed, so scaling is "weak" and `value` is the sum over ranks.

roofline : dominant kernel's ALGORITHMIC bytes (4 B read + 4/64 B written per input sample =
           4.0625 B/sample) / its average duration, measured with HIP events on the
           launch stream inside the library (sdrx_decim_get_timing), against 8 TB/s HBM3E.
cpu_baseline : the reference's own decimate64_cen (oracle/_ref/libsdrref.so, built from
           /root/reference in the build container; kind "reference") or, if that .so did not travel,
           the oracle port (oracle/libsdro_fast.so; kind "port"), one independent stream per thread
           on the host cores of this box, bounded sample.  The oracle is only the checker/baseline:
           nothing on the measured GPU path touches it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet (MI355X_MICROARCH.md); ~6300 achievable


def cpu_baseline(sample_cplx: int, reps: int, log2: int = 6):
    """Reference (or port) decimate64_cen, one stream per thread, on the host cores."""
    import numpy as np
    from tests import oracle_py as orc
    n_thr = max(1, min(os.cpu_count() or 1, 16))
    x = orc.synth_iq(sample_cplx, seed=1234, amp=2047)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libsdrref.so")
    kind = None
    if os.path.exists(ref_so):
        try:
            L = C.CDLL(ref_so)
            L.ref_decim_new.restype = C.c_void_p; L.ref_decim_new.argtypes = [C.c_int]
            L.ref_decim_process.restype = C.c_int
            L.ref_decim_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32, C.c_void_p]
            mk = lambda: L.ref_decim_new(12)
            run = lambda h, buf, out: L.ref_decim_process(h, log2, 2, buf.ctypes.data, buf.size, out.ctypes.data)
            kind = "reference"
        except OSError:
            kind = None
    if kind is None:
        fast = os.path.join(ROOT, "oracle", "libsdro_fast.so")
        if not os.path.exists(fast):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsdro_fast.so"])
        L = orc.lib(fast=True)
        mk = lambda: L.sdro_decim_new(log2, 2, 12)
        run = lambda h, buf, out: L.sdro_decim_process(h, buf.ctypes.data, buf.size, out.ctypes.data)
        kind = "port"

    def timed(n_threads):
        hs = [mk() for _ in range(n_threads)]
        bufs = [x.copy() for _ in range(n_threads)]
        outs = [np.empty(sample_cplx // (1 << log2) * 2 + 64, np.int16) for _ in range(n_threads)]
        for i in range(n_threads):
            run(hs[i], bufs[i][: 2 * 65536], outs[i])          # touch / warm
        def work(i):
            for _ in range(reps):
                run(hs[i], bufs[i], outs[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        return n_threads * reps * sample_cplx / dt / 1e6

    one = timed(1)
    allc = timed(n_thr) if n_thr > 1 else one
    return {"value": round(allc, 2), "unit": "MS/s", "cores": n_thr, "kind": kind,
            "single_thread_MSps": round(one, 2),
            "sample": f"decimate64_cen <16,12> on {sample_cplx} synthetic int16 I/Q samples x {reps} passes per thread, "
                      f"one independent stream per thread ({n_thr} threads); single_thread_MSps = 1 thread, as sdrangelbench runs it"}


def cpu_baseline_fi(sample_cplx: int, reps: int):
    """Reference (or port) DecimatorsFI::decimate64_cen, one stream per thread, on the host cores."""
    import numpy as np
    from tests import oracle_py as orc
    n_thr = max(1, min(os.cpu_count() or 1, 16))
    x = (orc.synth_iq(sample_cplx, seed=99, amp=2047).astype(np.float32) / np.float32(4096.0))
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libsdrref.so")
    kind = "port"
    L = None
    if os.path.exists(ref_so):
        try:
            L = C.CDLL(ref_so)
            L.ref_fdecim_new.restype = C.c_void_p; L.ref_fdecim_new.argtypes = [C.c_int] * 3
            L.ref_fdecim_process.restype = C.c_int; L.ref_fdecim_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int32, C.c_void_p]
            kind = "reference"
        except (OSError, AttributeError):
            L = None
    if L is not None:
        mk = lambda: L.ref_fdecim_new(0, 0, 16)
        run = lambda h, out: L.ref_fdecim_process(h, 6, 2, x.ctypes.data, x.size, out.ctypes.data)
    else:
        O = orc.lib(); orc._sig_fdecim(O)
        mk = lambda: O.sdro_fdecim_new(6, 2, 0, 0, 16)
        run = lambda h, out: O.sdro_fdecim_process(h, x.ctypes.data, x.size, out.ctypes.data)

    def timed(n_threads):
        hs = [mk() for _ in range(n_threads)]
        outs = [np.empty(x.size + 64, np.int16) for _ in range(n_threads)]
        def work(i):
            for _ in range(reps):
                run(hs[i], outs[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return n_threads * reps * sample_cplx / (time.perf_counter() - t0) / 1e6

    one = timed(1)
    allc = timed(n_thr) if n_thr > 1 else one
    return {"value": round(allc, 2), "unit": "MS/s", "cores": n_thr, "kind": kind, "single_thread_MSps": round(one, 2),
            "sample": f"DecimatorsFI::decimate64_cen on {sample_cplx} synthetic float I/Q samples x {reps} passes per thread, one stream per thread ({n_thr} threads)"}


def cpu_baseline_chan(fcs, sample_cplx: int):
    """Reference DownChannelizer stage chains (IntHalfbandFilterEO<qint32,qint32,48> objects driven by the loop of
    DownChannelizer::feed), every channel re-filtering the full-rate stream on its own thread like
    ThreadedBasebandSampleSink does (threadedbasebandsamplesink.cpp:74-78); value = input MS/s the host sustains
    for the WHOLE bank."""
    import numpy as np
    from tests import oracle_py as orc
    n_thr = max(1, min(os.cpu_count() or 1, 16, len(fcs)))
    x = orc.synth_iq(sample_cplx, seed=77, amp=2047)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libsdrref.so")
    plans = [orc.chan_plan(61_440_000, 48000, int(fc))[0] for fc in fcs]
    kind = "port"
    L = None
    if os.path.exists(ref_so):
        try:
            L = C.CDLL(ref_so)
            L.ref_chain_new.restype = C.c_void_p; L.ref_chain_new.argtypes = [C.c_int, C.c_void_p]
            L.ref_chain_feed.restype = C.c_int64; L.ref_chain_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
            kind = "reference"
        except OSError:
            L = None
    if L is not None:
        hs = [L.ref_chain_new(len(m), np.ascontiguousarray(m).ctypes.data) for m in plans]
        feed = lambda h, out: L.ref_chain_feed(h, x.ctypes.data, sample_cplx, out.ctypes.data)
    else:
        O = orc.lib(fast=os.path.exists(os.path.join(ROOT, "oracle", "libsdro_fast.so")))
        hs = [O.sdro_chain_new(len(m), np.ascontiguousarray(m).ctypes.data) for m in plans]
        feed = lambda h, out: O.sdro_chain_feed(h, x.ctypes.data, sample_cplx, out.ctypes.data)
    outs = [np.empty(sample_cplx // 64 + 64, np.int16) for _ in hs]
    def work(t):
        for c in range(t, len(hs), n_thr):
            feed(hs[c], outs[c])
    th = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    return {"value": round(sample_cplx / dt / 1e6, 3), "unit": "MS/s", "cores": n_thr, "kind": kind,
            "sample": f"{len(fcs)} DownChannelizer chains (one per channel, each over the full-rate stream) on {sample_cplx} synthetic samples, "
                      f"{n_thr} host threads, channels dealt round-robin; value = input rate sustained for the whole bank"}


def load_traffic(kernel: str, batch: int, workload: str):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (profiles/*traffic*.json), or None."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        for e in d.get("kernels", []):
            if e.get("workload") == workload and e.get("batch_cplx") == batch and e.get("kernel", "").startswith(kernel.split("<")[0]):
                best = e.get("hbm_bytes_per_step", e.get("hbm_bytes_per_launch"))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="decim64", choices=["decim64", "chan32", "chan128", "cfg4", "fi64"])
    ap.add_argument("--batch", type=int, default=None,
                    help="complex samples per step per GPU; default per workload: decim64 and chan32 1 Gi (4 GiB of int16 I/Q), chan128/cfg4 256 Mi, "
                         "fi64 512 Mi (4 GiB of float I/Q).  One wave of the decimator lives ~0.3 ms, so short launches lose a "
                         "large part of their time to the tail: 256 Mi samples run at 385 GS/s, 1 Gi at 494 GS/s (DESIGN.md 6)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary cfg-3 (32-channel bank) measurement")
    args = ap.parse_args()

    import sdrangel_amd as sa
    from sdrangel_amd import shard
    sa.lib()                                    # before torch: one HIP runtime (see sdrangel_amd/__init__.py)
    import torch

    rank, local, world = shard.world_from_env()
    # SDRX_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share the cards)
    backend = os.environ.get("SDRX_BENCH_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", (local % n_dev if backend == "gloo" else local) if world > 1 else 0)
    torch.cuda.set_device(dev)
    dist = shard.init_process_group(backend, rank, world, device=dev) if world > 1 else None   # "nccl" == RCCL on ROCm
    n_gpus = world
    my_streams = shard.streams_of_rank(n_gpus, rank, world)       # one stream per GPU: stream s -> GPU s
    assert my_streams == [rank]

    B = args.batch if args.batch else {"decim64": 1 << 30, "chan32": 1 << 30, "chan128": 1 << 28, "cfg4": 1 << 28, "fi64": 1 << 29}[args.workload]
    def make_input(n):
        # sdrbench-shaped data: uniform 12-bit noise, I/Q interleaved (mainbench.cpp:76-79) + an in-band tone
        g = torch.Generator(device=dev); g.manual_seed(5489 + rank)
        v = torch.randint(-2048, 2048, (2 * n,), generator=g, device=dev, dtype=torch.int32)
        t = torch.arange(n, device=dev, dtype=torch.int32).remainder_(10000).to(torch.float32)    # 0.0011 * 10000 = 11 whole cycles: exact at any n
        v[0::2] += (600 * torch.cos(2 * torch.pi * 0.0011 * t)).to(torch.int32)
        v[1::2] += (600 * torch.sin(2 * torch.pi * 0.0011 * t)).to(torch.int32)
        return v.clamp_(-32768, 32767).to(torch.int16)
    try:
        x = make_input(B)
    except torch.cuda.OutOfMemoryError:                      # a device with less free memory than an MI355X: same workload, smaller step
        if args.batch:
            raise
        torch.cuda.empty_cache()
        B = B // 4
        x = make_input(B)
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize(dev)                 # the library's streams are not ordered against torch's default stream (handle 0 = "own stream")

    if args.workload == "fi64":
        # DecimatorsFI::decimate64_cen (AirspyHF thread), float I/Q in, int16 Samples out
        xf = (x[: 2 * B].to(torch.float32) / 4096.0).contiguous()
        del x
        x = xf
        torch.cuda.synchronize(dev)
        h = sa.FloatDecimators("fi", 6, sa.FC_CEN, device=dev.index)
        out = torch.empty(2 * (B >> 6) + 64, dtype=torch.int16, device=dev)
        h.set_stream(stream)
        step = lambda: h.decimate_dev(x.data_ptr(), 2 * B, out.data_ptr())
        bytes_per_sample = 8.0 + 4.0 / 64
        workload = f"8f.4: DecimatorsFI::decimate64_cen (IntHalfbandFilterEOF<64> x 6), one stream per GPU, {B} complex float32 samples per step, device-resident"
    elif args.workload == "decim64":
        h = sa.Decimators(6, sa.FC_CEN, 12, device=dev.index)
        out = torch.empty(2 * (B >> 6) + 64, dtype=torch.int16, device=dev)
        h.set_stream(stream)
        step = lambda: h.decimate_dev(x.data_ptr(), 2 * B, out.data_ptr())
        bytes_per_sample = 4.0 + 4.0 / 64
        workload = f"cfg2: decimate64_cen Decimators<qint32,qint16,16,12>, one stream per GPU, {B} complex int16 samples per step, device-resident"
    else:
        # cfg 3 = 32 channels, cfg 5 = 128 channels per stream/GPU, cfg 4 = 256 channels + demod front
        n_ch = {"chan32": 32, "chan128": 128, "cfg4": 256}[args.workload]
        k = torch.arange(n_ch, dtype=torch.float64)
        if args.workload == "cfg4":
            fcs = (-25_000_000 + k * (50_000_000 / 255)).to(torch.int64).tolist()
        else:
            fcs = (-15_000_000 + k * (30_000_000 / (n_ch - 1)) + 137 * k).to(torch.int64).tolist()
        h = sa.ChannelizerBank(61_440_000, [48000] * n_ch, fcs, device=dev.index)
        h.set_stream(stream)
        be = None
        if args.workload == "cfg4":
            cfgs = []
            for c in range(n_ch):
                _m, out_rate, ofs = h.info(c)
                cfgs.append(sa.BackendCfg(in_rate=out_rate, nco_freq=-ofs, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5,
                                          filt_mode=2, f1=300 / 48000, f2=5000 / 48000, discri=1, fm_scaling=48000 / 2000))
            be = sa.BackendBank(cfgs, device=dev.index)
        def step():
            h.feed_dev(x.data_ptr(), B)
            if be is not None:
                be.feed_bank(h)                  # device-ordered hand-over; the back-end runs on its own stream
                if not os.environ.get("SDRX_BENCH_CFG4_PIPELINED"):
                    be.sync()                    # letting the back-end overlap the NEXT step's channelizer measured slower (2.97 vs 2.40 ms/step)
            for c in range(n_ch):               # consumer side: drop the queued outputs (host bookkeeping only)
                h.skip(c)
        depth = sum(4.0 / (1 << len(h.info(c)[0])) for c in range(n_ch))
        bytes_per_sample = 4.0 + depth + (n_ch * 4.0 * 48000 / 61_440_000 if be is not None else 0.0)
        workload = (f"{args.workload}: DownChannelizer bank, {n_ch} channels (48 kS/s each) from one 61.44 MS/s-shaped stream per GPU"
                    + (" + NCO/Interpolator/fftfilt-SSB/NFM-discriminator front per channel" if be is not None else "")
                    + f", {B} samples per step")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    h.set_timing(True)
    elapsed = shard.timed_region(step, args.steps, 0, lambda: torch.cuda.synchronize(dev), dist=dist, device=dev if backend != "gloo" else None)
    k_ms, k_n = h.get_timing()
    h.set_timing(False)

    if rank == 0:
        value = n_gpus * args.steps * B / elapsed / 1e6
        kern_ms = k_ms / max(k_n, 1)
        achieved = bytes_per_sample * B / (kern_ms * 1e-3) / 1e9
        ll = h.last_launch()
        line = {
            "metric": "MS/s complex int16 IQ through decim-64 + DownChannelizer, 1/2/4/8 GPU; % HBM roofline",
            "value": round(value, 1), "unit": "MS/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.workload == "fi64" else "int32", "data": "synthetic",
            "config": {"workload": workload, "streams": n_gpus, "parallelism": f"{n_gpus} independent stream(s), one per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": load_traffic(ll["kernel"], B, args.workload),
                         "kernel": ll["kernel"], "kernel_ms": round(kern_ms, 4), "launches": k_n,
                         "algorithmic_bytes_per_sample": bytes_per_sample,
                         "grid": ll["grid"], "block": ll["block"], "lds_bytes": ll["lds_bytes"]},
        }
        if n_gpus == 1 and args.workload == "decim64" and not args.no_also:
            # the other half of the metric's name: BASELINE configs[2], 32-channel DownChannelizer bank, same GPU, same run
            nb = B                                         # the same resident buffer
            k32 = torch.arange(32, dtype=torch.float64)
            fcs32 = (-15_000_000 + k32 * (30_000_000 / 31) + 137 * k32).to(torch.int64).tolist()
            bank = sa.ChannelizerBank(61_440_000, [48000] * 32, fcs32, device=dev.index)
            bank.set_stream(stream)
            def bstep():
                bank.feed_dev(x.data_ptr(), nb)
                for c in range(32):
                    bank.skip(c)
            for _ in range(2):
                bstep()
            torch.cuda.synchronize(dev)
            bank.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(5):
                bstep()
            torch.cuda.synchronize(dev)
            bel = time.perf_counter() - t0
            bms, bn = bank.get_timing()
            bms /= max(bn, 1)
            bps = 4.0 + 32 * 4.0 / 1024
            line["also"] = {"chan32": {"workload": "cfg3: DownChannelizer bank, 32 channels x 48 kS/s from one 61.44 MS/s-shaped stream, %d samples per feed" % nb,
                                       "value": round(5 * nb / bel / 1e6, 1), "unit": "MS/s", "kernel": "tree_kernel (all passes of a feed)",
                                       "kernel_ms": round(bms, 4), "algorithmic_bytes_per_sample": bps,
                                       "roofline_frac": round(bps * nb / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            del bank
        if n_gpus == 1:
            # the denominator twice -- datasheet (peak/frac above) and a read-only streaming kernel on this box
            del x
            torch.cuda.empty_cache()
            meas = sa.measure_hbm_read(dev.index, 4 << 30, 5)
            line["roofline"]["peak_measured"] = round(meas, 1)
            line["roofline"]["frac_of_measured"] = round(achieved / meas, 4)
        if n_gpus == 1 and not args.no_cpu:
            if args.workload == "fi64":
                line["cpu_baseline"] = cpu_baseline_fi(4 * 1024 * 1024, 8)
            elif args.workload == "decim64":
                line["cpu_baseline"] = cpu_baseline(8 * 1024 * 1024, 16)
            else:
                line["cpu_baseline"] = cpu_baseline_chan(fcs, 4 * 1024 * 1024)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
