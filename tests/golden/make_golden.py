#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json FROM THE REFERENCE ITSELF (build container only).

Runs the reference's own classes, compiled where they lie under /root/reference by oracle/Makefile
(`make -C oracle ref ref_qt` -> oracle/_ref/libsdrref.so, libsdrref_qt.so; wrappers in
oracle/ref_shim*.cpp), on deterministic integer-generated inputs (tests/synth.py) and stores
inputs' recipe + expected outputs.  The fixtures are data only; no reference source travels.

    LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 python tests/golden/make_golden.py
(the preload keeps conda's older libstdc++ -- pulled in by libQt5Core -- from shadowing the system one)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402

ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsdrref.so"))
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
ref.ref_decim_new.restype = vp; ref.ref_decim_new.argtypes = [C.c_int]
ref.ref_decim_free.argtypes = [vp]
ref.ref_decim_process.restype = C.c_int; ref.ref_decim_process.argtypes = [vp, C.c_int, C.c_int, vp, i32, vp]
ref.ref_chain_new.restype = vp; ref.ref_chain_new.argtypes = [C.c_int, vp]
ref.ref_chain_free.argtypes = [vp]
ref.ref_chain_feed.restype = i64; ref.ref_chain_feed.argtypes = [vp, vp, i64, vp]


def ref_decim(bits, log2, fc, x, cuts):
    h = ref.ref_decim_new(bits)
    outs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = np.ascontiguousarray(x[a:b])
        o = np.zeros(seg.size + 16, np.int16)
        n = ref.ref_decim_process(h, log2, fc, seg.ctypes.data, seg.size, o.ctypes.data)
        outs.append(o[: 2 * n].copy())
    ref.ref_decim_free(h)
    return np.concatenate(outs) if outs else np.zeros(0, np.int16)


def ref_chain(modes, x, cuts):
    m = np.ascontiguousarray(modes, dtype=np.uint8)
    h = ref.ref_chain_new(m.size, m.ctypes.data)
    outs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = np.ascontiguousarray(x[2 * a: 2 * b])
        o = np.zeros(seg.size + 16, np.int16)
        n = ref.ref_chain_feed(h, seg.ctypes.data, (b - a), o.ctypes.data)
        outs.append(o[: 2 * n].copy())
    ref.ref_chain_free(h)
    return np.concatenate(outs) if outs else np.zeros(0, np.int16)


def main():
    # ------------------------------------------------------------------ Decimators
    N = 12288 + 200                                # ragged on purpose
    cases = {"b12": synth.mix(N, 11, 2047, 900, 1), "b8": synth.mix(N, 12, 127, 60, 1),
             "b16": synth.mix(N, 13, 32767, 0), "wrap": None}
    w = np.empty(2 * N, np.int16); w[0::2] = -32768; w[1::2] = np.where(np.arange(N) % 3 == 0, 32767, -32768)
    w[:4000] = synth.noise_iq(2000, 14, 32767)
    cases["wrap"] = w
    # cut points in int16 units: an empty call, one shorter than a group, odd leftovers
    cuts = [0, 0, 6, 2 * 2048 + 2, 2 * 5001, 2 * 9000 + 128, 2 * N]
    dec = {"cuts": np.array(cuts, np.int64)}
    hashes = {}
    for name, x in cases.items():
        for bits in (8, 12, 16):
            for log2 in range(0, 7):
                for fc in (0, 1, 2):
                    y = ref_decim(bits, log2, fc, x, cuts)
                    key = f"{name}_bits{bits}_log{log2}_fc{fc}"
                    hashes[key] = {"n": int(y.size // 2), "fnv1a64": f"{synth.fnv1a64(y):016x}"}
                    if log2 >= 4 and (bits == 12 or name == "wrap"):
                        dec[key] = y                 # small enough to keep verbatim
    np.savez_compressed(os.path.join(HERE, "decim_golden.npz"), **dec)
    json.dump({"recipe": {"N": N, "b12": ["mix", 11, 2047, 900, 1], "b8": ["mix", 12, 127, 60, 1], "b16": ["mix", 13, 32767, 0, 1],
                          "wrap": "see make_golden.py"}, "cuts_int16": cuts, "hashes": hashes},
              open(os.path.join(HERE, "decim_golden.json"), "w"), indent=0, sort_keys=True)

    # ------------------------------------------------------------------ DecimatorsU<qint32, quint8, 16, 8, 127> (RTL-SDR)
    ref.ref_decimu_new.restype = vp; ref.ref_decimu_free.argtypes = [vp]
    ref.ref_decimu_process.restype = C.c_int; ref.ref_decimu_process.argtypes = [vp, C.c_int, C.c_int, vp, i32, vp]
    xu = (synth.lcg_u32(2 * N, 15) & 0xff).astype(np.uint8)
    xu[:2000] = 0; xu[2000:4000] = 255
    uh = {}
    for log2 in range(0, 7):
        for fc in (0, 1, 2):
            h = ref.ref_decimu_new(); outs = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                seg = np.ascontiguousarray(xu[a:b]); o = np.zeros(seg.size + 16, np.int16)
                n = ref.ref_decimu_process(h, log2, fc, seg.ctypes.data, seg.size, o.ctypes.data)
                outs.append(o[: 2 * n].copy())
            ref.ref_decimu_free(h)
            y = np.concatenate(outs)
            uh[f"u8_log{log2}_fc{fc}"] = {"n": int(y.size // 2), "fnv1a64": f"{synth.fnv1a64(y):016x}"}
    json.dump({"recipe": "(lcg_u32(2N,15) & 0xff), first 2000 bytes 0, next 2000 bytes 255; cuts as decim_golden", "hashes": uh},
              open(os.path.join(HERE, "decimu_golden.json"), "w"), indent=0, sort_keys=True)

    # ------------------------------------------------------------------ DownChannelizer plans (real QObject class)
    q = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsdrref_qt.so"))
    q.refqt_chan_new.restype = vp; q.refqt_chan_new.argtypes = [C.c_int] * 3
    q.refqt_chan_plan.argtypes = [vp, vp, vp, vp]; q.refqt_chan_free.argtypes = [vp]
    q.refqt_chan_feed.restype = i64; q.refqt_chan_feed.argtypes = [vp, vp, i64, vp]
    plans = []
    u = synth.lcg_u32(3 * 400, 99).astype(np.int64)
    rates_in = [61440000, 10000000, 2400000, 48000, 96000, 1000001, 3200000, 250000, 6000000, 30720000]
    rates_rq = [48000, 8000, 12500, 200000, 24000, 64000, 100, 1]
    fixed = [(61440000, 48000, 1234567), (61440000, 48000, -20000000), (61440000, 48000, 0), (2400000, 2400000, 0),
             (2400000, 1200000, 0), (2400000, 1200000, 600000), (48000, 48000, 1)]
    k = np.arange(32)
    fixed += [(61440000, 48000, int(-15_000_000 + kk * (30_000_000 / 31) + 137 * kk)) for kk in k]
    for i in range(400):
        ir = rates_in[u[3 * i] % len(rates_in)]; rr = rates_rq[u[3 * i + 1] % len(rates_rq)]
        fc = int(u[3 * i + 2] % (ir + 2000)) - ir // 2 - 1000
        fixed.append((ir, rr, fc))
    for ir, rr, fc in fixed:
        h = q.refqt_chan_new(ir, rr, fc); m = np.zeros(40, np.uint8); r = C.c_int(); f = C.c_int()
        n = q.refqt_chan_plan(h, m.ctypes.data, C.byref(r), C.byref(f)); q.refqt_chan_free(h)
        plans.append({"in": ir, "req": rr, "fc": fc, "modes": [int(v) for v in m[:n]], "out_rate": r.value, "ofs": f.value})
    json.dump(plans, open(os.path.join(HERE, "chan_plans.json"), "w"))

    # ------------------------------------------------------------------ DownChannelizer feed (real class) + stage chains
    NC = 1 << 16
    xs = {"n12": synth.mix(NC, 21, 2047, 1200, 3), "full": synth.noise_iq(NC, 22, 32767)}
    xs["full"][::11] = -32768
    ccuts = [0, 5, 4099, 4099, 30000, 30001, NC]
    chan = {"cuts": np.array(ccuts, np.int64)}
    for name, x in xs.items():
        for ci in (0, 5, 13, 16, 31):                 # a few of the cfg-3 channels, through DownChannelizer::feed itself
            ir, rr, fc = fixed[7 + ci]
            h = q.refqt_chan_new(ir, rr, fc)
            outs = []
            for a, b in zip(ccuts[:-1], ccuts[1:]):
                seg = np.ascontiguousarray(x[2 * a: 2 * b]); o = np.zeros(seg.size + 16, np.int16)
                n = q.refqt_chan_feed(h, seg.ctypes.data, b - a, o.ctypes.data)
                outs.append(o[: 2 * n].copy())
            q.refqt_chan_free(h)
            chan[f"{name}_cfg3ch{ci}"] = np.concatenate(outs)
        for modes in ([0], [1], [2], [1, 2, 0], [2, 2, 1, 0, 1], [0, 0, 0, 0, 0, 0, 0], [1, 0, 2, 1, 0, 2, 1, 0, 2, 1, 0, 2]):
            # short chains decimate little: keep their fixtures small by feeding only the first 4099 samples
            cc = ccuts if len(modes) > 3 else ccuts[:4]
            chan[f"{name}_chain_{''.join(map(str, modes))}"] = ref_chain(modes, x, cc)
    np.savez_compressed(os.path.join(HERE, "chan_golden.npz"), **chan)
    # ------------------------------------------------------------------ float back-end (strict-IEEE scalar build of the reference)
    f32 = C.c_float
    ref.ref_backend_new.restype = vp; ref.ref_backend_new.argtypes = [f32, f32, f32, C.c_int, f32, f32]
    ref.ref_backend_free.argtypes = [vp]
    ref.ref_backend_feed.restype = i64; ref.ref_backend_feed.argtypes = [vp, vp, i64, vp]
    ref.ref_fftfilt_new.restype = vp; ref.ref_fftfilt_new.argtypes = [f32, f32, C.c_int]
    ref.ref_fftfilt_run.restype = i64; ref.ref_fftfilt_run.argtypes = [vp, C.c_int, vp, i64, vp]
    ref.ref_gfft.argtypes = [vp, C.c_int, C.c_int]
    ref.ref_discri.argtypes = [C.c_int, f32, vp, i64, vp]
    ref.ref_nco_table.argtypes = [vp]
    fl = {}
    t = np.zeros(4096, np.float32); ref.ref_nco_table(t.ctypes.data); fl["nco_table"] = t
    xin = synth.mix(6000, 31, 12000, 6000, 1)
    for name, (nf, ir, orr, cut, tpp) in {"nfm": (-4567.0, 60000.0, 48000.0, 12500 / 2.2, 4.5), "ssb": (20000.0, 120000.0, 48000.0, 5000.0, 2.0)}.items():
        h = ref.ref_backend_new(nf, ir, orr, 16, cut, tpp)
        o = np.zeros(2 * 6000 + 8, np.float32); k = 0
        for a, b in ((0, 1234), (1234, 1235), (1235, 6000)):
            seg = np.ascontiguousarray(xin[2 * a: 2 * b])
            k += ref.ref_backend_feed(h, seg.ctypes.data, b - a, o[2 * k:].ctypes.data)
        ref.ref_backend_free(h)
        fl[f"resamp_{name}"] = o[: 2 * k].copy()
    # g_fft on an integer-valued vector, forward and inverse
    v = synth.noise_iq(1024, 32, 30000).astype(np.float32)
    a = v.copy(); ref.ref_gfft(a.ctypes.data, 1024, 0); fl["gfft1024_fwd"] = a
    a = v.copy(); ref.ref_gfft(a.ctypes.data, 1024, 1); fl["gfft1024_inv"] = a
    # fftfilt on the NFM-resampled stream, all run modes, + discriminators on the usb output
    xr = fl["resamp_nfm"]
    for mode, nm in ((0, "filt"), (1, "usb"), (2, "lsb")):
        h = ref.ref_fftfilt_new(300 / 48000, 3000 / 48000, 1024)
        o = np.zeros(xr.size + 2048, np.float32)
        k = ref.ref_fftfilt_run(h, mode, xr.ctypes.data, xr.size // 2, o.ctypes.data)
        fl[f"fftfilt_{nm}"] = o[: 2 * k].copy()
    h = ref.ref_fftfilt_new(-1.0, 2 * 3000 / 48000, 2048)              # DSBFilter (ssbdemod.cpp:92), runDSB
    o = np.zeros(xr.size + 4096, np.float32)
    k = ref.ref_fftfilt_run(h, 3, xr.ctypes.data, xr.size // 2, o.ctypes.data)
    fl["fftfilt_dsb2048"] = o[: 2 * k].copy()
    a = synth.noise_iq(2048, 33, 30000).astype(np.float32); ref.ref_gfft(a.ctypes.data, 2048, 0); fl["gfft2048_fwd"] = a
    y = fl["fftfilt_usb"]
    for kind, nm in ((0, "delta"), (1, "atan2")):
        o = np.zeros(y.size // 2, np.float32)
        ref.ref_discri(kind, 24.0, y.ctypes.data, y.size // 2, o.ctypes.data)
        fl[f"discri_{nm}"] = o
    # audio FIRs of the NFM tail (nfmdemod.cpp:428-429) on the discriminator output
    ref.ref_fir_new.restype = vp; ref.ref_fir_new.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    ref.ref_fir_run.argtypes = [vp, vp, i64, vp]
    d = np.ascontiguousarray(fl["discri_delta"])
    for kind, nm, a, b in ((0, "lowpass301", 250.0, 0.0), (1, "bandpass301", 300.0, 3000.0)):
        h = ref.ref_fir_new(kind, 301, 48000.0, a, b); o = np.zeros(d.size, np.float32)
        ref.ref_fir_run(h, d.ctypes.data, d.size, o.ctypes.data)
        fl[nm] = o
    np.savez_compressed(os.path.join(HERE, "float_golden.npz"), **fl)

    # ---- float half-band decimators: DecimatorsFI / FF / IF<qint16,12> (oracle/ref_shim_f.cpp)
    ref.ref_fdecim_new.restype = vp; ref.ref_fdecim_new.argtypes = [C.c_int] * 3
    ref.ref_fdecim_free.argtypes = [vp]
    ref.ref_fdecim_process.restype = C.c_int; ref.ref_fdecim_process.argtypes = [vp, C.c_int, C.c_int, vp, i32, vp]
    fd = {}
    for kind, (ik, ok, bits) in {"fi": (0, 0, 16), "ff": (0, 1, 16), "if12": (1, 1, 12)}.items():
        for L, fc in synth.FDECIM_CASES:
            n = 3000 if L <= 2 else 24000
            x = synth.fdecim_input(kind, n, 100 + 7 * L + fc)
            h = ref.ref_fdecim_new(ik, ok, bits)
            outs = []
            for a, b in synth.fdecim_cuts(n):
                seg = np.ascontiguousarray(x[2 * a: 2 * b])
                o = np.zeros(seg.size + 8, np.int16 if ok == 0 else np.float32)
                k = ref.ref_fdecim_process(h, L, fc, seg.ctypes.data, seg.size, o.ctypes.data)
                outs.append(o[: 2 * k].copy())
            ref.ref_fdecim_free(h)
            fd[f"{kind}_L{L}_fc{fc}"] = np.concatenate(outs)
    np.savez_compressed(os.path.join(HERE, "fdecim_golden.npz"), **fd)
    print("golden written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
