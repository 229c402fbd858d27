"""ctypes face of the CPU oracle (oracle/libsdro.so).  TEST INFRASTRUCTURE: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only -- never by sdrangel_amd/."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib(fast: bool = False):
    global _lib
    if fast:
        p = os.path.join(ROOT, "oracle", "libsdro_fast.so")
        L = C.CDLL(p)
        _sig(L)
        return L
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "libsdro.so"))
        _sig(_lib)
    return _lib


def _sig(L):
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.sdro_decim_new.restype = vp; L.sdro_decim_new.argtypes = [C.c_int] * 3
    L.sdro_decim_free.argtypes = [vp]; L.sdro_decim_reset.argtypes = [vp]
    L.sdro_decim_process.restype = i32; L.sdro_decim_process.argtypes = [vp, vp, i32, vp]
    L.sdro_decim_group_int16.restype = i32; L.sdro_decim_group_int16.argtypes = [C.c_int] * 2
    L.sdro_chan_plan.restype = i32; L.sdro_chan_plan.argtypes = [i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
    L.sdro_chain_new.restype = vp; L.sdro_chain_new.argtypes = [i32, vp]
    L.sdro_chain_free.argtypes = [vp]; L.sdro_chain_reset.argtypes = [vp]
    L.sdro_chain_feed.restype = i64; L.sdro_chain_feed.argtypes = [vp, vp, i64, vp]


class Decim:
    def __init__(self, log2, fcpos, bits, fast=False):
        self.L = lib(fast)
        self.h = self.L.sdro_decim_new(log2, fcpos, bits)
        assert self.h
        self.log2 = log2

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdro_decim_free(self.h); self.h = None

    def process(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.int16)
        out = np.empty(buf.size + 8, np.int16)
        n = self.L.sdro_decim_process(self.h, buf.ctypes.data, buf.size, out.ctypes.data)
        return out[: 2 * n].copy()


def chan_plan(in_rate, req_rate, req_fc):
    modes = np.zeros(40, np.uint8)
    r, f = C.c_int32(), C.c_int32()
    n = lib().sdro_chan_plan(in_rate, req_rate, req_fc, modes.ctypes.data, C.byref(r), C.byref(f))
    return modes[:n].copy(), r.value, f.value


class Chain:
    def __init__(self, modes):
        modes = np.ascontiguousarray(modes, dtype=np.uint8)
        self.n = modes.size
        self.h = lib().sdro_chain_new(self.n, modes.ctypes.data)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().sdro_chain_free(self.h); self.h = None

    def feed(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        out = np.empty(iq.size + 8, np.int16)
        n = lib().sdro_chain_feed(self.h, iq.ctypes.data, iq.size // 2, out.ctypes.data)
        return out[: 2 * n].copy()


def synth_iq(n_cplx, seed=1, amp=2047, tone=None):
    """Portable synthetic I/Q: uniform noise in [-amp, amp] (+ optional complex tone (freq, amplitude))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.integers(-amp, amp + 1, size=2 * n_cplx, dtype=np.int64)
    if tone is not None:
        f, a = tone
        t = np.arange(n_cplx, dtype=np.float64)
        x[0::2] += np.round(a * np.cos(2 * np.pi * f * t)).astype(np.int64)
        x[1::2] += np.round(a * np.sin(2 * np.pi * f * t)).astype(np.int64)
    return np.clip(x, -32768, 32767).astype(np.int16)
