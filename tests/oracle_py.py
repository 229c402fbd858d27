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
        # SDRO_LIB: another build of the same restatement (tests/test_sanitizers.py points it at the ASan/UBSan build)
        _lib = C.CDLL(os.environ.get("SDRO_LIB") or os.path.join(ROOT, "oracle", "libsdro.so"))
        _sig(_lib)
    return _lib


def _sig(L):
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.sdro_decim_new.restype = vp; L.sdro_decim_new.argtypes = [C.c_int] * 3
    L.sdro_decim_free.argtypes = [vp]; L.sdro_decim_reset.argtypes = [vp]
    L.sdro_decim_process.restype = i32; L.sdro_decim_process.argtypes = [vp, vp, i32, vp]
    L.sdro_decim_switch.argtypes = [vp, C.c_int, C.c_int]
    L.sdro_decimu_new.restype = vp; L.sdro_decimu_new.argtypes = [C.c_int] * 3
    L.sdro_decimu_process.restype = i32; L.sdro_decimu_process.argtypes = [vp, vp, i32, vp]
    L.sdro_decim_group_int16.restype = i32; L.sdro_decim_group_int16.argtypes = [C.c_int] * 2
    L.sdro_chan_plan.restype = i32; L.sdro_chan_plan.argtypes = [i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
    L.sdro_chain_new.restype = vp; L.sdro_chain_new.argtypes = [i32, vp]
    L.sdro_chain_free.argtypes = [vp]; L.sdro_chain_reset.argtypes = [vp]
    L.sdro_chain_feed.restype = i64; L.sdro_chain_feed.argtypes = [vp, vp, i64, vp]
    L.sdro_decim24_new.restype = vp; L.sdro_decim24_new.argtypes = [C.c_int] * 3
    L.sdro_decim24_process.restype = i32; L.sdro_decim24_process.argtypes = [vp, vp, i32, vp]
    L.sdro_chain24_new.restype = vp; L.sdro_chain24_new.argtypes = [i32, vp]
    L.sdro_chain24_feed.restype = i64; L.sdro_chain24_feed.argtypes = [vp, vp, i64, vp]


def _sig_fdecim(L):
    vp, i32 = C.c_void_p, C.c_int32
    L.sdro_fdecim_new.restype = vp; L.sdro_fdecim_new.argtypes = [C.c_int] * 5
    L.sdro_fdecim_free.argtypes = [vp]; L.sdro_fdecim_reset.argtypes = [vp]
    L.sdro_fdecim_process.restype = i32; L.sdro_fdecim_process.argtypes = [vp, vp, i32, vp]
    L.sdro_fdecim_group.restype = i32; L.sdro_fdecim_group.argtypes = [C.c_int] * 2
    L.sdro_fdecim_switch.argtypes = [vp, C.c_int, C.c_int]


class FDecim:
    """oracle DecimatorsFI ("fi") / DecimatorsFF ("ff") / DecimatorsIF<qint16,bits> ("if"), one (log2, fcpos)"""
    KINDS = {"fi": (0, 0), "ff": (0, 1), "if": (1, 1)}

    def __init__(self, kind, log2, fcpos, bits=16):
        self.L = lib(); _sig_fdecim(self.L)
        self.ik, self.ok = self.KINDS[kind]
        self.h = self.L.sdro_fdecim_new(log2, fcpos, self.ik, self.ok, bits)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdro_fdecim_free(self.h); self.h = None

    def process(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.float32 if self.ik == 0 else np.int16)
        out = np.zeros(buf.size + 8, np.int16 if self.ok == 0 else np.float32)
        n = self.L.sdro_fdecim_process(self.h, buf.ctypes.data, buf.size, out.ctypes.data)
        return out[: 2 * n]

    def switch(self, log2, fcpos):
        """the next call runs another decimateK_x of the same object (same six filters)"""
        self.L.sdro_fdecim_switch(self.h, log2, fcpos)


def _sig_float(L):
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.sdro_nco_table.argtypes = [vp]
    L.sdro_nco_inc.restype = i32; L.sdro_nco_inc.argtypes = [f32, f32]
    L.sdro_backend_new.restype = vp; L.sdro_backend_new.argtypes = [f32, f32, f32, i32, f32, f32]
    L.sdro_backend_free.argtypes = [vp]
    L.sdro_backend_feed.restype = i64; L.sdro_backend_feed.argtypes = [vp, vp, i64, vp]
    L.sdro_backend_ntaps.restype = i32; L.sdro_backend_ntaps.argtypes = [vp]
    L.sdro_backend_taps.restype = C.POINTER(C.c_float); L.sdro_backend_taps.argtypes = [vp]
    L.sdro_gfft.argtypes = [vp, i32, i32]
    L.sdro_fftfilt_new.restype = vp; L.sdro_fftfilt_new.argtypes = [f32, f32, i32]
    L.sdro_fftfilt_new_asym.restype = vp; L.sdro_fftfilt_new_asym.argtypes = [f32, f32, i32]
    L.sdro_fftfilt_free.argtypes = [vp]
    L.sdro_fftfilt_filter.restype = C.POINTER(C.c_float); L.sdro_fftfilt_filter.argtypes = [vp]
    L.sdro_fftfilt_run.restype = i64; L.sdro_fftfilt_run.argtypes = [vp, i32, vp, i64, vp]
    L.sdro_discri.argtypes = [i32, f32, vp, i64, vp]


class Fir:
    """oracle Lowpass (kind 0) / Bandpass (kind 1), streaming"""

    def __init__(self, kind, ntaps, rate, f1, f2=0.0):
        L = lib()
        L.sdro_fir_new.restype = C.c_void_p; L.sdro_fir_new.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]
        L.sdro_fir_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.sdro_fir_taps.restype = C.c_int32; L.sdro_fir_taps.argtypes = [C.c_void_p, C.c_void_p]
        self.L = L
        self.h = L.sdro_fir_new(kind, ntaps, rate, f1, f2)

    def taps(self):
        t = np.zeros(4096, np.float32)
        n = self.L.sdro_fir_taps(self.h, t.ctypes.data)
        return t[:n].copy()

    def run(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(max(x.size, 1), np.float32)
        self.L.sdro_fir_run(self.h, x.ctypes.data, x.size, out.ctypes.data)
        return out[: x.size].copy()


class Backend:
    """oracle of one channel's NCO -> Interpolator -> [fftfilt] -> [discriminator] chain, streaming"""

    def __init__(self, in_rate, nco_freq, out_rate, cutoff, tpp, filt_mode=0, f1=0.0, f2=0.0, discri=0, fm_scaling=1.0):
        L = lib(); _sig_float(L)
        self.L = L
        self.b = L.sdro_backend_new(float(nco_freq), float(in_rate), float(out_rate), 16, cutoff, tpp)
        self.flen = 2048 if filt_mode >= 4 else 1024
        self.f = (L.sdro_fftfilt_new_asym(f1, f2, 2048) if filt_mode >= 5 else L.sdro_fftfilt_new(-1.0, f2, 2048) if filt_mode == 4 else L.sdro_fftfilt_new(f1, f2, 1024)) if filt_mode else None
        self.filt_mode, self.discri, self.fm = filt_mode, discri, fm_scaling
        self.last = None      # last sample seen by the discriminator (carried state)
        self.prev_tail = np.zeros(0, np.float32)

    def taps(self):
        nt = self.L.sdro_backend_ntaps(self.b)
        return nt, np.ctypeslib.as_array(self.L.sdro_backend_taps(self.b), shape=(16 * nt,)).copy()

    def filter(self):
        return np.ctypeslib.as_array(self.L.sdro_fftfilt_filter(self.f), shape=(2 * self.flen,)).copy()

    def feed(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        n = iq.size // 2
        res = np.empty(2 * n + 8, np.float32)
        k = self.L.sdro_backend_feed(self.b, iq.ctypes.data, n, res.ctypes.data)
        x = res[: 2 * k].copy()
        if self.filt_mode:
            y = np.empty(x.size + 4096, np.float32)
            m = self.L.sdro_fftfilt_run(self.f, self.filt_mode - 1, x.ctypes.data, k, y.ctypes.data)   # 1 runFilt, 2 usb, 3 lsb -> 0, 1, 2
            x = y[: 2 * m].copy()
        if not self.discri:
            return x
        # the oracle's discriminator function is stateless per call: replay it over (carried sample + new ones)
        full = np.concatenate([self.prev_tail, x]).astype(np.float32)
        out = np.empty(full.size // 2 + 1, np.float32)
        self.L.sdro_discri(self.discri - 1, self.fm, full.ctypes.data, full.size // 2, out.ctypes.data)
        skip = self.prev_tail.size // 2
        if x.size:
            self.prev_tail = x[-2:].copy()
        return out[skip: full.size // 2].copy()


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


class DecimU:
    def __init__(self, log2, fcpos, shift=127):
        self.L = lib()
        self.h = self.L.sdro_decimu_new(log2, fcpos, shift)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdro_decim_free(self.h); self.h = None

    def process(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        out = np.empty(buf.size + 8, np.int16)
        n = self.L.sdro_decimu_process(self.h, buf.ctypes.data, buf.size, out.ctypes.data)
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


class Decim24:
    """Decimators<qint32, qint16, 24, bits> (the reference's SDR_RX_SAMPLE_24BIT build)"""

    def __init__(self, log2, fcpos, bits):
        self.h = lib().sdro_decim24_new(log2, fcpos, bits)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().sdro_decim_free(self.h); self.h = None

    def process(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.int16)
        out = np.empty(buf.size + 8, np.int32)
        n = lib().sdro_decim24_process(self.h, buf.ctypes.data, buf.size, out.ctypes.data)
        return out[: 2 * n].copy()


class Chain24:
    """one DownChannelizer stage chain of the 24-bit build: {int32, int32} in and out"""

    def __init__(self, modes):
        modes = np.ascontiguousarray(modes, dtype=np.uint8)
        self.n = modes.size
        self.h = lib().sdro_chain24_new(self.n, modes.ctypes.data)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().sdro_chain_free(self.h); self.h = None

    def feed(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int32)
        out = np.empty(iq.size + 8, np.int32)
        n = lib().sdro_chain24_feed(self.h, iq.ctypes.data, iq.size // 2, out.ctypes.data)
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


class DcCorr:
    """oracle DC offset correction (DSPDeviceSourceEngine::iqCorrections, DC only), streaming"""

    def __init__(self):
        self.L = lib()
        self.L.sdro_dccorr_new.restype = C.c_void_p
        self.L.sdro_dccorr_free.argtypes = [C.c_void_p]
        self.L.sdro_dccorr_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        self.h = self.L.sdro_dccorr_new()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdro_dccorr_free(self.h); self.h = None

    def process(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        out = np.empty_like(iq)
        self.L.sdro_dccorr_process(self.h, iq.ctypes.data, iq.size // 2, out.ctypes.data)
        return out


class IqImb:
    """oracle of DSPDeviceSourceEngine::iqCorrections(.., imbalanceCorrection=true), float flavour (oracle/sdro.c)"""

    def __init__(self):
        self.L = lib()
        self.L.sdro_iqimb_new.restype = C.c_void_p
        self.L.sdro_iqimb_free.argtypes = [C.c_void_p]
        self.L.sdro_iqimb_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        self.h = self.L.sdro_iqimb_new()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdro_iqimb_free(self.h); self.h = None

    def process(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        out = np.empty_like(iq)
        self.L.sdro_iqimb_process(self.h, iq.ctypes.data, iq.size // 2, out.ctypes.data)
        return out


class AudioTailOracle:
    """oracle/sdro_audio.c: kind 0 = NFM tail, 1 = SSB tail"""

    def __init__(self, kind, **k):
        self.L = lib(); self.kind = kind
        L = self.L
        L.sdro_nfmtail_new.restype = C.c_void_p; L.sdro_nfmtail_new.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_float]
        L.sdro_nfmtail_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; L.sdro_nfmtail_free.argtypes = [C.c_void_p]
        L.sdro_ssbtail_new.restype = C.c_void_p; L.sdro_ssbtail_new.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_float]
        L.sdro_ssbtail_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; L.sdro_ssbtail_free.argtypes = [C.c_void_p]
        if kind == 0:
            self.h = L.sdro_nfmtail_new(k["audio_rate"], k["fm_scaling"], k["squelch_level"], k["squelch_gate"], k["volume"], k["af_bandwidth"])
        else:
            self.h = L.sdro_ssbtail_new(k["agc_active"], k["agc_nb_samples"], k["agc_threshold"], k["agc_threshold_enable"], k["agc_gate"], k["agc_clamping"], k["volume"])

    def __del__(self):
        if getattr(self, "h", None):
            (self.L.sdro_nfmtail_free if self.kind == 0 else self.L.sdro_ssbtail_free)(self.h); self.h = None

    def feed(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(max(x.size // 2, 1), np.int16)
        (self.L.sdro_nfmtail_process if self.kind == 0 else self.L.sdro_ssbtail_process)(self.h, x.ctypes.data, x.size // 2, out.ctypes.data)
        return out[: x.size // 2]


class Iir:
    """oracle IIRFilter<float, Order> (oracle/sdro_audio.c)"""

    def __init__(self, order, a, b):
        self.L = lib()
        self.L.sdro_iir_new.restype = C.c_void_p; self.L.sdro_iir_new.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
        self.L.sdro_iir_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]; self.L.sdro_iir_free.argtypes = [C.c_void_p]
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        self.h = self.L.sdro_iir_new(order, a.ctypes.data, b.ctypes.data)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdro_iir_free(self.h); self.h = None

    def run(self, x):
        x = np.ascontiguousarray(x, np.float32); out = np.zeros(max(x.size, 1), np.float32)
        self.L.sdro_iir_run(self.h, x.ctypes.data, x.size, out.ctypes.data)
        return out[: x.size]
