"""CPU: float part of the oracle (oracle/sdro_float.c) against golden vectors produced by the
reference's own NCO / Interpolator / g_fft / fftfilt / PhaseDiscriminators (strict-IEEE scalar build,
tests/golden/make_golden.py).  Bit-identical, except atan2f-based outputs (libm version dependent)."""
import ctypes as C
import os

import numpy as np

from tests import oracle_py as orc
from tests import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_float_oracle_vs_reference_vectors():
    g = np.load(os.path.join(G, "float_golden.npz"))
    L = orc.lib(); orc._sig_float(L)
    t = np.zeros(4096, np.float32); L.sdro_nco_table(t.ctypes.data)
    assert np.array_equal(bits(t), bits(g["nco_table"]))
    xin = synth.mix(6000, 31, 12000, 6000, 1)
    for name, (nf, ir, orr, cut, tpp) in {"nfm": (-4567.0, 60000.0, 48000.0, 12500 / 2.2, 4.5), "ssb": (20000.0, 120000.0, 48000.0, 5000.0, 2.0)}.items():
        o = orc.Backend(ir, nf, orr, cut, tpp)
        y = np.concatenate([o.feed(xin[2 * a: 2 * b]) for a, b in ((0, 1234), (1234, 1235), (1235, 6000))])
        assert np.array_equal(bits(y), bits(g[f"resamp_{name}"])), name
    v = synth.noise_iq(1024, 32, 30000).astype(np.float32)
    for inv, nm in ((0, "fwd"), (1, "inv")):
        a = v.copy(); L.sdro_gfft(a.ctypes.data, 1024, inv)
        assert np.array_equal(bits(a), bits(g[f"gfft1024_{nm}"])), nm
    xr = g["resamp_nfm"]
    for mode, nm in ((0, "filt"), (1, "usb"), (2, "lsb")):
        h = L.sdro_fftfilt_new(300 / 48000, 3000 / 48000, 1024)
        o = np.zeros(xr.size + 2048, np.float32)
        k = L.sdro_fftfilt_run(h, mode, np.ascontiguousarray(xr).ctypes.data, xr.size // 2, o.ctypes.data)
        assert np.array_equal(bits(o[: 2 * k]), bits(g[f"fftfilt_{nm}"])), nm
        L.sdro_fftfilt_free(h)
    h = L.sdro_fftfilt_new(-1.0, 2 * 3000 / 48000, 2048)
    o = np.zeros(xr.size + 4096, np.float32)
    k = L.sdro_fftfilt_run(h, 3, np.ascontiguousarray(xr).ctypes.data, xr.size // 2, o.ctypes.data)
    assert np.array_equal(bits(o[: 2 * k]), bits(g["fftfilt_dsb2048"]))
    a = synth.noise_iq(2048, 33, 30000).astype(np.float32); L.sdro_gfft(a.ctypes.data, 2048, 0)
    assert np.array_equal(bits(a), bits(g["gfft2048_fwd"]))
    y = np.ascontiguousarray(g["fftfilt_usb"])
    o = np.zeros(y.size // 2, np.float32); L.sdro_discri(0, 24.0, y.ctypes.data, y.size // 2, o.ctypes.data)
    assert np.array_equal(bits(o), bits(g["discri_delta"]))
    dd = np.ascontiguousarray(g["discri_delta"])
    assert np.array_equal(bits(orc.Fir(0, 301, 48000.0, 250.0).run(dd)), bits(g["lowpass301"]))
    assert np.array_equal(bits(orc.Fir(1, 301, 48000.0, 300.0, 3000.0).run(dd)), bits(g["bandpass301"]))
    o = np.zeros(y.size // 2, np.float32); L.sdro_discri(1, 24.0, y.ctypes.data, y.size // 2, o.ctypes.data)
    assert np.max(np.abs(o - g["discri_atan2"])) <= 1e-5        # atan2f: same libm here, but do not rely on it


def test_gfft_agrees_with_numpy_fft_numerically():
    """independent sanity: the restated butterfly network IS a DFT (loose tolerance; parity is pinned above)"""
    L = orc.lib(); orc._sig_float(L)
    rng = np.random.default_rng(0)
    for n in (16, 128, 1024, 8192):
        x = rng.standard_normal(2 * n).astype(np.float32)
        a = x.copy(); L.sdro_gfft(a.ctypes.data, n, 0)
        want = np.fft.fft(x[0::2].astype(np.float64) + 1j * x[1::2].astype(np.float64))
        got = a[0::2] + 1j * a[1::2]
        assert np.max(np.abs(got - want)) < 1e-3 * np.sqrt(n)
        b = a.copy(); L.sdro_gfft(b.ctypes.data, n, 1)
        assert np.max(np.abs(b - x)) < 1e-4


def test_float_decimators_oracle_vs_reference_vectors():
    """DecimatorsFI / FF / IF<qint16,12>: oracle/sdro_fdecim.c against outputs of the compiled reference classes"""
    g = np.load(os.path.join(G, "fdecim_golden.npz"))
    for kind, okind, nbits in (("fi", "fi", 16), ("ff", "ff", 16), ("if12", "if", 12)):
        for L, fc in synth.FDECIM_CASES:
            n = 3000 if L <= 2 else 24000
            x = synth.fdecim_input(kind, n, 100 + 7 * L + fc)
            o = orc.FDecim(okind, L, fc, nbits)
            y = np.concatenate([o.process(x[2 * a: 2 * b]) for a, b in synth.fdecim_cuts(n)])
            want = g[f"{kind}_L{L}_fc{fc}"]
            assert y.size == want.size and np.array_equal(y.view(np.uint8), want.view(np.uint8)), (kind, L, fc)
