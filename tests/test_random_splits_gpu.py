"""Property tests (hypothesis): however a stream is cut into calls, the GPU objects produce what the oracle produces for
the same cuts -- Decimators (any K / fcPos / input width, cuts anywhere: tails are dropped per call like the reference),
the channelizer bank (cuts anywhere, nothing dropped) and the float decimators."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["1", "4"], autouse=True)
def _fast_flavour(request, monkeypatch):
    """both flavours of the FAST kernel: single-wave workgroups (long launches) and four-wave workgroups (short ones);
    the library picks by launch size, the tests pin each in turn"""
    monkeypatch.setenv("SDRX_DECIM_NW", request.param)
import os
N_EX = int(os.environ.get("SDRX_HYP_EXAMPLES", "30"))           # a soak run sets this to a few hundred and drops derandomize
SET = dict(max_examples=N_EX, deadline=None, derandomize=N_EX <= 30, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


def _cuts(draw, n, k):
    pts = sorted(draw(st.lists(st.integers(0, n), min_size=0, max_size=k)))
    return [0] + pts + [n]


@settings(**SET)
@given(st.data())
def test_decimators_any_cuts(data):
    log2 = data.draw(st.integers(0, 6)); fc = data.draw(st.sampled_from([sa.FC_INF, sa.FC_SUP, sa.FC_CEN]))
    bits = data.draw(st.sampled_from([8, 12, 16]))
    n = data.draw(st.integers(1, 60000))
    amp = {8: 127, 12: 2047, 16: 32767}[bits]
    x = orc.synth_iq(n, seed=data.draw(st.integers(0, 1 << 20)), amp=amp)
    cuts = _cuts(data.draw, 2 * n, 5)                      # cuts in int16 units: odd lengths and sub-group calls included
    g, o = sa.Decimators(log2, fc, bits), orc.Decim(log2, fc, bits)
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = x[a:b]
        assert np.array_equal(g.decimate(seg), o.process(seg)), (log2, fc, bits, a, b)


@settings(**SET)
@given(st.data())
def test_channelizer_bank_any_cuts(data):
    n_ch = data.draw(st.integers(1, 6))
    fs = data.draw(st.sampled_from([2_400_000, 61_440_000, 10_000_000]))
    rates = [data.draw(st.sampled_from([12000, 48000, 96000, 250000])) for _ in range(n_ch)]
    fcs = [data.draw(st.integers(-fs // 2 + 130000, fs // 2 - 130000)) for _ in range(n_ch)]
    n = data.draw(st.integers(1, 150000))
    x = orc.synth_iq(n, seed=data.draw(st.integers(0, 1 << 20)), amp=data.draw(st.sampled_from([2047, 32767])))
    cuts = _cuts(data.draw, n, 4)
    bank = sa.ChannelizerBank(fs, rates, fcs)
    chains = [orc.Chain(orc.chan_plan(fs, r, f)[0]) for r, f in zip(rates, fcs)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = x[2 * a: 2 * b]
        bank.feed(seg)
        for c in range(n_ch):
            assert np.array_equal(bank.read(c), chains[c].feed(seg)), (fs, rates, fcs, c, a, b)


@settings(**SET)
@given(st.data())
def test_float_decimators_any_cuts(data):
    kind = data.draw(st.sampled_from(["fi", "ff", "if"])); bits = data.draw(st.sampled_from([8, 12, 16]))
    log2 = data.draw(st.integers(0, 6)); fc = data.draw(st.sampled_from([sa.FC_INF, sa.FC_SUP, sa.FC_CEN]))
    if log2 == 0:
        fc = sa.FC_CEN
    n = data.draw(st.integers(1, 40000))
    rng = np.random.default_rng(data.draw(st.integers(0, 1 << 20)))
    x = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), 2 * n).astype(np.int16) if kind == "if" else rng.uniform(-0.9, 0.9, 2 * n).astype(np.float32)
    cuts = _cuts(data.draw, 2 * n, 4)
    g, o = sa.FloatDecimators(kind, log2, fc, bits), orc.FDecim(kind, log2, fc, bits)
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = x[a:b]
        got, want = g.decimate(seg), o.process(seg)
        assert got.size == want.size and np.array_equal(got.view(np.uint8), want.view(np.uint8)), (kind, bits, log2, fc, a, b)


@settings(**SET)
@given(st.data())
def test_backend_any_ratio_any_cuts(data):
    """the demod front with random resampling ratios (dyadic ones take the closed-form schedule, the others the serial walk),
    every filter mode, and feeds cut anywhere -- 0 ulp against the oracle"""
    n_ch = data.draw(st.integers(1, 3))
    cfgs = []
    for _ in range(n_ch):
        out_rate = data.draw(st.sampled_from([48000, 44100, 24000, 37123]))
        mult = data.draw(st.sampled_from([1.0, 1.25, 1.5625, 2.5, 1.302083, 3.0, 1.1, 2.0, 5.0]))
        in_rate = int(round(out_rate * mult))
        mode = data.draw(st.integers(0, 6))
        f1, f2 = (0.04, 0.35) if mode >= 5 else (0.0, 0.12) if mode == 4 else (300 / 48000, 3000 / 48000)
        cfgs.append(dict(in_rate=in_rate, nco_freq=data.draw(st.integers(-20000, 20000)), out_rate=out_rate,
                         interp_cutoff=data.draw(st.sampled_from([3000.0, 5681.8, 9000.0])), taps_per_phase=data.draw(st.sampled_from([2.0, 4.5])),
                         filt_mode=mode, f1=f1, f2=f2, discri=data.draw(st.integers(0, 1)), fm_scaling=12.0))
    gpu = sa.BackendBank([sa.BackendCfg(**c) for c in cfgs])
    ora = [orc.Backend(c["in_rate"], c["nco_freq"], c["out_rate"], c["interp_cutoff"], c["taps_per_phase"],
                       c["filt_mode"], c["f1"], c["f2"], c["discri"], c["fm_scaling"]) for c in cfgs]
    n = data.draw(st.integers(1, 30000))
    xs = [orc.synth_iq(n, seed=data.draw(st.integers(0, 1 << 20)), amp=12000) for _ in range(n_ch)]
    cuts = _cuts(data.draw, n, 3)
    for a, b in zip(cuts[:-1], cuts[1:]):
        segs = [x[2 * a: 2 * b] for x in xs]
        gpu.feed(segs)
        for c in range(n_ch):
            want, got = ora[c].feed(segs[c]), gpu.read(c)
            assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (cfgs[c], a, b)
