"""Property tests (hypothesis): however a stream is cut into calls, the GPU objects produce what the oracle produces for
the same cuts -- Decimators (any K / fcPos / input width, cuts anywhere: tails are dropped per call like the reference),
the channelizer bank (cuts anywhere, nothing dropped) and the float decimators."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu
SET = dict(max_examples=30, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


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
