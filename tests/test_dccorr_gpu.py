"""GPU parity of the device-stream DC offset correction (DSPDeviceSourceEngine::iqCorrections, DC-only branch) against the
oracle, which tests/test_oracle_vs_ref.py pins to the reference's own MovingAverageUtil<int32_t,int64_t,1024>.  Bit-exact."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu


def _x(n, seed, dc=(1500, -900), amp=20000):
    rng = np.random.default_rng(seed)
    x = rng.integers(-amp, amp + 1, 2 * n).astype(np.int64)
    x[0::2] += dc[0]; x[1::2] += dc[1]
    return x.clip(-32768, 32767).astype(np.int16)


def test_spans_of_any_length_match_oracle():
    g, o = sa.DcCorrection(), orc.DcCorr()
    for i, n in enumerate((5, 1000, 1023, 1024, 1, 0, 3071, 3072, 3073, 70001, 2047, 1 << 20)):
        x = _x(n, 10 + i)
        assert np.array_equal(g.process(x), o.process(x)), (i, n)


def test_full_scale_wrap_and_reset():
    g, o = sa.DcCorrection(), orc.DcCorr()
    x = np.full(2 * 5000, 32767, np.int16); x[1::2] = -32768          # re - avg wraps through int16 like the reference's `-=`
    assert np.array_equal(g.process(x), o.process(x))
    g.reset(); o = orc.DcCorr()
    y = _x(4000, 3, dc=(-32000, 32000), amp=700)
    assert np.array_equal(g.process(y), o.process(y))


def test_device_path_large_and_dc_removed():
    torch = pytest.importorskip("torch")
    n = 32 * 1024 * 1024
    x = _x(n, 77, dc=(1234, -777), amp=2047)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty_like(d_in)
    torch.cuda.synchronize()                         # handle 0 (torch's default stream) means "own stream" to the library: not ordered with it
    g = sa.DcCorrection(); g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.process_dev(d_in.data_ptr(), d_out.data_ptr(), n)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got, orc.DcCorr().process(x))
    assert abs(got[2 * 4096::2].astype(np.float64).mean()) < 2 and abs(got[2 * 4096 + 1::2].astype(np.float64).mean()) < 2
    with pytest.raises(sa.SdrxError):
        g.process_dev(d_in.data_ptr(), d_in.data_ptr(), n)           # in place on the device is refused


def test_iq_imbalance_correction_matches_oracle_many_streams():
    """sdrx_iqimb_*: DSPDeviceSourceEngine::iqCorrections(begin, end, true) (dspdevicesourceengine.cpp:175-181, 217-253, float
    flavour), one lane per stream, several streams side by side, ragged calls, state carried -- bit-identical to the oracle."""
    n_str = 5
    lens = [30_000, 4_000, 1, 2_500, 70_000]
    xs = []
    for i, n in enumerate(lens):
        x = orc.synth_iq(n, seed=60 + i, amp=6000 + 2000 * i, tone=(0.013 * (i + 1), 5000)).astype(np.int32)
        x[1::2] = (x[1::2] * (0.7 + 0.05 * i)).astype(np.int32) + 40 * i          # amplitude imbalance + DC on Q
        x[0::2] += 123 - 60 * i
        xs.append(np.clip(x, -32768, 32767).astype(np.int16))
    g = sa.IqImbalance(n_str)
    os_ = [orc.IqImb() for _ in range(n_str)]
    for frac in ((0.0, 0.2), (0.2, 0.2), (0.2, 0.21), (0.21, 1.0)):
        segs = [x[2 * int(frac[0] * n): 2 * int(frac[1] * n)] for x, n in zip(xs, lens)]
        got = g.process(segs)
        for i in range(n_str):
            want = os_[i].process(segs[i])
            assert got[i].size == want.size and np.array_equal(got[i], want), (i, frac, int((got[i] != want).sum()))
    g.reset()
    again = g.process([x[: 2 * 1000] if n >= 1000 else x for x, n in zip(xs, lens)])
    for i in range(n_str):
        seg = xs[i][: 2 * 1000] if lens[i] >= 1000 else xs[i]
        assert np.array_equal(again[i], orc.IqImb().process(seg)), i
