"""GPU parity: sdrx_decim_* (HIP, through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["1", "4"], autouse=True)
def _fast_flavour(request, monkeypatch):
    """both flavours of the FAST kernel: single-wave workgroups (long launches) and four-wave workgroups (short ones);
    the library picks by launch size, the tests pin each in turn"""
    monkeypatch.setenv("SDRX_DECIM_NW", request.param)

BITS = (12, 8, 16)


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("fcpos", (sa.FC_CEN, sa.FC_INF, sa.FC_SUP))
@pytest.mark.parametrize("log2", range(0, 7))
def test_decim_matches_oracle_split_calls(log2, fcpos, bits):
    n = 3 * 32768 + 4096 + 200            # ragged: not a multiple of anything useful
    amp = {8: 127, 12: 2047, 16: 32767}[bits]
    x = orc.synth_iq(n, seed=100 + log2 * 9 + fcpos * 3 + bits, amp=amp, tone=(0.0021, 0.5 * amp))
    g = sa.Decimators(log2, fcpos, bits)
    o = orc.Decim(log2, fcpos, bits)
    # split into calls of awkward sizes (int16 counts), incl. one shorter than a group and an empty one
    cuts = [0, 2 * 4096, 2 * 4096 + 6, 2 * 4096 + 6, 2 * 40000 + 2, 2 * 90001, 2 * n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = x[a:b]
        got = g.decimate(seg)
        want = o.process(seg)
        assert got.size == want.size, (log2, fcpos, bits, a, b)
        assert np.array_equal(got, want), (log2, fcpos, bits, a, b, int((got != want).sum()))


@pytest.mark.parametrize("path", ("fast", "exact"))
@pytest.mark.parametrize("fcpos", (sa.FC_CEN, sa.FC_INF, sa.FC_SUP))
@pytest.mark.parametrize("log2", range(1, 7))
def test_each_kernel_path_alone(log2, fcpos, path, monkeypatch):
    """SDRX_DECIM_PATH pins one kernel: `fast` = wave-private dot2 kernel WITHOUT its fallback (valid
    on contract-honouring 12-bit data, where no int16 overflow can occur), `exact` = int32 kernel."""
    monkeypatch.setenv("SDRX_DECIM_PATH", path)
    n = 2 * 32768 + 4096 * 3 + 1024 + 256
    x = orc.synth_iq(n, seed=500 + log2 + 7 * fcpos, amp=1400, tone=(0.0017, 600))
    g = sa.Decimators(log2, fcpos, 12)
    o = orc.Decim(log2, fcpos, 12)
    for a, b in ((0, 2 * 5000), (2 * 5000, 2 * 70016), (2 * 70016, 2 * n)):
        got, want = g.decimate(x[a:b]), o.process(x[a:b])
        assert got.size == want.size and np.array_equal(got, want), (log2, fcpos, path, a)


@pytest.mark.parametrize("fcpos", (sa.FC_CEN, sa.FC_INF))
def test_decim64_full_scale_wrap(fcpos):
    """int16 extremes (-32768 everywhere, alternating full scale): exercises int32 wrap + non-mad24 stages."""
    n = 2 * 65536
    x = np.empty(2 * n, np.int16)
    x[0::2] = -32768
    x[1::2] = np.where(np.arange(n) % 2 == 0, 32767, -32768)
    x[: 2 * 5000] = orc.synth_iq(5000, seed=7, amp=32767)
    for bits in (12, 16, 8):
        g = sa.Decimators(6, fcpos, bits)
        o = orc.Decim(6, fcpos, bits)
        assert np.array_equal(g.decimate(x), o.process(x))


def test_decim_reset_and_state_roundtrip():
    x = orc.synth_iq(50000, seed=3, amp=2047, tone=(0.001, 1500))
    g = sa.Decimators(6, sa.FC_CEN, 12)
    a1 = g.decimate(x[: 2 * 20032])
    st = g.get_state()
    a2 = g.decimate(x[2 * 20032:])
    g.set_state(st)
    a3 = g.decimate(x[2 * 20032:])
    assert np.array_equal(a2, a3)
    g.reset()
    b = g.decimate(x[: 2 * 20032])
    assert np.array_equal(a1, b)


def test_decim_device_path_large():
    """Device-resident call at a BASELINE-sized buffer (10 M samples, cfg 2) vs the oracle, bit-exact."""
    torch = pytest.importorskip("torch")
    n = 10_000_000
    x = orc.synth_iq(n, seed=11, amp=2047, tone=(0.0005, 1000))
    d_in = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()                         # handle 0 (torch's default stream) means "own stream" to the library: not ordered with it
    d_out = torch.empty(2 * (n >> 6) + 64, dtype=torch.int16, device="cuda")
    g = sa.Decimators(6, sa.FC_CEN, 12)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    n_out = g.decimate_dev(d_in.data_ptr(), x.size, d_out.data_ptr())
    torch.cuda.synchronize()
    got = d_out[: 2 * n_out].cpu().numpy()
    want = orc.Decim(6, sa.FC_CEN, 12).process(x)
    assert n_out == want.size // 2 == (n // 64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("fcpos", (sa.FC_CEN, sa.FC_INF, sa.FC_SUP))
@pytest.mark.parametrize("log2", range(0, 7))
def test_decimators_u8_rtlsdr_flavour(log2, fcpos):
    """DecimatorsU<qint32, quint8, 16, 8, 127>: unsigned 8-bit I/Q, split calls, ragged tails"""
    from tests import synth
    n = 2 * 32768 + 4096 + 777
    x = (synth.lcg_u32(2 * n, 900 + log2 * 3 + fcpos) & 0xff).astype(np.uint8)
    x[: 2 * 3000] = 0; x[2 * 3000: 2 * 6000] = 255
    g = sa.DecimatorsU(log2, fcpos, 127)
    o = orc.DecimU(log2, fcpos, 127)
    for a, b in ((0, 2 * 5000 + 2), (2 * 5000 + 2, 2 * 5000 + 2), (2 * 5000 + 2, 2 * 40001), (2 * 40001, 2 * n)):
        got, want = g.decimate(x[a:b]), o.process(x[a:b])
        assert got.size == want.size and np.array_equal(got, want), (log2, fcpos, a, b)
