"""BASELINE.json's full bench sizes, on the GPU: bit-exact against the C oracle where it finishes in seconds, and
size-independent properties (one call == the same stream in ragged calls; a decimated DC level is the DC level times
the chain gain) where it does not.  Inputs are built on the device so that nothing but the library touches them."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dev_noise(n_cplx, seed, amp=2047):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.randint(-amp, amp + 1, (2 * n_cplx,), generator=g, device="cuda", dtype=torch.int32)
    t = torch.arange(n_cplx, device="cuda", dtype=torch.float32)
    x[0::2] += (500 * torch.cos(2 * torch.pi * 0.0007 * t)).to(torch.int32)
    x[1::2] += (500 * torch.sin(2 * torch.pi * 0.0007 * t)).to(torch.int32)
    x = x.clamp_(-32768, 32767).to(torch.int16)
    # torch's default stream has the handle 0, which the library reads as "use the handle's own stream": the two are not
    # ordered against each other, so the generated data has to be complete before the library is pointed at it
    torch.cuda.synchronize()
    return x


def test_decim64_bench_batch_bit_exact_and_split_invariant():
    """cfg 2 at the bench's own batch: 1 Gi samples (4 GiB) resident, decimate64_cen -- the launch size at which the library
    switches to 64 sub-chunks per segment"""
    n = 1024 * 1024 * 1024
    x = _dev_noise(n, 5489)
    out = torch.empty(2 * (n >> 6) + 64, dtype=torch.int16, device="cuda")
    g = sa.Decimators(6, sa.FC_CEN, 12)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    n_out = g.decimate_dev(x.data_ptr(), 2 * n, out.data_ptr())
    torch.cuda.synchronize()
    assert n_out == n >> 6
    whole = out[: 2 * n_out].cpu().numpy().copy()
    # the reference's algorithm on the host over the SAME 4 GiB (about 15 s of one core)
    xh = x.cpu().numpy()
    o = orc.Decim(6, sa.FC_CEN, 12)                          # its C entry point takes the reference's qint32 `len`: feed it in four calls
    q = n // 4
    want = np.concatenate([o.process(xh[2 * i * q: 2 * (i + 1) * q]) for i in range(4)])
    bad = np.nonzero(whole != want)[0] if whole.size == want.size else np.arange(1)
    assert bad.size == 0, (whole.size, want.size, bad.size, bad[:8].tolist(), bad[-4:].tolist())
    # the same stream as three device-resident calls of uneven size (state carried, group-aligned cuts)
    g.reset()
    cuts = [0, 64 * 1_000_003, 64 * 9_000_001, n]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        o2 = torch.empty(2 * ((b - a) >> 6) + 64, dtype=torch.int16, device="cuda")
        k = g.decimate_dev(x[2 * a:].data_ptr(), 2 * (b - a), o2.data_ptr())
        torch.cuda.synchronize()
        parts.append(o2[: 2 * k].cpu().numpy())
    assert np.array_equal(np.concatenate(parts), whole)


def test_fi64_bench_batch_bit_exact():
    """SURVEY 8f.4 at the bench's batch: DecimatorsFI::decimate64_cen over 128 Mi float samples"""
    n = 128 * 1024 * 1024
    x = (_dev_noise(n, 77).to(torch.float32) / 4096.0).contiguous()
    torch.cuda.synchronize()
    out = torch.empty(2 * (n >> 6) + 64, dtype=torch.int16, device="cuda")
    g = sa.FloatDecimators("fi", 6, sa.FC_CEN)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    n_out = g.decimate_dev(x.data_ptr(), 2 * n, out.data_ptr())
    torch.cuda.synchronize()
    assert n_out == n >> 6
    want = orc.FDecim("fi", 6, sa.FC_CEN).process(x.cpu().numpy())
    assert np.array_equal(out[: 2 * n_out].cpu().numpy(), want)


def test_chan32_bench_feed_equals_ragged_feeds_and_oracle_prefix():
    """cfg 3 at the bench's feed size (64 Mi samples): one feed == the same stream in ragged feeds, channel by channel;
    the first 2 Mi input samples of four channels are also checked against the oracle chains"""
    fs, n = 61_440_000, 64 * 1024 * 1024
    k = np.arange(32)
    fcs = [int(v) for v in (-15_000_000 + k * (30_000_000 / 31) + 137 * k)]
    x = _dev_noise(n, 4242)
    stream = torch.cuda.current_stream().cuda_stream
    one = sa.ChannelizerBank(fs, [48000] * 32, fcs); one.set_stream(stream)
    one.feed_dev(x.data_ptr(), n)
    whole = [one.read(c) for c in range(32)]
    rag = sa.ChannelizerBank(fs, [48000] * 32, fcs); rag.set_stream(stream)
    cuts = [0, 1, 4099, 10_000_001, 10_000_001, 33_333_333, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        rag.feed_dev(x[2 * a:].data_ptr(), b - a)
    for c in range(32):
        assert np.array_equal(rag.read(c), whole[c]), c
    head = x[: 2 * (2 << 20)].cpu().numpy()
    for c in (0, 9, 17, 31):
        modes, out_rate, ofs = one.info(c)
        want = orc.Chain(modes).feed(head)
        assert np.array_equal(whole[c][: want.size], want), c


def test_dc_level_through_every_chain_length():
    """a constant input comes out as the constant times the chain's DC gain (2 per stage, then the post shift):
    decimation_shifts<16,12> keeps 12-bit full scale at 16-bit full scale for every K (decimators.h:116-131)"""
    n = 1 << 22
    x = torch.full((2 * n,), 1000, dtype=torch.int16, device="cuda")
    x[1::2] = -700
    torch.cuda.synchronize()
    for log2 in range(1, 7):
        g = sa.Decimators(log2, sa.FC_CEN, 12)
        out = torch.empty(2 * (n >> log2) + 64, dtype=torch.int16, device="cuda")
        k = g.decimate_dev(x.data_ptr(), 2 * n, out.data_ptr())
        g.sync()
        y = out[: 2 * k].cpu().numpy().astype(np.int64)
        want = orc.Decim(log2, sa.FC_CEN, 12).process(x[: 2 * 65536].cpu().numpy()).astype(np.int64)
        settled = y[2 * 4096:]                     # past the group delay
        assert np.all(settled[0::2] == settled[0]) and np.all(settled[1::2] == settled[1])
        assert settled[0] == want[-2] and settled[1] == want[-1]
        # truncated integer coefficients: the DC gain of a stage is a little under 2 (2 * sum(c) + 2048 < 4096)
        assert abs(settled[0] - 16 * 1000) <= 160 and abs(settled[1] + 16 * 700) <= 112
