"""GPU parity of sdrx_audiotail_* (audio-rate tail of the NFM / SSB demods: discriminator + squelch + delay line + Bandpass,
MagAGC + delay line + step value) against the oracle, which tests/test_oracle_vs_ref.py pins to the reference's own
PhaseDiscriminators / MovingAverageUtil / DoubleBufferFIFO / Bandpass / MagAGC classes.  qint16 output: bit-exact."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu


def bursts(n, seed, hi=9000.0, lo=30.0, period=15000, dphi=0.2):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    env = np.where((t // period) % 2 == 0, hi, lo)
    ph = np.cumsum(dphi * np.sin(2 * np.pi * t * 1000 / 48000))
    x = np.empty(2 * n, np.float32)
    x[0::2] = env * np.cos(ph) + rng.normal(0, 20, n)
    x[1::2] = env * np.sin(ph) + rng.normal(0, 20, n)
    return x


NFM = [dict(kind=0, audio_rate=48000, volume=2.0, fm_scaling=48000 / (2 * 5000.0), squelch_level=1e-6, squelch_gate=4800, af_bandwidth=3000.0),
       dict(kind=0, audio_rate=48000, volume=1.0, fm_scaling=4.8, squelch_level=3e-3, squelch_gate=100, af_bandwidth=5000.0),
       dict(kind=0, audio_rate=48000, volume=3.5, fm_scaling=9.6, squelch_level=1e-9, squelch_gate=30000, af_bandwidth=2500.0)]
SSB = [dict(kind=1, audio_rate=48000, volume=3.0, agc_active=1, agc_nb_samples=6144, agc_threshold=1e-2, agc_threshold_enable=1, agc_gate=0, agc_clamping=0),
       dict(kind=1, audio_rate=48000, volume=3.0, agc_active=1, agc_nb_samples=6144, agc_threshold=1e-4 * 32768.0 ** 2, agc_threshold_enable=1, agc_gate=192, agc_clamping=1),
       dict(kind=1, audio_rate=48000, volume=1.0, agc_active=0, agc_nb_samples=6144, agc_threshold=0.0, agc_threshold_enable=0, agc_gate=0, agc_clamping=0),
       dict(kind=1, audio_rate=48000, volume=2.0, agc_active=1, agc_nb_samples=384, agc_threshold=5e5, agc_threshold_enable=1, agc_gate=10, agc_clamping=0),
       dict(kind=1, audio_rate=48000, volume=2.0, agc_active=1, agc_nb_samples=12288, agc_threshold=0.0, agc_threshold_enable=0, agc_gate=0, agc_clamping=1)]


def cfg_struct(d):
    c = sa.AudioTailCfg()
    for k, v in d.items():
        setattr(c, k, v)
    return c


def test_nfm_and_ssb_tails_match_oracle_ragged_feeds():
    cfgs = NFM + SSB
    n_total = [120_000, 90_000, 100_000, 120_000, 80_000, 30_000, 70_000, 60_000]
    xs = [bursts(n, 10 + i, period=15000 if i % 2 == 0 else 7000) for i, n in enumerate(n_total)]
    g = sa.AudioTail([cfg_struct(c) for c in cfgs])
    os_ = [orc.AudioTailOracle(**c) for c in cfgs]
    nonzero = 0
    for a, b in ((0.0, 0.0001), (0.0001, 0.31), (0.31, 0.31), (0.31, 1.0)):
        segs = [x[2 * int(a * n): 2 * int(b * n)] for x, n in zip(xs, n_total)]
        got = g.feed(segs)
        for c in range(len(cfgs)):
            want = os_[c].feed(segs[c])
            assert got[c].size == want.size and np.array_equal(got[c], want), (c, a, b, int((got[c] != want).sum()))
            nonzero += int((want != 0).sum())
    assert nonzero > 300_000                                  # the squelch opened, the AGC passed audio
    g.reset()
    again = g.feed([x[: 2 * 6000] for x in xs])
    for c in range(len(cfgs)):
        assert np.array_equal(again[c], orc.AudioTailOracle(**cfgs[c]).feed(xs[c][: 2 * 6000])), c


def test_forty_channels_share_a_launch():
    """more channels than one workgroup holds (32 lanes); silence and full-scale inputs included"""
    cfgs = [dict(NFM[i % 3]) for i in range(25)] + [dict(SSB[i % 5]) for i in range(15)]
    xs = [bursts(9000, 100 + i, hi=20000.0 if i % 4 else 0.0) for i in range(40)]
    g = sa.AudioTail([cfg_struct(c) for c in cfgs])
    got = g.feed(xs)
    for c in range(40):
        assert np.array_equal(got[c], orc.AudioTailOracle(**cfgs[c]).feed(xs[c])), c


IIR_SPECS = [
    # FilterMbe's pair (sdrbase/dsp/filtermbe.cpp): 2nd-order low pass / high pass, a = feedback, b = feed-forward
    (2, [1.0, 1.5610180758, -0.6413515381], [0.0200833656, 0.0401667311, 0.0200833656]),
    (2, [1.0, 1.9111970674, -0.9149758348], [0.9565432255, -1.9130864510, 0.9565432255]),
    (3, [0.01, 0.03, 0.03, 0.01], [1.0, 0.9, -0.5, 0.1]),
    (4, [0.004, 0.016, 0.024, 0.016, 0.004], [1.0, 0.7, -0.3, 0.05, -0.01]),
    (8, [0.2, 0.1, 0.05, 0.02, 0.01, 0.0, -0.01, 0.0, 0.005], [1.0, 0.3, -0.2, 0.1, -0.05, 0.02, -0.01, 0.005, -0.002]),
]


def test_iir_bank_matches_oracle_bitwise():
    """sdrx_iir_*: IIRFilter<float, Order>, order 2 (specialisation) and the generic template (swapped coefficient copy),
    ragged feeds, state carried; float results bit-identical to the oracle (pinned to the reference's template)"""
    rng = np.random.default_rng(4)
    specs = IIR_SPECS * 15                                     # 75 channels: more than one wave
    g = sa.IirBank(specs)
    os_ = [orc.Iir(*s) for s in specs]
    for sizes in ([1], [0], [5000], [33], [12000]):
        xs = [rng.standard_normal(sizes[0] + (c % 3)).astype(np.float32) * 1000 for c in range(len(specs))]
        got = g.feed(xs)
        for c in range(len(specs)):
            want = os_[c].run(xs[c])
            assert got[c].size == want.size and np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), (c, sizes)
