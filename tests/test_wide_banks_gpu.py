"""GPU parity at the WIDE bank shapes of BASELINE configs 4 and 5 -- the shapes the tree planner treats differently
(several passes, the 64 KB LDS plan for >= 192 channels, four schedule waves in the back-end), every channel checked:

  cfg 5 (per GPU): 128-channel DownChannelizer bank with the cfg-3 spacing, ragged feeds, ALL channels vs orc.Chain
                   (reference: DownChannelizer::feed, sdrbase/dsp/downchannelizer.cpp:50-91)
  cfg 4          : 256 channels, ChannelizerBank -> sdrx_backend_feed_bank -> BackendBank (NCO -> Interpolator ->
                   fftfilt SSB -> NFM discriminator, plugins/channelrx/demodnfm/nfmdemod.cpp:150-163), every channel
                   0 ulp vs orc.Backend(orc.Chain(...)), closed-form (dyadic) and serial resampler schedule.
The oracle chains run on a thread pool (ctypes releases the GIL); sizes keep the CPU side at a few seconds.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc
from tests import synth
from tests.test_backend_gpu import mk, ulp_diff

pytestmark = pytest.mark.gpu

FS = 61_440_000
POOL = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))


def spaced(n_ch, lo, span):
    k = np.arange(n_ch)
    return [int(v) for v in (lo + k * (span / (n_ch - 1)) + 137 * k).astype(np.int64)]


def test_cfg5_bank128_all_channels_ragged_feeds():
    n_ch = 128
    fcs = spaced(n_ch, -15_000_000, 30_000_000)            # bench.py --workload chan128 uses exactly these centres
    n = (1 << 20) + 4099
    x = orc.synth_iq(n, seed=515, amp=2047, tone=(0.0371, 500))
    x[2 * 700_000: 2 * 700_064] = -32768                    # a burst that wraps in the int16 stage stores
    bank = sa.ChannelizerBank(FS, [48000] * n_ch, fcs)
    plans = [orc.chan_plan(FS, 48000, f) for f in fcs]
    depths = set()
    for c, (modes, out_rate, ofs) in enumerate(plans):
        m, r, o = bank.info(c)
        assert np.array_equal(m, modes) and (r, o) == (out_rate, ofs), c
        depths.add(len(modes))
    assert depths == {9, 10}                                # both chain lengths occur with this spacing
    cuts = [0, 3, 4096, 250_001, 250_001, 777_777, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        bank.feed(x[2 * a: 2 * b])
    want = list(POOL.map(lambda p: orc.Chain(p[0]).feed(x), plans))
    bad = []
    for c in range(n_ch):
        got = bank.read(c)
        if got.size != want[c].size or not np.array_equal(got, want[c]):
            bad.append(c)
    assert not bad, bad


@pytest.mark.parametrize("schedule", ["dyadic", "serial"])
def test_cfg4_full_width_256_channels_bank_to_backend(schedule, monkeypatch):
    n_ch = 256
    k = np.arange(n_ch)
    fcs = [int(v) for v in (-25_000_000 + k * (50_000_000 / 255))]            # bench.py --workload cfg4
    if schedule == "serial":
        monkeypatch.setenv("SDRX_BE_SERIAL_SCHEDULE", "1")                    # read by sdrx_backend_create
    else:
        monkeypatch.delenv("SDRX_BE_SERIAL_SCHEDULE", raising=False)
    bank = sa.ChannelizerBank(FS, [48000] * n_ch, fcs)
    cfgs, specs = [], []
    for c in range(n_ch):
        modes, out_rate, ofs = bank.info(c)
        om, orr, oo = orc.chan_plan(FS, 48000, fcs[c])
        assert np.array_equal(modes, om) and (out_rate, ofs) == (orr, oo), c
        cfg = dict(in_rate=out_rate, nco_freq=-ofs, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5,
                   filt_mode=2, f1=300 / 48000, f2=5000 / 48000, discri=1, fm_scaling=48000 / 2000)
        cfgs.append(mk(cfg)[0]); specs.append((modes, cfg))
    be = sa.BackendBank(cfgs)
    n = 1_400_000                                            # ~1367 channel samples at n = 10 -> two fftfilt blocks
    x = synth.mix(n, 4040, 3000, 1500, 1)
    cuts = ((0, 500_001), (500_001, 500_002), (500_002, n))
    segs = [x[2 * a: 2 * b] for a, b in cuts]
    got = [[] for _ in range(n_ch)]
    for seg in segs:
        bank.feed(seg)
        be.feed_bank(bank)                                   # device-ordered hand-over, no host sync in between
        for c in range(n_ch):
            bank.skip(c)
        for c in range(n_ch):
            got[c].append(be.read(c, 1 << 16))               # sdrx_backend_read: the outputs of the LAST feed

    def oracle(spec):
        modes, cfg = spec
        chain, o = orc.Chain(modes), mk(cfg)[1]
        return np.concatenate([o.feed(chain.feed(seg)) for seg in segs])

    want = list(POOL.map(oracle, specs))
    bad, n_out = [], 0
    for c in range(n_ch):
        g = np.concatenate(got[c])
        n_out += g.size
        if g.size != want[c].size or (g.size and ulp_diff(g, want[c]) != 0):
            bad.append(c)
    assert not bad, bad
    assert n_out >= n_ch * 512                               # every channel produced at least one fftfilt block
