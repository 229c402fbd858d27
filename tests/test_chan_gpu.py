"""GPU parity: sdrx_chan_bank_* (HIP tree kernel through the C ABI) vs the CPU oracle's
DownChannelizer restatement, bit-exact per channel, incl. int16 wrap, ragged feeds, reconfigure."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu

FS = 61_440_000


def cfg3_channels(n_ch=32):
    # SURVEY.md §8(d) cfg 3: fc_k = -15e6 + k*(30e6/31) + 137k, req 48 kS/s
    k = np.arange(n_ch)
    fc = (-15_000_000 + k * (30_000_000 / 31) + 137 * k).astype(np.int64)
    return [48000] * n_ch, [int(v) for v in fc]


def oracle_bank(in_rate, rates, fcs):
    chains = []
    for r, f in zip(rates, fcs):
        modes, out_rate, ofs = orc.chan_plan(in_rate, r, f)
        chains.append((orc.Chain(modes), modes, out_rate, ofs))
    return chains


def test_plan_matches_oracle():
    rng = np.random.default_rng(5)
    for _ in range(3000):
        ir = int(rng.choice([61440000, 10000000, 2400000, 48000, 96000, 1000001, 250000]))
        rr = int(rng.choice([48000, 8000, 12500, 200000, 64000, 100]))
        fc = int(rng.integers(-ir // 2 - 500, ir // 2 + 500))
        m1, r1, o1 = sa.chan_plan(ir, rr, fc)
        m2, r2, o2 = orc.chan_plan(ir, rr, fc)
        assert np.array_equal(m1, m2) and r1 == r2 and o1 == o2, (ir, rr, fc)


@pytest.mark.parametrize("amp,tone", [(2047, (0.0123, 600)), (32767, None), (20000, (0.2501, 12000))])
def test_bank32_matches_oracle_ragged_feeds(amp, tone):
    rates, fcs = cfg3_channels(32)
    n = 1 << 20
    x = orc.synth_iq(n, seed=21, amp=amp, tone=tone)
    if amp == 32767:
        x[::7] = -32768                                  # wrap-negation corner on both arms
    bank = sa.ChannelizerBank(FS, rates, fcs)
    ref = oracle_bank(FS, rates, fcs)
    for c, (_, modes, out_rate, ofs) in enumerate(ref):
        m, r, o = bank.info(c)
        assert np.array_equal(m, modes) and r == out_rate and o == ofs
    cuts = [0, 5, 4096, 4096 + 3, 70001, 70001, 300000, 300001, 777777, n]
    want = [[] for _ in ref]
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = x[2 * a: 2 * b]
        bank.feed(seg)
        for c, (chain, *_r) in enumerate(ref):
            want[c].append(chain.feed(seg))
    for c in range(len(ref)):
        w = np.concatenate(want[c])
        assert bank.available(c) == w.size // 2, c
        got = bank.read(c)
        assert np.array_equal(got, w), (c, int((got != w).sum()), ref[c][1])
        assert bank.available(c) == 0


def test_mixed_depths_shared_prefixes_and_passthrough():
    # different requested rates -> chains of different length sharing prefixes; one 0-stage channel
    in_rate = 2_400_000
    rates = [48000, 48000, 200000, 12500, 1_200_000, 2_400_000, 300000, 8000, 48000]
    fcs = [0, 100000, -400000, 512345, 0, 0, 600000, -1_000_000, 100000]
    n = 600_000
    x = orc.synth_iq(n, seed=9, amp=30000, tone=(0.041, 3000))
    bank = sa.ChannelizerBank(in_rate, rates, fcs)
    ref = oracle_bank(in_rate, rates, fcs)
    assert any(len(r[1]) == 0 for r in ref)              # the pass-through case is present
    for a, b in [(0, 123457), (123457, 123458), (123458, n)]:
        bank.feed(x[2 * a: 2 * b])
    for c, (chain, modes, *_r) in enumerate(ref):
        w = chain.feed(x)
        got = bank.read(c)
        assert np.array_equal(got, w), (c, modes)


def test_partial_reads_and_reconfigure():
    rates, fcs = cfg3_channels(8)
    x = orc.synth_iq(400_000, seed=4, amp=2047, tone=(0.11, 900))
    bank = sa.ChannelizerBank(FS, rates, fcs)
    ref = oracle_bank(FS, rates, fcs)
    bank.feed(x[: 2 * 250_000])
    w0 = ref[0][0].feed(x[: 2 * 250_000])
    part = bank.read(0, 10)
    assert np.array_equal(part, w0[:20])
    # reconfigure channel 3 mid-stream: new chain starts from zero history at this very sample
    bank.reconfigure(3, 48000, 2_000_000)
    m, r, o = bank.info(3)
    om, orr, oo = orc.chan_plan(FS, 48000, 2_000_000)
    assert np.array_equal(m, om) and (r, o) == (orr, oo)
    pre3 = bank.read(3)                                   # what the old chain had produced
    assert np.array_equal(pre3, ref[3][0].feed(x[: 2 * 250_000]))
    new3 = orc.Chain(om)
    bank.feed(x[2 * 250_000:])
    assert np.array_equal(bank.read(3), new3.feed(x[2 * 250_000:]))
    rest0 = np.concatenate([w0[20:], ref[0][0].feed(x[2 * 250_000:])])
    assert np.array_equal(bank.read(0), rest0)
    for c in (1, 2, 4, 5, 6, 7):                          # the others never noticed
        assert np.array_equal(bank.read(c), ref[c][0].feed(x))
    bank.reset()
    bank.feed(x[: 2 * 100_000])
    assert np.array_equal(bank.read(5), orc.Chain(ref[5][1]).feed(x[: 2 * 100_000]))


def test_bank256_cfg4_shape_spot_check():
    """256 channels (SURVEY cfg 4 spacing): planner splits the wide trie over several passes; check a spread of channels"""
    n_ch = 256
    k = np.arange(n_ch)
    fcs = [int(v) for v in (-25_000_000 + k * (50_000_000 / 255))]
    bank = sa.ChannelizerBank(FS, [48000] * n_ch, fcs)
    x = orc.synth_iq(1 << 20, seed=33, amp=2047, tone=(0.07, 700))
    bank.feed(x[: 2 * 400_003]); bank.feed(x[2 * 400_003:])
    for c in (0, 1, 17, 100, 127, 128, 200, 254, 255):
        modes, out_rate, ofs = orc.chan_plan(FS, 48000, fcs[c])
        m, r, o = bank.info(c)
        assert np.array_equal(m, modes) and (r, o) == (out_rate, ofs)
        assert np.array_equal(bank.read(c), orc.Chain(modes).feed(x)), c


def test_retune_fifty_times_stays_exact_and_bounded():
    """A live session retunes (DownChannelizer::configure per retune, downchannelizer.cpp:44-48): every reconfigure restarts
    that channel from zero history; the bank must not accumulate dead stage tries (launches and device memory per feed
    stay bounded) and the other channels must never notice."""
    rates, fcs = cfg3_channels(6)
    bank = sa.ChannelizerBank(FS, rates, fcs)
    ref = oracle_bank(FS, rates, fcs)
    assert bank.group_count == 1
    n_blk = 40_000
    x = orc.synth_iq(52 * n_blk, seed=1234, amp=2047, tone=(0.031, 800))
    other = [[] for _ in ref]
    for i in range(50):
        seg = x[2 * i * n_blk: 2 * (i + 1) * n_blk]
        fc_new = -20_000_000 + 777_001 * i
        bank.reconfigure(2, 48000, fc_new)
        assert bank.group_count <= 2, (i, bank.group_count)        # the original trie + channel 2's current one
        modes, out_rate, ofs = orc.chan_plan(FS, 48000, fc_new)
        m, r, o = bank.info(2)
        assert np.array_equal(m, modes) and (r, o) == (out_rate, ofs)
        bank.feed(seg)
        assert np.array_equal(bank.read(2), orc.Chain(modes).feed(seg)), i     # a FRESH chain each time
        for c in (0, 1, 3, 4, 5):
            other[c].append(bank.read(c))
    for c in (0, 1, 3, 4, 5):
        assert np.array_equal(np.concatenate(other[c]), ref[c][0].feed(x[: 2 * 50 * n_blk])), c
    # every original channel retuned once: the original trie has no live channel left and is retired
    for c in (0, 1, 3, 4, 5):
        bank.reconfigure(c, 48000, fcs[c] + 5000)
    assert bank.group_count == 6


def test_add_and_remove_channel_leave_the_others_alone():
    rates, fcs = cfg3_channels(4)
    bank = sa.ChannelizerBank(FS, rates, fcs)
    ref = oracle_bank(FS, rates, fcs)
    x = orc.synth_iq(300_000, seed=77, amp=2047, tone=(0.02, 900))
    a, b = x[: 2 * 100_001], x[2 * 100_001:]
    bank.feed(a)
    c_new = bank.add_channel(48000, 7_654_321)                      # addThreadedSink on a running device set
    assert c_new == 4 and bank.group_count == 2
    modes, out_rate, ofs = orc.chan_plan(FS, 48000, 7_654_321)
    assert (bank.info(4)[1], bank.info(4)[2]) == (out_rate, ofs)
    bank.remove_channel(1)
    bank.feed(b)
    assert np.array_equal(bank.read(4), orc.Chain(modes).feed(b))   # starts at the first sample after it was added
    assert bank.available(1) == 0
    for c in (0, 2, 3):
        assert np.array_equal(bank.read(c), ref[c][0].feed(x)), c   # history and queued output untouched
    bank.remove_channel(4)
    assert bank.group_count == 1
    bank.reconfigure(1, 48000, fcs[1])                              # a removed index can be revived
    bank.feed(a)
    assert np.array_equal(bank.read(1), orc.Chain(ref[1][1]).feed(a))


@pytest.mark.parametrize("levels,lds_kb", [(8, 80), (10, 150)])
def test_deep_pass_plans_stay_exact(levels, lds_kb, monkeypatch):
    """SDRX_CHAN_MAX_LEVELS / SDRX_CHAN_LDS_KB (the experiment of DESIGN 4.3: deeper first pass, fewer node-stream bytes):
    more than six levels per pass need several warm-up chunks and a longer stream history -- same samples"""
    monkeypatch.setenv("SDRX_CHAN_MAX_LEVELS", str(levels))
    monkeypatch.setenv("SDRX_CHAN_LDS_KB", str(lds_kb))
    rates, fcs = cfg3_channels(32)
    n = 600_000
    x = orc.synth_iq(n, seed=77, amp=32767, tone=None)
    x[::7] = -32768
    bank = sa.ChannelizerBank(FS, rates, fcs)
    ref = oracle_bank(FS, rates, fcs)
    cuts = [0, 5, 4099, 70001, 70001, 300000, 300001, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = x[2 * a: 2 * b]
        bank.feed(seg)
        for c, (chain, *_r) in enumerate(ref):
            got = bank.read(c)
            assert np.array_equal(got, chain.feed(seg)), (levels, c, a, b)
    bank.close()
