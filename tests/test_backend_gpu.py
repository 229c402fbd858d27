"""GPU parity of the float back-end (NCO -> Interpolator -> fftfilt/g_fft -> discriminator) against the
strict-IEEE oracle.  Bar (BASELINE north_star): <= 1 ulp float32; the arithmetic-only stages are
expected -- and asserted -- to be bit-identical; only atan2f (UDPSrc discriminator) uses a looser bound."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc
from tests import synth

pytestmark = pytest.mark.gpu


def ulp_diff(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai); bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return int(np.abs(ai - bi).max())


CFGS = [
    # in_rate, nco, out_rate, cutoff, tpp, filt, f1, f2, discri, fm   -- NFM front (nfmdemod.cpp:453-476)
    dict(in_rate=60000, nco_freq=-4567, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5, filt_mode=0, f1=0.0, f2=0.0, discri=1, fm_scaling=48000 / 2000),
    # SSB front (ssbdemod.cpp): tpp 2.0, runSSB usb 300..3000 Hz
    dict(in_rate=120000, nco_freq=20000, out_rate=48000, interp_cutoff=5000.0, taps_per_phase=2.0, filt_mode=2, f1=300 / 48000, f2=3000 / 48000, discri=0, fm_scaling=1.0),
    # SURVEY cfg 4: SSB filter + NFM discriminator
    dict(in_rate=60000, nco_freq=1234, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5, filt_mode=2, f1=300 / 48000, f2=5000 / 48000, discri=1, fm_scaling=24.0),
    # lsb, and a plain complex filter; resampler only; equal rates
    dict(in_rate=60000, nco_freq=-300, out_rate=48000, interp_cutoff=3000.0, taps_per_phase=2.0, filt_mode=3, f1=300 / 48000, f2=3000 / 48000, discri=0, fm_scaling=1.0),
    dict(in_rate=96000, nco_freq=0, out_rate=48000, interp_cutoff=8000.0, taps_per_phase=4.5, filt_mode=1, f1=0.0, f2=0.1, discri=0, fm_scaling=1.0),
    dict(in_rate=60000, nco_freq=777, out_rate=48000, interp_cutoff=6000.0, taps_per_phase=4.5, filt_mode=0, f1=0.0, f2=0.0, discri=0, fm_scaling=1.0),
    dict(in_rate=48000, nco_freq=-12000, out_rate=48000, interp_cutoff=10000.0, taps_per_phase=2.0, filt_mode=0, f1=0.0, f2=0.0, discri=1, fm_scaling=5.0),
    # DSB mode of the SSB demod: fftfilt(2*bw/rate, 2048).runDSB (ssbdemod.cpp:92,167)
    dict(in_rate=60000, nco_freq=2500, out_rate=48000, interp_cutoff=6000.0, taps_per_phase=2.0, filt_mode=4, f1=0.0, f2=2 * 3000 / 48000, discri=0, fm_scaling=1.0),
    # ATV demod's vestigial-sideband filter: fftfilt(fin, 2048) + create_asym_filter(fopp, fin), runAsym usb / lsb (atvdemod.cpp:262,647)
    dict(in_rate=60000, nco_freq=-1500, out_rate=48000, interp_cutoff=9000.0, taps_per_phase=2.0, filt_mode=5, f1=0.04, f2=0.35, discri=0, fm_scaling=1.0),
    dict(in_rate=60000, nco_freq=800, out_rate=48000, interp_cutoff=9000.0, taps_per_phase=2.0, filt_mode=6, f1=0.15, f2=0.08, discri=0, fm_scaling=1.0),
    # 62500 / 48000 is not a dyadic step: `distance += step` rounds on most emissions (the schedule recurrence must follow the float bits)
    dict(in_rate=62500, nco_freq=-9100, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5, filt_mode=0, f1=0.0, f2=0.0, discri=1, fm_scaling=24.0),
    dict(in_rate=250000, nco_freq=31000, out_rate=44100, interp_cutoff=9000.0, taps_per_phase=4.5, filt_mode=2, f1=300 / 44100, f2=3000 / 44100, discri=0, fm_scaling=1.0),
]


def mk(cfg):
    return sa.BackendCfg(**cfg), orc.Backend(cfg["in_rate"], cfg["nco_freq"], cfg["out_rate"], cfg["interp_cutoff"], cfg["taps_per_phase"],
                                             cfg["filt_mode"], cfg["f1"], cfg["f2"], cfg["discri"], cfg["fm_scaling"])


def test_design_products_match_oracle():
    pairs = [mk(c) for c in CFGS]
    bank = sa.BackendBank([p[0] for p in pairs])
    for c, (_, o) in enumerate(pairs):
        nt, taps, filt, inc = bank.design(c)
        ont, otaps = o.taps()
        assert nt == ont and np.array_equal(taps.view(np.uint32), otaps.view(np.uint32)), c
        assert inc == orc.lib().sdro_nco_inc(float(CFGS[c]["nco_freq"]), float(CFGS[c]["in_rate"]))
        if CFGS[c]["filt_mode"]:
            assert ulp_diff(filt[: o.filter().size], o.filter()) == 0, c          # forward g_fft on the GPU + host normalisation


def test_streaming_feeds_match_oracle():
    pairs = [mk(c) for c in CFGS]
    bank = sa.BackendBank([p[0] for p in pairs])
    n_total = [40000, 90000, 40000, 30000, 60000, 20000, 25000, 50000, 45000, 52000, 70001, 131072]
    xs = [synth.mix(n, 700 + i, 12000, 6000, 1 + i % 3) for i, n in enumerate(n_total)]
    # ragged feeds, different per channel, including empty and 1-sample ones
    cut_frac = [0.0, 0.00005, 0.013, 0.013, 0.41, 0.4101, 0.77, 1.0]
    worst = 0
    for a, b in zip(cut_frac[:-1], cut_frac[1:]):
        segs = [x[2 * int(a * n): 2 * int(b * n)] for x, n in zip(xs, n_total)]
        bank.feed(segs)
        for c, (_, o) in enumerate(pairs):
            want = o.feed(segs[c])
            got = bank.read(c)
            assert got.size == want.size, (c, a, b, got.size, want.size)
            d = ulp_diff(got, want)
            worst = max(worst, d)
            assert d == 0, (c, a, b, d)
    assert worst == 0


def test_many_channels_of_one_design_share_a_tap_table():
    """40 channels with the same Interpolator design (one tap table, staged once per FIR workgroup), different NCO
    frequencies and data, plus two odd ones so that one tile of 16 schedule columns straddles two designs."""
    base = dict(in_rate=60000, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5, filt_mode=0, f1=0.0, f2=0.0, discri=0, fm_scaling=1.0)
    cfgs = [dict(base, nco_freq=-7000 + 350 * i) for i in range(40)]
    cfgs.insert(5, dict(base, in_rate=120000, nco_freq=100))
    cfgs.insert(23, dict(base, interp_cutoff=4000.0, nco_freq=-100))
    pairs = [mk(c) for c in cfgs]
    bank = sa.BackendBank([p[0] for p in pairs])
    n_total = [9000 + 173 * i for i in range(len(cfgs))]
    xs = [synth.mix(n, 40 + i, 9000, 4000, 1 + i % 3) for i, n in enumerate(n_total)]
    for a, b in ((0.0, 0.37), (0.37, 1.0)):
        segs = [x[2 * int(a * n): 2 * int(b * n)] for x, n in zip(xs, n_total)]
        bank.feed(segs)
        for c, (_, o) in enumerate(pairs):
            want = o.feed(segs[c])
            got = bank.read(c)
            assert got.size == want.size, (c, got.size, want.size)
            assert ulp_diff(got, want) == 0, c


def test_udpsrc_atan2_discriminator_within_tolerance():
    cfg = dict(in_rate=60000, nco_freq=100, out_rate=48000, interp_cutoff=6000.0, taps_per_phase=4.5, filt_mode=0, f1=0.0, f2=0.0, discri=2, fm_scaling=10.0)
    g, o = mk(cfg)
    bank = sa.BackendBank([g])
    x = synth.mix(30000, 5, 9000, 9000, 1)
    bank.feed([x])
    got, want = bank.read(0), o.feed(x)
    assert got.size == want.size
    # std::arg -> atan2f comes from two math libraries (glibc on the host, the oracle's; ROCm's on the device).  The
    # arguments are bit-identical on both sides (every stage in front is 0 ulp), the device rounds a double atan2 once
    # (<= 0.5 ulp + 2^-29), glibc documents <= 2 ulp for atan2f on x86-64 (libm-test-ulps), and the common scaling
    # `(double) a / pi * fm_scaling -> float` adds one rounding: <= 3 ulp apart, at every magnitude (incl. angles near 0)
    d = ulp_diff(got, want)
    print("atan2 discriminator: max ulp distance", d, "max abs", float(np.max(np.abs(got - want))))
    assert d <= 3


def test_zero_and_constant_input():
    cfg = CFGS[2]
    g, o = mk(cfg)
    bank = sa.BackendBank([g])
    z = np.zeros(2 * 5000, np.int16)
    bank.feed([z])
    assert ulp_diff(bank.read(0), o.feed(z)) == 0
    k = np.empty(2 * 5000, np.int16); k[0::2] = 32767; k[1::2] = -32768
    bank.feed([k])
    assert ulp_diff(bank.read(0), o.feed(k)) == 0


@pytest.mark.parametrize("handover", ["host_sync", "device_ordered"])
def test_cfg4_pipeline_bank_to_backend_on_device(handover):
    """SURVEY cfg 4 shape at small scale: DownChannelizer bank -> (device hand-over) -> NCO -> Interpolator ->
    fftfilt SSB -> NFM discriminator, 16 channels, against oracle chain + oracle back-end per channel.
    device_ordered: sdrx_backend_feed_bank, no host synchronisation between the bank and the back-end, and the bank's
    queue is dropped and refilled by the next feed straight away (the stream ordering has to protect the samples)."""
    fs = 61_440_000
    n_ch = 16
    k = np.arange(n_ch)
    fcs = [int(v) for v in (-25_000_000 + k * (50_000_000 / 15) + 137 * k)]
    bank = sa.ChannelizerBank(fs, [48000] * n_ch, fcs)
    cfgs, oras, chains = [], [], []
    for c in range(n_ch):
        modes, out_rate, ofs = bank.info(c)
        cfg = dict(in_rate=out_rate, nco_freq=-ofs, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5,
                   filt_mode=2, f1=300 / 48000, f2=5000 / 48000, discri=1, fm_scaling=48000 / 2000)
        g, o = mk(cfg)
        cfgs.append(g); oras.append(o); chains.append(orc.Chain(modes))
    be = sa.BackendBank(cfgs)
    x = synth.mix(3_000_000, 77, 3000, 1500, 1)
    cuts = ((0, 1_000_001), (1_000_001, 2_100_000), (2_100_000, 3_000_000))
    segs = [x[2 * a: 2 * b] for a, b in cuts]

    def check(seg):
        for c in range(n_ch):
            want = oras[c].feed(chains[c].feed(seg))
            got = be.read(c)
            assert got.size == want.size, (c, got.size, want.size)
            assert ulp_diff(got, want) == 0, c

    if handover == "host_sync":
        for seg in segs:
            bank.feed(seg)
            ptrs, cnts = zip(*[bank.last_dev(c) for c in range(n_ch)])
            bank.sync()
            be.feed_dev(ptrs, cnts)
            check(seg)
    else:
        for i, seg in enumerate(segs):
            bank.feed(seg)                   # from the second round on this overwrites the queues the back-end was handed
            if i:
                check(segs[i - 1])           # ... before the back-end's results for the previous feed are looked at
            be.feed_bank(bank)
            for c in range(n_ch):
                bank.skip(c)
        check(segs[-1])


def test_audio_fir_bank_lowpass_bandpass():
    """a14: Lowpass<Real>/Bandpass<Real> as NFM uses them (301 taps @ 48 kS/s: 250 Hz CTCSS low pass, 300..3000 Hz audio
    band pass) + odd corner sizes; streaming with ragged feeds; taps and outputs bit-identical to the oracle"""
    specs = [(0, 301, 48000.0, 250.0, 0.0), (1, 301, 48000.0, 300.0, 3000.0), (0, 64, 48000.0, 3000.0, 0.0), (1, 21, 8000.0, 300.0, 2500.0)]
    bank = sa.FirBank([sa.FirCfg(k, n, r, a, b) for k, n, r, a, b in specs])
    oras = [orc.Fir(k, n, r, a, b) for k, n, r, a, b in specs]
    for c, o in enumerate(oras):
        assert np.array_equal(bank.taps(c).view(np.uint32), o.taps().view(np.uint32)), c
    rng = np.random.default_rng(11)
    for sizes in ([1, 0, 5, 3], [1000, 17, 0, 400], [4801, 4801, 4801, 4801], [0, 0, 0, 0]):
        xs = [rng.standard_normal(n).astype(np.float32) for n in sizes]
        got = bank.feed(xs)
        for c, o in enumerate(oras):
            want = o.run(xs[c])
            assert got[c].size == want.size and np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), (c, sizes)
