#!/usr/bin/env python3
"""End-to-end walk through the GPU RX path with the pieces of libsdrx.so, shaped like an SDRangel FileSource device set:

    .sdriq file (FileRecord header + int16 I/Q)  ->  SampleSinkFifo  ->  engine drain loop
        ->  DC offset correction (work()'s iqCorrections)  ->  DownChannelizer bank (N channels)
        ->  per channel NCO -> Interpolator -> phaseDiscriminatorDelta (the NFM demod front)  ->  float audio-rate streams
        ->  (second leg) the same front with complex output -> NFM audio tail: discriminator, power squelch, gate delay
            line, 301-tap Bandpass, volume -> qint16 PCM, what NFMDemod::feed pushes into its AudioFifo

    python examples/filesource_to_nfm.py [out_dir]          # writes a synthetic recording, replays it, saves the results

Everything numeric runs on the MI355X through the C ABI (include/sdrx.h); this script is host glue only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdrangel_amd as sa  # noqa: E402


def synth_recording(path, fs, fcs, seconds=0.25, dev_hz=2500.0):
    """a few FM carriers (1 kHz .. tones) + noise + a DC offset, as a .sdriq file"""
    n = int(fs * seconds)
    t = np.arange(n) / fs
    x = np.zeros(n, np.complex128)
    for i, fc in enumerate(fcs):
        tone = 700.0 + 300.0 * i
        phase = 2 * np.pi * fc * t + (dev_hz / tone) * np.sin(2 * np.pi * tone * t)
        x += 500.0 * np.exp(1j * phase)
    rng = np.random.default_rng(1)
    x += rng.normal(0, 40, n) + 1j * rng.normal(0, 40, n) + (90 - 60j)          # noise + DC offset
    iq = np.empty(2 * n, np.int16)
    iq[0::2] = np.clip(np.round(x.real), -2048, 2047); iq[1::2] = np.clip(np.round(x.imag), -2048, 2047)
    with open(path, "wb") as f:
        f.write(sa.sdriq_header_bytes(fs, 435_000_000, 1_700_000_000, 16))
        f.write(iq.tobytes())
    return n


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    fs = 2_400_000
    fcs = [-600_000, -150_000, 75_000, 480_000]
    rec = os.path.join(out_dir, "synthetic.sdriq")
    n = synth_recording(rec, fs, fcs)

    data = open(rec, "rb").read()
    hdr, payload = sa.sdriq_parse(data)                     # FileRecord::readHeader (filerecord.cpp:140-148) + the samples behind it
    assert hdr.sample_rate == fs and hdr.sample_size == 16

    fifo = sa.SampleSinkFifo(fs // 4)                       # the FileSource device FIFO (filesourceinput.cpp:143: rate * 4 ... here smaller)
    dc = sa.DcCorrection()
    bank = sa.ChannelizerBank(fs, [48000] * len(fcs), fcs)
    cfgs = []
    for c in range(len(fcs)):
        _modes, out_rate, ofs = bank.info(c)
        cfgs.append(sa.BackendCfg(in_rate=out_rate, nco_freq=-ofs, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5,
                                  filt_mode=0, f1=0.0, f2=0.0, discri=1, fm_scaling=48000 / 2500))
    front = sa.BackendBank(cfgs)
    # second leg: complex samples out of the front, then the audio-rate tail of NFMDemod::feed (nfmdemod.cpp:160-300)
    cfgs_c = []
    for c in range(len(fcs)):
        _modes, out_rate, ofs = bank.info(c)
        cfgs_c.append(sa.BackendCfg(in_rate=out_rate, nco_freq=-ofs, out_rate=48000, interp_cutoff=12500 / 2.2, taps_per_phase=4.5,
                                    filt_mode=0, f1=0.0, f2=0.0, discri=0, fm_scaling=1.0))
    front_c = sa.BackendBank(cfgs_c)
    tail = sa.AudioTail([sa.AudioTailCfg(kind=0, audio_rate=48000, volume=2.0, fm_scaling=48000 / (2 * 2500.0), squelch_level=1e-6,
                                         squelch_gate=480, af_bandwidth=3000.0) for _ in fcs])

    audio = [[] for _ in fcs]
    pcm = [[] for _ in fcs]
    block = 2 * 50_000                                      # int16 per "FileSourceThread tick"
    for pos in range(0, payload.size, block):
        fifo.write(payload[pos: pos + block])
        while fifo.fill:                                    # DSPDeviceSourceEngine::work: drain, correct, feed the sinks
            span = fifo.read(fifo.fill)
            span = dc.process(span)
            bank.feed(span)
            chans = [bank.read(c) for c in range(len(fcs))]
            front.feed(chans)
            front_c.feed(chans)
            cplx = [front_c.read(c) for c in range(len(fcs))]
            for c, y16 in enumerate(tail.feed(cplx)):
                pcm[c].append(y16.copy())
            for c in range(len(fcs)):
                audio[c].append(front.read(c))
    for c, fc in enumerate(fcs):
        y = np.concatenate(audio[c])
        np.save(os.path.join(out_dir, f"nfm_front_ch{c}.npy"), y)
        # the discriminator output of an FM carrier is its modulating tone: report the dominant frequency
        spec = np.abs(np.fft.rfft(y[2000:] - y[2000:].mean()))
        f_peak = np.argmax(spec) * 48000.0 / (2 * (spec.size - 1))
        print(f"channel {c}: fc {fc:+8d} Hz  {y.size} samples at 48 kS/s, dominant tone {f_peak:7.1f} Hz (sent {700 + 300 * c} Hz)")
    for c in range(len(fcs)):
        z = np.concatenate(pcm[c]).astype(np.float64)
        np.save(os.path.join(out_dir, f"nfm_pcm_ch{c}.npy"), z.astype(np.int16))
        zs = z[4000:]
        spec = np.abs(np.fft.rfft(zs - zs.mean()))
        f_peak = np.argmax(spec) * 48000.0 / (2 * (spec.size - 1))
        print(f"channel {c}: qint16 audio {z.size} samples, peak |s| {int(np.abs(z).max())}, dominant tone {f_peak:7.1f} Hz")
    print(f"{n} input samples replayed from {rec}")
    main.pcm = pcm
    return audio


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "examples_out")
