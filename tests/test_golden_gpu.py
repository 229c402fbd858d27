"""GPU: the HIP engine straight against the committed golden vectors (made from the compiled
reference by tests/golden/make_golden.py) -- no oracle in the loop."""
import json
import os

import numpy as np
import pytest

import sdrangel_amd as sa
from tests import synth
from tests.test_oracle_golden import G, cfg3_fc, chan_inputs, decim_inputs

pytestmark = pytest.mark.gpu


def test_decimators_vs_reference_hashes_on_gpu():
    meta, cases = decim_inputs()
    cuts = meta["cuts_int16"]
    full = np.load(os.path.join(G, "decim_golden.npz"))
    for key, want in meta["hashes"].items():
        name, b, l, f = key.split("_")
        bits, log2, fc = int(b[4:]), int(l[3:]), int(f[2:])
        d = sa.Decimators(log2, fc, bits)
        y = np.concatenate([d.decimate(cases[name][a:b2]) for a, b2 in zip(cuts[:-1], cuts[1:])])
        assert y.size // 2 == want["n"], key
        assert f"{synth.fnv1a64(y):016x}" == want["fnv1a64"], key
        if key in full.files:
            assert np.array_equal(y, full[key]), key


def test_decimators_u8_vs_reference_hashes_on_gpu():
    from tests.test_oracle_golden import decimu_input
    meta = json.load(open(os.path.join(G, "decim_golden.json")))
    cuts, N = meta["cuts_int16"], meta["recipe"]["N"]
    xu = decimu_input(N)
    for key, want in json.load(open(os.path.join(G, "decimu_golden.json")))["hashes"].items():
        _u, l, f = key.split("_")
        d = sa.DecimatorsU(int(l[3:]), int(f[2:]), 127)
        y = np.concatenate([d.decimate(xu[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
        assert y.size // 2 == want["n"] and f"{synth.fnv1a64(y):016x}" == want["fnv1a64"], key


def test_channelizer_bank_vs_reference_feed_on_gpu():
    g = np.load(os.path.join(G, "chan_golden.npz"))
    cuts = [int(v) for v in g["cuts"]]
    xs = chan_inputs()
    plans = {(p["in"], p["req"], p["fc"]): p for p in json.load(open(os.path.join(G, "chan_plans.json")))}
    chans = [0, 5, 13, 16, 31]
    for name, x in xs.items():
        bank = sa.ChannelizerBank(61440000, [48000] * len(chans), [cfg3_fc(c) for c in chans])
        for i, c in enumerate(chans):
            m, r, o = bank.info(i)
            p = plans[(61440000, 48000, cfg3_fc(c))]
            assert list(m) == p["modes"] and r == p["out_rate"] and o == p["ofs"]
        for a, b in zip(cuts[:-1], cuts[1:]):
            bank.feed(x[2 * a: 2 * b])
        for i, c in enumerate(chans):
            assert np.array_equal(bank.read(i), g[f"{name}_cfg3ch{c}"]), (name, c)
