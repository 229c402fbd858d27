"""CPU: the oracle (oracle/sdro.c) against the golden vectors that tests/golden/make_golden.py
produced from the compiled reference.  Bit-exact."""
import json
import os

import numpy as np
import pytest

from tests import oracle_py as orc
from tests import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def decim_inputs():
    meta = json.load(open(os.path.join(G, "decim_golden.json")))
    N = meta["recipe"]["N"]
    cases = {"b12": synth.mix(N, 11, 2047, 900, 1), "b8": synth.mix(N, 12, 127, 60, 1), "b16": synth.mix(N, 13, 32767, 0)}
    w = np.empty(2 * N, np.int16); w[0::2] = -32768; w[1::2] = np.where(np.arange(N) % 3 == 0, 32767, -32768)
    w[:4000] = synth.noise_iq(2000, 14, 32767)
    cases["wrap"] = w
    return meta, cases


def chan_inputs():
    NC = 1 << 16
    xs = {"n12": synth.mix(NC, 21, 2047, 1200, 3), "full": synth.noise_iq(NC, 22, 32767)}
    xs["full"][::11] = -32768
    return xs


def cfg3_fc(k):
    return int(-15_000_000 + k * (30_000_000 / 31) + 137 * k)


def test_synth_is_stable():
    # the fixtures depend on these exact bytes
    x = synth.mix(1000, 3, 2047, 600)
    assert list(x[:8]) == [1989, -554, 1738, -647, -298, 716, -1195, 460]
    assert synth.fnv1a64(synth.mix(100000, 3, 2047, 600)[:1000]) == 0x6c114860fde043ce


def test_decimators_all_variants_vs_reference_hashes():
    meta, cases = decim_inputs()
    cuts = meta["cuts_int16"]
    full = np.load(os.path.join(G, "decim_golden.npz"))
    n_checked = 0
    for key, want in meta["hashes"].items():
        name, b, l, f = key.split("_")
        bits, log2, fc = int(b[4:]), int(l[3:]), int(f[2:])
        o = orc.Decim(log2, fc, bits)
        y = np.concatenate([o.process(cases[name][a:b2]) for a, b2 in zip(cuts[:-1], cuts[1:])])
        assert y.size // 2 == want["n"], key
        assert f"{synth.fnv1a64(y):016x}" == want["fnv1a64"], key
        if key in full.files:
            assert np.array_equal(y, full[key]), key
        n_checked += 1
    assert n_checked == 4 * 3 * 7 * 3


def decimu_input(N):
    xu = (synth.lcg_u32(2 * N, 15) & 0xff).astype(np.uint8)
    xu[:2000] = 0; xu[2000:4000] = 255
    return xu


def test_decimators_u8_vs_reference_hashes():
    meta = json.load(open(os.path.join(G, "decim_golden.json")))
    cuts, N = meta["cuts_int16"], meta["recipe"]["N"]
    xu = decimu_input(N)
    for key, want in json.load(open(os.path.join(G, "decimu_golden.json")))["hashes"].items():
        _u, l, f = key.split("_")
        o = orc.DecimU(int(l[3:]), int(f[2:]), 127)
        y = np.concatenate([o.process(xu[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
        assert y.size // 2 == want["n"] and f"{synth.fnv1a64(y):016x}" == want["fnv1a64"], key


def test_group_strides_match_reference_loops():
    # `pos +=` of decimateK_* (decimators.h): the tail-drop granularity
    want = {(0, 2): 2, (1, 0): 8, (1, 2): 8, (2, 1): 16, (2, 2): 16, (3, 0): 32, (3, 2): 16,
            (4, 1): 64, (4, 2): 32, (5, 0): 128, (5, 2): 64, (6, 0): 256, (6, 1): 256, (6, 2): 128}
    for (l, f), g in want.items():
        assert orc.lib().sdro_decim_group_int16(l, f) == g


def test_channelizer_plans_vs_reference():
    plans = json.load(open(os.path.join(G, "chan_plans.json")))
    assert len(plans) > 400
    for p in plans:
        modes, out_rate, ofs = orc.chan_plan(p["in"], p["req"], p["fc"])
        assert list(modes) == p["modes"] and out_rate == p["out_rate"] and ofs == p["ofs"], p
    # the survey's hand-checked example
    m, r, o = orc.chan_plan(61440000, 48000, 1234567)
    assert list(m) == [2, 1, 1, 1, 1, 2, 1, 2, 1, 1] and (r, o) == (60000, 4567)


def test_channelizer_feed_vs_reference():
    g = np.load(os.path.join(G, "chan_golden.npz"))
    cuts = [int(v) for v in g["cuts"]]
    xs = chan_inputs()
    n = 0
    for key in g.files:
        if key == "cuts":
            continue
        parts = key.split("_")                     # "<input>_cfg3ch<k>"  or  "<input>_chain_<modes>"
        x = xs[parts[0]]
        if parts[1].startswith("cfg3ch"):
            modes, _, _ = orc.chan_plan(61440000, 48000, cfg3_fc(int(parts[1][6:])))
            cc = cuts
        else:
            modes = [int(c) for c in parts[2]]
            cc = cuts if len(modes) > 3 else cuts[:4]
        ch = orc.Chain(modes)
        y = np.concatenate([ch.feed(x[2 * a: 2 * b]) for a, b in zip(cc[:-1], cc[1:])])
        assert np.array_equal(y, g[key]), key
        n += 1
    assert n == 2 * (5 + 7)
