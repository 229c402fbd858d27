"""CPU: the oracle's 24-bit flavour (oracle/sdro.c sdro_decim24_* / sdro_chain24_*) against fixtures generated from the
reference's own SDR_RX_SAMPLE_24BIT build (tests/golden/make_golden24.py -> wide24_golden.{json,npz})."""
import json
import os

import numpy as np
import pytest

from tests import oracle_py as orc
from tests import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HASH = json.load(open(os.path.join(G, "wide24_golden.json")))["hashes"]
KEEP = np.load(os.path.join(G, "wide24_golden.npz"))


def _check(key, y):
    assert y.size // 2 == HASH[key]["n"], key
    assert f"{synth.fnv1a64(y):016x}" == HASH[key]["fnv1a64"], key
    if key in KEEP.files:
        assert np.array_equal(y, KEEP[key]), key


@pytest.mark.parametrize("bits", (8, 12, 16))
def test_decimators24_golden(bits):
    for name, x in synth.w24_dec_inputs().items():
        for log2 in range(7):
            for fc in range(3):
                o = orc.Decim24(log2, fc, bits)
                cuts = synth.W24_DEC_CUTS
                y = np.concatenate([o.process(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
                _check(f"dec_{name}_bits{bits}_log{log2}_fc{fc}", y)


def test_chains24_golden():
    for name, x in synth.w24_chan_inputs().items():
        for modes in synth.W24_CH_MODES:
            cuts = synth.W24_CH_CUTS if len(modes) > 3 else synth.W24_CH_CUTS[:4]
            o = orc.Chain24(modes)
            y = np.concatenate([o.feed(x[2 * a: 2 * b]) for a, b in zip(cuts[:-1], cuts[1:])])
            _check(f"chain_{name}_{''.join(map(str, modes))}", y)


def test_centre_tap_wraps_at_32_bits():
    """the build's `((int32_t) x) << 11` (inthalfbandfiltereo.h:818-827) is an int shift: a constant 2^20 stream through one
    stage comes out as -512, not 2^20 (the centre tap lands on -2^31), and 2^22 comes out halved (the tap vanishes).  The
    fixtures above hold such samples; this test names the reason they look the way they do."""
    for v, want in ((1 << 20, -512), (1 << 22, 2095104), (1 << 19, 524032)):
        y = orc.Chain24([0]).feed(np.full(2 * 400, v, np.int32))
        assert y.size == 400 and int(y[-2]) == want and int(y[-1]) == want, (v, y[-2:])
