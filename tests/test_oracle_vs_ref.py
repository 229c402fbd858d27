"""CPU, build container only: the oracle against the reference's own classes compiled from
/root/reference (oracle/_ref/*.so).  Skipped where the reference build did not happen (GPU box).
Randomised, wider than the committed fixtures."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle_py as orc
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libsdrref.so")
pytestmark = [pytest.mark.ref, pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (no /root/reference here)")]


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(REF)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ref_decim_new.restype = vp; L.ref_decim_new.argtypes = [C.c_int]
    L.ref_decim_free.argtypes = [vp]
    L.ref_decim_process.restype = C.c_int; L.ref_decim_process.argtypes = [vp, C.c_int, C.c_int, vp, i32, vp]
    L.ref_chain_new.restype = vp; L.ref_chain_new.argtypes = [C.c_int, vp]
    L.ref_chain_free.argtypes = [vp]
    L.ref_chain_feed.restype = i64; L.ref_chain_feed.argtypes = [vp, vp, i64, vp]
    return L


@pytest.mark.parametrize("bits", (8, 12, 16))
def test_decimators_random_splits(ref, bits):
    rng = np.random.default_rng(bits)
    for log2 in range(7):
        for fc in range(3):
            n = 30000
            x = synth.mix(n, 1000 + bits + log2 * 3 + fc, int(rng.choice([127, 2047, 32767])), 500, 1)
            cuts = sorted(set([0, 2 * n] + [2 * int(v) + int(rng.integers(0, 2)) * 2 for v in rng.integers(0, n, size=4)]))
            h = ref.ref_decim_new(bits); o = orc.Decim(log2, fc, bits)
            for a, b in zip(cuts[:-1], cuts[1:]):
                seg = np.ascontiguousarray(x[a:b]); out = np.zeros(seg.size + 16, np.int16)
                k = ref.ref_decim_process(h, log2, fc, seg.ctypes.data, seg.size, out.ctypes.data)
                assert np.array_equal(o.process(seg), out[: 2 * k]), (bits, log2, fc, a, b)
            ref.ref_decim_free(h)


def test_chains_random(ref):
    rng = np.random.default_rng(77)
    for trial in range(25):
        ns = int(rng.integers(1, 12))
        modes = rng.integers(0, 3, size=ns).astype(np.uint8)
        n = 1 << 15
        x = synth.noise_iq(n, 300 + trial, [2047, 32767, 20000][trial % 3])
        if trial % 4 == 0:
            x[::5] = -32768
        h = ref.ref_chain_new(ns, modes.ctypes.data); o = orc.Chain(modes)
        cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n, size=4)]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            seg = np.ascontiguousarray(x[2 * a: 2 * b]); out = np.zeros(seg.size + 16, np.int16)
            k = ref.ref_chain_feed(h, seg.ctypes.data, b - a, out.ctypes.data)
            assert np.array_equal(o.feed(seg), out[: 2 * k]), (trial, list(modes))
        ref.ref_chain_free(h)
