"""GPU parity of the float half-band decimators (DecimatorsFI / DecimatorsFF / DecimatorsIF over
IntHalfbandFilterEOF<64>) against the oracle restatement, which tests/test_oracle_vs_ref.py pins bit for bit to the
compiled reference classes.  Bar: bit-identical (same float operation order, no FMA) -- float outputs are compared
as raw bits, int16 outputs as integers."""
import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc

pytestmark = pytest.mark.gpu


def _input(kind, n, seed, bits=16):
    rng = np.random.default_rng(seed)
    if kind == "if":
        lim = 1 << (bits - 1)
        return rng.integers(-lim, lim, n).astype(np.int16)
    t = np.arange(n // 2)
    x = np.empty(n, np.float32)
    x[0::2] = (0.6 * np.cos(2 * np.pi * 0.0031 * t) + 0.3 * rng.uniform(-1, 1, t.size)).astype(np.float32)
    x[1::2] = (0.6 * np.sin(2 * np.pi * 0.0031 * t) + 0.3 * rng.uniform(-1, 1, t.size)).astype(np.float32)
    return x


def _same(a, b):
    return a.size == b.size and np.array_equal(a.view(np.uint8), b.view(np.uint8))


CASES = [(k, L, fc, bits) for k, bl in (("fi", (16,)), ("ff", (16,)), ("if", (8, 12, 16))) for bits in bl
         for L in range(7) for fc in (sa.FC_INF, sa.FC_SUP, sa.FC_CEN) if not (L == 0 and fc != sa.FC_CEN)]


@pytest.mark.parametrize("kind,L,fc,bits", CASES)
def test_streaming_blocks_match_oracle(kind, L, fc, bits):
    g = sa.FloatDecimators(kind, L, fc, bits)
    o = orc.FDecim(kind, L, fc, bits)
    # ragged device-thread blocks: tails are dropped, state is carried; one block shorter than the warm-up history
    for i, n in enumerate((2 * 40000 + 6, 2, 2 * 300, 2 * 70001, 0, 2 * 2048 * 5)):
        x = _input(kind, n, 1000 * L + 10 * fc + i, bits)
        got, want = g.decimate(x), o.process(x)
        assert _same(got, want), (kind, L, fc, bits, i, n, got.size, want.size)


def test_golden_vectors_from_the_compiled_reference():
    import os
    from tests import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fdecim_golden.npz"))
    for kind, gk, nbits in (("fi", "fi", 16), ("ff", "ff", 16), ("if12", "if", 12)):
        for L, fc in synth.FDECIM_CASES:
            n = 3000 if L <= 2 else 24000
            x = synth.fdecim_input(kind, n, 100 + 7 * L + fc)
            d = sa.FloatDecimators(gk, L, fc, nbits)
            y = np.concatenate([d.decimate(x[2 * a: 2 * b]) for a, b in synth.fdecim_cuts(n)])
            assert _same(y, g[f"{kind}_L{L}_fc{fc}"]), (kind, L, fc)


def test_reset_and_long_run_property():
    """a long run (4 Mi samples, many segments per launch) equals the same stream cut into device-thread blocks"""
    n = 2 * (1 << 22)
    x = _input("fi", n, 7)
    a = sa.FloatDecimators("fi", 6, sa.FC_CEN)
    whole = a.decimate(x)
    a.reset()
    parts = np.concatenate([a.decimate(x[i: i + 2 * 32768]) for i in range(0, n, 2 * 32768)])
    assert _same(whole, parts)
    o = orc.FDecim("fi", 6, sa.FC_CEN)
    assert _same(whole[: 2 * 4096], o.process(x[: 2 * 4096 * 64]))


def test_bad_arguments_fail_loudly():
    with pytest.raises(sa.SdrxError):
        sa.FloatDecimators("fi", 7)
    h = sa.lib()
    import ctypes as C
    p = C.c_void_p()
    assert h.sdrx_fdecim_create(C.byref(p), 0, 3, 2, 1, 0, 12) == -1          # int16 in -> int16 out is not a reference class
