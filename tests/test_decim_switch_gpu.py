"""GPU: one Decimators object, several decimateK_x.  The reference runs every cascade on the SAME six filters
(decimators.h:326-333), so a change of K / fcPos at run time continues on the other cascade's leftovers.
sdrx_decim_save_stages / sdrx_decim_load_stages reproduce that; the oracle's model of it (sdro_decim_switch) is pinned
against the compiled reference object in tests/test_oracle_vs_ref.py::test_decimators_variant_switch."""
import ctypes as C
import os

import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc
from tests import synth

pytestmark = pytest.mark.gpu


def _switch(o, log2, fc):
    orc.lib().sdro_decim_switch(o.h, log2, fc)
    o.log2 = log2


@pytest.mark.parametrize("bits", (8, 12, 16))
def test_variant_changes_on_one_object(bits):
    rng = np.random.default_rng(100 + bits)
    for trial in range(int(os.environ.get("SDRX_FUZZ_SEEDS", "6"))):
        amp = int(rng.choice([127, 2047, 32767]))
        x = synth.mix(400_000, 4000 + trial + bits, amp, 500, 1)
        obj = sa.DecimatorsObject(bits)
        o, pos = None, 0
        for seg in range(8):
            log2, fc = int(rng.integers(0, 7)), int(rng.integers(0, 3))
            # short stays (still inside the 4096-sample hand-over), exact multiples of it, long stays (parallel kernels take over)
            n = int(rng.choice([8, 64, 200, 1000, 2048, 4096, 5000, 9000, 20000])) * 2
            buf = x[pos: pos + n]; pos += n
            if o is None:
                o = orc.Decim(log2, fc, bits)
            else:
                _switch(o, log2, fc)
            # two calls per stay: the second one continues the hand-over or runs on the parallel path
            cut = (buf.size // 3) & ~1
            got = np.concatenate([obj.decimate(log2, fc, buf[:cut]), obj.decimate(log2, fc, buf[cut:])])
            want = np.concatenate([o.process(buf[:cut]), o.process(buf[cut:])])
            assert np.array_equal(got, want), (bits, trial, seg, log2, fc, n)
        obj.close()


def test_variant_changes_u8():
    rng = np.random.default_rng(7)
    xu = (synth.lcg_u32(400_000, 15) & 0xff).astype(np.uint8)
    obj = sa.DecimatorsObject(8, u8_shift=127)
    o, pos = None, 0
    for seg in range(10):
        log2, fc = int(rng.integers(1, 7)), int(rng.integers(0, 3))
        n = int(rng.choice([64, 1000, 4096, 9000])) * 2
        buf = xu[pos: pos + n]; pos += n
        if o is None:
            o = orc.DecimU(log2, fc, 127)
        else:
            _switch(o, log2, fc)
        assert np.array_equal(obj.decimate(log2, fc, buf), o.process(buf)), (seg, log2, fc, n)
    obj.close()


def test_return_to_a_variant_sees_what_the_others_left():
    """A -> B -> A: A's deeper stages still hold what A left, its first stages hold what B left"""
    x = synth.mix(120_000, 9, 2047, 900, 1)
    obj = sa.DecimatorsObject(12)
    o = orc.Decim(6, 2, 12)
    plan = [(6, 2, 30000), (2, 0, 1000), (6, 2, 3000), (3, 1, 8), (6, 2, 20000)]
    pos = 0
    for k, (log2, fc, n) in enumerate(plan):
        buf = x[pos: pos + 2 * n]; pos += 2 * n
        if k:
            _switch(o, log2, fc)
        assert np.array_equal(obj.decimate(log2, fc, buf), o.process(buf)), k
    obj.close()


def test_checkpoint_refused_while_on_loaded_stages():
    d = sa.Decimators(4, 2, 12); st = sa.DecimStages()
    d.load_stages(st)
    buf = np.zeros(sa.lib().sdrx_decim_state_bytes(d._h), np.uint8)
    assert sa.lib().sdrx_decim_get_state(d._h, buf.ctypes.data) != 0
    d.decimate(synth.mix(5000, 1, 2047, 0, 1))                      # past the hand-over: the input history is the state again
    assert sa.lib().sdrx_decim_get_state(d._h, buf.ctypes.data) == 0
    d.close(); st.close()


@pytest.mark.parametrize("kind,bits", [("fi", 16), ("ff", 16), ("if", 12)])
def test_float_decimators_variant_changes_on_one_object(kind, bits):
    """DecimatorsFI / FF / IF: the six IntHalfbandFilterEOF members are shared by all cascades of an object as well
    (oracle model pinned by tests/test_oracle_vs_ref.py::test_float_decimators_variant_switch)"""
    rng = np.random.default_rng(31 + bits)
    for trial in range(int(os.environ.get("SDRX_FUZZ_SEEDS", "5"))):
        obj = sa.FloatDecimatorsObject(kind, bits)
        o = None
        for seg in range(8):
            log2 = int(rng.integers(0, 7)); fc = int(rng.integers(0, 3)) if log2 else 2
            blk = int(rng.choice([16, 130, 2 * 777, 5000, 40000, 300000]))
            x = rng.uniform(-0.95, 0.95, blk).astype(np.float32) if kind != "if" else rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), blk).astype(np.int16)
            if o is None:
                o = orc.FDecim(kind, log2, fc, bits)
            else:
                o.switch(log2, fc)
            cut = (blk // 3) & ~1
            got = np.concatenate([obj.decimate(log2, fc, x[:cut]), obj.decimate(log2, fc, x[cut:])])
            want = np.concatenate([o.process(x[:cut]), o.process(x[cut:])])
            assert got.size == want.size and np.array_equal(got.view(np.uint8), want.view(np.uint8)), (kind, trial, seg, log2, fc, blk)
        obj.close()
