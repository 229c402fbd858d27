"""GPU: API edge cases of the C ABI -- empty and tiny calls, wrong-flavour calls, misaligned device pointers,
two handles driven from two host threads, state hand-over between handles, maximum-depth chains."""
import ctypes as C
import threading

import numpy as np
import pytest

import sdrangel_amd as sa
from tests import oracle_py as orc
from tests import synth

pytestmark = pytest.mark.gpu


def test_empty_and_sub_group_calls_change_nothing():
    x = synth.mix(20000, 2, 2047, 700)
    g, o = sa.Decimators(6, sa.FC_INF, 12), orc.Decim(6, sa.FC_INF, 12)
    assert g.decimate(np.zeros(0, np.int16)).size == 0
    assert g.decimate(x[:254]).size == 0                 # one int16 pair short of a 256-int16 group: dropped, not carried
    assert np.array_equal(g.decimate(x), o.process(x))   # ... and the state is untouched by it
    bank = sa.ChannelizerBank(2_400_000, [48000, 48000], [0, 300_000])
    bank.feed(np.zeros(0, np.int16))
    assert bank.available(0) == 0 and bank.read(0).size == 0
    bank.feed(x[:2])                                      # a single sample: stored, no output yet
    assert bank.available(0) == 0


def test_wrong_flavour_and_bad_pointers_are_refused():
    L = sa.lib()
    g = sa.Decimators(3, sa.FC_CEN, 12)
    u = sa.DecimatorsU(3, sa.FC_CEN, 127)
    buf = np.zeros(64, np.int16); out = np.zeros(64, np.int16); n = C.c_int32()
    assert L.sdrx_decim_process_u8(g._h, buf.ctypes.data, 64, out.ctypes.data, C.byref(n)) == -5     # SDRX_ESTATE
    assert L.sdrx_decim_process(u._h, buf.ctypes.data, 64, out.ctypes.data, C.byref(n)) == -5
    assert L.sdrx_decim_process(g._h, None, 64, out.ctypes.data, C.byref(n)) == -1
    n64 = C.c_int64()
    torch = pytest.importorskip("torch")
    d = torch.zeros(4096, dtype=torch.int16, device="cuda")
    assert L.sdrx_decim_process_dev(g._h, d.data_ptr() + 4, 1024, d.data_ptr(), C.byref(n64)) == -1  # not 16-byte aligned
    assert b"aligned" in L.sdrx_last_error()
    bank = sa.ChannelizerBank(2_400_000, [48000], [0])
    assert L.sdrx_chan_bank_read(bank._h, 5, out.ctypes.data, 10) == -1
    assert L.sdrx_chan_bank_info(bank._h, -1, None, None, None, None) == -1


def test_two_handles_two_threads():
    xs = [synth.mix(300_000, 40 + i, 2047, 500) for i in range(2)]
    want = [orc.Decim(6, sa.FC_CEN, 12).process(x) for x in xs]
    got = [None, None]

    def work(i):
        g = sa.Decimators(6, sa.FC_CEN, 12)
        parts = [g.decimate(xs[i][a: a + 2 * 50_048]) for a in range(0, xs[i].size, 2 * 50_048)]
        got[i] = np.concatenate(parts)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    # 50 048 is a multiple of 64, so no tail is dropped between the calls
    for i in range(2):
        assert np.array_equal(got[i], want[i]), i


def test_state_moves_between_handles():
    x = synth.mix(100_000, 9, 2047, 900)
    a = sa.Decimators(5, sa.FC_SUP, 12)
    first = a.decimate(x[: 2 * 40_064])
    b = sa.Decimators(5, sa.FC_SUP, 12)
    b.set_state(a.get_state())
    o = orc.Decim(5, sa.FC_SUP, 12)
    want = o.process(x[: 2 * 40_064]), o.process(x[2 * 40_064:])
    assert np.array_equal(first, want[0]) and np.array_equal(b.decimate(x[2 * 40_064:]), want[1])


def test_deep_chain_three_passes():
    """2.4 MS/s -> 100 S/s request: 14 half-band stages, i.e. three passes of the tree kernel for one channel"""
    modes, out_rate, ofs = orc.chan_plan(2_400_000, 100, 123_456)
    assert len(modes) >= 13
    bank = sa.ChannelizerBank(2_400_000, [100, 48000], [123_456, 123_456])
    x = synth.mix(1 << 20, 5, 3000, 2500)
    bank.feed(x[: 2 * 333_333]); bank.feed(x[2 * 333_333:])
    assert np.array_equal(bank.read(0), orc.Chain(modes).feed(x))
    m2, _, _ = orc.chan_plan(2_400_000, 48000, 123_456)
    assert np.array_equal(bank.read(1), orc.Chain(m2).feed(x))
