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


def test_round2_entry_points_refuse_bad_arguments_and_take_empty_calls():
    """null handles / pointers, out-of-range channels and empty inputs on the entry points added in round 2"""
    L = sa.lib()
    out = np.zeros(64, np.int32); n = C.c_int32(); n64 = C.c_int64(); p = C.c_void_p()
    # 24-bit flavour
    d = sa.Decimators24(3, sa.FC_INF, 12)
    assert d.decimate(np.zeros(0, np.int16)).size == 0
    assert d.decimate(np.zeros(30, np.int16)).size == 0                      # shorter than one 32-int16 group of decimate8_inf
    assert L.sdrx_decim24_process(d._h, None, 64, out.ctypes.data, C.byref(n)) == -1
    assert L.sdrx_decim24_process(None, out.ctypes.data, 64, out.ctypes.data, C.byref(n)) == -1
    assert L.sdrx_decim24_process_dev(d._h, None, 64, None, C.byref(n64)) == -1
    b = sa.ChannelizerBank24(2_400_000, [48000, 2_400_000], [100_000, 0])     # second channel: no stage, pass-through
    assert [o.size for o in b.feed(np.zeros(0, np.int32))] == [0, 0]
    x = synth.noise24(3000, 5)
    y = b.feed(x)
    assert np.array_equal(y[1], x)                                            # pass-through hands the input back
    assert L.sdrx_chan24_bank_read(b._h, 7, out.ctypes.data, 10) == -1
    assert L.sdrx_chan24_bank_info(b._h, 2, None, None, None, None) == -1
    assert L.sdrx_chan24_bank_feed(b._h, None, 10) == -1
    assert L.sdrx_chan24_bank_out_dev(b._h, 0, None, C.byref(n64)) == -1
    # shared stage states
    g = sa.Decimators(4, sa.FC_CEN, 12); st = sa.DecimStages()
    assert L.sdrx_decim_save_stages(None, st._h) == -1 and L.sdrx_decim_load_stages(g._h, None) == -1
    assert L.sdrx_decim_stages_create(None, 0) == -1
    g1 = sa.Decimators(0, sa.FC_CEN, 12)                                      # decimate1 has no stage: save / load are no-ops
    g1.save_stages(st); g1.load_stages(st)
    assert np.array_equal(g1.decimate(np.arange(8, dtype=np.int16)), (np.arange(8) << 4).astype(np.int16))
    # batch, ring
    assert L.sdrx_decim_process_dev_batch(None, 1, None, None, None, None) == -1
    assert L.sdrx_decim_ring_create(g._h, 100, 1, 1) == -1                     # needs two slots
    assert L.sdrx_decim_ring_acquire(g._h) in (None, 0)                        # no ring yet
    # corrections, audio tail, IIR
    assert L.sdrx_iqimb_create(C.byref(p), 0, 0) == -1
    assert L.sdrx_audiotail_create(C.byref(p), 0, 0, None) == -1
    assert L.sdrx_iir_create(C.byref(p), 0, 0, None) == -1
    bad = (sa.IirCfg * 1)(); bad[0].order = 9
    assert L.sdrx_iir_create(C.byref(p), 0, 1, bad) == -1
    for h in (d, b, g, g1, st):
        h.close()


def test_checkpoints_of_bank_and_float_decimators():
    """get_state / set_state: a fresh object that is given the state continues exactly where the first one was"""
    L = sa.lib()
    x = synth.mix(300_000, 61, 20000, 9000)
    rates = [48000, 48000, 12500, 2_400_000]; fcs = [100_000, -733_000, 555_555, 0]
    a = sa.ChannelizerBank(2_400_000, rates, fcs)
    a.feed(x[: 2 * 123_457])
    st = np.zeros(L.sdrx_chan_bank_state_bytes(a._h), np.uint8)
    assert L.sdrx_chan_bank_get_state(a._h, st.ctypes.data) == 0
    for c in range(4):
        a.skip(c)
    a.feed(x[2 * 123_457:])
    b = sa.ChannelizerBank(2_400_000, rates, fcs)
    b.feed(x[:2000])                                       # something else first: set_state drops it
    assert L.sdrx_chan_bank_set_state(b._h, st.ctypes.data) == 0
    assert all(b.available(c) == 0 for c in range(3))
    b.feed(x[2 * 123_457:])
    for c in range(3):
        assert np.array_equal(a.read(c), b.read(c)), c
    other = sa.ChannelizerBank(2_400_000, rates[:2], fcs[:2])
    assert L.sdrx_chan_bank_set_state(other._h, st.ctypes.data) == -1 and b"another configuration" in L.sdrx_last_error()
    for h in (a, b, other):
        h.close()
    xf = np.random.default_rng(3).uniform(-0.9, 0.9, 200_000).astype(np.float32)
    f1 = sa.FloatDecimators("fi", 6, sa.FC_CEN); f2 = sa.FloatDecimators("fi", 6, sa.FC_CEN)
    f1.decimate(xf[:77_056])
    fs = np.zeros(L.sdrx_fdecim_state_bytes(f1._h), np.uint8)
    assert L.sdrx_fdecim_get_state(f1._h, fs.ctypes.data) == 0 and L.sdrx_fdecim_set_state(f2._h, fs.ctypes.data) == 0
    assert np.array_equal(f1.decimate(xf[77_056:]), f2.decimate(xf[77_056:]))
    f1.close(); f2.close()
