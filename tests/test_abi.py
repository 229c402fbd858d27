"""CPU: the C-ABI library loads, exports every symbol include/sdrx.h declares, its host-only logic
(channel plan, SampleSinkFifo mirror) behaves like the reference, and GPU objects fail LOUDLY
without a device (no CPU fallback)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import sdrangel_amd as sa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def _lib_built():
    if not os.path.exists(sa.LIB_PATH):
        subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "sdrangel_amd", "csrc")])


def test_every_declared_symbol_is_exported():
    names = sa.exported_symbols()
    assert len(names) >= 45
    out = subprocess.check_output(["nm", "-D", "--defined-only", sa.LIB_PATH], text=True)
    defined = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    missing = [n for n in names if n not in defined]
    assert not missing, missing
    L = sa.lib()                                   # binds argtypes for all of them
    assert L.sdrx_version().startswith(b"sdrx")


def test_product_never_links_the_oracle():
    out = subprocess.check_output(["ldd", sa.LIB_PATH], text=True)
    assert "sdro" not in out and "sdrref" not in out
    for root, _d, files in os.walk(os.path.join(ROOT, "sdrangel_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "libsdro" not in txt and "sdro_" not in txt and "oracle_py" not in txt and "libsdrref" not in txt, f


def test_channel_plan_host_logic_matches_reference_fixture():
    for p in json.load(open(os.path.join(G, "chan_plans.json"))):
        modes, out_rate, ofs = sa.chan_plan(p["in"], p["req"], p["fc"])
        assert list(modes) == p["modes"] and out_rate == p["out_rate"] and ofs == p["ofs"], p


def test_group_strides():
    L = sa.lib()
    want = {(0, 2): 2, (1, 0): 8, (2, 2): 16, (3, 0): 32, (3, 2): 16, (6, 1): 256, (6, 2): 128}
    for (l, f), g in want.items():
        assert L.sdrx_decim_group_int16(l, f) == g


def test_gpu_objects_fail_loudly_without_device():
    if sa.lib().sdrx_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(sa.SdrxError) as e:
        sa.Decimators(6, sa.FC_CEN, 12)
    assert "no CPU fallback" in str(e.value) or "rc=-2" in str(e.value)
    with pytest.raises(sa.SdrxError):
        sa.ChannelizerBank(61440000, [48000], [0])


def test_bad_arguments_are_rejected():
    h = C.c_void_p()
    L = sa.lib()
    assert L.sdrx_decim_create(C.byref(h), 0, 7, 2, 12) == -1      # log2 out of range
    assert L.sdrx_decim_create(C.byref(h), 0, 3, 5, 12) == -1      # fcpos
    assert L.sdrx_decim_create(C.byref(h), 0, 3, 2, 10) == -1      # input bits
    assert b"log2" in L.sdrx_last_error()


# ---------------------------------------------------------------- SampleSinkFifo mirror
class RefFifoModel:
    """The reference's semantics (samplesinkfifo.cpp:70-231) in ten lines of Python."""

    def __init__(self, size):
        self.size, self.buf = size, []

    def write(self, samples):
        room = self.size - len(self.buf)
        take = samples[:room]
        self.buf += take
        return len(take)

    def read(self, n):
        out, self.buf = self.buf[:n], self.buf[n:]
        return out


def _s(vals):
    a = np.zeros(2 * len(vals), np.int16)
    a[0::2] = vals
    a[1::2] = [-v for v in vals]
    return a


def test_fifo_matches_reference_semantics():
    f = sa.SampleSinkFifo(10)
    m = RefFifoModel(10)
    rng = np.random.default_rng(1)
    nxt = 1
    for _ in range(400):
        if rng.random() < 0.55:
            k = int(rng.integers(0, 8))
            vals = list(range(nxt, nxt + k)); nxt += k
            assert f.write(_s(vals)) == m.write(vals)
        else:
            k = int(rng.integers(0, 8))
            got = f.read(k)
            want = m.read(k)
            assert list(got[0::2]) == want and list(got[1::2]) == [-v for v in want]
        assert f.fill == len(m.buf)
    assert f.dropped > 0                                # overflowing writes dropped their tail, counted


def test_fifo_read_begin_commit_two_spans_and_write_bytes():
    f = sa.SampleSinkFifo(8)
    assert f.write(_s([1, 2, 3, 4, 5, 6])) == 6
    assert list(f.read(5)[0::2]) == [1, 2, 3, 4, 5]
    assert f.write_bytes(_s([7, 8, 9, 10, 11]).tobytes() + b"\x01\x02\x03") == 5   # count /= sizeof(Sample)
    tot, p1, p2 = f.read_begin(100)                     # asks for more than there is: "underflow", gets the fill
    assert tot == 6 and list(p1[0::2]) == [6, 7, 8] and list(p2[0::2]) == [9, 10, 11]
    assert f.read_commit(100) == 6 and f.fill == 0      # cannot commit more than available
    tot, p1, p2 = f.read_begin(3)
    assert tot == 0 and p1.size == 0 and p2.size == 0
    f.set_size(4)                                       # setSize empties
    assert f.size == 4 and f.fill == 0
    assert f.write(_s([1, 2, 3, 4, 5, 6])) == 4


def test_fifo_data_ready_callback_fires_after_nonempty_write():
    L = sa.lib()
    CB = C.CFUNCTYPE(None, C.c_void_p)
    hits = []
    cb = CB(lambda u: hits.append(1))
    L.sdrx_fifo_on_data_ready.argtypes = [C.c_void_p, CB, C.c_void_p]
    L.sdrx_fifo_on_data_ready.restype = None
    f = sa.SampleSinkFifo(4)
    L.sdrx_fifo_on_data_ready(f._h, cb, None)
    f.write(_s([]))
    assert hits == []
    f.write(_s([1]))
    f.write(_s([2, 3, 4, 5]))                           # partly dropped, still non-empty -> still signalled
    assert len(hits) == 2


def test_sdriq_header_roundtrip_and_filesource_to_fifo():
    """.sdriq layout of FileRecord::writeHeader (24 bytes, field by field) and the FileSource pump
    (file bytes straight into SampleSinkFifo::write(const quint8*, uint), filesourcethread.cpp:213)."""
    import struct
    from tests import synth
    x = synth.mix(3000, 8, 2047, 500)
    blob = sa.sdriq_header_bytes(2_400_000, 433_920_000, 1_700_000_000, 16) + x.tobytes()
    # the same 24 bytes struct.pack would give for qint32 | quint64 | time_t | quint32, little endian, unpadded
    assert blob[:24] == struct.pack("<iQqI", 2_400_000, 433_920_000, 1_700_000_000, 16)
    h, body = sa.sdriq_parse(blob)
    assert (h.sample_rate, h.center_frequency, h.start_timestamp, h.sample_size) == (2_400_000, 433_920_000, 1_700_000_000, 16)
    assert np.array_equal(body, x)
    garbage = bytearray(blob); garbage[20:24] = struct.pack("<I", 12345)
    assert sa.sdriq_parse(bytes(garbage))[0].sample_size == 16          # old files: "assume 16 bits if garbage"
    with pytest.raises(sa.SdrxError):
        sa.sdriq_parse(blob[:10])
    # FileSourceThread::tick: chunks of file bytes -> fifo (rate * 4 samples large, filesourceinput.cpp:143)
    fifo = sa.SampleSinkFifo(2_400_000 * 4)
    data = blob[24:]
    for off in range(0, len(data), 1000):
        fifo.write_bytes(data[off: off + 1000])
    assert fifo.fill == 3000 and np.array_equal(fifo.read(3000), x)
