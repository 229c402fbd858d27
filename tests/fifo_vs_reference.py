"""Run by tests/test_oracle_vs_ref.py in a child process (LD_PRELOAD of the system libstdc++, see tests/golden/make_golden.py):
random operation sequences on the reference's real SampleSinkFifo (oracle/_ref/libsdrref_qt.so) and on the sdrx_fifo_* mirror;
every return value, fill level and sample handed out must agree."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdrangel_amd as sa  # noqa: E402

q = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsdrref_qt.so"))
vp, u32 = C.c_void_p, C.c_uint32
q.refqt_fifo_new.restype = vp; q.refqt_fifo_new.argtypes = [C.c_int]
q.refqt_fifo_free.argtypes = [vp]
q.refqt_fifo_fill.restype = u32; q.refqt_fifo_fill.argtypes = [vp]
q.refqt_fifo_write.restype = u32; q.refqt_fifo_write.argtypes = [vp, vp, u32]
q.refqt_fifo_write_bytes.restype = u32; q.refqt_fifo_write_bytes.argtypes = [vp, vp, u32]
q.refqt_fifo_read.restype = u32; q.refqt_fifo_read.argtypes = [vp, vp, u32]
q.refqt_fifo_read_begin.restype = u32; q.refqt_fifo_read_begin.argtypes = [vp, u32, vp, C.POINTER(u32), C.POINTER(u32)]
q.refqt_fifo_read_commit.restype = u32; q.refqt_fifo_read_commit.argtypes = [vp, u32]

rng = np.random.default_rng(2024)
n_ops = 0
for size in (1, 7, 64, 1000):
    ref = q.refqt_fifo_new(size); mine = sa.SampleSinkFifo(size)
    for _ in range(600):
        op = rng.integers(0, 5)
        n = int(rng.integers(0, 2 * size + 3))
        if op == 0:
            x = rng.integers(-32768, 32768, 2 * n).astype(np.int16)
            a = q.refqt_fifo_write(ref, x.ctypes.data, n); b = mine.write(x)
            assert a == b, ("write", size, n, a, b)
        elif op == 1:
            nb = int(rng.integers(0, 4 * size + 7))          # byte count, not a multiple of 4 on purpose
            d = rng.integers(0, 256, nb).astype(np.uint8)
            a = q.refqt_fifo_write_bytes(ref, d.ctypes.data, nb); b = mine.write_bytes(d.tobytes())
            assert a == b, ("write_bytes", size, nb, a, b)
        elif op == 2:
            out = np.zeros(2 * n + 2, np.int16)
            a = q.refqt_fifo_read(ref, out.ctypes.data, n); got = mine.read(n)
            assert a == got.size // 2 and np.array_equal(out[: 2 * a], got), ("read", size, n, a, got.size)
        else:
            out = np.zeros(2 * n + 2, np.int16); n1, n2 = u32(), u32()
            tot = q.refqt_fifo_read_begin(ref, n, out.ctypes.data, C.byref(n1), C.byref(n2))
            t2, s1, s2 = mine.read_begin(n)
            assert tot == t2 and n1.value == s1.size // 2 and n2.value == s2.size // 2, ("read_begin", size, n, tot, t2, n1.value, n2.value)
            assert np.array_equal(out[: 2 * tot], np.concatenate([s1, s2])), ("read_begin data", size, n)
            c = int(rng.integers(0, tot + 1)) if op == 3 else tot
            assert q.refqt_fifo_read_commit(ref, c) == mine.read_commit(c), ("read_commit", size, c)
        assert q.refqt_fifo_fill(ref) == mine.fill, ("fill", size)
        n_ops += 1
    q.refqt_fifo_free(ref)
print("fifo vs reference:", n_ops, "operations agree")

# ---- .sdriq: the reference's FileRecord writes a recording, sdrx_sdriq_* reads it; sdrx writes a header, FileRecord::readHeader reads it
import tempfile
q.refqt_filerecord_write.restype = C.c_int; q.refqt_filerecord_write.argtypes = [C.c_char_p, C.c_int, C.c_longlong, vp, u32]
q.refqt_filerecord_read_header.restype = C.c_int
q.refqt_filerecord_read_header.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_longlong), C.POINTER(u32)]
tmp = tempfile.mkdtemp()
x = rng.integers(-32768, 32768, 2 * 5000).astype(np.int16)
path = os.path.join(tmp, "ref.sdriq").encode()
assert q.refqt_filerecord_write(path, 2_400_000, 435_123_456, x.ctypes.data, 5000) == 0
hdr, body = sa.sdriq_parse(open(path, "rb").read())
assert (hdr.sample_rate, hdr.center_frequency, hdr.sample_size) == (2_400_000, 435_123_456, 16) and hdr.start_timestamp > 1_500_000_000
assert np.array_equal(body, x)
mine_path = os.path.join(tmp, "mine.sdriq")
open(mine_path, "wb").write(sa.sdriq_header_bytes(61_440_000, 1_296_000_000, 1_700_000_123, 16) + x.tobytes())
r, cf, ts, ss = C.c_int(), C.c_ulonglong(), C.c_longlong(), u32()
assert q.refqt_filerecord_read_header(mine_path.encode(), C.byref(r), C.byref(cf), C.byref(ts), C.byref(ss)) == 0
assert (r.value, cf.value, ts.value, ss.value) == (61_440_000, 1_296_000_000, 1_700_000_123, 16)
open(mine_path, "wb").write(sa.sdriq_header_bytes(48000, 1, 2, 12345))     # garbage sample size: the reference assumes 16 bits
assert q.refqt_filerecord_read_header(mine_path.encode(), C.byref(r), C.byref(cf), C.byref(ts), C.byref(ss)) == 0 and ss.value == 16
h2, _ = sa.sdriq_parse(open(mine_path, "rb").read())
assert h2.sample_size == 16
print("sdriq vs reference FileRecord: headers and samples agree")
