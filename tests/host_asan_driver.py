"""Driver for the sanitizer build of the product library's HOST-ONLY slice (oracle/Makefile target `asan`):
sdrx_fifo_* against a Python model of SampleSinkFifo's contract (samplesinkfifo.cpp:70-231: overwrite-oldest on
overflow is NOT done -- the write is truncated and counted as dropped), the .sdriq header round trip, and sdrx_chan_plan
against the fixture taken from the real DownChannelizer (tests/golden/chan_plans.json).  Run by tests/test_sanitizers.py
with the sanitizer runtime preloaded; any ASan/UBSan report makes the process exit non-zero."""
import ctypes as C
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(sys.argv[1])
vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32

# ---- the float bisection of DownChannelizer::createFilterChain
L.sdrx_chan_plan.restype = C.c_int
L.sdrx_chan_plan.argtypes = [i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
n_plans = 0
for p in json.load(open(os.path.join(ROOT, "tests", "golden", "chan_plans.json"))):
    modes = (C.c_uint8 * 32)(); rate = i32(); ofs = i32()
    n = L.sdrx_chan_plan(p["in"], p["req"], p["fc"], modes, C.byref(rate), C.byref(ofs))
    assert list(modes[:n]) == p["modes"] and rate.value == p["out_rate"] and ofs.value == p["ofs"], p
    n_plans += 1

# ---- SampleSinkFifo mirror: random writes / reads / two-part reads against a model
L.sdrx_fifo_create.argtypes = [C.POINTER(vp), u32]
L.sdrx_fifo_write.restype = u32; L.sdrx_fifo_write.argtypes = [vp, vp, u32]
L.sdrx_fifo_read.restype = u32; L.sdrx_fifo_read.argtypes = [vp, vp, u32]
L.sdrx_fifo_fill.restype = u32; L.sdrx_fifo_fill.argtypes = [vp]
L.sdrx_fifo_size.restype = u32; L.sdrx_fifo_size.argtypes = [vp]
L.sdrx_fifo_set_size.argtypes = [vp, u32]
L.sdrx_fifo_destroy.argtypes = [vp]
L.sdrx_fifo_read_begin.restype = u32
L.sdrx_fifo_read_begin.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u32), C.POINTER(vp), C.POINTER(u32)]
L.sdrx_fifo_read_commit.restype = u32; L.sdrx_fifo_read_commit.argtypes = [vp, u32]
rnd = random.Random(7)
for size in (1, 7, 64, 1000):
    f = vp(); assert L.sdrx_fifo_create(C.byref(f), size) == 0
    model = []; seq = 0
    for _ in range(600):
        op = rnd.random()
        if op < 0.45:
            n = rnd.randint(0, size + 5)
            buf = (C.c_int16 * (2 * n))(*[((seq + k // 2) * (1 if k % 2 == 0 else -1)) % 30000 for k in range(2 * n)])
            w = L.sdrx_fifo_write(f, buf, n)
            assert w == min(n, size - len(model)), (w, n, size, len(model))
            model += [(buf[2 * k], buf[2 * k + 1]) for k in range(w)]
            seq += n
        elif op < 0.8:
            n = rnd.randint(0, size + 3)
            out = (C.c_int16 * (2 * max(n, 1)))()
            r = L.sdrx_fifo_read(f, out, n)
            assert r == min(n, len(model))
            assert [(out[2 * k], out[2 * k + 1]) for k in range(r)] == model[:r]
            del model[:r]
        else:
            n = rnd.randint(0, size + 3)
            p1, p2, n1, n2 = vp(), vp(), u32(), u32()
            r = L.sdrx_fifo_read_begin(f, n, C.byref(p1), C.byref(n1), C.byref(p2), C.byref(n2))
            assert r == min(n, len(model)) and n1.value + n2.value == r
            got = []
            for p, k in ((p1, n1.value), (p2, n2.value)):
                if k:
                    a = C.cast(p, C.POINTER(C.c_int16))
                    got += [(a[2 * q], a[2 * q + 1]) for q in range(k)]
            assert got == model[:r]
            assert L.sdrx_fifo_read_commit(f, r) == r
            del model[:r]
        assert L.sdrx_fifo_fill(f) == len(model)
    L.sdrx_fifo_set_size(f, size + 3)
    assert L.sdrx_fifo_fill(f) == 0 and L.sdrx_fifo_size(f) == size + 3
    L.sdrx_fifo_destroy(f)

# ---- .sdriq header: write -> parse round trip, and a truncated buffer is refused
class Hdr(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("center_frequency", C.c_uint64), ("start_timestamp", C.c_int64), ("sample_size", C.c_uint32)]
L.sdrx_sdriq_write_header.argtypes = [vp, C.POINTER(Hdr)]
L.sdrx_sdriq_parse_header.argtypes = [vp, C.c_uint64, C.POINTER(Hdr)]
raw = (C.c_uint8 * 24)()
h = Hdr(61_440_000, 433_920_000_123, 1_600_000_000, 16)
assert L.sdrx_sdriq_write_header(raw, C.byref(h)) == 0
g = Hdr()
assert L.sdrx_sdriq_parse_header(raw, 24, C.byref(g)) == 0
assert (g.sample_rate, g.center_frequency, g.start_timestamp) == (h.sample_rate, h.center_frequency, h.start_timestamp)
assert L.sdrx_sdriq_parse_header(raw, 23, C.byref(g)) != 0
print(f"host slice under sanitizers: {n_plans} channel plans, 4 FIFOs x 600 operations, .sdriq header round trip: ok")
