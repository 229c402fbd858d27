#!/usr/bin/env python3
"""Rate of experiments/mfma_decim (decimate64_cen on the matrix cores, NOT the product path) next to the product
kernel on the same resident buffer.  HIP-event timing on one stream; prints one JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdrangel_amd as sa
sa.lib()
import torch

L = C.CDLL(os.path.join(ROOT, "experiments", "mfma_decim", "libmfx.so"))
L.mfx_decim64.restype = C.c_int
L.mfx_decim64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
spws = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [32, 64]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5489)
x = torch.randint(-2048, 2048, (2 * n,), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
hist = torch.zeros(2 * 4096, dtype=torch.int16, device=dev)
out = torch.zeros(2 * (n // 64) + 64, dtype=torch.int16, device=dev)
out2 = torch.zeros_like(out)
flags = torch.zeros(n // 4096 + 2, dtype=torch.int32, device=dev)
s = torch.cuda.Stream()
res = {"n_cplx": n, "lds_bytes": L.mfx_lds_bytes()}
with torch.cuda.stream(s):
    for spw in spws:
        for _ in range(2):
            L.mfx_decim64(hist.data_ptr(), x.data_ptr(), out.data_ptr(), flags.data_ptr(), n, spw, s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            L.mfx_decim64(hist.data_ptr(), x.data_ptr(), out.data_ptr(), flags.data_ptr(), n, spw, s.cuda_stream)
        e1.record(s); e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res[f"mfma_spw{spw}"] = {"ms": round(ms, 4), "GSps": round(n / ms / 1e6, 1), "hbm_frac": round(4.0625 * n / (ms * 1e-3) / 8e12, 4)}
    h = sa.Decimators(6, sa.FC_CEN, 12)
    h.set_stream(s.cuda_stream)
    for _ in range(2):
        h.decimate_dev(x.data_ptr(), 2 * n, out2.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5):
        h.decimate_dev(x.data_ptr(), 2 * n, out2.data_ptr())
    e1.record(s); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res["product_valu"] = {"ms": round(ms, 4), "GSps": round(n / ms / 1e6, 1), "hbm_frac": round(4.0625 * n / (ms * 1e-3) / 8e12, 4)}
s.synchronize()
# same data, zero history on both sides: the first call of each must agree bit for bit
L.mfx_decim64(hist.data_ptr(), x.data_ptr(), out.data_ptr(), flags.data_ptr(), n, spws[0], s.cuda_stream)
h2 = sa.Decimators(6, sa.FC_CEN, 12); h2.set_stream(s.cuda_stream)
h2.decimate_dev(x.data_ptr(), 2 * n, out2.data_ptr())
s.synchronize()
res["bit_exact_vs_product"] = bool(torch.equal(out[: 2 * (n // 64)], out2[: 2 * (n // 64)]))
res["flags_raised"] = int(flags.sum().item())
print(json.dumps(res))
