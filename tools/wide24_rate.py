#!/usr/bin/env python3
"""Rate of the 24-bit sample flavour (sdrx_decim24_*, sdrx_chan24_bank_*) on inputs resident in HBM -- a completeness
number for DESIGN.md 9, not a bench line.  python tools/wide24_rate.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sdrangel_amd as sa  # noqa: E402

res = {}
n = 1 << 27
x = torch.randint(-2048, 2048, (2 * n,), dtype=torch.int16, device="cuda")
out = torch.zeros(2 * ((n >> 1) + 1), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for log2 in (6, 3, 1):
    d = sa.Decimators24(log2, sa.FC_CEN, 12)
    d.decimate_dev(x.data_ptr(), n, out.data_ptr()); d.sync()
    t = time.perf_counter()
    for _ in range(3):
        d.decimate_dev(x.data_ptr(), n, out.data_ptr())
    d.sync()
    dt = (time.perf_counter() - t) / 3
    res[f"decim24_{1 << log2}_cen_GSps"] = n / dt / 1e9
    d.close()
del x, out
m = 1 << 25
y = torch.randint(-(1 << 23), 1 << 23, (2 * m,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
fcs = [int(-15_000_000 + k * (30_000_000 / 31) + 137 * k) for k in range(32)]
b = sa.ChannelizerBank24(61_440_000, [48000] * 32, fcs)
b.feed_dev(y.data_ptr(), m); b.sync()
t = time.perf_counter()
for _ in range(3):
    b.feed_dev(y.data_ptr(), m)
b.sync()
dt = (time.perf_counter() - t) / 3
res["chan24_32ch_61.44M_to_48k_GSps"] = m / dt / 1e9
b.close()
res["note"] = "plain 64-bit kernel, stages shared between channels (trie of <= 6-stage segments); inputs in HBM; 1 GPU"
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
