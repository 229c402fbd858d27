"""PCIe-inclusive rate of the host-pointer API (sdrx_decim_process): H2D copy + kernel + D2H copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdrangel_amd as sa
from tests import synth

for n in (32768, 1 << 20, 1 << 24, 1 << 26):
    x = synth.noise_iq(n, 5, 2047)
    d = sa.Decimators(6, sa.FC_CEN, 12)
    d.decimate(x)
    reps = max(2, min(50, (1 << 27) // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        d.decimate(x)
    dt = (time.perf_counter() - t0) / reps
    print(f"host-buffer decimate64_cen: n={n} samples  {dt*1e3:.3f} ms/call  {n/dt/1e6:.1f} MS/s  ({4*n/dt/1e9:.2f} GB/s over PCIe, pageable host memory)")
