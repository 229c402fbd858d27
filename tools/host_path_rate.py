"""PCIe-inclusive rate of the host-side APIs of the decimator (decimate64_cen):
  * sdrx_decim_process        : pageable host memory, synchronous H2D copy + kernel + D2H copy per call
  * sdrx_decim_ring_*         : pinned ring, blocks written straight into the ring (what a device thread's receive call
                                does), asynchronous H2D / kernel / D2H, `flush` full blocks per launch
The ring numbers include writing the block into the pinned slot (numpy copy) only in the `+fill` column."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdrangel_amd as sa
from tests import synth

for n in (32768, 1 << 20, 1 << 24):
    x = synth.noise_iq(n, 5, 2047)
    d = sa.Decimators(6, sa.FC_CEN, 12)
    d.decimate(x)
    reps = max(2, min(50, (1 << 27) // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        d.decimate(x)
    dt = (time.perf_counter() - t0) / reps
    print(f"sdrx_decim_process (pageable, sync): n={n} samples  {dt*1e3:.3f} ms/call  {n/dt/1e6:.1f} MS/s  ({4*n/dt/1e9:.2f} GB/s)")

for n, slots, flush in ((32768, 64, 1), (32768, 64, 16), (32768, 128, 32), (1 << 20, 8, 1), (1 << 20, 8, 2), (1 << 22, 6, 1)):
    x = synth.noise_iq(n, 5, 2047)
    for fill in (False, True):
        d = sa.Decimators(6, sa.FC_CEN, 12)
        d.ring_create(2 * n, slots, flush)
        total = max(4 * slots, min(4096, (1 << 29) // n))
        def run(k):
            inflight = 0
            for _ in range(k):
                if inflight == slots - 1:
                    d.ring_retire(); inflight -= 1
                s = d.ring_acquire()
                if fill:
                    s[:] = x
                d.ring_submit(2 * n); inflight += 1
            while inflight:
                d.ring_retire(); inflight -= 1
        run(2 * slots)
        t0 = time.perf_counter()
        run(total)
        dt = (time.perf_counter() - t0) / total
        print(f"sdrx_decim_ring ({'fill + ' if fill else ''}submit/retire, {slots} slots, flush {flush}): n={n} samples per block  {dt*1e6:.1f} us/block  "
              f"{n/dt/1e6:.1f} MS/s  ({4*n/dt/1e9:.2f} GB/s over PCIe)")
