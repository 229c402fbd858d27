// Rate of the pinned-ring host path (sdrx_decim_ring_*) driven from C++, the way a device thread would call it:
// acquire a slot, (optionally) fill it, submit, retire the oldest when the ring is nearly full.
//   g++ -O2 -std=c++17 -Iinclude tools/host_ring_rate.cpp -Lsdrangel_amd -lsdrx -Wl,-rpath,$PWD/sdrangel_amd -o tools/host_ring_rate
#include "sdrx.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double run(sdrx_decim_t* h, int n, int slots, long total, const int16_t* fill)
{
    long inflight = 0;
    const int16_t* out; int32_t no;
    auto t0 = std::chrono::steady_clock::now();
    for (long k = 0; k < total; k++) {
        if (inflight == slots - 1) { if (sdrx_decim_ring_retire(h, &out, &no)) return -1; inflight--; }
        void* s = sdrx_decim_ring_acquire(h);
        if (!s) return -1;
        if (fill) std::memcpy(s, fill, (size_t)n * 4);
        if (sdrx_decim_ring_submit(h, 2 * n)) return -1;
        inflight++;
    }
    while (inflight) { if (sdrx_decim_ring_retire(h, &out, &no)) return -1; inflight--; }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (double)total;
}

int main()
{
    struct Cfg { int n, slots, flush; } cfgs[] = { { 32768, 64, 1 }, { 32768, 64, 8 }, { 32768, 64, 16 }, { 32768, 128, 32 },
                                                   { 1 << 20, 8, 1 }, { 1 << 20, 8, 2 }, { 1 << 22, 6, 1 } };
    for (const Cfg& c : cfgs) {
        std::vector<int16_t> x((size_t)c.n * 2);
        for (size_t i = 0; i < x.size(); i++) x[i] = (int16_t)((i * 2654435761u >> 20) & 0xfff) - 2048;
        for (int fill = 0; fill < 2; fill++) {
            sdrx_decim_t* h = nullptr;
            if (sdrx_decim_create(&h, 0, 6, SDRX_FC_CEN, 12)) { std::printf("create: %s\n", sdrx_last_error()); return 1; }
            if (sdrx_decim_ring_create(h, 2 * c.n, c.slots, c.flush)) { std::printf("ring: %s\n", sdrx_last_error()); return 1; }
            long total = (1L << 29) / c.n; if (total > 8192) total = 8192; if (total < 4 * c.slots) total = 4 * c.slots;
            run(h, c.n, c.slots, 2 * c.slots, fill ? x.data() : nullptr);
            const double dt = run(h, c.n, c.slots, total, fill ? x.data() : nullptr);
            if (dt < 0) { std::printf("error: %s\n", sdrx_last_error()); return 1; }
            std::printf("sdrx_decim_ring decimate64_cen (C++ caller%s, %d slots, flush %d): %d samples per block  %.2f us/block  %.1f MS/s  (%.2f GB/s over PCIe)\n",
                        fill ? ", memcpy into the slot" : "", c.slots, c.flush, c.n, dt * 1e6, c.n / dt / 1e6, 4.0 * c.n / dt / 1e9);
            sdrx_decim_destroy(h);
        }
    }
    return 0;
}
