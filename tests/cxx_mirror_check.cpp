// Compile-and-link check of the C++ mirror (include/sdrx/dsp.hpp) written the way a reference device
// thread uses the real classes (limesdrinputthread.cpp:103-135).  Run: exits 0 on success; with a GPU
// it also pushes one block through decimate64_cen and a 2-channel bank.
#include <cstdio>
#include <vector>
#include "sdrx/dsp.hpp"

int main()
{
    sdrx::SampleSinkFifo fifo(1024);
    sdrx::SampleVector conv(512);
    std::vector<int16_t> buf(2 * 512);
    for (size_t i = 0; i < buf.size(); i++) buf[i] = (int16_t)((i * 37) % 4096 - 2048);
    if (sdrx_device_count() == 0) {
        // host-only part: FIFO round trip
        sdrx::SampleVector v(10, sdrx::Sample(3, -3));
        if (fifo.write(v.begin(), v.end()) != 10 || fifo.fill() != 10) return 2;
        sdrx::SampleVector r(4);
        if (fifo.read(r.begin(), r.end()) != 4 || r[0].real() != 3 || r[0].imag() != -3) return 3;
        std::puts("cxx mirror: host-only checks ok (no GPU)");
        return 0;
    }
    sdrx::Decimators<int32_t, int16_t, 16, 12> dec;
    sdrx::SampleVector::iterator it = conv.begin();
    dec.decimate64_cen(&it, buf.data(), (int32_t)buf.size());
    if (it - conv.begin() != 8) return 4;
    fifo.write(conv.begin(), it);
    sdrx::DownChannelizerBank bank(2400000, { 48000, 48000 }, { 0, 300000 });
    if (!bank.ok()) return 5;
    sdrx::SampleVector in(4096, sdrx::Sample(100, -100)), out;
    bank.feed(in.begin(), in.end(), false);
    bank.pull(0, out);
    std::printf("cxx mirror: decimated %ld samples, channel 0 produced %zu at %d S/s\n", (long)(it - conv.begin()), out.size(), bank.getOutputSampleRate(0));
    return out.empty() ? 6 : 0;
}
