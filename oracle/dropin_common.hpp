// TEST INFRASTRUCTURE ONLY: helpers shared by the translation units of the drop-in harness (dropin_test*.cpp).
// decimators.h and decimatorsu.h of the reference both define decimation_shifts<>, so they cannot meet in one TU.
#ifndef SDRX_DROPIN_COMMON_HPP
#define SDRX_DROPIN_COMMON_HPP
#include <stdint.h>
#include <stdio.h>
#include <vector>

extern uint32_t dropin_rng_state;
extern int dropin_fails;
inline uint32_t rng() { dropin_rng_state = dropin_rng_state * 1664525u + 1013904223u; return dropin_rng_state >> 8; }
inline void report(const char* what, bool ok, long n) { printf("%-62s %s  (%ld)\n", what, ok ? "OK" : "MISMATCH", n); if (!ok) dropin_fails++; }

template <typename T, typename RefFn, typename GpuFn>
void producer_case(const char* name, RefFn ref_fn, GpuFn gpu_fn, int blocks, int block_len, int lo, int span)
{
    SampleVector refOut((size_t) blocks * block_len), gpuOut((size_t) blocks * block_len);
    SampleVector::iterator itR = refOut.begin(), itG = gpuOut.begin();
    std::vector<T> buf((size_t) block_len);
    for (int b = 0; b < blocks; b++) {
        const int len = (b % 3 == 1) ? block_len - 6 : block_len;           // ragged block: the tail is dropped, not carried
        for (int i = 0; i < len; i++) buf[i] = (T)((int)(rng() % span) + lo);
        ref_fn(&itR, buf.data(), len);
        gpu_fn(&itG, buf.data(), len);
    }
    bool same = (itR - refOut.begin()) == (itG - gpuOut.begin());
    const long n = (long)(itR - refOut.begin());
    for (long i = 0; same && i < n; i++) same = refOut[i].real() == gpuOut[i].real() && refOut[i].imag() == gpuOut[i].imag();
    report(name, same && n > 0, n);
}

#define PRODUCER(REFT, GPUT, ELEM, METHOD, LO, SPAN)                                                            \
    {                                                                                                           \
        REFT ref; GPUT gpu(device);                                                                             \
        producer_case<ELEM>(#REFT "::" #METHOD,                                                                  \
                      [&](SampleVector::iterator* it, const ELEM* b, qint32 len) { ref.METHOD(it, b, len); },   \
                      [&](SampleVector::iterator* it, const ELEM* b, qint32 len) { gpu.METHOD(it, b, len); },   \
                      7, 65536, LO, SPAN);                                                                      \
    }

#endif
