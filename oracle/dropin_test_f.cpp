// TEST INFRASTRUCTURE ONLY.  Drop-in harness, float decimators (DecimatorsFI / FF / IF) next to their sdrx:: mirrors.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "dsp/dsptypes.h"
#include "dsp/decimatorsfi.h"
#include "dsp/decimatorsff.h"
#include "dsp/decimatorsif.h"
#define SDRX_HOST_SAMPLE ::Sample
#define SDRX_HOST_FSAMPLE ::FSample
#include "sdrx/dsp.hpp"
#include "dropin_common.hpp"

namespace {
template <typename OutVec, typename T, typename RefFn, typename GpuFn>
void fcase(const char* name, RefFn ref_fn, GpuFn gpu_fn, bool int_in, int span)
{
    const int blocks = 5, block_len = 65536;
    OutVec refOut((size_t) blocks * block_len / 2 + 8), gpuOut((size_t) blocks * block_len / 2 + 8);
    typename OutVec::iterator itR = refOut.begin(), itG = gpuOut.begin();
    std::vector<T> buf((size_t) block_len);
    for (int b = 0; b < blocks; b++) {
        const int len = (b % 3 == 1) ? block_len - 10 : block_len;          // ragged block: the tail is dropped, not carried
        for (int i = 0; i < len; i++) buf[i] = int_in ? (T)((int)(rng() % span) - span / 2) : (T)(((int)(rng() % 60000) - 30000) / 32768.0f);
        ref_fn(&itR, buf.data(), len);
        gpu_fn(&itG, buf.data(), len);
    }
    const long n = (long)(itR - refOut.begin());
    const bool same = n == (long)(itG - gpuOut.begin()) && n > 0 && memcmp(&refOut[0], &gpuOut[0], (size_t) n * sizeof(refOut[0])) == 0;
    report(name, same, n);
}
}

#define FCASE(REFT, GPUT, OUTV, ELEM, METHOD, INT_IN, SPAN)                                                            \
    {                                                                                                                  \
        REFT ref; GPUT gpu(device);                                                                                    \
        fcase<OUTV, ELEM>(#REFT "::" #METHOD,                                                                           \
              [&](OUTV::iterator* it, const ELEM* b, qint32 len) { ref.METHOD(it, b, len); },                          \
              [&](OUTV::iterator* it, const ELEM* b, qint32 len) { gpu.METHOD(it, b, len); }, INT_IN, SPAN);           \
    }

void producer_side_f(int device)
{
    typedef DecimatorsIF<qint16, 12> RefIF12;
    typedef sdrx::DecimatorsIF<qint16, 12> GpuIF12;
    FCASE(DecimatorsFI, sdrx::DecimatorsFI, SampleVector, float, decimate64_cen, false, 0)     // AirspyHF thread (airspyhfthread.cpp:129)
    FCASE(DecimatorsFI, sdrx::DecimatorsFI, SampleVector, float, decimate8_cen, false, 0)
    FCASE(DecimatorsFI, sdrx::DecimatorsFI, SampleVector, float, decimate16_sup, false, 0)
    FCASE(DecimatorsFI, sdrx::DecimatorsFI, SampleVector, float, decimate2_inf, false, 0)
    FCASE(DecimatorsFI, sdrx::DecimatorsFI, SampleVector, float, decimate1, false, 0)
    FCASE(DecimatorsFF, sdrx::DecimatorsFF, FSampleVector, float, decimate32_cen, false, 0)
    FCASE(DecimatorsFF, sdrx::DecimatorsFF, FSampleVector, float, decimate8_inf, false, 0)
    FCASE(RefIF12, GpuIF12, FSampleVector, qint16, decimate16_cen, true, 4096)
    FCASE(RefIF12, GpuIF12, FSampleVector, qint16, decimate64_sup, true, 4096)
    FCASE(RefIF12, GpuIF12, FSampleVector, qint16, decimate4_inf, true, 4096)
    {
        // ONE DecimatorsFI object, the AirspyHF thread changes its decimation while running: the cascades share the object's filters
        DecimatorsFI ref; sdrx::DecimatorsFI gpu(device);
        int call = 0;
        fcase<SampleVector, float>("DecimatorsFI: one object, K / fcPos changed at run time",
              [&](SampleVector::iterator* it, const float* b, qint32 len) {
                  switch (call) { case 0: ref.decimate64_cen(it, b, len); break; case 1: ref.decimate8_inf(it, b, len); break;
                                  case 2: ref.decimate64_cen(it, b, len); break; case 3: ref.decimate2_sup(it, b, len); break;
                                  default: ref.decimate16_sup(it, b, len); break; } },
              [&](SampleVector::iterator* it, const float* b, qint32 len) {
                  switch (call) { case 0: gpu.decimate64_cen(it, b, len); break; case 1: gpu.decimate8_inf(it, b, len); break;
                                  case 2: gpu.decimate64_cen(it, b, len); break; case 3: gpu.decimate2_sup(it, b, len); break;
                                  default: gpu.decimate16_sup(it, b, len); break; }
                  call++; },
              false, 0);
    }
}
