// TEST INFRASTRUCTURE.  The 24-bit sample build of the reference (-DSDR_RX_SAMPLE_24BIT: FixReal = qint32, Sample = 8 bytes,
// IntHalfbandFilterEO<qint64,qint64,N> everywhere -- dsptypes.h:24-34, decimators.h:326-333, downchannelizer.h:78-81), compiled
// from the same headers into a library of its own (_ref/libsdrref24.so): Decimators<qint32,qint16,24,{8,12,16}> and the
// DownChannelizer stage chain driven the way DownChannelizer::feed drives it (downchannelizer.cpp:50-91).
#ifndef SDR_RX_SAMPLE_24BIT
#error "compile with -DSDR_RX_SAMPLE_24BIT"
#endif
#include <stdint.h>
#include <vector>
#include "dsp/dsptypes.h"
#include "dsp/decimators.h"
#include "dsp/inthalfbandfiltereo.h"

namespace {

struct DecimBase {
    virtual ~DecimBase() {}
    virtual void run(int log2, int fcpos, SampleVector::iterator* it, const qint16* buf, qint32 len) = 0;
};

template<uint InputBits>
struct DecimImpl : DecimBase {
    Decimators<qint32, qint16, SDR_RX_SAMP_SZ, InputBits> d;
    void run(int log2, int fcpos, SampleVector::iterator* it, const qint16* buf, qint32 len)
    {
        if (log2 == 0) { d.decimate1(it, buf, len); return; }
#define SDRX_CASES(sfx) switch (log2) { case 1: d.decimate2_##sfx(it, buf, len); break; case 2: d.decimate4_##sfx(it, buf, len); break; \
        case 3: d.decimate8_##sfx(it, buf, len); break; case 4: d.decimate16_##sfx(it, buf, len); break; \
        case 5: d.decimate32_##sfx(it, buf, len); break; case 6: d.decimate64_##sfx(it, buf, len); break; }
        if (fcpos == 0) { SDRX_CASES(inf) } else if (fcpos == 1) { SDRX_CASES(sup) } else { SDRX_CASES(cen) }
#undef SDRX_CASES
    }
};

typedef IntHalfbandFilterEO<qint64, qint64, 48> ChanStage;           // DownChannelizer::FilterStage::m_filter of this build
typedef bool (ChanStage::*ChanWork)(Sample*);
struct ChanChain {
    std::vector<ChanStage*> stages; std::vector<ChanWork> work;
    ~ChanChain() { for (size_t i = 0; i < stages.size(); i++) delete stages[i]; }
};

} // namespace

extern "C" {

int ref24_sample_bytes() { return (int) sizeof(Sample); }

void* ref24_decim_new(int input_bits)
{
    switch (input_bits) {
    case 8:  return static_cast<DecimBase*>(new DecimImpl<8>());
    case 12: return static_cast<DecimBase*>(new DecimImpl<12>());
    case 16: return static_cast<DecimBase*>(new DecimImpl<16>());
    }
    return 0;
}
void ref24_decim_free(void* h) { delete static_cast<DecimBase*>(h); }

// out: 2 x int32 per complex sample
int ref24_decim_process(void* h, int log2, int fcpos, const int16_t* buf, int32_t len, int32_t* out)
{
    SampleVector v(len / 2 + 8);
    SampleVector::iterator it = v.begin();
    static_cast<DecimBase*>(h)->run(log2, fcpos, &it, buf, len);
    const int n = (int)(it - v.begin());
    for (int i = 0; i < n; i++) { out[2*i] = v[i].real(); out[2*i+1] = v[i].imag(); }
    return n;
}

void* ref24_chain_new(int n_stages, const uint8_t* modes)
{
    ChanChain* c = new ChanChain;
    for (int i = 0; i < n_stages; i++) {
        c->stages.push_back(new ChanStage);
        ChanWork w;                                                   // the Sample* overloads -- the ones FilterStage binds
        if (modes[i] == 0) w = &ChanStage::workDecimateCenter;
        else if (modes[i] == 1) w = &ChanStage::workDecimateLowerHalf;
        else w = &ChanStage::workDecimateUpperHalf;
        c->work.push_back(w);
    }
    return c;
}
void ref24_chain_free(void* h) { delete static_cast<ChanChain*>(h); }

int64_t ref24_chain_feed(void* h, const int32_t* iq, int64_t n_cplx, int32_t* out)
{
    ChanChain* c = static_cast<ChanChain*>(h);
    const size_t ns = c->stages.size();
    int64_t n_out = 0;
    for (int64_t i = 0; i < n_cplx; i++) {
        Sample s(iq[2*i], iq[2*i+1]);
        size_t k = 0;
        for (; k < ns; k++) if (!((c->stages[k])->*(c->work[k]))(&s)) break;
        if (k == ns) {
            s.m_real /= (1 << ns);                                     // downchannelizer.cpp:80-81
            s.m_imag /= (1 << ns);
            out[2*n_out] = s.real(); out[2*n_out+1] = s.imag(); n_out++;
        }
    }
    return n_out;
}

}
