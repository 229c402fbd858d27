// TEST INFRASTRUCTURE ONLY.  Reference DecimatorsU<qint32, quint8, 16, 8, 127> (RTL-SDR thread,
// plugins/samplesource/rtlsdr/rtlsdrthread.h:55; sdrbase/dsp/decimatorsu.h) behind a C ABI.  Own translation
// unit because decimatorsu.h and decimators.h both define `decimation_shifts`.
#include <stdint.h>
#include "dsp/dsptypes.h"
#include "dsp/decimatorsu.h"

typedef DecimatorsU<qint32, quint8, SDR_RX_SAMP_SZ, 8, 127> DecU;

extern "C" {
void* ref_decimu_new() { return new DecU; }
void ref_decimu_free(void* h) { delete static_cast<DecU*>(h); }
int ref_decimu_process(void* h, int log2, int fcpos, const uint8_t* buf, int32_t len, int16_t* out)
{
    DecU& d = *static_cast<DecU*>(h);
    SampleVector v(len / 2 + 8);
    SampleVector::iterator it = v.begin();
#define CASE(K, L) case L: if (fcpos == 0) d.decimate##K##_inf(&it, buf, len); else if (fcpos == 1) d.decimate##K##_sup(&it, buf, len); else d.decimate##K##_cen(&it, buf, len); break;
    switch (log2) {
    case 0: d.decimate1(&it, buf, len); break;
    CASE(2, 1) CASE(4, 2) CASE(8, 3) CASE(16, 4) CASE(32, 5) CASE(64, 6)
    }
#undef CASE
    int n = (int)(it - v.begin());
    for (int i = 0; i < n; i++) { out[2*i] = v[i].real(); out[2*i+1] = v[i].imag(); }
    return n;
}
}
