// TEST INFRASTRUCTURE ONLY.  The reference's float half-band decimators behind a C ABI:
// DecimatorsFI (sdrbase/dsp/decimatorsfi.h, the AirspyHF thread's member), DecimatorsFF (decimatorsff.h) and
// DecimatorsIF<qint16, {8,12,16}> (decimatorsif.h), all over IntHalfbandFilterEOF<64>.  Compiled from the sources
// where they lie into oracle/_ref/libsdrref.so; validates oracle/sdro_fdecim.c and generates tests/golden fixtures.
#include <stdint.h>
#include "dsp/dsptypes.h"
#include "dsp/decimatorsfi.h"
#include "dsp/decimatorsff.h"
#include "dsp/decimatorsif.h"

namespace {
struct Holder {
    int in_kind, out_kind, bits;
    DecimatorsFI fi;
    DecimatorsFF ff;
    DecimatorsIF<qint16, 8> if8;
    DecimatorsIF<qint16, 12> if12;
    DecimatorsIF<qint16, 16> if16;
};

#define CALL(obj, K) do { if (fcpos == 0) (obj).decimate##K##_inf(&it, buf, n); else if (fcpos == 1) (obj).decimate##K##_sup(&it, buf, n); else (obj).decimate##K##_cen(&it, buf, n); } while (0)
#define DISPATCH(obj) switch (log2) { case 0: (obj).decimate1(&it, buf, n); break; case 1: CALL(obj, 2); break; case 2: CALL(obj, 4); break; \
    case 3: CALL(obj, 8); break; case 4: CALL(obj, 16); break; case 5: CALL(obj, 32); break; case 6: CALL(obj, 64); break; }
}

extern "C" {
void* ref_fdecim_new(int in_kind, int out_kind, int bits) { Holder* h = new Holder; h->in_kind = in_kind; h->out_kind = out_kind; h->bits = bits; return h; }
void ref_fdecim_free(void* p) { delete static_cast<Holder*>(p); }
int ref_fdecim_process(void* p, int log2, int fcpos, const void* in, int32_t n, void* out)
{
    Holder& h = *static_cast<Holder*>(p);
    if (h.in_kind == 0 && h.out_kind == 0) {
        const float* buf = static_cast<const float*>(in);
        SampleVector v(n / 2 + 8); SampleVector::iterator it = v.begin();
        DISPATCH(h.fi)
        const int cnt = (int)(it - v.begin()); int16_t* o = static_cast<int16_t*>(out);
        for (int i = 0; i < cnt; i++) { o[2*i] = v[i].real(); o[2*i+1] = v[i].imag(); }
        return cnt;
    }
    FSampleVector v(n / 2 + 8); FSampleVector::iterator it = v.begin();
    if (h.in_kind == 0) { const float* buf = static_cast<const float*>(in); DISPATCH(h.ff) }
    else {
        const qint16* buf = static_cast<const qint16*>(in);
        if (h.bits == 8) { DISPATCH(h.if8) } else if (h.bits == 12) { DISPATCH(h.if12) } else { DISPATCH(h.if16) }
    }
    const int cnt = (int)(it - v.begin()); float* o = static_cast<float*>(out);
    for (int i = 0; i < cnt; i++) { o[2*i] = v[i].real(); o[2*i+1] = v[i].imag(); }
    return cnt;
}
}
