// TEST INFRASTRUCTURE ONLY.  Thin C-ABI wrapper that *instantiates the reference's own
// classes* (compiled from the sources where they lie under /root/reference -- nothing is
// copied into this repo).  Built by oracle/Makefile into oracle/_ref/libsdrref.so.
// Used for (1) pinning oracle/sdro.c against the real reference, (2) generating
// tests/golden/*.bin, (3) optionally the "reference" CPU baseline in bench.py.
//
// Header-only / Qt-link-free part of the hot path:
//   Decimators<qint32,qint16,16,{8,12,16}>      sdrbase/dsp/decimators.h:279-341
//   IntHalfbandFilterEO<qint32,qint32,48>       sdrbase/dsp/inthalfbandfiltereo.h:31-934
//      (the stage type DownChannelizer instantiates, downchannelizer.h:83-84)
//   NCO, Interpolator, fftfilt, g_fft, PhaseDiscriminators  (float back-end)
// Only Qt *headers* (qint16 & friends from QtGlobal) are needed; no Qt library is linked.
#include <stdint.h>
#include <string.h>
#include <vector>
#include <complex>

#include "dsp/dsptypes.h"
#include "dsp/decimators.h"
#include "dsp/inthalfbandfiltereo.h"
#include "dsp/nco.h"
#include "dsp/interpolator.h"
#include "dsp/fftfilt.h"
#include "dsp/phasediscri.h"
#include "dsp/lowpass.h"
#include "dsp/bandpass.h"

namespace {

struct DecimBase {
    virtual ~DecimBase() {}
    virtual void run(int log2, int fcpos, SampleVector::iterator* it, const qint16* buf, qint32 len) = 0;
    SampleVector keep;            // ref_decim_run: the output vector of the bench loop, allocated once (sdrbench/mainbench.cpp:83-104)
};

// fcpos: 0 = infradyne (_inf), 1 = supradyne (_sup), 2 = centre (_cen) -- same coding as the
// device plugins' m_fcPos (e.g. limesdrinputthread.cpp:103-135).
template<uint InputBits>
struct DecimImpl : DecimBase {
    Decimators<qint32, qint16, SDR_RX_SAMP_SZ, InputBits> d;
    void run(int log2, int fcpos, SampleVector::iterator* it, const qint16* buf, qint32 len) override
    {
        if (log2 == 0) { d.decimate1(it, buf, len); return; }
        switch (fcpos) {
        case 0:
            switch (log2) {
            case 1: d.decimate2_inf(it, buf, len); break;
            case 2: d.decimate4_inf(it, buf, len); break;
            case 3: d.decimate8_inf(it, buf, len); break;
            case 4: d.decimate16_inf(it, buf, len); break;
            case 5: d.decimate32_inf(it, buf, len); break;
            case 6: d.decimate64_inf(it, buf, len); break;
            }
            break;
        case 1:
            switch (log2) {
            case 1: d.decimate2_sup(it, buf, len); break;
            case 2: d.decimate4_sup(it, buf, len); break;
            case 3: d.decimate8_sup(it, buf, len); break;
            case 4: d.decimate16_sup(it, buf, len); break;
            case 5: d.decimate32_sup(it, buf, len); break;
            case 6: d.decimate64_sup(it, buf, len); break;
            }
            break;
        default:
            switch (log2) {
            case 1: d.decimate2_cen(it, buf, len); break;
            case 2: d.decimate4_cen(it, buf, len); break;
            case 3: d.decimate8_cen(it, buf, len); break;
            case 4: d.decimate16_cen(it, buf, len); break;
            case 5: d.decimate32_cen(it, buf, len); break;
            case 6: d.decimate64_cen(it, buf, len); break;
            }
        }
    }
};

typedef IntHalfbandFilterEO<qint32, qint32, 48> ChanStage;
typedef bool (ChanStage::*ChanWork)(Sample*);

struct ChanChain {
    std::vector<ChanStage*> stages;
    std::vector<ChanWork> work;
    ~ChanChain() { for (auto* s : stages) delete s; }
};

struct BackEnd {
    NCO nco;
    Interpolator interp;
    Real distance;
    Real step;
};

} // namespace

extern "C" {

void* ref_decim_new(int input_bits)
{
    switch (input_bits) {
    case 8:  return static_cast<DecimBase*>(new DecimImpl<8>());
    case 12: return static_cast<DecimBase*>(new DecimImpl<12>());
    case 16: return static_cast<DecimBase*>(new DecimImpl<16>());
    }
    return 0;
}

void ref_decim_free(void* h) { delete static_cast<DecimBase*>(h); }

// returns number of complex outputs written to out (int16 re,im interleaved); out must hold len/2 cplx
int ref_decim_process(void* h, int log2, int fcpos, const int16_t* buf, int32_t len, int16_t* out)
{
    SampleVector v(len / 2 + 8);
    SampleVector::iterator it = v.begin();
    static_cast<DecimBase*>(h)->run(log2, fcpos, &it, buf, len);
    int n = (int)(it - v.begin());
    for (int i = 0; i < n; i++) { out[2*i] = v[i].real(); out[2*i+1] = v[i].imag(); }
    return n;
}

// The call as sdrangelbench times it (sdrbench/mainbench.cpp:83-104): the output SampleVector lives across calls and the
// result stays in it -- no allocation and no copy-out inside the timed loop.  Returns the number of outputs.
int ref_decim_run(void* h, int log2, int fcpos, const int16_t* buf, int32_t len)
{
    DecimBase* d = static_cast<DecimBase*>(h);
    if ((int)d->keep.size() < len / 2 + 8) d->keep.resize((size_t)(len / 2 + 8));
    SampleVector::iterator it = d->keep.begin();
    d->run(log2, fcpos, &it, buf, len);
    return (int)(it - d->keep.begin());
}

// modes[i]: 0 = centre, 1 = lower half, 2 = upper half (DownChannelizer::FilterStage::Mode order,
// downchannelizer.h:76-80)
void* ref_chain_new(int n_stages, const uint8_t* modes)
{
    ChanChain* c = new ChanChain;
    for (int i = 0; i < n_stages; i++) {
        c->stages.push_back(new ChanStage);
        ChanWork w;   // the Sample* overloads -- the ones FilterStage binds (downchannelizer.cpp:214-226)
        if (modes[i] == 0) w = &ChanStage::workDecimateCenter;
        else if (modes[i] == 1) w = &ChanStage::workDecimateLowerHalf;
        else w = &ChanStage::workDecimateUpperHalf;
        c->work.push_back(w);
    }
    return c;
}

void ref_chain_free(void* h) { delete static_cast<ChanChain*>(h); }

// The per-sample loop of DownChannelizer::feed (downchannelizer.cpp:65-84) driven over the
// reference's own stage objects; returns #outputs.
int64_t ref_chain_feed(void* h, const int16_t* iq, int64_t n_cplx, int16_t* out)
{
    ChanChain* c = static_cast<ChanChain*>(h);
    const size_t ns = c->stages.size();
    int64_t n_out = 0;
    for (int64_t i = 0; i < n_cplx; i++) {
        Sample s(iq[2*i], iq[2*i+1]);
        size_t k = 0;
        for (; k < ns; k++) {
            if (!((c->stages[k])->*(c->work[k]))(&s)) break;
        }
        if (k == ns) {
            s.m_real /= (1 << ns);
            s.m_imag /= (1 << ns);
            out[2*n_out] = s.real(); out[2*n_out+1] = s.imag();
            n_out++;
        }
    }
    return n_out;
}

// ---------------------------------------------------------------- float back-end
void ref_nco_table(float* tbl4096)
{
    NCO n; n.setFreq(1.0f, 4096.0f); n.setPhase(-1);   // inc = 1 -> walks the table
    for (int i = 0; i < 4096; i++) { Complex c = n.nextIQ(); tbl4096[i] = c.real(); }
}

void ref_nco_run(float freq, float rate, int n, float* out_iq)
{
    NCO nco; nco.setFreq(freq, rate);
    for (int i = 0; i < n; i++) { Complex c = nco.nextIQ(); out_iq[2*i] = c.real(); out_iq[2*i+1] = c.imag(); }
}

// NCO mix + Interpolator::decimate exactly as the channel plugins open their feed()
// (nfmdemod.cpp:150-160): c = Complex(re,im) * nco.nextIQ(); if (interp.decimate(&d, c, &ci)) {...; d += step}
void* ref_backend_new(float nco_freq, float in_rate, float out_rate, int phase_steps, float cutoff, float taps_per_phase)
{
    BackEnd* b = new BackEnd;
    b->nco.setFreq(nco_freq, in_rate);
    b->interp.create(phase_steps, in_rate, cutoff, taps_per_phase);
    b->distance = 0;
    b->step = (Real) in_rate / (Real) out_rate;
    return b;
}
void ref_backend_free(void* h) { delete static_cast<BackEnd*>(h); }

int64_t ref_backend_feed(void* h, const int16_t* iq, int64_t n_cplx, float* out_iq)
{
    BackEnd* b = static_cast<BackEnd*>(h);
    int64_t n_out = 0;
    for (int64_t i = 0; i < n_cplx; i++) {
        Complex c(iq[2*i], iq[2*i+1]);
        c *= b->nco.nextIQ();
        Complex ci;
        if (b->interp.decimate(&b->distance, c, &ci)) {
            out_iq[2*n_out] = ci.real(); out_iq[2*n_out+1] = ci.imag();
            n_out++;
            b->distance += b->step;
        }
    }
    return n_out;
}

// fftfilt overlap-add SSB/complex filter (fftfilt.cpp:261-325)
void* ref_fftfilt_new(float f1, float f2, int len) { return f1 < 0 ? new fftfilt(f2, len) : new fftfilt(f1, f2, len); }
void* ref_fftfilt_new_asym(float fopp, float fin, int len) { fftfilt* f = new fftfilt(fin, len); f->create_asym_filter(fopp, fin); return f; }
void ref_fftfilt_free(void* h) { delete static_cast<fftfilt*>(h); }
// mode 0: runFilt, 1: runSSB usb, 2: runSSB lsb, 3: runDSB, 4: runAsym usb, 5: runAsym lsb
int64_t ref_fftfilt_run(void* h, int mode, const float* in_iq, int64_t n, float* out_iq)
{
    fftfilt* f = static_cast<fftfilt*>(h);
    int64_t n_out = 0;
    for (int64_t i = 0; i < n; i++) {
        fftfilt::cmplx c(in_iq[2*i], in_iq[2*i+1]);
        fftfilt::cmplx* o = 0;
        int r = mode == 0 ? f->runFilt(c, &o) : mode == 1 ? f->runSSB(c, &o, true)
              : mode == 2 ? f->runSSB(c, &o, false) : mode == 3 ? f->runDSB(c, &o) : f->runAsym(c, &o, mode == 4);
        for (int k = 0; k < r; k++) { out_iq[2*n_out] = o[k].real(); out_iq[2*n_out+1] = o[k].imag(); n_out++; }
    }
    return n_out;
}

// g_fft forward / inverse, in place, n complex floats (n = power of two)
void ref_gfft(float* iq, int n, int inverse)
{
    g_fft<float> f(n);
    if (inverse) f.InverseComplexFFT((std::complex<float>*)iq); else f.ComplexFFT((std::complex<float>*)iq);
}

// PhaseDiscriminators::phaseDiscriminatorDelta (phasediscri.h:61-78) / phaseDiscriminator (:50-55)
void ref_discri(int kind, float fm_scaling, const float* in_iq, int64_t n, float* out)
{
    PhaseDiscriminators d = PhaseDiscriminators();   // value-init: m_prevArg has no initialiser in the class
    d.setFMScaling(fm_scaling); d.reset();
    for (int64_t i = 0; i < n; i++) {
        Complex c(in_iq[2*i], in_iq[2*i+1]);
        if (kind == 0) { double magsq; Real fmDev; out[i] = d.phaseDiscriminatorDelta(c, magsq, fmDev); }
        else out[i] = d.phaseDiscriminator(c);
    }
}

// Lowpass<Real> / Bandpass<Real> (lowpass.h, bandpass.h) as NFMDemod uses them (nfmdemod.cpp:88,239,279,428-429)
struct RefFir { int kind; Lowpass<Real> lp; Bandpass<Real> bp; };
void* ref_fir_new(int kind, int ntaps, double rate, double f1, double f2)
{
    RefFir* f = new RefFir; f->kind = kind;
    if (kind == 0) f->lp.create(ntaps, rate, f1); else f->bp.create(ntaps, rate, f1, f2);
    return f;
}
void ref_fir_free(void* h) { delete static_cast<RefFir*>(h); }
void ref_fir_run(void* h, const float* in, int64_t n, float* out)
{
    RefFir* f = static_cast<RefFir*>(h);
    for (int64_t i = 0; i < n; i++) out[i] = f->kind == 0 ? f->lp.filter(in[i]) : f->bp.filter(in[i]);
}

} // extern "C"


// DC offset correction as DSPDeviceSourceEngine::iqCorrections(begin, end, false) runs it (dspdevicesourceengine.cpp:175-181,
// 255-259) on the reference's own MovingAverageUtil<int32_t, int64_t, 1024> members (dspdevicesourceengine.h:106-107).
// The engine class itself (a QThread wired to the device and sink registries) is not instantiated; its four lines are.
#include "util/movingaverage.h"
namespace { struct DcCorr { MovingAverageUtil<int32_t, int64_t, 1024> m_iBeta, m_qBeta; }; }
extern "C" {
void* ref_dccorr_new() { return new DcCorr; }
void ref_dccorr_free(void* h) { delete static_cast<DcCorr*>(h); }
void ref_dccorr_process(void* h, const int16_t* iq, int64_t n_cplx, int16_t* out)
{
    DcCorr& d = *static_cast<DcCorr*>(h);
    SampleVector v((size_t) n_cplx);
    for (int64_t i = 0; i < n_cplx; i++) v[i] = Sample(iq[2*i], iq[2*i+1]);
    for (SampleVector::iterator it = v.begin(); it < v.end(); it++) {
        d.m_iBeta(it->real());
        d.m_qBeta(it->imag());
        it->m_real -= (int32_t) d.m_iBeta;
        it->m_imag -= (int32_t) d.m_qBeta;
    }
    for (int64_t i = 0; i < n_cplx; i++) { out[2*i] = v[i].real(); out[2*i+1] = v[i].imag(); }
}
}


// DC + I/Q imbalance correction as DSPDeviceSourceEngine::iqCorrections(begin, end, true) runs it with IMBALANCE_INT
// undefined (dspdevicesourceengine.cpp:175-181, 217-253): the loop body on the reference's own MovingAverageUtil members
// (dspdevicesourceengine.h:106-107, 120-125).  As for the DC-only case above, the engine class itself (a QThread wired to
// the device and sink registries, iqCorrections private) is not instantiated; the loop's statements are.
namespace { struct IqImb {
    MovingAverageUtil<int32_t, int64_t, 1024> m_iBeta, m_qBeta;
    MovingAverageUtil<float, double, 128> m_avgII, m_avgIQ, m_avgII2, m_avgQQ2;
    MovingAverageUtil<double, double, 128> m_avgPhi, m_avgAmp;
}; }
extern "C" {
void* ref_iqimb_new() { return new IqImb; }
void ref_iqimb_free(void* h) { delete static_cast<IqImb*>(h); }
void ref_iqimb_process(void* h, const int16_t* iq, int64_t n_cplx, int16_t* out)
{
    IqImb& d = *static_cast<IqImb*>(h);
    SampleVector v((size_t) n_cplx);
    for (int64_t i = 0; i < n_cplx; i++) v[i] = Sample(iq[2*i], iq[2*i+1]);
    for (SampleVector::iterator it = v.begin(); it < v.end(); it++) {
        d.m_iBeta(it->real());
        d.m_qBeta(it->imag());
        float xi = (it->m_real - (int32_t) d.m_iBeta) / SDR_RX_SCALEF;
        float xq = (it->m_imag - (int32_t) d.m_qBeta) / SDR_RX_SCALEF;
        d.m_avgII(xi*xi);
        d.m_avgIQ(xi*xq);
        if (d.m_avgII.asDouble() != 0) {
            d.m_avgPhi(d.m_avgIQ.asDouble()/d.m_avgII.asDouble());
        }
        float& yi = xi;
        float yq = xq - d.m_avgPhi.asDouble()*xi;
        d.m_avgII2(yi*yi);
        d.m_avgQQ2(yq*yq);
        if (d.m_avgQQ2.asDouble() != 0) {
            d.m_avgAmp(sqrt(d.m_avgII2.asDouble() / d.m_avgQQ2.asDouble()));
        }
        float& zi = yi;
        float zq = d.m_avgAmp.asDouble() * yq;
        it->m_real = zi * SDR_RX_SCALEF;
        it->m_imag = zq * SDR_RX_SCALEF;
    }
    for (int64_t i = 0; i < n_cplx; i++) { out[2*i] = v[i].real(); out[2*i+1] = v[i].imag(); }
}
}
