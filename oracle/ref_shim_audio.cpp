// TEST INFRASTRUCTURE.  The audio-rate tails of NFMDemod::feed (plugins/channelrx/demodnfm/nfmdemod.cpp:150-300) and
// SSBDemod::feed (plugins/channelrx/demodssb/ssbdemod.cpp:181-250) on the reference's OWN member classes --
// PhaseDiscriminators, MovingAverageUtil<Real,double,32>, DoubleBufferFIFO, Bandpass<Real>, MagAGC (sdrbase/dsp/agc.cpp
// compiled where it lies).  The demod plugin objects themselves cannot be instantiated outside the application (they
// attach to a DeviceSourceAPI, an audio device manager and a Qt thread), so the loop bodies are restated around the real
// members, with the default switches (m_deltaSquelch, m_ctcssOn, m_audioMute off; mono, not binaural).
#include <complex>
#include <cmath>
#include <stdint.h>
#include "dsp/dsptypes.h"
#include "dsp/phasediscri.h"
#include "dsp/bandpass.h"
#include "dsp/agc.h"
#include "util/movingaverage.h"
#include "util/doublebufferfifo.h"

namespace {

struct NfmTail {
    PhaseDiscriminators m_phaseDiscri;
    MovingAverageUtil<Real, double, 32> m_movingAverage;
    DoubleBufferFIFO<Real> m_squelchDelayLine;
    Bandpass<Real> m_bandpass;
    int m_squelchCount, m_squelchGate; Real m_squelchLevel; float m_discriCompensation; Real m_volume;
    // m_prevArg of PhaseDiscriminators has no initialiser in the reference (phasediscri.h:139, reset() leaves it alone): the
    // first deviation of a fresh demod is undefined there; value-initialisation pins it to 0 here, like the oracle and the GPU
    NfmTail() : m_phaseDiscri(), m_squelchDelayLine(24000), m_squelchCount(0) {}
};

struct SsbTail {
    MagAGC m_agc;
    DoubleBufferFIFO<std::complex<float> > m_squelchDelayLine;
    bool m_agcActive; Real m_volume;
    SsbTail() : m_agc(12000, 3276.8, 1e-2), m_squelchDelayLine(2 * 48000) {}
};

} // namespace

extern "C" {

void* ref_nfmtail_new(int32_t audio_rate, float fm_scaling, float squelch_level, int32_t squelch_gate, float volume, float af_bandwidth)
{
    NfmTail* t = new NfmTail;
    t->m_phaseDiscri.setFMScaling(fm_scaling);
    t->m_squelchLevel = squelch_level; t->m_squelchGate = squelch_gate; t->m_volume = volume;
    t->m_discriCompensation = (audio_rate / 48000.0f);
    t->m_discriCompensation *= sqrt(t->m_discriCompensation);
    t->m_bandpass.create(301, audio_rate, 300.0, af_bandwidth);
    for (int i = 0; i < 24000; i++) t->m_squelchDelayLine.write(0);            // new T[] leaves the line uninitialised; the demod runs long before it matters
    return t;
}
void ref_nfmtail_free(void* h) { delete static_cast<NfmTail*>(h); }

void ref_nfmtail_process(void* h, const float* ci_iq, int64_t n, int16_t* audio)
{
    NfmTail& d = *static_cast<NfmTail*>(h);
    for (int64_t k = 0; k < n; k++) {
        Complex ci(ci_iq[2 * k], ci_iq[2 * k + 1]);
        qint16 sample;
        double magsqRaw;
        Real deviation;
        Real demod = d.m_phaseDiscri.phaseDiscriminatorDelta(ci, magsqRaw, deviation);
        Real magsq = magsqRaw / (SDR_RX_SCALED*SDR_RX_SCALED);
        d.m_movingAverage(magsq);
        if ((Real) d.m_movingAverage < d.m_squelchLevel) {
            d.m_squelchDelayLine.write(0);
            if (d.m_squelchCount > 0) d.m_squelchCount--;
        } else {
            d.m_squelchDelayLine.write(demod * d.m_discriCompensation);
            if (d.m_squelchCount < 2*d.m_squelchGate) d.m_squelchCount++;
        }
        const bool squelchOpen = (d.m_squelchCount > d.m_squelchGate);
        if (squelchOpen) sample = d.m_bandpass.filter(d.m_squelchDelayLine.readBack(d.m_squelchGate)) * d.m_volume;
        else sample = 0;
        audio[k] = sample;
    }
}

void* ref_ssbtail_new(int32_t agc_active, int32_t agc_nb_samples, double agc_threshold, int32_t agc_threshold_enable,
                      int32_t agc_gate, int32_t agc_clamping, float volume)
{
    SsbTail* t = new SsbTail;
    t->m_agc.setClampMax(SDR_RX_SCALED/100.0);                                // ssbdemod.cpp:88-89
    t->m_agc.setClamping(agc_clamping != 0);
    t->m_agc.resize(agc_nb_samples, agc_nb_samples/2, 3276.8);                // :413-414 (agcTarget)
    t->m_agc.setStepDownDelay(agc_nb_samples);
    t->m_agc.setThresholdEnable(agc_threshold_enable != 0);                   // :503-525
    t->m_agc.setThreshold(agc_threshold);
    t->m_agc.setGate(agc_gate);
    t->m_agcActive = agc_active != 0; t->m_volume = volume;
    for (int i = 0; i < 2 * 48000; i++) t->m_squelchDelayLine.write(std::complex<float>(0, 0));
    return t;
}
void ref_ssbtail_free(void* h) { delete static_cast<SsbTail*>(h); }

void ref_ssbtail_process(void* h, const float* sb, int64_t n, int16_t* audio)
{
    SsbTail& d = *static_cast<SsbTail*>(h);
    for (int64_t k = 0; k < n; k++) {
        std::complex<float> sideband(sb[2 * k], sb[2 * k + 1]);
        float agcVal = d.m_agcActive ? d.m_agc.feedAndGetValue(sideband) : 10.0;
        std::complex<float>& delayedSample = d.m_squelchDelayLine.readBack(d.m_agc.getStepDownDelay());
        std::complex<float> delayed = delayedSample;                           // the reference keeps a reference; write() below may alias it
        d.m_squelchDelayLine.write(sideband*agcVal);
        std::complex<float> z = delayedSample * d.m_agc.getStepValue();
        (void) delayed;
        Real demod = (z.real() + z.imag()) * 0.7;
        qint16 sample = (qint16)(demod * d.m_volume);
        audio[k] = sample;
    }
}

}


// IIRFilter<float, Order> (sdrbase/dsp/iirfilter.h): the reference's own template, instantiated for the orders offered
#include "dsp/iirfilter.h"
namespace {
struct IirBase { virtual ~IirBase() {} virtual float run(float s) = 0; };
template<uint32_t O> struct IirImpl : IirBase { IIRFilter<float, O> f; IirImpl(const float* a, const float* b) : f(a, b) {} virtual float run(float s) { return f.run(s); } };
}
extern "C" {
void* ref_iir_new(int32_t order, const float* a, const float* b)
{
    switch (order) {
    case 2: return static_cast<IirBase*>(new IirImpl<2>(a, b));
    case 3: return static_cast<IirBase*>(new IirImpl<3>(a, b));
    case 4: return static_cast<IirBase*>(new IirImpl<4>(a, b));
    case 5: return static_cast<IirBase*>(new IirImpl<5>(a, b));
    case 6: return static_cast<IirBase*>(new IirImpl<6>(a, b));
    case 7: return static_cast<IirBase*>(new IirImpl<7>(a, b));
    case 8: return static_cast<IirBase*>(new IirImpl<8>(a, b));
    }
    return 0;
}
void ref_iir_free(void* h) { delete static_cast<IirBase*>(h); }
void ref_iir_run(void* h, const float* in, int64_t n, float* out) { IirBase* f = static_cast<IirBase*>(h); for (int64_t k = 0; k < n; k++) out[k] = f->run(in[k]); }
}
