/* TEST INFRASTRUCTURE -- CPU restatement of the audio-rate tail of the NFM and SSB demodulators:
 *   NFM : plugins/channelrx/demodnfm/nfmdemod.cpp:150-300 (m_deltaSquelch, m_ctcssOn, m_audioMute all off = the defaults):
 *         phaseDiscriminatorDelta (phasediscri.h:61-78) -> magsq moving average (MovingAverageUtil<Real,double,32>) ->
 *         power squelch with gate counter -> DoubleBufferFIFO<Real>(24000) delay line -> Bandpass<Real>(301, rate, 300, bw)
 *         (called only while the squelch is open) -> * volume -> qint16
 *   SSB : plugins/channelrx/demodssb/ssbdemod.cpp:181-250 (mono, not muted): MagAGC::feedAndGetValue (sdrbase/dsp/agc.cpp:96-175)
 *         -> DoubleBufferFIFO<cmplx>(96000) delay line -> getStepValue -> (re + im) * 0.7 * volume -> qint16
 * Strict IEEE, scalar.  Pinned against the reference's own classes by tests/test_oracle_vs_ref.py (oracle/ref_shim_audio.cpp). */
#include "sdro.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int16_t to_q16(float v)
{
    /* (qint16) of a float on x86-64: cvttss2si to int32 (0x80000000 when out of range or NaN), then the low 16 bits */
    int32_t i = (v >= -2147483648.0f && v < 2147483648.0f) ? (int32_t)v : (int32_t)0x80000000u;
    return (int16_t)i;
}

static float atan2_approx2(float y, float x)               /* phasediscri.h:172-197 */
{
    if (x == 0.0f) { if (y > 0.0f) return 1.5707963f; if (y == 0.0f) return 0.0f; return -1.5707963f; }
    float at; const float z = y / x;
    if (fabsf(z) < 1.0f) {
        at = z / (1.0f + 0.28f * z * z);
        if (x < 0.0f) { if (y < 0.0f) return at - 3.14159265f; return at + 3.14159265f; }
    } else {
        at = 1.5707963f - z / (z * z + 0.28f);
        if (y < 0.0f) return at - 3.14159265f;
    }
    return at;
}

/* ------------------------------------------------------------------ NFM */
struct sdro_nfmtail {
    float prev_arg, fm_scaling, level, volume, comp;
    int gate, count;
    float ma_s[32]; int ma_n; unsigned ma_idx; double ma_total;
    float* dl; int dl_size, dl_w, dl_cur;
    sdro_fir* bp;
};

sdro_nfmtail* sdro_nfmtail_new(int32_t audio_rate, float fm_scaling, float squelch_level, int32_t squelch_gate, float volume, float af_bandwidth)
{
    sdro_nfmtail* t = (sdro_nfmtail*)calloc(1, sizeof *t);
    t->fm_scaling = fm_scaling; t->level = squelch_level; t->gate = squelch_gate; t->volume = volume;
    t->comp = (float)audio_rate / 48000.0f;                /* nfmdemod.cpp:82-83 */
    t->comp *= sqrtf(t->comp);
    t->dl_size = 24000; t->dl = (float*)calloc((size_t)(2 * t->dl_size), sizeof(float));
    t->bp = sdro_fir_new(1, 301, (double)audio_rate, 300.0, (double)af_bandwidth);     /* nfmdemod.cpp:428-429 */
    return t;
}
void sdro_nfmtail_free(sdro_nfmtail* t) { if (t) { free(t->dl); sdro_fir_free(t->bp); free(t); } }

void sdro_nfmtail_process(sdro_nfmtail* t, const float* ci, int64_t n, int16_t* audio)
{
    for (int64_t k = 0; k < n; k++) {
        const float fI = ci[2 * k], fQ = ci[2 * k + 1];
        const double magsq_raw = (double)(fI * fI + fQ * fQ);
        const float cur = atan2_approx2(fQ, fI);
        float dev = (float)((double)(cur - t->prev_arg) / 3.14159265358979323846);
        t->prev_arg = cur;
        if (dev < -1.0f) dev += 2.0f; else if (dev > 1.0f) dev -= 2.0f;
        const float demod = dev * t->fm_scaling;
        const float magsq = (float)(magsq_raw / (32768.0 * 32768.0));
        if (t->ma_n < 32) { t->ma_s[t->ma_n++] = magsq; t->ma_total += magsq; }
        else { const float d = magsq - t->ma_s[t->ma_idx]; t->ma_total += d; t->ma_s[t->ma_idx] = magsq; t->ma_idx = (t->ma_idx + 1) % 32; }
        float w;
        if ((float)(t->ma_total / 32) < t->level) { w = 0.0f; if (t->count > 0) t->count--; }
        else { w = demod * t->comp; if (t->count < 2 * t->gate) t->count++; }
        t->dl[t->dl_w] = w; t->dl[t->dl_w + t->dl_size] = w; t->dl_cur = t->dl_w;          /* DoubleBufferFIFO::write */
        t->dl_w = t->dl_w < t->dl_size - 1 ? t->dl_w + 1 : 0;
        if (t->count > t->gate) {
            const int delay = t->gate > t->dl_size ? t->dl_size : t->gate;
            const float x = t->dl[t->dl_cur + t->dl_size - delay];                       /* readBack(m_squelchGate) */
            float y;
            sdro_fir_run(t->bp, &x, 1, &y);
            audio[k] = to_q16(y * t->volume);
        } else audio[k] = 0;
    }
}

/* ------------------------------------------------------------------ SSB */
struct sdro_ssbtail {
    /* MagAGC */
    double u0, R, magsq, threshold, step_delta, clamp_max;
    double* hist; int hist_n; unsigned hist_idx; double sum;
    int threshold_enable, gate, step_length, step_up, step_down, gate_counter, step_down_delay, clamping, count;
    int agc_active; float volume;
    float* dl; int dl_size, dl_w, dl_cur;
};

static float smootherstep(float x)                          /* util/stepfunctions.h:23-36 */
{
    if (x == 1.0f) return 1.0f; else if (x == 0.0f) return 0.0f;
    const double x3 = x * x * x, x4 = x * x3, x5 = x * x4;
    return (float)(6.0 * x5 - 15.0 * x4 + 10.0 * x3);
}

sdro_ssbtail* sdro_ssbtail_new(int32_t agc_active, int32_t agc_nb_samples, double agc_threshold, int32_t agc_threshold_enable,
                               int32_t agc_gate, int32_t agc_clamping, float volume)
{
    sdro_ssbtail* t = (sdro_ssbtail*)calloc(1, sizeof *t);
    /* MagAGC(12000, agcTarget, 1e-2) then resize(n, n / 2, agcTarget) with R narrowed to Real (agc.cpp:59-68, ssbdemod.cpp:411-414) */
    const float Rf = (float)3276.8;
    t->R = (double)Rf; t->u0 = 1.0;
    t->hist_n = agc_nb_samples; t->hist = (double*)calloc((size_t)agc_nb_samples, sizeof(double));
    t->step_length = agc_nb_samples / 2; t->step_delta = 1.0 / t->step_length;
    t->step_up = 0; t->step_down = t->step_length;
    t->step_down_delay = agc_nb_samples;
    t->threshold = agc_threshold; t->threshold_enable = agc_threshold_enable; t->gate = agc_gate;
    t->clamping = agc_clamping; t->clamp_max = 32768.0 / 100.0;
    t->agc_active = agc_active; t->volume = volume;
    t->dl_size = 2 * 48000; t->dl = (float*)calloc((size_t)(4 * t->dl_size), sizeof(float));
    return t;
}
void sdro_ssbtail_free(sdro_ssbtail* t) { if (t) { free(t->hist); free(t->dl); free(t); } }

static double magagc_feed(sdro_ssbtail* t, float re, float im)      /* MagAGC::feedAndGetValue, m_squared = false */
{
    t->magsq = (double)(re * re + im * im);
    { double* o = &t->hist[t->hist_idx]; t->sum += t->magsq - *o; *o = t->magsq; t->hist_idx = t->hist_idx < (unsigned)t->hist_n - 1 ? t->hist_idx + 1 : 0; }
    const double avg = t->sum / (double)t->hist_n;
    if (t->clamping) {
        if (sqrt(t->magsq) > t->clamp_max) t->u0 = t->clamp_max / sqrt(t->magsq);
        else t->u0 = t->R / sqrt(avg);
    } else t->u0 = t->R / sqrt(avg);
    if (!t->threshold_enable) return t->u0;
    if (t->magsq > t->threshold) { if (t->gate_counter < t->gate) t->gate_counter++; else t->count = 0; }
    else { if (t->count < t->step_down_delay) t->count++; t->gate_counter = 0; }
    if (t->count < t->step_down_delay) {
        t->step_down = t->step_up;
        if (t->step_up < t->step_length) { t->step_up++; return t->u0 * smootherstep((float)(t->step_up * t->step_delta)); }
        return t->u0;
    }
    t->step_up = t->step_down;
    if (t->step_down > 0) { t->step_down--; return t->u0 * smootherstep((float)(t->step_down * t->step_delta)); }
    return 0.0;
}

void sdro_ssbtail_process(sdro_ssbtail* t, const float* sideband, int64_t n, int16_t* audio)
{
    for (int64_t k = 0; k < n; k++) {
        const float re = sideband[2 * k], im = sideband[2 * k + 1];
        const float agc = t->agc_active ? (float)magagc_feed(t, re, im) : 10.0f;
        const int delay = t->step_down_delay > t->dl_size ? t->dl_size : t->step_down_delay;
        /* readBack BEFORE this sample's write: m_currentIndex is still the previous write's slot */
        const float dr = t->dl[2 * (t->dl_cur + t->dl_size - delay)], di = t->dl[2 * (t->dl_cur + t->dl_size - delay) + 1];
        const float wr = re * agc, wi = im * agc;
        t->dl[2 * t->dl_w] = wr; t->dl[2 * t->dl_w + 1] = wi;
        t->dl[2 * (t->dl_w + t->dl_size)] = wr; t->dl[2 * (t->dl_w + t->dl_size) + 1] = wi;
        t->dl_cur = t->dl_w; t->dl_w = t->dl_w < t->dl_size - 1 ? t->dl_w + 1 : 0;
        /* getStepValue (agc.cpp:189-199) */
        const float sv = t->count < t->step_down_delay ? smootherstep((float)(t->step_up * t->step_delta)) : smootherstep((float)(t->step_down * t->step_delta));
        const float zr = dr * sv, zi = di * sv;
        const float demod = (float)((double)(zr + zi) * 0.7);
        audio[k] = to_q16(demod * t->volume);
    }
}

/* ------------------------------------------------------------------ IIRFilter<float, Order> (sdrbase/dsp/iirfilter.h)
 * Order 2 is the specialisation (:121-160): y = b0 s + b1 x0 + b2 x1 + a1 y0 + a2 y1, summed left to right.
 * Other orders use the generic template (:59-105), whose setCoeffs stores `b` in m_a and `a` in m_b (sic) and whose run()
 * walks i = Order .. 1:  y = m_b[0] s;  y += m_b[i] x[i-1] + m_a[i] y[i-1]. */
struct sdro_iir { int order; float ma[9], mb[9], x[8], y[8]; };

sdro_iir* sdro_iir_new(int32_t order, const float* a, const float* b)
{
    sdro_iir* f = (sdro_iir*)calloc(1, sizeof *f);
    f->order = order;
    for (int i = 0; i <= order; i++) {
        if (order == 2) { f->ma[i] = a[i]; f->mb[i] = b[i]; }
        else { f->ma[i] = b[i]; f->mb[i] = a[i]; }
    }
    return f;
}
void sdro_iir_free(sdro_iir* f) { free(f); }
void sdro_iir_run(sdro_iir* f, const float* in, int64_t n, float* out)
{
    const int O = f->order;
    for (int64_t k = 0; k < n; k++) {
        const float s = in[k];
        float y;
        if (O == 2) {
            y = f->mb[0] * s + f->mb[1] * f->x[0] + f->mb[2] * f->x[1] + f->ma[1] * f->y[0] + f->ma[2] * f->y[1];
            f->x[1] = f->x[0]; f->x[0] = s; f->y[1] = f->y[0]; f->y[0] = y;
        } else {
            y = f->mb[0] * s;
            for (int i = O; i > 0; i--) {
                y += f->mb[i] * f->x[i - 1] + f->ma[i] * f->y[i - 1];
                if (i > 1) { f->x[i - 1] = f->x[i - 2]; f->y[i - 1] = f->y[i - 2]; }
            }
            f->x[0] = s; f->y[0] = y;
        }
        out[k] = y;
    }
}
