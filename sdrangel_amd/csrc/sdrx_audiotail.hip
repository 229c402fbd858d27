// libsdrx.so: sdrx_audiotail_* -- the audio-rate tail of the NFM and SSB demodulators:
//   NFM (plugins/channelrx/demodnfm/nfmdemod.cpp:150-300, default switches): phaseDiscriminatorDelta -> 32-deep power average
//       -> squelch with gate counter -> 24000-entry delay line -> Bandpass<Real>(301 taps), called only while the squelch is
//       open -> volume -> qint16
//   SSB (plugins/channelrx/demodssb/ssbdemod.cpp:181-250, mono): MagAGC::feedAndGetValue (sdrbase/dsp/agc.cpp:96-175: a
//       running sum in double, a square root and a division per sample, step-up / step-down counters) -> 96000-entry
//       delay line -> getStepValue -> (re + im) * 0.7 * volume -> qint16
// Both are serial state machines whose floating-point sums do not re-associate, so the only bit-exact form is the
// reference's statement order: ONE LANE PER CHANNEL walks its channel's samples, 32 channels per workgroup side by side
// (the FIR ring and taps of the NFM flavour sit in LDS, lane-interleaved).  256 channels x 1 s of 48 kS/s audio is ~12 M
// lane-steps: milliseconds, which is why "stays on the host" (round 1) was a choice, not a constraint.
#include "sdrx_common.hpp"
#include <new>
#include <vector>
#include <cmath>
#include <cstring>

using namespace sdrx;

namespace {

constexpr int AT_LANES = 32;                 // channels per workgroup
constexpr int AT_TAPS = 301, AT_H = 150;     // Bandpass<Real>::create(301, ...)
constexpr int NFM_DL = 24000, SSB_DL = 2 * 48000;

struct AtChan {                              // everything a channel carries between feeds (device memory)
    int kind;                                // 0 NFM, 1 SSB
    // NFM
    float prev_arg, fm_scaling, level, volume, comp;
    int gate, count;
    float ma_s[32]; int ma_n; unsigned ma_idx; double ma_total;
    int dl_w, dl_cur;
    float taps[AT_H + 1]; float ring[AT_TAPS]; int ring_p;        // ring_p: slot of the newest FIR input
    // SSB (MagAGC)
    double u0, R, magsq, threshold, step_delta, clamp_max, sum;
    int hist_n; unsigned hist_idx;
    int threshold_enable, agate, step_length, step_up, step_down, gate_counter, step_down_delay, clamping, acount, agc_active;
    // per-channel buffers
    float* dl;                               // NFM: 2 * 24000 floats; SSB: 2 * 96000 complex
    double* hist;                            // SSB: hist_n doubles
};

struct AtJob { const float* in; int16_t* out; long n; };

__device__ __forceinline__ int to_q16(float v)
{
    // (qint16) of a float as x86-64 does it: cvttss2si (0x80000000 when out of range or NaN), then the low 16 bits
    const int i = (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000u;
    return (int)(short)i;
}

__device__ __forceinline__ float at_atan2_approx2(float y, float x)      // phasediscri.h:172-197
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

__device__ __forceinline__ float at_smootherstep(float x)                // util/stepfunctions.h:23-36
{
    if (x == 1.0f) return 1.0f; else if (x == 0.0f) return 0.0f;
    const double x3 = x * x * x, x4 = x * x3, x5 = x * x4;
    return (float)(6.0 * x5 - 15.0 * x4 + 10.0 * x3);
}

__global__ __launch_bounds__(AT_LANES)
void audiotail_kernel(AtChan* __restrict__ chans, const AtJob* __restrict__ jobs, int n_ch)
{
    // FIR ring and folded taps of the NFM flavour, lane-interleaved: entry j of lane l at [j * 32 + l]
    __shared__ float s_ring[AT_TAPS * AT_LANES];
    __shared__ float s_taps[(AT_H + 1) * AT_LANES];
    const int lane = threadIdx.x, c = blockIdx.x * AT_LANES + lane;
    if (c >= n_ch) return;
    AtChan& s = chans[c];
    const AtJob jb = jobs[c];
    if (s.kind == 0) {
        for (int j = 0; j < AT_TAPS; j++) s_ring[j * AT_LANES + lane] = s.ring[j];
        for (int j = 0; j <= AT_H; j++) s_taps[j * AT_LANES + lane] = s.taps[j];
        float prev_arg = s.prev_arg; int count = s.count, ma_n = s.ma_n; unsigned ma_idx = s.ma_idx; double ma_total = s.ma_total;
        int dl_w = s.dl_w, dl_cur = s.dl_cur, p = s.ring_p;
        const int gate = s.gate; const float level = s.level, comp = s.comp, vol = s.volume, fms = s.fm_scaling;
        for (long k = 0; k < jb.n; k++) {
            const float fI = jb.in[2 * k], fQ = jb.in[2 * k + 1];
            const double magsq_raw = (double)(fI * fI + fQ * fQ);
            const float cur = at_atan2_approx2(fQ, fI);
            float dev = (float)((double)(cur - prev_arg) / 3.14159265358979323846);
            prev_arg = cur;
            if (dev < -1.0f) dev += 2.0f; else if (dev > 1.0f) dev -= 2.0f;
            const float demod = dev * fms;
            const float magsq = (float)(magsq_raw / (32768.0 * 32768.0));
            if (ma_n < 32) { s.ma_s[ma_n++] = magsq; ma_total += (double)magsq; }
            else { const float d = magsq - s.ma_s[ma_idx]; ma_total += (double)d; s.ma_s[ma_idx] = magsq; ma_idx = (ma_idx + 1) & 31; }
            float w;
            if ((float)(ma_total / 32) < level) { w = 0.0f; if (count > 0) count--; }
            else { w = demod * comp; if (count < 2 * gate) count++; }
            s.dl[dl_w] = w; s.dl[dl_w + NFM_DL] = w; dl_cur = dl_w;
            dl_w = dl_w < NFM_DL - 1 ? dl_w + 1 : 0;
            int y16 = 0;
            if (count > gate) {
                const int delay = gate > NFM_DL ? NFM_DL : gate;
                const float x = s.dl[dl_cur + NFM_DL - delay];
                // Bandpass<Real>::filter (bandpass.h:77-122): newest first pair, then (oldest + next-newest) pairs, centre last
                p = p + 1 < AT_TAPS ? p + 1 : 0;
                s_ring[p * AT_LANES + lane] = x;
                int b = p - 1; if (b < 0) b += AT_TAPS;
                float acc = 0.0f;
                acc += (x + s_ring[b * AT_LANES + lane]) * s_taps[lane];
                int a = p + 1; if (a >= AT_TAPS) a = 0;                    // oldest
                b = b - 1; if (b < 0) b += AT_TAPS;
                for (int i = 1; i < AT_H; i++) {
                    acc += (s_ring[a * AT_LANES + lane] + s_ring[b * AT_LANES + lane]) * s_taps[i * AT_LANES + lane];
                    a = a + 1 < AT_TAPS ? a + 1 : 0;
                    b = b > 0 ? b - 1 : AT_TAPS - 1;
                }
                acc += s_ring[a * AT_LANES + lane] * s_taps[AT_H * AT_LANES + lane];
                y16 = to_q16(acc * vol);
            }
            jb.out[k] = (int16_t)y16;
        }
        s.prev_arg = prev_arg; s.count = count; s.ma_n = ma_n; s.ma_idx = ma_idx; s.ma_total = ma_total;
        s.dl_w = dl_w; s.dl_cur = dl_cur; s.ring_p = p;
        for (int j = 0; j < AT_TAPS; j++) s.ring[j] = s_ring[j * AT_LANES + lane];
    } else {
        double u0 = s.u0, sum = s.sum; const double R = s.R, thr = s.threshold, sd = s.step_delta, cmax = s.clamp_max;
        unsigned hi = s.hist_idx; const int hn = s.hist_n;
        int step_up = s.step_up, step_down = s.step_down, gc = s.gate_counter, cnt = s.acount, dl_w = s.dl_w, dl_cur = s.dl_cur;
        const int sdd = s.step_down_delay, sl = s.step_length, ag = s.agate;
        const float vol = s.volume;
        for (long k = 0; k < jb.n; k++) {
            const float re = jb.in[2 * k], im = jb.in[2 * k + 1];
            float agc = 10.0f;
            if (s.agc_active) {
                // MagAGC::feedAndGetValue (agc.cpp:96-175), m_squared = false
                const double magsq = (double)(re * re + im * im);
                { const double o = s.hist[hi]; sum += magsq - o; s.hist[hi] = magsq; hi = hi < (unsigned)hn - 1 ? hi + 1 : 0; }
                const double avg = sum / (double)hn;
                if (s.clamping) { const double rm = __builtin_sqrt(magsq); if (rm > cmax) u0 = cmax / rm; else u0 = R / __builtin_sqrt(avg); }
                else u0 = R / __builtin_sqrt(avg);
                double val;
                if (!s.threshold_enable) val = u0;
                else {
                    if (magsq > thr) { if (gc < ag) gc++; else cnt = 0; }
                    else { if (cnt < sdd) cnt++; gc = 0; }
                    if (cnt < sdd) {
                        step_down = step_up;
                        if (step_up < sl) { step_up++; val = u0 * (double)at_smootherstep((float)(step_up * sd)); } else val = u0;
                    } else {
                        step_up = step_down;
                        if (step_down > 0) { step_down--; val = u0 * (double)at_smootherstep((float)(step_down * sd)); } else val = 0.0;
                    }
                }
                agc = (float)val;
            }
            const int delay = sdd > SSB_DL ? SSB_DL : sdd;
            const float dr = s.dl[2 * (dl_cur + SSB_DL - delay)], di = s.dl[2 * (dl_cur + SSB_DL - delay) + 1];   // readBack before this sample's write
            const float wr = re * agc, wi = im * agc;
            s.dl[2 * dl_w] = wr; s.dl[2 * dl_w + 1] = wi; s.dl[2 * (dl_w + SSB_DL)] = wr; s.dl[2 * (dl_w + SSB_DL) + 1] = wi;
            dl_cur = dl_w; dl_w = dl_w < SSB_DL - 1 ? dl_w + 1 : 0;
            const float sv = cnt < sdd ? at_smootherstep((float)(step_up * sd)) : at_smootherstep((float)(step_down * sd));   // getStepValue
            const float zr = dr * sv, zi = di * sv;
            const float demod = (float)((double)(zr + zi) * 0.7);
            jb.out[k] = (int16_t)to_q16(demod * vol);
        }
        s.u0 = u0; s.sum = sum; s.hist_idx = hi; s.step_up = step_up; s.step_down = step_down; s.gate_counter = gc; s.acount = cnt;
        s.dl_w = dl_w; s.dl_cur = dl_cur;
    }
}

// Bandpass<Real>::create(nTaps = 301, sampleRate, lowCutoff, highCutoff) (bandpass.h:14-75), folded taps [0 .. 150]
void bandpass_taps(double rate, double f1, double f2, float* t)
{
    const double PI = 3.14159265358979323846;
    const int ntaps = AT_TAPS, nt = AT_H + 1;
    const double mid = ((double)ntaps - 1.0) / 2.0;
    const double Wcl = 2.0 * PI * f1 / rate, Wch = 2.0 * PI * f2 / rate;
    std::vector<float> lp((size_t)nt), hp((size_t)nt);
    for (int i = 0; i < nt; i++) {
        if (i == (ntaps - 1) / 2) { lp[(size_t)i] = (float)(Wch / PI); hp[(size_t)i] = (float)(-(Wcl / PI)); }
        else { lp[(size_t)i] = (float)(std::sin(((double)i - mid) * Wch) / (((double)i - mid) * PI)); hp[(size_t)i] = (float)(-std::sin(((double)i - mid) * Wcl) / (((double)i - mid) * PI)); }
    }
    hp[(size_t)((ntaps - 1) / 2)] += 1;
    for (int i = 0; i < nt; i++) {
        const double w = 0.54 + 0.46 * std::cos((2.0 * PI * ((double)i - mid)) / (double)ntaps);
        lp[(size_t)i] = (float)(lp[(size_t)i] * w); hp[(size_t)i] = (float)(hp[(size_t)i] * w);
        t[i] = -(lp[(size_t)i] + hp[(size_t)i]);
    }
    t[(ntaps - 1) / 2] += 1;
    float sum = 0; int i;
    for (i = 0; i < nt - 1; i++) sum += t[i] * 2;
    sum += t[i];
    for (i = 0; i < nt; i++) t[i] /= sum;
}

} // namespace

struct sdrx_audiotail {
    int device = 0, n_ch = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<sdrx_audiotail_cfg> cfg;
    AtChan* d_chan = nullptr;
    std::vector<void*> bufs;                 // per-channel delay lines / histories
    AtJob* d_jobs = nullptr; AtJob* h_jobs = nullptr; hipEvent_t jobs_ev = nullptr;
    std::vector<DevBuf> d_in, d_out;
    std::vector<long> last_n;
};

static int audiotail_init_state(sdrx_audiotail* h)
{
    std::vector<AtChan> hc((size_t)h->n_ch);
    size_t bi = 0;
    for (int c = 0; c < h->n_ch; c++) {
        const sdrx_audiotail_cfg& k = h->cfg[(size_t)c];
        AtChan& s = hc[(size_t)c];
        std::memset(&s, 0, sizeof s);
        s.kind = k.kind; s.volume = k.volume;
        if (k.kind == 0) {
            s.fm_scaling = k.fm_scaling; s.level = k.squelch_level; s.gate = k.squelch_gate;
            s.comp = (float)k.audio_rate / 48000.0f; s.comp *= std::sqrt(s.comp);      // nfmdemod.cpp:82-83
            bandpass_taps((double)k.audio_rate, 300.0, (double)k.af_bandwidth, s.taps); // m_bandpass.create(301, rate, 300.0, bw) (:428-429)
            s.ring_p = 0;
            s.dl = static_cast<float*>(h->bufs[bi++]);
            SDRX_HIP(hipMemsetAsync(s.dl, 0, sizeof(float) * 2 * NFM_DL, h->stream));
        } else {
            const float Rf = (float)3276.8;                                            // MagAGC::resize(n, n / 2, Real agcTarget)
            s.R = (double)Rf; s.u0 = 1.0;
            s.hist_n = k.agc_nb_samples; s.step_length = k.agc_nb_samples / 2; s.step_delta = 1.0 / s.step_length;
            s.step_up = 0; s.step_down = s.step_length; s.step_down_delay = k.agc_nb_samples;
            s.threshold = k.agc_threshold; s.threshold_enable = k.agc_threshold_enable; s.agate = k.agc_gate;
            s.clamping = k.agc_clamping; s.clamp_max = 32768.0 / 100.0; s.agc_active = k.agc_active;
            s.dl = static_cast<float*>(h->bufs[bi++]);
            s.hist = static_cast<double*>(h->bufs[bi++]);
            SDRX_HIP(hipMemsetAsync(s.dl, 0, sizeof(float) * 4 * SSB_DL, h->stream));
            SDRX_HIP(hipMemsetAsync(s.hist, 0, sizeof(double) * (size_t)k.agc_nb_samples, h->stream));
        }
    }
    SDRX_HIP(hipMemcpyAsync(h->d_chan, hc.data(), sizeof(AtChan) * (size_t)h->n_ch, hipMemcpyHostToDevice, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

extern "C" {

int sdrx_audiotail_create(sdrx_audiotail_t** out, int device, int32_t n_ch, const sdrx_audiotail_cfg* cfg)
{
    if (!out || n_ch <= 0 || !cfg) { set_error("sdrx_audiotail_create: bad argument"); return SDRX_EINVAL; }
    *out = nullptr;
    for (int c = 0; c < n_ch; c++) {
        const sdrx_audiotail_cfg& k = cfg[c];
        if ((k.kind != 0 && k.kind != 1) || k.audio_rate <= 0 || (k.kind == 0 && (k.squelch_gate < 0 || k.af_bandwidth <= 300.0f)) ||
            (k.kind == 1 && (k.agc_nb_samples < 2 || k.agc_nb_samples > (1 << 20)))) {
            set_error("sdrx_audiotail_create: bad channel configuration"); return SDRX_EINVAL;
        }
    }
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_audiotail* h = new (std::nothrow) sdrx_audiotail;
    if (!h) return SDRX_ENOMEM;
    h->device = device; h->n_ch = n_ch; h->cfg.assign(cfg, cfg + n_ch);
    h->d_in.resize((size_t)n_ch); h->d_out.resize((size_t)n_ch); h->last_n.assign((size_t)n_ch, 0);
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) { h->stream = h->own_stream; e = hipMalloc(reinterpret_cast<void**>(&h->d_chan), sizeof(AtChan) * (size_t)n_ch); }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_jobs), sizeof(AtJob) * (size_t)n_ch);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->h_jobs), sizeof(AtJob) * (size_t)n_ch, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->jobs_ev, hipEventDisableTiming);
    for (int c = 0; c < n_ch && e == hipSuccess; c++) {
        void* p = nullptr;
        if (cfg[c].kind == 0) { e = hipMalloc(&p, sizeof(float) * 2 * NFM_DL); if (e == hipSuccess) h->bufs.push_back(p); }
        else {
            e = hipMalloc(&p, sizeof(float) * 4 * SSB_DL); if (e == hipSuccess) h->bufs.push_back(p);
            if (e == hipSuccess) { e = hipMalloc(&p, sizeof(double) * (size_t)cfg[c].agc_nb_samples); if (e == hipSuccess) h->bufs.push_back(p); }
        }
    }
    if (e != hipSuccess) { sdrx_audiotail_destroy(h); return hip_fail(e, "sdrx_audiotail_create", __FILE__, __LINE__); }
    rc = audiotail_init_state(h);
    if (rc) { sdrx_audiotail_destroy(h); return rc; }
    *out = h;
    return SDRX_OK;
}

int sdrx_audiotail_destroy(sdrx_audiotail_t* h)
{
    if (!h) return SDRX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->bufs) (void)hipFree(p);
    if (h->d_chan) (void)hipFree(h->d_chan);
    if (h->d_jobs) (void)hipFree(h->d_jobs);
    if (h->h_jobs) (void)hipHostFree(h->h_jobs);
    if (h->jobs_ev) (void)hipEventDestroy(h->jobs_ev);
    for (auto& b : h->d_in) b.release();
    for (auto& b : h->d_out) b.release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SDRX_OK;
}

int sdrx_audiotail_reset(sdrx_audiotail_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return audiotail_init_state(h);
}

int sdrx_audiotail_feed_dev(sdrx_audiotail_t* h, const float* const* d_in, const int64_t* n, int16_t* const* d_audio)
{
    if (!h || !d_in || !n || !d_audio) { set_error("sdrx_audiotail_feed_dev: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipEventSynchronize(h->jobs_ev));
    for (int c = 0; c < h->n_ch; c++) {
        if (n[c] < 0 || (n[c] > 0 && (!d_in[c] || !d_audio[c]))) { set_error("sdrx_audiotail_feed_dev: bad channel argument"); return SDRX_EINVAL; }
        h->h_jobs[c] = AtJob{ d_in[c], d_audio[c], (long)n[c] };
    }
    SDRX_HIP(hipMemcpyAsync(h->d_jobs, h->h_jobs, sizeof(AtJob) * (size_t)h->n_ch, hipMemcpyHostToDevice, h->stream));
    SDRX_HIP(hipEventRecord(h->jobs_ev, h->stream));
    hipLaunchKernelGGL(audiotail_kernel, dim3((unsigned)((h->n_ch + AT_LANES - 1) / AT_LANES)), dim3(AT_LANES), 0, h->stream, h->d_chan, h->d_jobs, h->n_ch);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

int sdrx_audiotail_feed(sdrx_audiotail_t* h, const float* const* in, const int64_t* n, int16_t* const* audio)
{
    if (!h || !in || !n || !audio) { set_error("sdrx_audiotail_feed: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    std::vector<const float*> di((size_t)h->n_ch); std::vector<int16_t*> dout((size_t)h->n_ch);
    for (int c = 0; c < h->n_ch; c++) {
        if (n[c] < 0 || (n[c] > 0 && (!in[c] || !audio[c]))) { set_error("sdrx_audiotail_feed: bad channel argument"); return SDRX_EINVAL; }
        int rc = h->d_in[(size_t)c].reserve((size_t)(n[c] > 0 ? n[c] : 1) * 8); if (rc) return rc;
        rc = h->d_out[(size_t)c].reserve((size_t)(n[c] > 0 ? n[c] : 1) * 2); if (rc) return rc;
        if (n[c]) SDRX_HIP(hipMemcpyAsync(h->d_in[(size_t)c].p, in[c], (size_t)n[c] * 8, hipMemcpyHostToDevice, h->stream));
        di[(size_t)c] = static_cast<const float*>(h->d_in[(size_t)c].p); dout[(size_t)c] = static_cast<int16_t*>(h->d_out[(size_t)c].p);
    }
    int rc = sdrx_audiotail_feed_dev(h, di.data(), n, dout.data()); if (rc) return rc;
    for (int c = 0; c < h->n_ch; c++)
        if (n[c]) SDRX_HIP(hipMemcpyAsync(audio[c], h->d_out[(size_t)c].p, (size_t)n[c] * 2, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_audiotail_sync(sdrx_audiotail_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

} // extern "C"


/* =====================================================================================================================
 * sdrx_iir_* -- IIRFilter<float, Order> (sdrbase/dsp/iirfilter.h; users: FilterMbe's 2nd-order low/high pass pair,
 * sdrbase/dsp/filtermbe.h:76-77).  A recursive filter is serial along time by nature; one lane per channel, the
 * reference's operation order (order 2: the specialisation :150-160; other orders: the generic template :88-105, including
 * its swapped coefficient copy :78-81), float multiply and add kept separate.
 * ===================================================================================================================== */
namespace {
struct IirChan { int order; float ma[9], mb[9], x[8], y[8]; };
struct IirJob { const float* in; float* out; long n; };

__global__ __launch_bounds__(64)
void iir_kernel(IirChan* __restrict__ chans, const IirJob* __restrict__ jobs, int n_ch)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_ch) return;
    IirChan s = chans[c];
    const IirJob jb = jobs[c];
    const int O = s.order;
    for (long k = 0; k < jb.n; k++) {
        const float v = jb.in[k];
        float y;
        if (O == 2) {
            y = s.mb[0] * v + s.mb[1] * s.x[0] + s.mb[2] * s.x[1] + s.ma[1] * s.y[0] + s.ma[2] * s.y[1];
            s.x[1] = s.x[0]; s.x[0] = v; s.y[1] = s.y[0]; s.y[0] = y;
        } else {
            y = s.mb[0] * v;
#pragma unroll
            for (int i = 8; i > 0; i--) {
                if (i > O) continue;
                y += s.mb[i] * s.x[i - 1] + s.ma[i] * s.y[i - 1];
                if (i > 1) { s.x[i - 1] = s.x[i - 2]; s.y[i - 1] = s.y[i - 2]; }
            }
            s.x[0] = v; s.y[0] = y;
        }
        jb.out[k] = y;
    }
    chans[c] = s;
}
} // namespace

struct sdrx_iir {
    int device = 0, n_ch = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<IirChan> init;
    IirChan* d_chan = nullptr; IirJob* d_jobs = nullptr; IirJob* h_jobs = nullptr; hipEvent_t jobs_ev = nullptr;
    std::vector<DevBuf> d_in, d_out;
};

extern "C" {

int sdrx_iir_create(sdrx_iir_t** out, int device, int32_t n_ch, const sdrx_iir_cfg* cfg)
{
    if (!out || n_ch <= 0 || !cfg) { set_error("sdrx_iir_create: bad argument"); return SDRX_EINVAL; }
    *out = nullptr;
    for (int c = 0; c < n_ch; c++) if (cfg[c].order < 2 || cfg[c].order > 8) { set_error("sdrx_iir_create: order 2..8"); return SDRX_EINVAL; }
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_iir* h = new (std::nothrow) sdrx_iir;
    if (!h) return SDRX_ENOMEM;
    h->device = device; h->n_ch = n_ch; h->init.resize((size_t)n_ch); h->d_in.resize((size_t)n_ch); h->d_out.resize((size_t)n_ch);
    for (int c = 0; c < n_ch; c++) {
        IirChan& s = h->init[(size_t)c];
        std::memset(&s, 0, sizeof s);
        s.order = cfg[c].order;
        for (int i = 0; i <= s.order; i++) {
            if (s.order == 2) { s.ma[i] = cfg[c].a[i]; s.mb[i] = cfg[c].b[i]; }
            else { s.ma[i] = cfg[c].b[i]; s.mb[i] = cfg[c].a[i]; }              // iirfilter.h:78-81 (sic)
        }
    }
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) { h->stream = h->own_stream; e = hipMalloc(reinterpret_cast<void**>(&h->d_chan), sizeof(IirChan) * (size_t)n_ch); }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_jobs), sizeof(IirJob) * (size_t)n_ch);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->h_jobs), sizeof(IirJob) * (size_t)n_ch, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->jobs_ev, hipEventDisableTiming);
    if (e != hipSuccess) { sdrx_iir_destroy(h); return hip_fail(e, "sdrx_iir_create", __FILE__, __LINE__); }
    *out = h;
    return sdrx_iir_reset(h);
}

int sdrx_iir_destroy(sdrx_iir_t* h)
{
    if (!h) return SDRX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->d_chan) (void)hipFree(h->d_chan);
    if (h->d_jobs) (void)hipFree(h->d_jobs);
    if (h->h_jobs) (void)hipHostFree(h->h_jobs);
    if (h->jobs_ev) (void)hipEventDestroy(h->jobs_ev);
    for (auto& b : h->d_in) b.release();
    for (auto& b : h->d_out) b.release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SDRX_OK;
}

int sdrx_iir_reset(sdrx_iir_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemcpyAsync(h->d_chan, h->init.data(), sizeof(IirChan) * (size_t)h->n_ch, hipMemcpyHostToDevice, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_iir_feed(sdrx_iir_t* h, const float* const* in, const int64_t* n, float* const* out)
{
    if (!h || !in || !n || !out) { set_error("sdrx_iir_feed: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipEventSynchronize(h->jobs_ev));
    for (int c = 0; c < h->n_ch; c++) {
        if (n[c] < 0 || (n[c] > 0 && (!in[c] || !out[c]))) { set_error("sdrx_iir_feed: bad channel argument"); return SDRX_EINVAL; }
        int rc = h->d_in[(size_t)c].reserve((size_t)(n[c] > 0 ? n[c] : 1) * 4); if (rc) return rc;
        rc = h->d_out[(size_t)c].reserve((size_t)(n[c] > 0 ? n[c] : 1) * 4); if (rc) return rc;
        if (n[c]) SDRX_HIP(hipMemcpyAsync(h->d_in[(size_t)c].p, in[c], (size_t)n[c] * 4, hipMemcpyHostToDevice, h->stream));
        h->h_jobs[c] = IirJob{ static_cast<const float*>(h->d_in[(size_t)c].p), static_cast<float*>(h->d_out[(size_t)c].p), (long)n[c] };
    }
    SDRX_HIP(hipMemcpyAsync(h->d_jobs, h->h_jobs, sizeof(IirJob) * (size_t)h->n_ch, hipMemcpyHostToDevice, h->stream));
    SDRX_HIP(hipEventRecord(h->jobs_ev, h->stream));
    hipLaunchKernelGGL(iir_kernel, dim3((unsigned)((h->n_ch + 63) / 64)), dim3(64), 0, h->stream, h->d_chan, h->d_jobs, h->n_ch);
    SDRX_HIP(hipGetLastError());
    for (int c = 0; c < h->n_ch; c++)
        if (n[c]) SDRX_HIP(hipMemcpyAsync(out[c], h->d_out[(size_t)c].p, (size_t)n[c] * 4, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

} // extern "C"
