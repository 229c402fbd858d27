// libsdrx.so: sdrx_backend_* -- per-channel NCO -> Interpolator -> fftfilt -> discriminator bank, the
// common front of the channelrx demods' feed() (nfmdemod.cpp:150-163, ssbdemod.cpp:158-172).
// Host side: filter / tap / table design exactly as the reference does it at configure time
// (Interpolator::create interpolator.cpp:21-129, fftfilt::create_filter fftfilt.cpp:108-146,
// NCO::initTable nco.cpp:30-39, g_fft::fftCosInit gfft.h:141-150), launches, state bookkeeping.
#include "sdrx_common.hpp"
#include "backend_kernels.hpp"
#include <vector>
#include <cmath>
#include <cstring>
#include <new>
#include <algorithm>

using namespace sdrx;

namespace {

const double PI_D = 3.14159265358979323846;

// Interpolator::createPolyphaseLowPass + reorder + per-phase normalisation
void design_interp(int phase_steps, double sample_rate, double cutoff, double tpp, std::vector<float>& poly, int* ntaps_per_phase)
{
    double gain = 1.0;
    const double fs = phase_steps * sample_rate;
    int ntaps = (int)(tpp * phase_steps);
    if ((ntaps % 2) != 0) ntaps++;
    ntaps *= phase_steps;
    std::vector<float> taps((size_t)ntaps, 0.0f), window((size_t)ntaps);
    for (int n = 0; n < ntaps; n++) window[(size_t)n] = (float)(0.54 - 0.46 * std::cos((2 * PI_D * n) / (ntaps - 1)));
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2 * PI_D * cutoff / fs;
    for (int n = -M; n <= M; n++) {
        if (n == 0) taps[(size_t)(n + M)] = (float)(fwT0 / PI_D * window[(size_t)(n + M)]);
        else taps[(size_t)(n + M)] = (float)(std::sin(n * fwT0) / (n * PI_D) * window[(size_t)(n + M)]);
    }
    double mx = taps[(size_t)M];
    for (int n = 1; n <= M; n++) mx += 2.0 * taps[(size_t)(n + M)];
    gain /= mx;
    for (int i = 0; i < ntaps; i++) taps[(size_t)i] = (float)(taps[(size_t)i] * gain);
    const int nt = ntaps / phase_steps;
    poly.assign((size_t)ntaps, 0.0f);
    for (int ph = 0; ph < phase_steps; ph++)
        for (int i = 0; i < nt; i++) poly[(size_t)(ph * nt + i)] = taps[(size_t)(i * phase_steps + ph)];
    for (int ph = 0; ph < phase_steps; ph++) {
        float sum = 0;
        for (int i = 0; i < nt; i++) sum += poly[(size_t)(ph * nt + i)];
        for (int i = 0; i < nt; i++) poly[(size_t)(ph * nt + i)] /= sum;
    }
    *ntaps_per_phase = nt;
}

float fsinc(float fc, int i, int len)
{
    const int len2 = len / 2;
    return (i == len2) ? (float)(2.0 * fc) : (float)(std::sin(2 * PI_D * fc * (i - len2)) / (PI_D * (i - len2)));
}
float blackman(int i, int len)
{
    return (float)(0.42 - 0.50 * std::cos(2.0 * PI_D * i / len) + 0.08 * std::cos(4.0 * PI_D * i / len));
}

struct ChanHost {
    sdrx_backend_cfg cfg;
    DevBuf mixed, res, head, tail, cplx_out, real_out;
    uint32_t* hist[2] = { nullptr, nullptr };
    int cur = 0;
    int64_t cap_in = 0;
    DevBuf stage_in;              // host-pointer feeds
};

} // namespace

struct sdrx_backend {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int n_ch = 0;
    std::vector<ChanHost> ch;
    std::vector<BeChan> h_chan;   // host mirror of the config part (state lives on the device)
    BeChan* d_chan = nullptr;
    BeBufs* d_bufs = nullptr;
    DevBuf sched;                 // bank-wide, channel-interleaved: entry o of channel c at [o * n_ch + c]
    int64_t sched_cap = 0;        // entries per channel
    BeBufs* h_bufs = nullptr;     // pinned: the per-feed table goes to the device in one async copy
    hipEvent_t bufs_ev = nullptr; // recorded behind that copy; waited on before the table is rewritten
    hipEvent_t prod_ev = nullptr, cons_ev = nullptr;       // device-side ordering against a producer stream (sdrx_backend_feed_bank)
    float* d_nco = nullptr; float* d_taps = nullptr; float2* d_filters = nullptr;
    float* d_utbl = nullptr; float* d_utbl2 = nullptr;     // g_fft cosine tables for N = 1024 / 2048
    bool any1024 = false, any2048 = false;
    bool any_dyadic = false;      // some resampling ratio has a closed-form schedule
    std::vector<float> taps_all; std::vector<float> filters_all;     // kept for inspection (tests)
    std::vector<int> taps_off, filt_off, ntaps;
    std::vector<int> perm, col_of;    // schedule column q holds channel perm[q]; col_of[c] is its inverse
    int* d_perm = nullptr;
    bool state_valid = false;     // d_chan state initialised
};

static int ensure_capacity(sdrx_backend* b, int c, int64_t n_in)
{
    ChanHost& h = b->ch[(size_t)c];
    if (n_in <= h.cap_in) return SDRX_OK;
    int64_t cap = h.cap_in ? h.cap_in : 4096;
    while (cap < n_in) cap *= 2;
    // pending (<512) + at most one resampler output per input (distance step >= 1 by construction of decimate())
    const size_t n_res_max = (size_t)cap + 1024;
    const size_t half = h.cfg.filt_mode >= 4 ? BE_FFT : BE_FFT / 2;
    const size_t n_blk_max = n_res_max / half + 2;
    // buffers that carry state (res: pending, tail: ovlbuf) must keep their content
    auto grow_keep = [&](DevBuf& buf, size_t bytes, size_t keep) -> int {
        if (bytes <= buf.cap) return SDRX_OK;
        void* np = nullptr;
        SDRX_HIP(hipMalloc(&np, bytes));
        SDRX_HIP(hipMemsetAsync(np, 0, bytes, b->stream));
        if (buf.p && keep) SDRX_HIP(hipMemcpyAsync(np, buf.p, keep, hipMemcpyDeviceToDevice, b->stream));
        SDRX_HIP(hipStreamSynchronize(b->stream));
        if (buf.p) (void)hipFree(buf.p);
        buf.p = np; buf.cap = bytes;
        return SDRX_OK;
    };
    int rc;
    if ((rc = grow_keep(h.mixed, (size_t)(BE_HIST + cap) * 8, 0))) return rc;
    if ((rc = grow_keep(h.res, (n_res_max + BE_FFT) * 8, BE_FFT * 8))) return rc;
    if ((rc = grow_keep(h.head, n_blk_max * half * 8, 0))) return rc;
    if ((rc = grow_keep(h.tail, (n_blk_max + 1) * half * 8, half * 8))) return rc;
    if ((rc = grow_keep(h.cplx_out, n_res_max * 8, 0))) return rc;
    if ((rc = grow_keep(h.real_out, n_res_max * 4, 0))) return rc;
    h.cap_in = cap;
    return SDRX_OK;
}

extern "C" {

int sdrx_backend_destroy(sdrx_backend_t* b)
{
    if (!b) return SDRX_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (auto& h : b->ch) {
        h.mixed.release(); h.res.release(); h.head.release(); h.tail.release();
        h.cplx_out.release(); h.real_out.release(); h.stage_in.release();
        for (int i = 0; i < 2; i++) if (h.hist[i]) (void)hipFree(h.hist[i]);
    }
    b->sched.release();
    if (b->d_chan) (void)hipFree(b->d_chan);
    if (b->d_bufs) (void)hipFree(b->d_bufs);
    if (b->h_bufs) (void)hipHostFree(b->h_bufs);
    if (b->bufs_ev) (void)hipEventDestroy(b->bufs_ev);
    if (b->prod_ev) (void)hipEventDestroy(b->prod_ev);
    if (b->cons_ev) (void)hipEventDestroy(b->cons_ev);
    if (b->d_nco) (void)hipFree(b->d_nco);
    if (b->d_taps) (void)hipFree(b->d_taps);
    if (b->d_perm) (void)hipFree(b->d_perm);
    if (b->d_filters) (void)hipFree(b->d_filters);
    if (b->d_utbl) (void)hipFree(b->d_utbl);
    if (b->d_utbl2) (void)hipFree(b->d_utbl2);
    if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
    delete b;
    return SDRX_OK;
}

int sdrx_backend_create(sdrx_backend_t** out, int device, int32_t n_ch, const sdrx_backend_cfg* cfg)
{
    if (!out) { set_error("sdrx_backend_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    if (n_ch <= 0 || !cfg) { set_error("sdrx_backend_create: bad argument"); return SDRX_EINVAL; }
    for (int c = 0; c < n_ch; c++) {
        const sdrx_backend_cfg& k = cfg[c];
        if (k.in_rate <= 0 || k.out_rate <= 0 || k.out_rate > k.in_rate || k.filt_mode < 0 || k.filt_mode > 6 ||
            k.discri < 0 || k.discri > 2 || k.taps_per_phase <= 0 || k.taps_per_phase * 16 > BE_HIST) {
            set_error("sdrx_backend_create: bad channel configuration (need out_rate <= in_rate, taps_per_phase*16 <= 256)");
            return SDRX_EINVAL;
        }
    }
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_backend* b = new (std::nothrow) sdrx_backend;
    if (!b) return SDRX_ENOMEM;
    b->device = device; b->n_ch = n_ch;
    hipError_t e = hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete b; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    b->stream = b->own_stream;
    b->ch.resize((size_t)n_ch); b->h_chan.resize((size_t)n_ch);
    b->taps_off.resize((size_t)n_ch); b->filt_off.resize((size_t)n_ch); b->ntaps.resize((size_t)n_ch);

#define BE_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = hip_fail(e_, #call, __FILE__, __LINE__); sdrx_backend_destroy(b); return r_; } } while (0)
    // NCO table (nco.cpp:30-39) and g_fft cosine table (gfft.h:141-150)
    std::vector<float> nco(BE_NCO_N);
    for (int i = 0; i < BE_NCO_N; i++) nco[(size_t)i] = (float)std::cos((2.0 * PI_D * i) / BE_NCO_N);
    BE_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_nco), BE_NCO_N * 4));
    BE_TRY(hipMemcpy(b->d_nco, nco.data(), BE_NCO_N * 4, hipMemcpyHostToDevice));
    for (int n : { BE_FFT, BE_FFT_MAX }) {
        std::vector<float> utbl((size_t)n / 4 + 1);
        utbl[0] = 1.0f;
        for (int i = 1; i < n / 4; i++) utbl[(size_t)i] = (float)std::cos((2.0 * 3.141592653589793238462643383279502884197 * (float)i) / (float)n);
        utbl[(size_t)n / 4] = 0.0f;
        float*& dst = n == BE_FFT ? b->d_utbl : b->d_utbl2;
        BE_TRY(hipMalloc(reinterpret_cast<void**>(&dst), ((size_t)n / 4 + 1) * 4));
        BE_TRY(hipMemcpy(dst, utbl.data(), ((size_t)n / 4 + 1) * 4, hipMemcpyHostToDevice));
    }

    // per channel design
    BE_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_filters), (size_t)n_ch * 2 * BE_FFT_MAX * 8));   // [filter | filterOpp] per channel
    b->filters_all.assign((size_t)n_ch * 2 * BE_FFT_MAX * 2, 0.0f);
    for (int c = 0; c < n_ch; c++) {
        const sdrx_backend_cfg& k = cfg[c];
        ChanHost& h = b->ch[(size_t)c];
        h.cfg = k;
        // channels with the same Interpolator::create arguments share one tap table
        int same = -1;
        for (int p = 0; p < c && same < 0; p++) {
            const sdrx_backend_cfg& o = cfg[p];
            if (o.in_rate == k.in_rate && o.interp_cutoff == k.interp_cutoff && o.taps_per_phase == k.taps_per_phase) same = p;
        }
        if (same >= 0) {
            b->taps_off[(size_t)c] = b->taps_off[(size_t)same]; b->ntaps[(size_t)c] = b->ntaps[(size_t)same];
        } else {
            std::vector<float> poly; int nt = 0;
            design_interp(16, (double)k.in_rate, (double)k.interp_cutoff, (double)k.taps_per_phase, poly, &nt);
            b->taps_off[(size_t)c] = (int)b->taps_all.size(); b->ntaps[(size_t)c] = nt;
            b->taps_all.insert(b->taps_all.end(), poly.begin(), poly.end());
        }
        const int nt = b->ntaps[(size_t)c];
        b->filt_off[(size_t)c] = c * 2 * BE_FFT_MAX;
        const int flen = k.filt_mode >= 4 ? BE_FFT_MAX : BE_FFT;          // runDSB / runAsym: "double the FFT size used for SSB"
        // fftfilt::create_filter / create_dsb_filter / create_asym_filter: windowed sinc in the first flen2 bins, forward FFT
        // (on the GPU, with the same kernel code the data path uses), normalise to max |H| over bins 0..flen2-1.
        // which: 0 = filter, 1 = filterOpp (runAsym only: low pass at f1 = the opposite band's width)
        auto design = [&](int which) -> int {
            std::vector<float> f((size_t)flen * 2, 0.0f);
            const int h2 = flen / 2;
            if (k.filt_mode >= 4) {
                const float fc = which ? k.f1 : k.f2;                                            // fftfilt(f2, len) / create_asym_filter(fopp = f1, fin = f2)
                for (int i = 0; i < h2; i++) f[(size_t)(2 * i)] = fsinc(fc, i, h2);
            } else {
                const bool lp = k.f2 != 0, hp = k.f1 != 0;
                for (int i = 0; i < h2; i++) {
                    float v = 0;
                    if (lp) v += fsinc(k.f2, i, h2);
                    if (hp) v -= fsinc(k.f1, i, h2);
                    f[(size_t)(2 * i)] = v;
                }
                if (hp && k.f2 < k.f1) f[(size_t)(2 * (h2 / 2))] += 1;
            }
            for (int i = 0; i < h2; i++) { const float w = blackman(i, h2); f[(size_t)(2 * i)] *= w; f[(size_t)(2 * i + 1)] *= w; }
            float2* dst = b->d_filters + (size_t)c * 2 * BE_FFT_MAX + (size_t)which * BE_FFT_MAX;
            SDRX_HIP(hipMemcpy(dst, f.data(), (size_t)flen * 8, hipMemcpyHostToDevice));
            if (flen == BE_FFT) hipLaunchKernelGGL(be_fft_design_kernel<BE_FFT>, dim3(1), dim3(BE_FFT / 8), 0, b->stream, dst, b->d_utbl);
            else hipLaunchKernelGGL(be_fft_design_kernel<BE_FFT_MAX>, dim3(1), dim3(BE_FFT_MAX / 8), 0, b->stream, dst, b->d_utbl2);
            SDRX_HIP(hipGetLastError());
            SDRX_HIP(hipStreamSynchronize(b->stream));
            SDRX_HIP(hipMemcpy(f.data(), dst, (size_t)flen * 8, hipMemcpyDeviceToHost));
            float scale = 0;
            for (int i = 0; i < h2; i++) { const float mag = hypotf(f[(size_t)(2 * i)], f[(size_t)(2 * i + 1)]); if (mag > scale) scale = mag; }
            if (scale != 0) for (int i = 0; i < flen * 2; i++) f[(size_t)i] /= scale;
            SDRX_HIP(hipMemcpy(dst, f.data(), (size_t)flen * 8, hipMemcpyHostToDevice));
            std::memcpy(&b->filters_all[((size_t)c * 2 + (size_t)which) * BE_FFT_MAX * 2], f.data(), (size_t)flen * 8);
            return SDRX_OK;
        };
        if (k.filt_mode) {
            int drc = design(0);
            if (!drc && k.filt_mode >= 5) drc = design(1);
            if (drc) { sdrx_backend_destroy(b); return drc; }
            (flen == BE_FFT ? b->any1024 : b->any2048) = true;
        }
        BeChan& s = b->h_chan[(size_t)c];
        std::memset(&s, 0, sizeof s);
        s.nco_inc = (int)(((float)k.nco_freq * BE_NCO_N) / (float)k.in_rate);          // NCO::setFreq (float math, truncation)
        s.step = (float)k.in_rate / (float)k.out_rate;
        s.dy_q = -1; s.dy_S = 0;
        if (s.step >= 1.0f && !getenv("SDRX_BE_SERIAL_SCHEDULE"))
            for (int q = 0; q <= 10; q++) {
                const float v = s.step * (float)(1 << q);                                  // exact (power of two)
                if (v == std::floor(v) && v < (float)(1 << 20)) { s.dy_q = q; s.dy_S = (int)v; break; }
            }
        s.ntaps = nt; s.phase_steps = 16;
        s.taps_off = b->taps_off[(size_t)c]; s.filt_mode = k.filt_mode; s.filt_off = b->filt_off[(size_t)c];
        s.discri = k.discri; s.fm_scaling = k.fm_scaling;
        if (s.dy_q >= 0) b->any_dyadic = true;
        s.half = flen / 2;
        for (int i = 0; i < 2; i++) {
            BE_TRY(hipMalloc(reinterpret_cast<void**>(&h.hist[i]), BE_HIST * 4));
            BE_TRY(hipMemsetAsync(h.hist[i], 0, BE_HIST * 4, b->stream));   // on the handle's own (non-blocking) stream: ordered before its kernels
        }
    }
    // schedule columns: channels sorted by tap table, so that a FIR tile of 16 columns normally sees one design
    b->perm.resize((size_t)n_ch); b->col_of.resize((size_t)n_ch);
    for (int c = 0; c < n_ch; c++) b->perm[(size_t)c] = c;
    std::stable_sort(b->perm.begin(), b->perm.end(), [&](int x, int y) { return b->taps_off[(size_t)x] < b->taps_off[(size_t)y]; });
    for (int q = 0; q < n_ch; q++) b->col_of[(size_t)b->perm[(size_t)q]] = q;
    BE_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_perm), (size_t)n_ch * sizeof(int)));
    BE_TRY(hipMemcpy(b->d_perm, b->perm.data(), (size_t)n_ch * sizeof(int), hipMemcpyHostToDevice));
    BE_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_taps), b->taps_all.size() * 4));
    BE_TRY(hipMemcpy(b->d_taps, b->taps_all.data(), b->taps_all.size() * 4, hipMemcpyHostToDevice));
    BE_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_chan), (size_t)n_ch * sizeof(BeChan)));
    BE_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_bufs), (size_t)n_ch * sizeof(BeBufs)));
    BE_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_bufs), (size_t)n_ch * sizeof(BeBufs), hipHostMallocDefault));
    BE_TRY(hipEventCreateWithFlags(&b->bufs_ev, hipEventDisableTiming));
    BE_TRY(hipEventRecord(b->bufs_ev, b->stream));
    BE_TRY(hipEventCreateWithFlags(&b->prod_ev, hipEventDisableTiming));
    BE_TRY(hipEventCreateWithFlags(&b->cons_ev, hipEventDisableTiming));
    BE_TRY(hipMemcpy(b->d_chan, b->h_chan.data(), (size_t)n_ch * sizeof(BeChan), hipMemcpyHostToDevice));
#undef BE_TRY
    *out = b;
    return SDRX_OK;
}

// producer != nullptr: the input samples are being written on that stream.  The schedule kernel (needs the counts only) is
// launched first and overlaps the producer; the readers of `in` (be_mix: NCO mix + raw history) wait for the producer on the
// device, and the producer's stream waits until they are done before it may run anything queued after this call.
static int feed_common(sdrx_backend* b, const int16_t* const* d_iq, const int64_t* n_per_ch, hipStream_t producer = nullptr)
{
    int64_t n_max = 0, n_res_bound = 0;
    for (int c = 0; c < b->n_ch; c++) {
        if (n_per_ch[c] < 0 || n_per_ch[c] > 0x0fffffff) { set_error("sdrx_backend_feed: bad length"); return SDRX_EINVAL; }
        int rc = ensure_capacity(b, c, std::max<int64_t>(n_per_ch[c], 1)); if (rc) return rc;
        n_max = std::max(n_max, n_per_ch[c]);
        // every resampler output after the first two of a stream consumes >= floor(step) inputs (distance >= step before the `-= 1` walk)
        const int64_t per_out = std::max<int64_t>(1, (int64_t)std::floor(b->h_chan[(size_t)c].step));
        n_res_bound = std::max(n_res_bound, n_per_ch[c] / per_out + 4);
    }
    if (n_max + 1024 > b->sched_cap) {
        int64_t cap = b->sched_cap ? b->sched_cap : 8192;
        while (cap < n_max + 1024) cap *= 2;
        SDRX_HIP(hipStreamSynchronize(b->stream));
        int rc = b->sched.reserve((size_t)cap * (size_t)b->n_ch * sizeof(uint2)); if (rc) return rc;
        b->sched_cap = cap;
    }
    // per-feed table (pointers + n_in), pinned, one async copy
    SDRX_HIP(hipEventSynchronize(b->bufs_ev));            // previous feed's copy has read the table
    for (int c = 0; c < b->n_ch; c++) {
        ChanHost& h = b->ch[(size_t)c];
        BeBufs& u = b->h_bufs[c];
        u.in = reinterpret_cast<const uint32_t*>(d_iq[c]);
        u.hist = h.hist[h.cur]; u.hist_next = h.hist[h.cur ^ 1];
        u.mixed = static_cast<float2*>(h.mixed.p); u.sched = static_cast<uint2*>(b->sched.p) + b->col_of[(size_t)c]; u.sched_stride = b->n_ch;
        u.res = static_cast<float2*>(h.res.p); u.head = static_cast<float2*>(h.head.p); u.tail = static_cast<float2*>(h.tail.p);
        u.cplx_out = static_cast<float2*>(h.cplx_out.p); u.real_out = static_cast<float*>(h.real_out.p);
        u.n_in = n_per_ch[c];
    }
    SDRX_HIP(hipMemcpyAsync(b->d_bufs, b->h_bufs, (size_t)b->n_ch * sizeof(BeBufs), hipMemcpyHostToDevice, b->stream));
    SDRX_HIP(hipEventRecord(b->bufs_ev, b->stream));
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (n_max + BE_HIST + 255) / 256));
    // closed-form schedule for dyadic ratios (prep decides per channel and feed), the serial walk for the rest
    hipLaunchKernelGGL(be_sched_dyadic_prep_kernel, dim3((unsigned)((b->n_ch + 63) / 64)), dim3(64), 0, b->stream, b->d_chan, b->d_bufs, b->d_perm, b->n_ch);
    SDRX_HIP(hipGetLastError());
    if (b->any_dyadic) {
        hipLaunchKernelGGL(be_sched_dyadic_fill_kernel, dim3((unsigned)((n_res_bound + 255) / 256), (unsigned)((b->n_ch + 63) / 64)), dim3(256), 0,
                           b->stream, b->d_chan, b->d_bufs, b->d_perm, b->n_ch);
        SDRX_HIP(hipGetLastError());
    }
    // always launched: lanes whose channel the closed form took exit at once (prep's off-grid guard can hand a channel back)
    hipLaunchKernelGGL(be_schedule_kernel, dim3((unsigned)((b->n_ch + 63) / 64)), dim3(64), 0, b->stream, b->d_chan, b->d_bufs, b->d_perm, b->n_ch);
    SDRX_HIP(hipGetLastError());
    if (producer && producer != b->stream) {
        SDRX_HIP(hipEventRecord(b->prod_ev, producer));
        SDRX_HIP(hipStreamWaitEvent(b->stream, b->prod_ev, 0));
    }
    // be_mix stages the 16 KB NCO table per workgroup: at least 8192 samples each
    const unsigned gx_mix = (unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (n_max + BE_HIST + 8191) / 8192));
    hipLaunchKernelGGL(be_mix_kernel, dim3(gx_mix, (unsigned)b->n_ch), dim3(256), 0, b->stream, b->d_chan, b->d_bufs, b->d_nco);
    SDRX_HIP(hipGetLastError());
    if (producer && producer != b->stream) {
        SDRX_HIP(hipEventRecord(b->cons_ev, b->stream));
        SDRX_HIP(hipStreamWaitEvent(producer, b->cons_ev, 0));
    }
    hipLaunchKernelGGL(be_fir_kernel, dim3((unsigned)((n_res_bound + BE_FIR_TO - 1) / BE_FIR_TO), (unsigned)((b->n_ch + BE_FIR_TC - 1) / BE_FIR_TC)), dim3(256), 0,
                       b->stream, b->d_chan, b->d_bufs, b->d_taps, b->d_perm, b->n_ch);
    SDRX_HIP(hipGetLastError());
    if (b->any1024) {
        const unsigned max_blocks = (unsigned)((n_max + BE_FFT) / (BE_FFT / 2) + 1);
        hipLaunchKernelGGL(be_fft_kernel<BE_FFT>, dim3(max_blocks, (unsigned)b->n_ch), dim3(BE_FFT / 8), 0, b->stream, b->d_chan, b->d_bufs, b->d_filters, b->d_utbl);
        SDRX_HIP(hipGetLastError());
    }
    if (b->any2048) {
        const unsigned max_blocks = (unsigned)((n_max + BE_FFT_MAX) / (BE_FFT_MAX / 2) + 1);
        hipLaunchKernelGGL(be_fft_kernel<BE_FFT_MAX>, dim3(max_blocks, (unsigned)b->n_ch), dim3(BE_FFT_MAX / 8), 0, b->stream, b->d_chan, b->d_bufs, b->d_filters, b->d_utbl2);
        SDRX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(be_finish_kernel, dim3(gx, (unsigned)b->n_ch), dim3(256), 0, b->stream, b->d_chan, b->d_bufs);
    SDRX_HIP(hipGetLastError());
    hipLaunchKernelGGL(be_carry_kernel, dim3((unsigned)b->n_ch), dim3(256), 0, b->stream, b->d_chan, b->d_bufs);
    SDRX_HIP(hipGetLastError());
    for (auto& h : b->ch) h.cur ^= 1;
    return SDRX_OK;
}

int sdrx_backend_feed_dev(sdrx_backend_t* b, const int16_t* const* d_iq, const int64_t* n_per_ch)
{
    if (!b || !d_iq || !n_per_ch) { set_error("sdrx_backend_feed_dev: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    return feed_common(b, d_iq, n_per_ch);
}

int sdrx_backend_feed_bank(sdrx_backend_t* b, sdrx_chan_bank_t* bank)
{
    if (!b || !bank) { set_error("sdrx_backend_feed_bank: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    void* ps = nullptr;
    int rc = sdrx_chan_bank_get_stream(bank, &ps); if (rc) return rc;
    std::vector<const int16_t*> d((size_t)b->n_ch);
    std::vector<int64_t> n((size_t)b->n_ch);
    for (int c = 0; c < b->n_ch; c++) {
        rc = sdrx_chan_bank_last_dev(bank, c, &d[(size_t)c], &n[(size_t)c]);
        if (rc) { set_error("sdrx_backend_feed_bank: the bank has fewer channels than the back-end"); return rc; }
    }
    return feed_common(b, d.data(), n.data(), static_cast<hipStream_t>(ps));
}

int sdrx_backend_feed(sdrx_backend_t* b, const int16_t* const* iq, const int64_t* n_per_ch)
{
    if (!b || !iq || !n_per_ch) { set_error("sdrx_backend_feed: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    std::vector<const int16_t*> d((size_t)b->n_ch);
    for (int c = 0; c < b->n_ch; c++) {
        ChanHost& h = b->ch[(size_t)c];
        int rc = h.stage_in.reserve((size_t)std::max<int64_t>(n_per_ch[c], 1) * 4); if (rc) return rc;
        if (n_per_ch[c] > 0) SDRX_HIP(hipMemcpyAsync(h.stage_in.p, iq[c], (size_t)n_per_ch[c] * 4, hipMemcpyHostToDevice, b->stream));
        d[(size_t)c] = static_cast<const int16_t*>(h.stage_in.p);
    }
    return feed_common(b, d.data(), n_per_ch);
}

int64_t sdrx_backend_read(sdrx_backend_t* b, int32_t c, float* out, int64_t cap_floats)
{
    if (!b || c < 0 || c >= b->n_ch || cap_floats < 0 || (cap_floats > 0 && !out)) { set_error("sdrx_backend_read: bad argument"); return SDRX_EINVAL; }
    if (hipSetDevice(b->device) != hipSuccess) return SDRX_EHIP;
    BeChan s;
    hipError_t e = hipMemcpyAsync(&s, b->d_chan + c, sizeof s, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return hip_fail(e, "read state", __FILE__, __LINE__);
    const bool real = s.discri != 0;
    int64_t n_floats = (int64_t)s.n_out * (real ? 1 : 2);
    if (n_floats > cap_floats) n_floats = cap_floats;
    if (n_floats == 0) return 0;
    const void* src = real ? b->ch[(size_t)c].real_out.p : b->ch[(size_t)c].cplx_out.p;
    e = hipMemcpy(out, src, (size_t)n_floats * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "read data", __FILE__, __LINE__);
    return n_floats;
}

int sdrx_backend_get_design(sdrx_backend_t* b, int32_t c, int32_t* ntaps_per_phase, float* taps, int32_t taps_cap,
                            float* filter_iq, int32_t* nco_inc)
{
    if (!b || c < 0 || c >= b->n_ch) { set_error("sdrx_backend_get_design: bad channel"); return SDRX_EINVAL; }
    const int nt = b->ntaps[(size_t)c];
    if (ntaps_per_phase) *ntaps_per_phase = nt;
    if (taps) std::memcpy(taps, &b->taps_all[(size_t)b->taps_off[(size_t)c]], (size_t)std::min(taps_cap, nt * 16) * 4);
    if (filter_iq) std::memcpy(filter_iq, &b->filters_all[(size_t)c * 2 * BE_FFT_MAX * 2], BE_FFT_MAX * 8);
    if (nco_inc) *nco_inc = b->h_chan[(size_t)c].nco_inc;
    return SDRX_OK;
}

int sdrx_backend_sync(sdrx_backend_t* b)
{
    if (!b) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(b->device));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    return SDRX_OK;
}

} // extern "C"
