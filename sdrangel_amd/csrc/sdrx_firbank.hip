// libsdrx.so: sdrx_firbank_* -- Lowpass<Real> / Bandpass<Real> audio FIRs (sdrbase/dsp/lowpass.h, bandpass.h), the
// 301-tap filters of the NFM demod's audio tail (nfmdemod.cpp:88,239,279,428-429), for N channels at once.
// The reference's ring walk (lowpass.h:55-99) sums, with x[n] the newest sample, N taps and h = N/2,
//     y[n] = (x[n] + x[n-1]) t[0] + sum_{i=1}^{h-1} (x[n-N+i] + x[n-1-i]) t[i] + x[n-N+h] t[h]
// in that order in float (x[n-N] has just been overwritten by x[n], hence the odd first pair); one lane per output
// sample keeps exactly that order.  Tap design on the host with the reference's float/double mix.
#include "sdrx_common.hpp"
#include <vector>
#include <cmath>
#include <cstring>
#include <new>
#include <algorithm>

using namespace sdrx;

namespace {

struct FirChan { const float* in; float* out; float* hist; float* hist_next; long n; int N, h, taps_off, pad; };

__global__ void fir_fold_kernel(const FirChan* __restrict__ ch, const float* __restrict__ taps)
{
    const FirChan c = ch[blockIdx.y];
    const float* t = taps + c.taps_off;
    const int N = c.N, h = c.h;
    auto x = [&](long k) -> float { return k >= 0 ? c.in[k] : c.hist[N + k]; };     // hist[0..N): oldest first
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < c.n; k += (long)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        acc += (x(k) + x(k - 1)) * t[0];
        for (int i = 1; i < h; i++) acc += (x(k - N + i) + x(k - 1 - i)) * t[i];
        acc += x(k - N + h) * t[h];
        c.out[k] = acc;
    }
}

__global__ void fir_hist_kernel(const FirChan* __restrict__ ch)
{
    const FirChan c = ch[blockIdx.x];
    for (int i = threadIdx.x; i < c.N; i += blockDim.x) {
        const long src = (long)i + c.n - c.N;
        c.hist_next[i] = src >= 0 ? c.in[src] : c.hist[i + c.n];
    }
}

const double PI_D = 3.14159265358979323846;

void design(const sdrx_fir_cfg& k, std::vector<float>& t, int* N_out)
{
    int ntaps = k.ntaps;
    if (!(ntaps & 1)) ntaps++;                                 // "has to have an odd number of taps"
    const int nt = ntaps / 2 + 1;
    t.assign((size_t)nt, 0.0f);
    const double mid = ((double)ntaps - 1.0) / 2.0, rate = k.sample_rate;
    if (k.kind == 0) {                                         // Lowpass::create (lowpass.h:16-52)
        const double Wc = 2.0 * PI_D * (double)k.f1 / rate;
        for (int i = 0; i < nt; i++)
            t[(size_t)i] = (i == (ntaps - 1) / 2) ? (float)(Wc / PI_D) : (float)(std::sin(((double)i - mid) * Wc) / (((double)i - mid) * PI_D));
        for (int i = 0; i < nt; i++) t[(size_t)i] = (float)(t[(size_t)i] * (0.54 + 0.46 * std::cos((2.0 * PI_D * ((double)i - mid)) / (double)ntaps)));
    } else {                                                   // Bandpass::create (bandpass.h:14-75)
        const double Wcl = 2.0 * PI_D * (double)k.f1 / rate, Wch = 2.0 * PI_D * (double)k.f2 / rate;
        std::vector<float> lp((size_t)nt), hp((size_t)nt);
        for (int i = 0; i < nt; i++) {
            if (i == (ntaps - 1) / 2) { lp[(size_t)i] = (float)(Wch / PI_D); hp[(size_t)i] = (float)(-(Wcl / PI_D)); }
            else {
                lp[(size_t)i] = (float)(std::sin(((double)i - mid) * Wch) / (((double)i - mid) * PI_D));
                hp[(size_t)i] = (float)(-std::sin(((double)i - mid) * Wcl) / (((double)i - mid) * PI_D));
            }
        }
        hp[(size_t)((ntaps - 1) / 2)] += 1;
        for (int i = 0; i < nt; i++) {
            const double w = 0.54 + 0.46 * std::cos((2.0 * PI_D * ((double)i - mid)) / (double)ntaps);
            lp[(size_t)i] = (float)(lp[(size_t)i] * w); hp[(size_t)i] = (float)(hp[(size_t)i] * w);
            t[(size_t)i] = -(lp[(size_t)i] + hp[(size_t)i]);
        }
        t[(size_t)((ntaps - 1) / 2)] += 1;
    }
    float sum = 0; int i;
    for (i = 0; i < nt - 1; i++) sum += t[(size_t)i] * 2;
    sum += t[(size_t)i];
    for (i = 0; i < nt; i++) t[(size_t)i] /= sum;
    *N_out = ntaps;
}

} // namespace

struct sdrx_firbank {
    int device = 0, n_ch = 0;
    hipStream_t stream = nullptr;
    std::vector<int> N, taps_off;
    std::vector<float> taps_all;
    float* d_taps = nullptr;
    std::vector<float*> hist[2];
    int cur = 0;
    std::vector<DevBuf> d_in, d_out;
    FirChan* d_ch = nullptr; FirChan* h_ch = nullptr;
};

extern "C" {

int sdrx_firbank_destroy(sdrx_firbank_t* b)
{
    if (!b) return SDRX_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (int q = 0; q < 2; q++) for (float* p : b->hist[q]) if (p) (void)hipFree(p);
    for (auto& d : b->d_in) d.release();
    for (auto& d : b->d_out) d.release();
    if (b->d_taps) (void)hipFree(b->d_taps);
    if (b->d_ch) (void)hipFree(b->d_ch);
    if (b->h_ch) (void)hipHostFree(b->h_ch);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return SDRX_OK;
}

int sdrx_firbank_create(sdrx_firbank_t** out, int device, int32_t n_ch, const sdrx_fir_cfg* cfg)
{
    if (!out) { set_error("sdrx_firbank_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    if (n_ch <= 0 || !cfg) { set_error("sdrx_firbank_create: bad argument"); return SDRX_EINVAL; }
    for (int c = 0; c < n_ch; c++)
        if (cfg[c].kind < 0 || cfg[c].kind > 1 || cfg[c].ntaps < 3 || cfg[c].ntaps > 4096 || cfg[c].sample_rate <= 0) {
            set_error("sdrx_firbank_create: kind 0|1, ntaps 3..4096, sample_rate > 0"); return SDRX_EINVAL;
        }
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_firbank* b = new (std::nothrow) sdrx_firbank;
    if (!b) return SDRX_ENOMEM;
    b->device = device; b->n_ch = n_ch;
#define FB_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = hip_fail(e_, #call, __FILE__, __LINE__); sdrx_firbank_destroy(b); return r_; } } while (0)
    FB_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    b->N.resize((size_t)n_ch); b->taps_off.resize((size_t)n_ch);
    b->hist[0].assign((size_t)n_ch, nullptr); b->hist[1].assign((size_t)n_ch, nullptr);
    b->d_in.resize((size_t)n_ch); b->d_out.resize((size_t)n_ch);
    for (int c = 0; c < n_ch; c++) {
        std::vector<float> t; int N = 0;
        design(cfg[c], t, &N);
        b->N[(size_t)c] = N; b->taps_off[(size_t)c] = (int)b->taps_all.size();
        b->taps_all.insert(b->taps_all.end(), t.begin(), t.end());
        for (int q = 0; q < 2; q++) {
            FB_TRY(hipMalloc(reinterpret_cast<void**>(&b->hist[q][(size_t)c]), (size_t)N * 4));
            FB_TRY(hipMemsetAsync(b->hist[q][(size_t)c], 0, (size_t)N * 4, b->stream));   // the handle's own stream orders it before the kernels
        }
    }
    FB_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_taps), b->taps_all.size() * 4));
    FB_TRY(hipMemcpy(b->d_taps, b->taps_all.data(), b->taps_all.size() * 4, hipMemcpyHostToDevice));
    FB_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_ch), (size_t)n_ch * sizeof(FirChan)));
    FB_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_ch), (size_t)n_ch * sizeof(FirChan), hipHostMallocDefault));
#undef FB_TRY
    *out = b;
    return SDRX_OK;
}

int sdrx_firbank_get_taps(const sdrx_firbank_t* b, int32_t c, float* taps, int32_t cap)
{
    if (!b || c < 0 || c >= b->n_ch) { set_error("sdrx_firbank_get_taps: bad channel"); return SDRX_EINVAL; }
    const int nt = b->N[(size_t)c] / 2 + 1;
    if (taps) std::memcpy(taps, &b->taps_all[(size_t)b->taps_off[(size_t)c]], (size_t)std::min(cap, nt) * 4);
    return nt;
}

// in[c] / out[c]: n_per_ch[c] host floats each; state (the last N inputs of every channel) is carried
int sdrx_firbank_feed(sdrx_firbank_t* b, const float* const* in, const int64_t* n_per_ch, float* const* out)
{
    if (!b || !in || !n_per_ch || !out) { set_error("sdrx_firbank_feed: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    int64_t n_max = 0;
    for (int c = 0; c < b->n_ch; c++) {
        const int64_t n = n_per_ch[c];
        if (n < 0) { set_error("sdrx_firbank_feed: negative length"); return SDRX_EINVAL; }
        int rc = b->d_in[(size_t)c].reserve((size_t)std::max<int64_t>(n, 1) * 4); if (rc) return rc;
        rc = b->d_out[(size_t)c].reserve((size_t)std::max<int64_t>(n, 1) * 4); if (rc) return rc;
        if (n) SDRX_HIP(hipMemcpyAsync(b->d_in[(size_t)c].p, in[c], (size_t)n * 4, hipMemcpyHostToDevice, b->stream));
        FirChan& k = b->h_ch[c];
        k.in = static_cast<const float*>(b->d_in[(size_t)c].p); k.out = static_cast<float*>(b->d_out[(size_t)c].p);
        k.hist = b->hist[b->cur][(size_t)c]; k.hist_next = b->hist[b->cur ^ 1][(size_t)c];
        k.n = n; k.N = b->N[(size_t)c]; k.h = k.N / 2; k.taps_off = b->taps_off[(size_t)c]; k.pad = 0;
        n_max = std::max(n_max, n);
    }
    SDRX_HIP(hipMemcpyAsync(b->d_ch, b->h_ch, (size_t)b->n_ch * sizeof(FirChan), hipMemcpyHostToDevice, b->stream));
    if (n_max > 0) {
        const unsigned gx = (unsigned)std::min<int64_t>(1024, (n_max + 255) / 256);
        hipLaunchKernelGGL(fir_fold_kernel, dim3(gx, (unsigned)b->n_ch), dim3(256), 0, b->stream, b->d_ch, b->d_taps);
        SDRX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(fir_hist_kernel, dim3((unsigned)b->n_ch), dim3(256), 0, b->stream, b->d_ch);
    SDRX_HIP(hipGetLastError());
    for (int c = 0; c < b->n_ch; c++)
        if (n_per_ch[c]) SDRX_HIP(hipMemcpyAsync(out[c], b->d_out[(size_t)c].p, (size_t)n_per_ch[c] * 4, hipMemcpyDeviceToHost, b->stream));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    b->cur ^= 1;
    return SDRX_OK;
}

} // extern "C"
