// Channel back-end kernels: the first lines of every channelrx demod's feed()
// (plugins/channelrx/demodnfm/nfmdemod.cpp:150-163, demodssb/ssbdemod.cpp:158-172):
//     c = Complex(re, im) * m_nco.nextIQ();                       NCO            sdrbase/dsp/nco.cpp:30-64
//     if (m_interpolator.decimate(&dist, c, &ci)) { ...; dist += step }   Interpolator   interpolator.h:23-36,182-195
//     n = SSBFilter->runSSB(ci, &sideband, usb)                    fftfilt        fftfilt.cpp:261-361 (g_fft, gfft.h)
//     demod = m_phaseDiscri.phaseDiscriminatorDelta(...)           discriminator  phasediscri.h:50-78,172-197
//
// Float path: the parity bar is <= 1 ulp against the strict-IEEE scalar reference build, so every
// expression keeps the reference's operand order and the file is compiled with -ffp-contract=off
// (no FMA contraction), correctly rounded division, denormals on.  Cosine/twiddle/tap tables are
// computed on the host with the same double-precision libm calls the reference makes and uploaded.
//
// These streams run at channel rate (tens of kS/s per channel); they are latency/launch bound, not
// bandwidth bound, so the kernels are written for exactness and clarity: one lane per output sample
// for the FIR (sequential tap order), one 128-thread workgroup per 1024-point FFT block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrx {

constexpr int BE_HIST = 256;            // raw samples of history kept per channel (>= taps per phase)
constexpr int BE_NCO_N = 4096;
constexpr int BE_FFT = 1024;            // SSB fftfilt length (ssbFftLen, ssbdemod.h:36); the DSB filter runs at 2 * BE_FFT
constexpr int BE_FFT_MAX = 2048;

struct BeChan {                         // device resident: config + carried state of one channel
    // --- config
    int nco_inc;                        // NCO::setFreq: (int)((freq * 4096) / rate)
    float step;                         // (Real) inRate / (Real) outRate
    int ntaps;                          // taps per phase
    int phase_steps;
    int taps_off;                       // float offset into the taps table: [phase][ntaps]
    int filt_mode;                      // 0 none, 1 runFilt, 2 runSSB usb, 3 runSSB lsb, 4 runDSB (fft length 2048)
    int filt_off;                       // complex offset into the filter table (BE_FFT_MAX entries per channel)
    int discri;                         // 0 none, 1 phaseDiscriminatorDelta, 2 phaseDiscriminator
    float fm_scaling;
    // --- state
    int nco_phase;                      // phase after the last consumed input sample
    float distance;
    int pending;                        // resampled samples waiting for a full fftfilt block
    float prev_arg, m1r, m1i;
    // --- per feed (written by the kernels)
    int n_in;                           // new input samples of this feed            (copied from BeBufs by be_schedule)
    int n_res;                          // resampler outputs of this feed            (written by be_schedule)
    int n_blocks;                       // complete fftfilt blocks of this feed       (written by be_schedule)
    int n_out;                          // samples in the output buffer after this feed
    int half;                           // fftfilt block = flen / 2: 512, or 1024 for runDSB
};

struct BeBufs {                         // per channel device pointers (per feed capacity ensured by the host)
    const uint32_t* in;                 // n_in packed Samples
    uint32_t* hist;                     // BE_HIST packed Samples: tail of the previous feeds
    uint32_t* hist_next;
    float2* mixed;                      // BE_HIST + n_in: NCO-mixed samples, index BE_HIST + k
    uint32_t* sched;                    // per resampler output o: sched[o * sched_stride] = k * 16 + phase (channel-interleaved
                                        // so that the serial schedule lanes of a wave store to neighbouring addresses)
    float2* res;                        // [pending | new resampler outputs]
    float2* head;                       // n_blocks * 512
    float2* tail;                       // (1 + n_blocks) * 512; slot 0 = ovlbuf carried from the previous feed
    float2* cplx_out;                   // output when the last stage is complex (resampler or fftfilt)
    float* real_out;                    // output when a discriminator is enabled
    long n_in;                          // new input samples of this feed (host -> device with the pointers, one copy)
    long sched_stride;                  // = number of channels
};

// ---- 1. schedule: the float `distance` recurrence, one lane per channel (serial by nature).
// Walks emission by emission instead of input by input: while d >= 1 the reference's `d -= 1.0` is exact, so
// after an emission leaves d_new the next one happens m = max(1, floor(d_new)) inputs later with
// d = d_new - m (exact) -- or fl(d_new - 1) when d_new < 1 -- which is bit-for-bit what the per-input loop holds.
__global__ void be_schedule_kernel(BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs, int n_ch)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    BeChan& s = ch[c];
    uint32_t* sched = bufs[c].sched;
    const long stride = bufs[c].sched_stride;
    const int n_in = (int)bufs[c].n_in;
    s.n_in = n_in;
    float d = s.distance;                                   // value BEFORE the next input's `-= 1.0`
    const float step = s.step, ps = (float)s.phase_steps;
    int cnt = 0;
    long k = -1;                                            // index of the last consumed input
    for (;;) {
        // inputs until the next emission
        const int m = d >= 1.0f ? (int)floorf(d) : 1;
        if (k + m >= n_in) {                                // the feed ends first: consume what is left, no emission
            const int left = (int)(n_in - 1 - k);           // inputs still to consume (each: d -= 1, all stay >= 1 or it would emit)
            d = d - (float)left;                            // exact: d - left >= 1 unless left == 0
            break;
        }
        k += m;
        d = d >= 1.0f ? d - (float)m : d - 1.0f;
        int ph = (int)floorf(d * ps);
        if (ph < 0) ph = 0;
        sched[(long)cnt * stride] = (uint32_t)k * 16u + (uint32_t)ph;
        cnt++;
        d = d + step;
    }
    s.distance = d;
    s.n_res = cnt;
    s.n_blocks = s.filt_mode ? (s.pending + cnt) / s.half : 0;
}

// ---- 2a. NCO mix of history + new samples into float (exact: int16 -> float, complex product)
__global__ void be_mix_kernel(const BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs, const float* __restrict__ nco_tbl)
{
    const int c = blockIdx.y;
    const BeChan s = ch[c];
    const BeBufs b = bufs[c];
    const int total = BE_HIST + s.n_in;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i - BE_HIST;                          // sample index relative to this feed; < 0: history
        const uint32_t v = j < 0 ? b.hist[i] : b.in[j];
        // phase after nextPhase() for this sample; s.nco_phase is still the phase BEFORE this feed
        long p = ((long)s.nco_phase + (long)(j + 1) * (long)s.nco_inc) % BE_NCO_N;
        if (p < 0) p += BE_NCO_N;
        const float o_r = nco_tbl[p], o_i = -nco_tbl[(p + BE_NCO_N / 4) % BE_NCO_N];
        const float a = (float)(int16_t)(v & 0xffffu), q = (float)(int16_t)(v >> 16);
        float2 m; m.x = a * o_r - q * o_i; m.y = a * o_i + q * o_r;         // std::complex<float> operator*=
        b.mixed[i] = m;
    }
}

// ---- 2b. polyphase FIR, one lane per output, taps summed newest-first like the ring walk
__global__ void be_fir_kernel(const BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs, const float* __restrict__ taps)
{
    const int c = blockIdx.y;
    const BeChan s = ch[c];
    const BeBufs b = bufs[c];
    const bool direct = s.filt_mode == 0 && s.discri == 0;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < s.n_res; o += gridDim.x * blockDim.x) {
        const uint32_t e = b.sched[(long)o * b.sched_stride];
        const int k = (int)(e >> 4), ph = (int)(e & 15u);
        const float* t = taps + s.taps_off + ph * s.ntaps;
        const float2* x = b.mixed + BE_HIST + k;
        float ra = 0.0f, ia = 0.0f;
        for (int i = 0; i < s.ntaps; i++) {
            const float2 v = x[-i];
            ra += t[i] * v.x;
            ia += t[i] * v.y;
        }
        float2 r; r.x = ra; r.y = ia;
        b.res[s.pending + o] = r;
        if (direct) b.cplx_out[o] = r;
    }
}

// ---- g_fft network on a 1024-point block held in LDS.
// John Green's FFT as the reference runs it for N = 1024 (gfft.h: ffts1 :1189-1224, iffts1 :2238-2275):
// bit-reversed load fused with one radix-2 stage (bitrevR2 :185-317; scbitrevR2 :1231-1363 scales by 1/N),
// then three radix-8 passes (bfstages :843-1158 / ibfstages :1889-2209) with D = 2, 16, 128.  The reference's
// in-place index choreography does not change values; the arithmetic FORMS do.  With multiplier m = (mr, mi):
//     PLUS (a,b,m): r = (a.r + b.r*mr) - b.i*mi ;  i = (a.i + b.r*mi) + b.i*mr      (= a + b*m)
//     MINUS(a,b,m): r = (a.r - b.r*mr) + b.i*mi ;  i = (a.i - b.r*mi) - b.i*mr      (= a - b*m)
//     and the partner of every butterfly is formed as 2*a - result.
// Forward multipliers are conj(w), i*conj(w); inverse ones their conjugates (IEEE negation is exact, so a
// sign flip of mi reproduces the reference's explicit +/- variants bit for bit).  Twiddles come from the
// quarter-wave float cosine table of fftCosInit (:141-150); w0 crosses pi/2 at u = D/2 and is mirrored.
__device__ __forceinline__ float2 c_plus(float2 a, float2 b, float mr, float mi)  { float2 t; t.x = (a.x + b.x * mr) - b.y * mi; t.y = (a.y + b.x * mi) + b.y * mr; return t; }
__device__ __forceinline__ float2 c_minus(float2 a, float2 b, float mr, float mi) { float2 t; t.x = (a.x - b.x * mr) + b.y * mi; t.y = (a.y - b.x * mi) - b.y * mr; return t; }
__device__ __forceinline__ float2 c_two_minus(float2 a, float2 t)                 { float2 f; f.x = a.x * 2.0f - t.x; f.y = a.y * 2.0f - t.y; return f; }
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) { float2 t; t.x = a.x * b.x - a.y * b.y; t.y = a.x * b.y + a.y * b.x; return t; }

// y: LDS work array (N float2); x: LDS input copy (N float2); u: cosine table (N/4+1 floats); N/8 threads.
// N = 1024: stage list R2(bitrev) + 3 x radix-8 (D = 2, 16, 128);  N = 2048: R2(bitrev) + bfR2 (:531-635 /
// ibfR2 :1577-1681: twiddles 1 and -/+i) + 3 x radix-8 (D = 4, 32, 256).  inverse: first stage scaled by 1/N,
// multipliers conjugated.
template<int N, bool INVERSE>
__device__ __forceinline__ void gfft(float2* __restrict__ y, const float2* __restrict__ x, const float* __restrict__ u, int tid)
{
    constexpr int NT = N / 8;
    constexpr int M = N == 1024 ? 10 : 11;
    static_assert(N == 1024 || N == 2048, "fftfilt lengths of the demods");
    const float scale = (float)(1.0 / N);
    for (int j = tid; j < N / 2; j += NT) {
        const unsigned r = __brev((unsigned)(2 * j)) >> (32 - M);
        const float2 a = x[r], b = x[r + N / 2];
        float2 s, d; s.x = a.x + b.x; s.y = a.y + b.y; d.x = a.x - b.x; d.y = a.y - b.y;
        if (INVERSE) { s.x = scale * s.x; s.y = scale * s.y; d.x = scale * d.x; d.y = scale * d.y; }
        y[2 * j] = s; y[2 * j + 1] = d;
    }
    __syncthreads();
    constexpr int D0 = N == 1024 ? 2 : 4;
    if constexpr (N == 2048) {
        for (int k = 4 * tid; k < N; k += 4 * NT) {
            const float2 a = y[k], b = y[k + 2], c = y[k + 1], d = y[k + 3];
            float2 t;
            t.x = a.x + b.x; t.y = a.y + b.y; y[k] = t;
            t.x = a.x - b.x; t.y = a.y - b.y; y[k + 2] = t;
            if (!INVERSE) { t.x = c.x + d.y; t.y = c.y - d.x; y[k + 1] = t; t.x = c.x - d.y; t.y = c.y + d.x; y[k + 3] = t; }
            else          { t.x = c.x - d.y; t.y = c.y + d.x; y[k + 1] = t; t.x = c.x + d.y; t.y = c.y - d.x; y[k + 3] = t; }
        }
        __syncthreads();
    }
    const float sg = INVERSE ? 1.0f : -1.0f;
#pragma unroll
    for (int D = D0; D < N; D *= 8) {
        const int uinc = N / 8 / D;
        const int uu = tid % D, g = tid / D;                           // N/8 butterflies per pass
        const int i2 = uu * uinc, i1 = 2 * i2;
        int i0 = 4 * i2; float w0r;
        if (uu < D / 2) w0r = u[i0]; else { i0 = N / 2 - i0; w0r = -u[i0]; }
        const float w0i = u[N / 4 - i0];
        const float w1r = u[i1], w1i = u[N / 4 - i1];
        const float w2r = u[i2], w2i = u[N / 4 - i2];
        const float w3r = u[i2 + N / 8], w3i = u[N / 4 - i2 - N / 8];
        float2* p = y + g * 8 * D + uu;
        float2 f0 = p[0], f1 = p[D], f2 = p[2 * D], f3 = p[3 * D], f4 = p[4 * D], f5 = p[5 * D], f6 = p[6 * D], f7 = p[7 * D];
        float2 t0, t1;
        t0 = c_plus(f0, f1, w0r, sg * w0i);  f1 = c_two_minus(f0, t0);
        t1 = c_minus(f2, f3, w0r, sg * w0i); f2 = c_two_minus(f2, t1);
        f0 = c_plus(t0, f2, w1r, sg * w1i);  f2 = c_two_minus(t0, f0);
        f3 = c_plus(f1, t1, w1i, -sg * w1r); f1 = c_two_minus(f1, f3);
        t0 = c_plus(f4, f5, w0r, sg * w0i);  f5 = c_two_minus(f4, t0);
        t1 = c_minus(f6, f7, w0r, sg * w0i); f6 = c_two_minus(f6, t1);
        f4 = c_plus(t0, f6, w1r, sg * w1i);  f6 = c_two_minus(t0, f4);
        f7 = c_plus(f5, t1, w1i, -sg * w1r); f5 = c_two_minus(f5, f7);
        t0 = c_minus(f0, f4, w2r, sg * w2i); f0 = c_two_minus(f0, t0);
        t1 = c_minus(f1, f5, w3r, sg * w3i); f1 = c_two_minus(f1, t1);
        const float2 n4 = c_minus(f2, f6, w2i, -sg * w2r); f6 = c_two_minus(f2, n4);
        const float2 n5 = c_minus(f3, f7, w3i, -sg * w3r); f7 = c_two_minus(f3, n5);
        p[0] = f0; p[D] = f1; p[2 * D] = n4; p[3 * D] = n5; p[4 * D] = t0; p[5 * D] = t1; p[6 * D] = f6; p[7 * D] = f7;
        __syncthreads();
    }
}

// ---- 3. fftfilt block: zero-padded forward FFT, filter, inverse FFT; writes head / tail halves.
// One instantiation per FFT length; a workgroup whose channel uses the other length exits at once.
template<int N>
__global__ __launch_bounds__(N / 8)
void be_fft_kernel(const BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs,
                   const float2* __restrict__ filters, const float* __restrict__ utbl)
{
    constexpr int NT = N / 8, H = N / 2;
    __shared__ float2 xa[N], ya[N];
    __shared__ float us[N / 4 + 1];
    const int c = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const BeChan s = ch[c];
    if (s.half != H || blk >= s.n_blocks) return;
    const BeBufs b = bufs[c];
    for (int i = tid; i <= N / 4; i += NT) us[i] = utbl[i];
    for (int i = tid; i < H; i += NT) { xa[i] = b.res[blk * H + i]; xa[H + i] = make_float2(0.0f, 0.0f); }
    __syncthreads();
    gfft<N, false>(ya, xa, us, tid);
    const float2* filt = filters + s.filt_off;
    for (int i = tid; i < H; i += NT) {
        float2 lo = ya[i], hi = ya[H + i];
        if (s.filt_mode == 1 || s.filt_mode == 4) { lo = c_mul(lo, filt[i]); hi = c_mul(hi, filt[H + i]); }   // runFilt / runDSB (getDC)
        else if (i == 0) { lo = c_mul(lo, filt[0]); /* bin N/2 is left untouched (fftfilt.cpp:294-311) */ }
        else if (s.filt_mode == 2) { lo = c_mul(lo, filt[i]); hi = make_float2(0.0f, 0.0f); }    // usb
        else { lo = make_float2(0.0f, 0.0f); hi = c_mul(hi, filt[H + i]); }                      // lsb
        xa[i] = lo; xa[H + i] = hi;
    }
    __syncthreads();
    gfft<N, true>(ya, xa, us, tid);
    for (int i = tid; i < H; i += NT) {
        b.head[blk * H + i] = ya[i];
        b.tail[(blk + 1) * H + i] = ya[H + i];
    }
}

// phasediscri.h:172-197
__device__ __forceinline__ float atan2_approx2(float y, float x)
{
    const float PI_F = 3.14159265f, PIBY2_F = 1.5707963f;
    if (x == 0.0f) { if (y > 0.0f) return PIBY2_F; if (y == 0.0f) return 0.0f; return -PIBY2_F; }
    float at;
    const float z = y / x;
    if (fabsf(z) < 1.0f) {
        at = z / (1.0f + 0.28f * z * z);
        if (x < 0.0f) { if (y < 0.0f) return at - PI_F; return at + PI_F; }
    } else {
        at = PIBY2_F - z / (z * z + 0.28f);
        if (y < 0.0f) return at - PI_F;
    }
    return at;
}

// ---- 4. overlap-add + discriminator.  Sample j of the feed's filtered stream = tail[blk][i] + head[blk][i]
// (tail slot 0 = ovlbuf of the previous feed).  Without a filter the stream is the resampler output.
__global__ void be_finish_kernel(BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs)
{
    const int c = blockIdx.y;
    const BeChan s = ch[c];
    const BeBufs b = bufs[c];
    const int H = s.half;
    const int n = s.filt_mode ? s.n_blocks * H : s.n_res;
    auto sample = [&](int j) -> float2 {
        if (!s.filt_mode) return b.res[s.pending + j];
        const float2 o = b.tail[j], h = b.head[j];             // tail slots are shifted by one block
        float2 r; r.x = o.x + h.x; r.y = o.y + h.y;            // output[i] = ovlbuf[i] + data[i]
        return r;
    };
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float2 v = sample(j);
        if (s.discri == 0) { b.cplx_out[j] = v; continue; }
        if (s.discri == 1) {
            const float cur = atan2_approx2(v.y, v.x);
            const float prev = j == 0 ? s.prev_arg : [&] { const float2 pv = sample(j - 1); return atan2_approx2(pv.y, pv.x); }();
            float dev = (float)((double)(cur - prev) / 3.14159265358979323846);
            if (dev < -1.0f) dev += 2.0f; else if (dev > 1.0f) dev -= 2.0f;
            b.real_out[j] = dev * s.fm_scaling;
        } else {
            float2 m1; if (j == 0) { m1.x = s.m1r; m1.y = s.m1i; } else m1 = sample(j - 1);
            const float dr = m1.x * v.x - (-m1.y) * v.y, di = m1.x * v.y + (-m1.y) * v.x;    // conj(prev) * cur
            b.real_out[j] = (float)(((double)atan2f(di, dr) / 3.14159265358979323846) * (double)s.fm_scaling);
        }
    }
}

// ---- 5. carry state to the next feed (one workgroup per channel; runs after everything else)
__global__ __launch_bounds__(256)
void be_carry_kernel(BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs)
{
    const int c = blockIdx.x, tid = threadIdx.x;
    BeChan& s = ch[c];
    const BeBufs b = bufs[c];
    const int H = s.half;
    const int n_in = s.n_in, n_res = s.n_res, pending = s.pending, nb = s.n_blocks;
    const int n = s.filt_mode ? nb * H : n_res;
    // discriminator memory = last sample of the finished stream
    float prev_arg = s.prev_arg, m1r = s.m1r, m1i = s.m1i;
    if (tid == 0 && n > 0 && s.discri) {
        float2 v;
        if (!s.filt_mode) v = b.res[pending + n - 1];
        else { const float2 o = b.tail[n - 1], h = b.head[n - 1]; v.x = o.x + h.x; v.y = o.y + h.y; }
        prev_arg = atan2_approx2(v.y, v.x); m1r = v.x; m1i = v.y;
    }
    // raw history: last BE_HIST of (old history ++ new input)
    for (int i = tid; i < BE_HIST; i += 256) {
        const long src = (long)i + n_in - BE_HIST;
        b.hist_next[i] = src >= 0 ? b.in[src] : b.hist[i + n_in];
    }
    // fftfilt: unconsumed resampler outputs move to the front; ovlbuf = tail of the last block
    const int left = s.filt_mode ? (pending + n_res) - nb * H : 0;
    float2 keep[4];
    for (int q = 0; q < 4; q++) { const int i = q * 256 + tid; keep[q] = (s.filt_mode && i < left) ? b.res[nb * H + i] : make_float2(0.0f, 0.0f); }
    float2 ov[4];
    for (int q = 0; q < 4; q++) { const int i = q * 256 + tid; ov[q] = (s.filt_mode && nb > 0 && i < H) ? b.tail[nb * H + i] : make_float2(0.0f, 0.0f); }
    __syncthreads();
    for (int q = 0; q < 4; q++) { const int i = q * 256 + tid; if (s.filt_mode && i < left) b.res[i] = keep[q]; }
    if (s.filt_mode && nb > 0) for (int q = 0; q < 4; q++) { const int i = q * 256 + tid; if (i < H) b.tail[i] = ov[q]; }
    if (tid == 0) {
        long p = ((long)s.nco_phase + (long)n_in * (long)s.nco_inc) % BE_NCO_N;
        if (p < 0) p += BE_NCO_N;
        s.nco_phase = (int)p;
        s.pending = left;
        s.prev_arg = prev_arg; s.m1r = m1r; s.m1i = m1i;
        s.n_out = n;
    }
}

// single forward FFT of one block (filter design: fft->ComplexFFT(filter), fftfilt.cpp:131,158)
template<int N>
__global__ __launch_bounds__(N / 8)
void be_fft_design_kernel(float2* __restrict__ data, const float* __restrict__ utbl)
{
    constexpr int NT = N / 8;
    __shared__ float2 xa[N], ya[N];
    __shared__ float us[N / 4 + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i <= N / 4; i += NT) us[i] = utbl[i];
    for (int i = tid; i < N; i += NT) xa[i] = data[i];
    __syncthreads();
    gfft<N, false>(ya, xa, us, tid);
    for (int i = tid; i < N; i += NT) data[i] = ya[i];
}

} // namespace sdrx
