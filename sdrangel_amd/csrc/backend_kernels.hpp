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
// These streams run at channel rate (tens of kS/s per channel, hundreds of channels per device stream).  What bounds
// them: the resampler schedule is a serial float recurrence (closed form when the ratio is dyadic), the
// polyphase FIR is 72 taps per output and runs from LDS (tile of 64 outputs x 16 channels, tap table staged once per
// workgroup), the fftfilt block is two 1024/2048-point g_fft passes in LDS per 512/1024 outputs.
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
    int filt_mode;                      // 0 none, 1 runFilt, 2 runSSB usb, 3 runSSB lsb, 4 runDSB, 5 runAsym usb, 6 runAsym lsb (4..6: fft length 2048)
    int filt_off;                       // complex offset into the filter table (2 * BE_FFT_MAX entries per channel: filter, filterOpp)
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
    // --- dyadic resampling ratio (step * 2^dy_q is an integer; dy_q < 0: not dyadic): closed-form schedule
    int dy_q, dy_S;                     // config: Q = 1 << dy_q, S = step * Q
    int dy_active;                      // per feed: 1 = this feed's schedule comes from the closed form (be_sched_dyadic_*)
    int dy_mode;                        //   0: k_j = dy_kb + ((dy_P0 + j*S) >> q), d_j = ((dy_P0 + j*S) & (Q-1)) / Q;  1 (S == Q, d < 1): k_j = dy_kb + j, d_j = dy_P0 / Q
    int dy_pre, dy_cnt;                 //   entries written one by one in front (start-up, d < 1) / entries of the closed form
    int dy_kb, dy_P0;
};

struct BeBufs {                         // per channel device pointers (per feed capacity ensured by the host)
    const uint32_t* in;                 // n_in packed Samples
    uint32_t* hist;                     // BE_HIST packed Samples: tail of the previous feeds
    uint32_t* hist_next;
    float2* mixed;                      // BE_HIST + n_in: NCO-mixed samples, index BE_HIST + k
    uint2* sched;                       // per resampler output o: sched[o * sched_stride] = {k, bits(distance)} (channel-interleaved
                                        // so that the serial schedule lanes of a wave store to neighbouring addresses)
    float2* res;                        // [pending | new resampler outputs]
    float2* head;                       // n_blocks * 512
    float2* tail;                       // (1 + n_blocks) * 512; slot 0 = ovlbuf carried from the previous feed
    float2* cplx_out;                   // output when the last stage is complex (resampler or fftfilt)
    float* real_out;                    // output when a discriminator is enabled
    long n_in;                          // new input samples of this feed (host -> device with the pointers, one copy)
    long sched_stride;                  // = number of channels
};

// ---- 1. schedule: the float `distance` recurrence, one lane per channel (serial by nature; 256 channels = 4 waves, so the
// kernel's time is instructions-per-emission x single-wave issue latency -- the loop body is kept to the recurrence itself).
// Walks emission by emission instead of input by input: while d >= 1 the reference's `d -= 1.0` is exact, so
// after an emission leaves d_new the next one happens m = max(1, floor(d_new)) inputs later with
// d = d_new - m (exact) -- or fl(d_new - 1) when d_new < 1 -- which is bit-for-bit what the per-input loop holds.
// Entry o of a channel = {k, bits(d)}: k = index of the input that completes the emission, d = the distance the
// reference passes to doInterpolate(); the FIR kernel derives the phase floor(d * phase_steps) from it.
// Unchecked batches: once one emission has happened d_post < 1, so every later emission consumes
// m <= M = max(1, floor(fl(1 + step))) inputs; with R inputs left, the next R / M emissions cannot run past the
// feed and need no end test.  The wave runs min-over-lanes(R / M) of them with a scalar trip count, re-evaluates,
// and finishes the last few emissions of each lane in the checked loop.
struct BeSchedLane {
    float d; int k; uint2* p;
};
__device__ __forceinline__ void be_sched_emit(BeSchedLane& l, const float step, const long stride)
{
    const float m_f = fmaxf(floorf(l.d), 1.0f);             // inputs until the next emission: floor(d) if d >= 1, else 1
    l.k += (int)m_f;
    l.d -= m_f;                                             // d - floor(d) is exact; d - 1.0f is the reference's own op when d < 1
    *l.p = make_uint2((uint32_t)l.k, __float_as_uint(l.d));
    l.p += stride;
    l.d += step;
}
__device__ __forceinline__ int be_wave_min(int v)
{
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__global__ void __launch_bounds__(64) be_schedule_kernel(BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs, const int* __restrict__ perm, int n_ch)
{
    const int c_raw = blockIdx.x * 64 + threadIdx.x;       // = schedule column; perm[column] = channel (columns are sorted by filter design)
    const int c = perm[c_raw < n_ch ? c_raw : n_ch - 1];   // surplus lanes shadow the last column (same values to the same addresses)
    BeChan& s = ch[c];
    const long stride = bufs[c].sched_stride;
    const bool mine = !s.dy_active;                         // the closed-form kernels own dyadic channels (decided per feed by be_sched_dyadic_prep)
    const int n_in_real = (int)bufs[c].n_in;                // < 2^28
    const int n_in = mine ? n_in_real : 0;
    const float step = s.step;
    BeSchedLane l; l.d = s.distance; l.k = -1; l.p = bufs[c].sched;     // d: value BEFORE the next input's `-= 1.0`
    uint2* const p0 = l.p;
    const int M = max(1, (int)fminf(floorf(1.0f + step), 1.0e6f));
    bool first = true;
    for (;;) {
        // checked emissions: always the first of a feed (the carried-in d is not bounded by M), then whatever the batches left
        int budget = first ? 1 : 0x7fffffff;
        bool done = false;
        while (budget-- > 0) {
            const int kn = l.k + (int)fmaxf(floorf(l.d), 1.0f);
            if (kn >= n_in) { done = true; break; }
            be_sched_emit(l, step, stride);
        }
        if (!first) break;
        first = false;
        // unchecked batches
        for (;;) {
            const int safe = done ? 0 : (n_in - 1 - l.k) / M;
            const bool parked = safe < 16;                  // nearly finished lanes sit out and do not hold the others back
            const int wave_safe = be_wave_min(parked ? 0x7fffffff : safe);
            if (wave_safe == 0x7fffffff) break;
            if (!parked) {
                for (int i = wave_safe >> 2; i > 0; --i) {
                    be_sched_emit(l, step, stride); be_sched_emit(l, step, stride);
                    be_sched_emit(l, step, stride); be_sched_emit(l, step, stride);
                }
            }
        }
        if (done) break;
    }
    if (c_raw >= n_ch || !mine) return;
    s.n_in = n_in;
    s.distance = l.d - (float)(n_in - 1 - l.k);             // inputs consumed without an emission: each `-= 1.0` exact (d stays >= 1)
    const int cnt = (int)((l.p - p0) / stride);
    s.n_res = cnt;
    s.n_blocks = s.filt_mode ? (s.pending + cnt) / s.half : 0;
}

// ---- 1b. the same schedule in closed form when the ratio is dyadic (60000/48000 = 1.25, 120000/48000 = 2.5, 75000/48000 =
// 25/16 ...: every channelizer output rate that is 48 kHz times a dyadic number).  With Q = 2^q, S = step * Q and the
// distance on the 1/Q grid (it starts at 0 and every operation below keeps it there), all of the reference's float
// operations are exact, so integer arithmetic reproduces them bit for bit.  In D = d * Q:
//     emission: m = D >= Q ? D >> q : 1;  k += m;  D_post = D - m * Q;  entry {k, D_post / Q};  D = D_post + S.
// Once D >= Q it stays so (S >= Q), and with P_0 = D mod Q, kb = index of the first such emission:
//     k_j = kb + floor((P_0 + j * S) / Q),   d_j = ((P_0 + j * S) mod Q) / Q          (the floor sum telescopes)
// so the emissions are independent of each other: `prep` (one lane per channel) walks the start-up emissions (d < 1, at
// most Q + |D| of them), solves for the count and the end state; `fill` writes the entries in parallel.  A distance
// that is not on the grid (never seen; a guard) hands the channel to the serial kernel for this feed.
__global__ void be_sched_dyadic_prep_kernel(BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs, const int* __restrict__ perm, int n_ch)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_ch) return;
    const int c = perm[col];
    BeChan& s = ch[c];
    s.dy_active = 0;
    if (s.dy_q < 0) return;
    const int q = s.dy_q;
    const long Q = 1L << q, S = s.dy_S;
    const float d0 = s.distance;
    long D = (long)(d0 * (float)Q);                         // exact when d0 is on the grid
    if ((float)D / (float)Q != d0 || D > (1L << 40) || D < -(1L << 40)) return;
    const long n_in = bufs[c].n_in;
    uint2* p = bufs[c].sched;
    const long stride = bufs[c].sched_stride;
    long k = -1; int pre = 0;
    bool ended = false;
    while (D < Q && S != Q) {                               // start-up: one input per emission until the distance passes 1
        if (k + 1 >= n_in) { ended = true; break; }
        k += 1; D -= Q;
        *p = make_uint2((uint32_t)k, __float_as_uint((float)D / (float)Q)); p += stride; pre++;
        D += S;
    }
    long cnt = 0, kb = 0, P0 = 0; int mode = 0;
    if (!ended) {
        if (D < Q) {                                        // S == Q and d < 1: one input per emission for ever, constant distance
            mode = 1; kb = k + 1; P0 = D - Q;
            cnt = n_in - 1 - k; if (cnt < 0) cnt = 0;
            k += cnt;                                       // D unchanged: (D - Q) + S = D
        } else {
            const long m0 = D >> q;
            P0 = D & (Q - 1); kb = k + m0;
            if (kb <= n_in - 1) {
                const long R = n_in - 1 - kb;
                const long J = (R * Q + Q - 1 - P0) / S;
                cnt = J + 1;
                k = kb + ((P0 + J * S) >> q);
                D = ((P0 + J * S) & (Q - 1)) + S;
            }
        }
    }
    D -= (n_in - 1 - k) * Q;                                // inputs consumed without an emission: each `-= 1.0` exact
    s.dy_active = 1; s.dy_mode = mode; s.dy_pre = pre; s.dy_cnt = (int)cnt; s.dy_kb = (int)kb; s.dy_P0 = (int)P0;
    s.n_in = (int)n_in;
    s.distance = (float)D / (float)Q;
    s.n_res = pre + (int)cnt;
    s.n_blocks = s.filt_mode ? (s.pending + s.n_res) / s.half : 0;
}

// 64 columns x 4 rows per workgroup; a lane writes 64 consecutive entries of its column, so every store instruction of a
// wave covers 64 neighbouring columns of one schedule row (512 contiguous bytes)
__global__ void __launch_bounds__(256) be_sched_dyadic_fill_kernel(const BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs,
                                                                   const int* __restrict__ perm, int n_ch)
{
    const int col = blockIdx.y * 64 + (threadIdx.x & 63);
    if (col >= n_ch) return;
    const BeChan& s = ch[perm[col]];
    if (!s.dy_active) return;
    const int q = s.dy_q, mode = s.dy_mode, cnt = s.dy_cnt;
    const long Q = 1L << q, S = s.dy_S, P0 = s.dy_P0, kb = s.dy_kb;
    uint2* __restrict__ base = bufs[perm[0]].sched + col;   // bank-wide base (column 0) + this column
    const long j0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    const float inv = 1.0f / (float)Q;                      // a power of two: the product below is the exact quotient
    for (int i = 0; i < 64; i++) {
        const long j = j0 + i;
        if (j >= cnt) break;
        const long t = P0 + j * S;
        const uint32_t k = mode ? (uint32_t)(kb + j) : (uint32_t)(kb + (t >> q));
        const float d = mode ? (float)P0 * inv : (float)(t & (Q - 1)) * inv;
        base[(long)(s.dy_pre + j) * n_ch] = make_uint2(k, __float_as_uint(d));
    }
}

// ---- 2a. NCO mix of history + new samples into float (exact: int16 -> float, complex product)
__global__ __launch_bounds__(256)
void be_mix_kernel(const BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs, const float* __restrict__ nco_tbl)
{
    // the cosine table (16 KB) is gathered twice per sample at a lane stride of `inc` entries: from LDS, not through the
    // texture path; a workgroup takes enough samples (the host sizes the grid) to pay for staging it
    __shared__ float tbl[BE_NCO_N];
    for (int i = threadIdx.x; i < BE_NCO_N; i += blockDim.x) tbl[i] = nco_tbl[i];
    __syncthreads();
    const int c = blockIdx.y;
    const BeChan s = ch[c];
    const BeBufs b = bufs[c];
    const int total = BE_HIST + s.n_in;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i - BE_HIST;                          // sample index relative to this feed; < 0: history
        const uint32_t v = j < 0 ? b.hist[i] : b.in[j];
        // phase after nextPhase() for this sample; s.nco_phase is still the phase BEFORE this feed
        // (phase + (j + 1) * inc) mod 4096, non-negative -- what NCO::nextPhase's add-and-wrap loop arrives at; in 24-bit
        // arithmetic on the residues (the table size is a power of two, so the masks are exact for negative j + 1 and inc too)
        const uint32_t p = ((uint32_t)s.nco_phase + ((uint32_t)(j + 1) & (BE_NCO_N - 1)) * ((uint32_t)s.nco_inc & (BE_NCO_N - 1))) & (BE_NCO_N - 1);
        const float o_r = tbl[p], o_i = -tbl[(p + BE_NCO_N / 4) & (BE_NCO_N - 1)];
        const float a = (float)(int16_t)(v & 0xffffu), q = (float)(int16_t)(v >> 16);
        float2 m; m.x = a * o_r - q * o_i; m.y = a * o_i + q * o_r;         // std::complex<float> operator*=
        b.mixed[i] = m;
    }
    // raw history for the next feed: last BE_HIST of (old history ++ new input); double-buffered, so nothing read here is overwritten
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < BE_HIST; i += blockDim.x) {
            const long src = (long)i + s.n_in - BE_HIST;
            b.hist_next[i] = src >= 0 ? b.in[src] : b.hist[i + s.n_in];
        }
    }
}

// ---- 2b. polyphase FIR, one lane per output, taps summed newest-first like the ring walk (mul and add separate, in order).
// The schedule is column-interleaved (entry o of column q at [o * n_ch + q]; perm[q] = channel, columns sorted by
// interpolator design), so a workgroup takes a tile of 64 outputs x 16 columns: 128-byte rows in, transposed through
// LDS; each wave then walks 4 columns with its 64 lanes on 64 consecutive outputs of one channel.
// Interpolator::create's "taps per phase" argument 4.5 yields 72 taps per phase (interpolator.cpp: ntaps =
// (int)(4.5 * 16), then * 16 phases), so an output is 72 complex-by-real MACs: the 16 x nt tap table (odd pitch) is
// staged in LDS once per workgroup -- channels with the same design share one table (the host de-duplicates them and
// the column order keeps them together) -- and each wave stages the input window [kmin - nt + 1, kmax] of its 64
// outputs; the tap loop runs from LDS.  A tile that straddles two designs reads its taps from global memory, and a
// window that does not fit is read from global memory, with the same loop.  After the tile barrier the waves are
// independent (wave-private windows, wave-level ordering only), so one wave's staging latency overlaps the others' taps.
constexpr int BE_FIR_TO = 64, BE_FIR_TC = 16;
constexpr int BE_FIR_XCAP = 384;                            // float2 per wave: 64 * (inputs per output) + nt
constexpr int BE_FIR_NT_MAX = 80, BE_FIR_TCAP = 16 * (BE_FIR_NT_MAX + 8);
// tap-table row pitch: a multiple of 4 floats whose quarter is odd -- rows start 16-byte aligned (float4 reads) and the 16
// phases read at the same tap index fall into 16 different 4-bank groups
__host__ __device__ constexpr int be_fir_pitch(int nt) { const int p = (nt + 3) & ~3; return (p / 4) % 2 ? p : p + 4; }
__device__ __forceinline__ void be_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool XL, bool TL>                                 // input window / tap table in LDS
__device__ __forceinline__ float2 be_fir_taps(const float2* x, const float* t, int nt)
{
    float ra = 0.0f, ia = 0.0f;
    int i = 0;
    if constexpr (TL) {                                      // LDS table: rows are 16-byte aligned, four taps per ds_read_b128
        for (; i + 4 <= nt; i += 4) {
            const float4 c = *reinterpret_cast<const float4*>(t + i);
            const float2 v0 = x[-i], v1 = x[-i - 1], v2 = x[-i - 2], v3 = x[-i - 3];
            ra += c.x * v0.x; ia += c.x * v0.y;
            ra += c.y * v1.x; ia += c.y * v1.y;
            ra += c.z * v2.x; ia += c.z * v2.y;
            ra += c.w * v3.x; ia += c.w * v3.y;
        }
    }
#pragma unroll 8
    for (; i < nt; i++) {
        const float2 v = x[-i];
        ra += t[i] * v.x;
        ia += t[i] * v.y;
    }
    float2 r; r.x = ra; r.y = ia;
    return r;
}
__global__ void __launch_bounds__(256) be_fir_kernel(const BeChan* __restrict__ ch, const BeBufs* __restrict__ bufs,
                                                     const float* __restrict__ taps, const int* __restrict__ perm, int n_ch)
{
    __shared__ uint2 tile[BE_FIR_TO][BE_FIR_TC + 1];
    __shared__ float2 xw_all[4][BE_FIR_XCAP];
    __shared__ __attribute__((aligned(16))) float tw[BE_FIR_TCAP];
    const int o0 = blockIdx.x * BE_FIR_TO, q0 = blockIdx.y * BE_FIR_TC;
    const int q_end = min(q0 + BE_FIR_TC, n_ch);
    int n_res_max = 0;
    const BeChan& first = ch[perm[q0]];
    const int nt0 = first.ntaps, off0 = first.taps_off;
    bool shared = nt0 <= BE_FIR_NT_MAX;
    for (int q = q0; q < q_end; q++) {
        const BeChan& s = ch[perm[q]];
        n_res_max = max(n_res_max, s.n_res);
        shared = shared && s.taps_off == off0 && s.ntaps == nt0 && s.phase_steps == 16;
    }
    if (o0 >= n_res_max) return;                            // uniform
    const uint2* __restrict__ sched = bufs[perm[0]].sched;  // bank-wide base (column 0)
    for (int i = threadIdx.x; i < BE_FIR_TO * BE_FIR_TC; i += 256) {
        const int row = i / BE_FIR_TC, col = i % BE_FIR_TC;
        if (q0 + col < n_ch && o0 + row < n_res_max) tile[row][col] = sched[(long)(o0 + row) * n_ch + q0 + col];
    }
    if (shared) {
        const int pitch = be_fir_pitch(nt0);
        const float* tg = taps + off0;
        for (int q = threadIdx.x >> 6; q < 16; q += 4)
            for (int i = threadIdx.x & 63; i < nt0; i += 64) tw[q * pitch + i] = tg[q * nt0 + i];
    }
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2* const xw = xw_all[w];
    const int o = o0 + lane;
    for (int j = 0; j < BE_FIR_TC / 4; j++) {
        const int col = w * (BE_FIR_TC / 4) + j;
        if (q0 + col >= n_ch) break;                        // wave-uniform
        const int c = perm[q0 + col];
        const BeChan& s = ch[c];
        const BeBufs& b = bufs[c];
        const int n_res = s.n_res;
        if (o0 >= n_res) continue;                          // wave-uniform
        const bool valid = o < n_res;
        const uint2 e = tile[lane][col];
        const int k = valid ? (int)e.x : 0;
        int ph = (int)floorf(__uint_as_float(e.y) * (float)s.phase_steps);     // Interpolator::decimate's phase (interpolator.h:33)
        ph = valid && ph > 0 ? ph : 0;
        const int nt = s.ntaps;
        // k grows with the output index (every emission consumes at least one input): the window's ends are the first and the
        // last valid output's k -- two uniform LDS reads instead of a 12-step cross-lane min/max
        const int last_lane = min(63, n_res - 1 - o0);
        const int kmin = (int)tile[0][col].x, kmax = (int)tile[last_lane][col].x;
        const int win = kmax - kmin + nt;
        const bool x_lds = win <= BE_FIR_XCAP;
        const float* tg = taps + s.taps_off + ph * nt;
        const float* tl = tw + ph * be_fir_pitch(nt);
        const float2* xg = b.mixed + BE_HIST;
        if (x_lds) {
            const float2* src = xg + (kmin - nt + 1);
            for (int i = lane; i < win; i += 64) xw[i] = src[i];
        }
        be_wave_sync();
        if (valid) {
            const float2* xl = xw + (k - kmin) + (nt - 1);
            float2 r;
            if (x_lds) r = shared ? be_fir_taps<true, true>(xl, tl, nt) : be_fir_taps<true, false>(xl, tg, nt);
            else       r = shared ? be_fir_taps<false, true>(xg + k, tl, nt) : be_fir_taps<false, false>(xg + k, tg, nt);
            b.res[s.pending + o] = r;
            if (s.filt_mode == 0 && s.discri == 0) b.cplx_out[o] = r;
        }
        be_wave_sync();                                     // the next column's staging overwrites xw
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
        else if (i == 0) { lo = c_mul(lo, filt[0]); /* bin N/2 is left untouched (fftfilt.cpp:294-311, 374-392) */ }
        else if (s.filt_mode == 5) { lo = c_mul(lo, filt[i]); hi = c_mul(hi, filt[BE_FFT_MAX + H + i]); }         // runAsym usb: lsb side through filterOpp
        else if (s.filt_mode == 6) { lo = c_mul(lo, filt[BE_FFT_MAX + i]); hi = c_mul(hi, filt[H + i]); }         // runAsym lsb
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
            // (taking the left neighbour's `cur` instead of recomputing it changes nothing: the kernel moves 36 bytes per
            // output at ~6 TB/s -- it is at the memory roofline)
            const float prev = j == 0 ? s.prev_arg : [&] { const float2 pv = sample(j - 1); return atan2_approx2(pv.y, pv.x); }();
            float dev = (float)((double)(cur - prev) / 3.14159265358979323846);
            if (dev < -1.0f) dev += 2.0f; else if (dev > 1.0f) dev -= 2.0f;
            b.real_out[j] = dev * s.fm_scaling;
        } else {
            float2 m1; if (j == 0) { m1.x = s.m1r; m1.y = s.m1i; } else m1 = sample(j - 1);
            const float dr = m1.x * v.x - (-m1.y) * v.y, di = m1.x * v.y + (-m1.y) * v.x;    // conj(prev) * cur
            // std::arg(complex<float>) = atan2f: evaluated in double and rounded once, i.e. the correctly rounded float
            // (up to double rounding, ~2^-29 of the cases) -- glibc's atan2f is within 1-2 ulp of that, the device libm's
            // float atan2f is not better, so this is the closest a different libm can get to the host's value
            const float ang = (float)atan2((double)di, (double)dr);
            b.real_out[j] = (float)(((double)ang / 3.14159265358979323846) * (double)s.fm_scaling);
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
