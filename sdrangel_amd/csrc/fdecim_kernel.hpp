// Float half-band decimator chains: DecimatorsFI / DecimatorsFF / DecimatorsIF over IntHalfbandFilterEOF<64>
// (reference: sdrbase/dsp/decimatorsfi.cpp, decimatorsff.cpp, decimatorsif.h, inthalfbandfiltereof.h:65-72,141-188).
//
// One stage (myDecimate -> storeSample x2 + doFIR) on a stream x[n], output k from x[2k], x[2k+1]; with the polyphase
// arms O[j] = x[2j+1], E[j] = x[2j] the ring indices of doFIR (:153-171, a = tip, b = tail, odd branch) resolve to
//     acc = 0;  for i = 0..15:  acc = acc + (O[k-i] + O[k-31+i]) * c[i];      y[k] = acc + E[k-15] * 0.5f
// in float, in exactly this order, add and multiply separate (the file is compiled with -ffp-contract=off), so the
// results are the reference's bit for bit.  c[] = (float) of the order-64 decimals (hbfiltertraits.cpp:173-190).
//
// Chain shapes (call patterns of decimatorsfi.cpp): _cen = L stages on the raw stream; _inf/_sup = a memoryless
// "4x downsample and rotate" front end (sums of 4 consecutive samples, association kept as written there) followed
// by L-2 stages; L = 1, 2 _inf/_sup and decimate1 have no filter at all (fd_pointwise_kernel).
//
// Tried and not adopted (experiments/fdecim_wave_kernel.hpp.txt): a single-wave flavour (LDS-bandwidth bound: every lane
// reads (R + 32) / R window entries per output) and giving the two components to different waves (R = 8 windows cost
// 224 VGPRs, or spills under a launch bound).
//
// Work decomposition = decim_chain_kernel's: the stream that enters stage 1 ("pre-samples") is cut into chunks of
// 2048; a workgroup owns a segment of consecutive chunks and carries every stage's 32-entry arm history in LDS from
// chunk to chunk.  The FIRST segment of a call starts from the handle's explicit filter state (`seed`: per stage the
// 32 + 32 last inputs as odd / even arm, I and Q -- the contents of the reference filter's ring); every other segment
// replays warm-up chunks (>= 62 * (2^NS - 1) pre-samples: the chain's memory) taken from the call's own input in front
// of it; the workgroup that owns the last chunk writes the state at the end of the data to `dump`.  The state being the
// filters' own rings, one DecimatorsFI object's cascades can hand it to each other (sdrx_fdecim_save/load_stages).
// HBM: reads the input once (+ warm-up re-reads that hit L2/MALL), writes 1/2^L of it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace sdrx {

constexpr int FD_CHUNK = 2048;          // pre-samples per chunk
constexpr int FD_THREADS = 256;
constexpr int FD_H = 32;                // history entries kept in front of every arm (31 are read)

// front-end codes
enum { FD_FE_ID = 0, FD_FE_INF4 = 1, FD_FE_SUP4_A = 2, FD_FE_SUP4_B = 3, FD_FE_INF2 = 4, FD_FE_SUP2 = 5 };

__host__ __device__ constexpr int fd_arr(int s) { return FD_H + (FD_CHUNK >> s); }     // floats of one arm array of stage s's input
__host__ __device__ constexpr int fd_off(int s) { int o = 0; for (int u = 1; u < s; u++) o += 4 * fd_arr(u); return o; }
__host__ __device__ constexpr int fd_lds_floats(int ns) { return fd_off(ns + 1); }
__host__ __device__ constexpr int fd_warm_chunks(int ns) { return (62 * ((1 << ns) - 1) + FD_CHUNK - 1) / FD_CHUNK; }
constexpr int FD_STATE = 4 * FD_H;      // floats of one filter's state: oI, oQ, eI, eQ, 32 entries each, oldest first

// raw element access: IN = 0 float I/Q, IN = 1 int16 I/Q (converted exactly; sums of the front end stay in int)
template<int IN> struct FdIn;
template<> struct FdIn<0> {
    typedef float T;
    static __device__ __forceinline__ float2 sample(const void* in, long s) { return static_cast<const float2*>(in)[s]; }
    // b[0..7] = 4 consecutive complex samples starting at sample 4g
    static __device__ __forceinline__ float2 combo(const void* in, long g, int fe)
    {
        const float4 lo = static_cast<const float4*>(in)[2 * g], hi = static_cast<const float4*>(in)[2 * g + 1];
        const float b0 = lo.x, b1 = lo.y, b2 = lo.z, b3 = lo.w, b4 = hi.x, b5 = hi.y, b6 = hi.z, b7 = hi.w;
        float2 r;
        if (fe == FD_FE_INF4)        { r.x = ((b0 - b3) + b7) - b4; r.y = ((b1 - b5) + b2) - b6; }
        else if (fe == FD_FE_SUP4_A) { r.x = ((b1 - b2) - b5) + b6; r.y = ((-b0 - b3) + b4) + b7; }
        else                         { r.x = ((b1 - b2) - b5) + b6; r.y = ((b4 + b7) - b0) - b3; }
        return r;
    }
    // the two outputs of decimate2_inf / _sup per group of 4 samples (decimatorsfi.cpp:53-93)
    static __device__ __forceinline__ float2 combo2(const void* in, long g, int which, int fe)
    {
        const float4 v = static_cast<const float4*>(in)[2 * g + which];
        float2 r;
        if (fe == FD_FE_INF2) { if (which == 0) { r.x = v.x - v.w; r.y = v.y + v.z; } else { r.x = v.w - v.x; r.y = -v.y - v.z; } }
        else                  { if (which == 0) { r.x = v.y - v.z; r.y = -v.x - v.w; } else { r.x = v.z - v.y; r.y = v.x + v.w; } }
        return r;
    }
};
template<> struct FdIn<1> {
    typedef int16_t T;
    static __device__ __forceinline__ float2 sample(const void* in, long s)
    {
        const uint32_t v = static_cast<const uint32_t*>(in)[s];
        return make_float2((float)(int16_t)(v & 0xffffu), (float)(int16_t)(v >> 16));
    }
    static __device__ __forceinline__ float2 combo(const void* in, long g, int fe)
    {
        const uint4 v = static_cast<const uint4*>(in)[g];
        const int b0 = (int16_t)(v.x & 0xffffu), b1 = (int16_t)(v.x >> 16), b2 = (int16_t)(v.y & 0xffffu), b3 = (int16_t)(v.y >> 16);
        const int b4 = (int16_t)(v.z & 0xffffu), b5 = (int16_t)(v.z >> 16), b6 = (int16_t)(v.w & 0xffffu), b7 = (int16_t)(v.w >> 16);
        float2 r;
        if (fe == FD_FE_INF4) { r.x = (float)(b0 - b3 + b7 - b4); r.y = (float)(b1 - b5 + b2 - b6); }
        else                  { r.x = (float)(b1 - b2 - b5 + b6); r.y = (float)(-b0 - b3 + b4 + b7); }
        return r;
    }
    static __device__ __forceinline__ float2 combo2(const void* in, long g, int which, int fe)
    {
        const uint2 v = static_cast<const uint2*>(in)[2 * g + which];
        const int x = (int16_t)(v.x & 0xffffu), y = (int16_t)(v.x >> 16), z = (int16_t)(v.y & 0xffffu), w = (int16_t)(v.y >> 16);
        float2 r;
        if (fe == FD_FE_INF2) { if (which == 0) { r.x = (float)(x - w); r.y = (float)(y + z); } else { r.x = (float)(w - x); r.y = (float)(-y - z); } }
        else                  { if (which == 0) { r.x = (float)(y - z); r.y = (float)(-x - w); } else { r.x = (float)(z - y); r.y = (float)(x + w); } }
        return r;
    }
};

// pre-sample p of the call (p >= 0): what enters stage 1
template<int IN> __device__ __forceinline__ float2 fd_pre(const void* in, long p, int fe)
{
    return fe == FD_FE_ID ? FdIn<IN>::sample(in, p) : FdIn<IN>::combo(in, p, fe);
}

// output conversion.  out_kind 0: DecimatorsFI, (int16)(v * 32768.0) -- the double product of a float and 2^15 is the
// float product, conversion truncates; 1: float, times `scale` for DecimatorsIF (scale = 1 otherwise: exact)
__device__ __forceinline__ void fd_store(void* out, long k, float2 v, int out_kind, float scale)
{
    if (out_kind == 0) {
        const int re = (int)(v.x * 32768.0f), im = (int)(v.y * 32768.0f);
        static_cast<uint32_t*>(out)[k] = ((uint32_t)re & 0xffffu) | ((uint32_t)im << 16);
    } else {
        static_cast<float2*>(out)[k] = make_float2(v.x * scale, v.y * scale);
    }
}

// order-64 half-band decimals (hbfiltertraits.cpp:173-190) narrowed to float like hbCoeffsF.  Compile-time
// literals on purpose: on gfx950 a VALU op with an SGPR operand issues at half rate (profiles/r01_valu_issue_rates.txt).
__host__ __device__ constexpr float fd_c(int i)
{
    constexpr double d[16] = {
        -0.0004653050334792540416659067936677729449, 0.0007120490624526883919470643391491648799,
        -0.0012303473710125558716887983479182366864, 0.0019716520179919017584369012041634050547,
        -0.0029947484165425580261710170049127555103, 0.0043703902150498061263128590780979720876,
        -0.0061858352927315653213558022116558277048, 0.0085554408639278121950777489246320328675,
        -0.0116397924445187355563247066925214312505, 0.0156852221106748394852115069397768820636,
        -0.0211070832238078286147153761476147337817, 0.0286850846890029896607554604770484729670,
        -0.0400956173930921908055147184768429724500, 0.0597215923200692666572564348825835622847,
        -0.1036982054813635201195864965484361164272, 0.3175014394028848885298543791577685624361,
    };
    return (float)d[i];
}

template<int I, int N, typename F> __device__ __forceinline__ void fd_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); fd_static_for<I + 1, N>(f); }
}

// R consecutive outputs (I and Q) of one stage for lane t.  Arm arrays hold O[j] / E[j] at index FD_H + j.
template<int R>
__device__ __forceinline__ void fd_stage(const float* __restrict__ oI, const float* __restrict__ oQ,
                                         const float* __restrict__ eI, const float* __restrict__ eQ,
                                         int t, float (&yI)[R], float (&yQ)[R])
{
    const int b = R * t;                         // w[i] = O[k0 - 32 + i], k0 = R * t
    float wI[R + 32], wQ[R + 32], cI[R], cQ[R];
    if constexpr (R == 4) {
#pragma unroll
        for (int q = 0; q < (R + 32) / 4; q++) {
            const float4 a = reinterpret_cast<const float4*>(oI + b)[q], c = reinterpret_cast<const float4*>(oQ + b)[q];
            wI[4*q] = a.x; wI[4*q+1] = a.y; wI[4*q+2] = a.z; wI[4*q+3] = a.w;
            wQ[4*q] = c.x; wQ[4*q+1] = c.y; wQ[4*q+2] = c.z; wQ[4*q+3] = c.w;
        }
    } else if constexpr (R == 2) {
#pragma unroll
        for (int q = 0; q < (R + 32) / 2; q++) {
            const float2 a = reinterpret_cast<const float2*>(oI + b)[q], c = reinterpret_cast<const float2*>(oQ + b)[q];
            wI[2*q] = a.x; wI[2*q+1] = a.y; wQ[2*q] = c.x; wQ[2*q+1] = c.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < R + 32; q++) { wI[q] = oI[b + q]; wQ[q] = oQ[b + q]; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) { cI[r] = eI[b + r + FD_H - 15]; cQ[r] = eQ[b + r + FD_H - 15]; }
    fd_static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        float aI = 0.0f, aQ = 0.0f;
        fd_static_for<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            aI = aI + (wI[32 + r - i] + wI[1 + r + i]) * fd_c(i);
            aQ = aQ + (wQ[32 + r - i] + wQ[1 + r + i]) * fd_c(i);
        });
        yI[r] = aI + cI[r] * 0.5f;
        yQ[r] = aQ + cQ[r] * 0.5f;
    });
}

template<int NS, int IN>
__global__ __launch_bounds__(FD_THREADS)
void fdecim_chain_kernel(const float* __restrict__ seed,    // NS x FD_STATE floats: the filters' state in front of the call
                         float* __restrict__ dump,          // same layout: the state behind the call's last pre-sample
                         const void* __restrict__ in, void* __restrict__ out,
                         long n_pre, long n_out, int n_chunks, int cps, int fe, int out_kind, float scale)
{
    constexpr int C = FD_CHUNK, NT = FD_THREADS, WARM = fd_warm_chunks(NS);
    constexpr int PPT = C / 2 / NT;                        // pre-sample PAIRS per lane per chunk (4)
    __shared__ __attribute__((aligned(16))) float lds[fd_lds_floats(NS)];
    const int tid = threadIdx.x;
    const long first = (long)blockIdx.x * cps;             // the host keeps cps >= WARM: only segment 0 reaches in front of the call
    long last = first + cps; if (last > n_chunks) last = n_chunks;

    for (int i = tid; i < fd_lds_floats(NS); i += NT) lds[i] = 0.0f;
    __syncthreads();
    if (blockIdx.x == 0) {
#pragma unroll
        for (int s = 1; s <= NS; s++) {
            float* a = lds + fd_off(s);
            for (int i = tid; i < FD_STATE; i += NT) a[(i / FD_H) * fd_arr(s) + (i % FD_H)] = seed[(s - 1) * FD_STATE + i];
        }
    }

    float2 pe[PPT], po[PPT];                               // even / odd pre-sample of each pair
    auto fetch = [&](long chunk) {
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            const long g = chunk * C + 2 * (j * NT + tid);  // global pre-sample index of the pair's first member
            pe[j] = g < n_pre ? fd_pre<IN>(in, g, fe) : make_float2(0.0f, 0.0f);
            po[j] = g + 1 < n_pre ? fd_pre<IN>(in, g + 1, fe) : make_float2(0.0f, 0.0f);
        }
    };
    const long start = blockIdx.x == 0 ? 0 : first - WARM;
    fetch(start);
    __syncthreads();

    for (long chunk = start; chunk < last; ++chunk) {
        {
            float* oI = lds + fd_off(1), *oQ = oI + fd_arr(1), *eI = oQ + fd_arr(1), *eQ = eI + fd_arr(1);
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                const int q = FD_H + j * NT + tid;
                eI[q] = pe[j].x; eQ[q] = pe[j].y; oI[q] = po[j].x; oQ[q] = po[j].y;
            }
        }
        if (chunk + 1 < last) fetch(chunk + 1);            // in flight during the whole chunk
        __syncthreads();

        const bool live = chunk >= first;
#pragma unroll
        for (int s = 1; s <= NS; s++) {
            const int NOUT = C >> s;
            const float* oI = lds + fd_off(s), *oQ = oI + fd_arr(s), *eI = oQ + fd_arr(s), *eQ = eI + fd_arr(s);
            float* nI = lds + fd_off(s + 1);               // next stage's oI, oQ, eI, eQ
            const int na = fd_arr(s + 1);
            auto emit = [&](int k, float yi, float yq) {   // output k of this stage
                if (s < NS) {
                    const int j = FD_H + (k >> 1);
                    if (k & 1) { nI[j] = yi; nI[na + j] = yq; } else { nI[2 * na + j] = yi; nI[3 * na + j] = yq; }
                } else if (live) {
                    const long gk = chunk * NOUT + k;
                    if (gk < n_out) fd_store(out, gk, make_float2(yi, yq), out_kind, scale);
                }
            };
            if (s == 1) {
                float yI[4], yQ[4];
                fd_stage<4>(oI, oQ, eI, eQ, tid, yI, yQ);
#pragma unroll
                for (int r = 0; r < 4; r++) emit(4 * tid + r, yI[r], yQ[r]);
            } else if (s == 2) {
                float yI[2], yQ[2];
                fd_stage<2>(oI, oQ, eI, eQ, tid, yI, yQ);
                emit(2 * tid, yI[0], yQ[0]); emit(2 * tid + 1, yI[1], yQ[1]);
            } else if (tid < NOUT) {
                float yI[1], yQ[1];
                fd_stage<1>(oI, oQ, eI, eQ, tid, yI, yQ);
                emit(tid, yI[0], yQ[0]);
            }
            __syncthreads();
        }

        // the state behind the data: the FD_H arm entries that end at the call's last sample (a partial last chunk holds
        // n_pre - chunk * C pre-samples = a whole number of outputs of every stage: the call is whole groups)
        if (chunk == (long)n_chunks - 1) {
            const long left = n_pre - chunk * C;
#pragma unroll
            for (int s = 1; s <= NS; s++) {
                const int have = (int)(left >> s);         // entries this chunk put into each arm of stage s
                const float* a = lds + fd_off(s);
                for (int i = tid; i < FD_STATE; i += NT) dump[(s - 1) * FD_STATE + i] = a[(i / FD_H) * fd_arr(s) + have + (i % FD_H)];
            }
            __syncthreads();                               // the carry below overwrites entries 0..31, which a short chunk just read
        }
        // carry: the last FD_H entries of every arm become the next chunk's history
#pragma unroll
        for (int s = 1; s <= NS; s++) {
            const int ND = C >> s;                         // chunk entries per arm
            float* a = lds + fd_off(s);
            for (int i = tid; i < 4 * FD_H; i += NT) {
                const int arr = i / FD_H, e = i % FD_H;
                a[arr * fd_arr(s) + e] = a[arr * fd_arr(s) + ND + e];
            }
        }
        __syncthreads();
    }
}

// no-filter variants: decimate1, decimate2_inf/_sup, decimate4_inf/_sup
template<int IN>
__global__ void fd_pointwise_kernel(const void* __restrict__ in, void* __restrict__ out, long n_out, int fe, int out_kind, float scale, int dec1)
{
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n_out; k += (long)gridDim.x * blockDim.x) {
        float2 v;
        if (fe == FD_FE_INF2 || fe == FD_FE_SUP2) v = FdIn<IN>::combo2(in, k >> 1, (int)(k & 1), fe);
        else v = fd_pre<IN>(in, k, fe);
        // decimate1 of DecimatorsFI multiplies by SDR_RX_SCALEF in float; same value as the 32768.0 product
        (void)dec1;
        fd_store(out, k, v, out_kind, scale);
    }
}

} // namespace sdrx
