// Fused Decimators<qint32,qint16,16,InputBits>::decimate{2..64}_{cen,inf,sup} chain
// (reference: sdrbase/dsp/decimators.h:463-3885) as ONE gfx950 kernel.
//
// Work decomposition ("segments of chunks"):
//   * the consumed input (n_in complex samples) is cut into chunks of C = 4096 samples;
//   * workgroup b owns `cps` consecutive chunks and walks them in order, carrying every stage's
//     FIR history in LDS from chunk to chunk exactly like the CPU carries its ring buffers;
//   * before its first chunk it replays ONE warm-up chunk (the 4096 samples in front of the
//     segment; for segment 0 these come from the handle's history buffer = tail of the previous
//     call).  4096 >= 62*(2^L-1) = the chain's total memory, so after the warm-up every LDS
//     history is exactly what a sequential run would hold (the filters are feed-forward).
//   Per chunk: coalesced 16 B/lane global loads (prefetched one chunk ahead into registers),
//   de-interleave with v_perm_b32 into the four packed-int16 polyphase arrays of stage 1, then
//   stages 1..L back to back through LDS, final stage stores packed Sample dwords.
//
// HBM traffic: 4 B/sample read once (+1/cps warm-up re-read, L2/MALL resident) + 4/2^L written.
#pragma once
#include "hb_common.hpp"

namespace sdrx {

constexpr int DC_CHUNK = 4096;       // input samples per chunk
constexpr int DC_THREADS = 256;

// One launch serves up to DJ_MAX independent device streams (blockIdx.y = stream): the per-stream arguments travel
// as a by-value table in the kernarg segment (scalar loads, no extra copy or allocation per call).
constexpr int DJ_MAX = 64;
struct DecimJob {
    const void* hist;                // DC_CHUNK samples: tail of this stream's previous call
    const void* in;                  // n_in samples (quad aligned)
    uint32_t*   out;                 // n_in >> L packed Samples
    uint32_t*   flags;               // one per DC_CHUNK-sample chunk (FAST kernel writes, EXACT kernel reads; nullptr: recompute all)
    long        n_in;
    int         n_units;             // sub-chunks (FAST) or chunks (EXACT) of this stream
    int         pad_;
};
struct DecimJobs { DecimJob j[DJ_MAX]; };

// outputs per lane of stage s (1-based) -- chosen so that the first three stages keep all 256
// lanes busy (2048/8, 1024/4, 512/2) and the low-rate tail still uses >= 32 lanes.
__host__ __device__ constexpr int dc_R(int s) { return s == 1 ? 8 : s == 2 ? 4 : 2; }
// LDS dwords of ONE polyphase array of stage s's input
__host__ __device__ constexpr int dc_arr(int s)
{
    return s == 1 ? (HIST / 2 + DC_CHUNK / 4)            // packed int16 pairs
                  : (HIST + (DC_CHUNK >> s));             // int32
}
__host__ __device__ constexpr int dc_off(int s)           // dword offset of stage s's 4 arrays
{
    int o = 0;
    for (int u = 1; u < s; u++) o += 4 * dc_arr(u);
    return o;
}
__host__ __device__ constexpr int dc_lds_dwords(int L) { return dc_off(L + 1); }

// static bound on |input of stage s| for raw int16 input shifted left by pre
__host__ __device__ constexpr long dc_in_bound(int s, int pre)
{
    long b = 32768L << pre;
    for (int i = 1; i < s; i++) { b = b * hb_l1<64>() / 2048 + 1; if (b > (1L << 40)) b = 1L << 40; }
    return b;
}
__host__ __device__ constexpr bool dc_m24(int s, int pre) { return 2 * dc_in_bound(s, pre) < (1L << 23); }

// stage modes of decimateK_{inf,sup,cen} (call pattern of decimators.h:463-2584):
//   cen: all centre;  inf: Inf,Sup,..,Sup,Cen;  sup: Sup,Inf,..,Inf,Cen;  (L=1: single, L=2: pair)
__host__ __device__ constexpr int dc_mode(int L, int fc, int s)
{
    if (fc == 2) return MODE_CEN;
    const int first = fc == 0 ? MODE_INF : MODE_SUP, other = fc == 0 ? MODE_SUP : MODE_INF;
    if (s == 1) return first;
    if (L >= 3 && s == L) return MODE_CEN;
    return other;
}

template<int L, int FC, int PRE, bool U8>
__global__ __launch_bounds__(DC_THREADS, 3)
void decim_chain_kernel(const DecimJobs jobs, int cps, int post, int in_shift)
{
    typedef typename Quad<U8>::T QT;
    const DecimJob& job = jobs.j[blockIdx.y];             // wave-uniform: scalar loads from the kernarg segment
    const QT* __restrict__ hist = static_cast<const QT*>(job.hist);
    const QT* __restrict__ in = static_cast<const QT*>(job.in);
    uint32_t* __restrict__ out = job.out;
    const uint32_t* __restrict__ flags = job.flags;       // per chunk: recompute? (nullptr: all) -- set by the FAST kernel
    const long n_in = job.n_in;
    const int n_chunks = job.n_units;
    constexpr int C = DC_CHUNK, NT = DC_THREADS;
    constexpr int LPT = C / 4 / NT;                       // uint4 loads per lane per chunk (4)
    __shared__ __attribute__((aligned(16))) uint32_t lds[dc_lds_dwords(L)];

    const int tid = threadIdx.x;
    // Plain run (flags == nullptr): one workgroup per segment.  Fallback run behind the FAST kernel: a
    // small grid strides over the segments and only works on those holding a flagged chunk, so clean
    // input costs one short launch.
    const long n_seg = (n_chunks + cps - 1) / cps;
  for (long seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
    const long first = seg * cps;
    long last = first + cps; if (last > n_chunks) last = n_chunks;
    if (flags) {
        bool any = false;
        for (long c = first + tid; c < last; c += NT) any = any || flags[c] != 0;
        if (!__syncthreads_or(any)) continue;
    }
    const long n_in4 = n_in >> 2;                         // n_in is a multiple of 4 (group sizes are)
    const long n_out = n_in >> L;

    for (int i = tid; i < dc_lds_dwords(L); i += NT) lds[i] = 0;

    QT pre[LPT];
    auto fetch = [&](long chunk) {
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            const int q = j * NT + tid;                   // quad index inside the chunk
            if (chunk < 0) pre[j] = hist[q];
            else {
                const long g = chunk * (C / 4) + q;
                pre[j] = g < n_in4 ? in[g] : Quad<U8>::zero();
            }
        }
    };
    fetch(first - 1);
    __syncthreads();

    for (long chunk = first - 1; chunk < last; ++chunk) {
        // ---- raw samples -> stage-1 packed polyphase arrays (index 16 dwords = 32 int16 of history)
        {
            uint32_t* oI = lds + dc_off(1), *oQ = oI + dc_arr(1), *eI = oQ + dc_arr(1), *eQ = eI + dc_arr(1);
#pragma unroll
            for (int j = 0; j < LPT; j++) {
                const int q = HIST / 2 + j * NT + tid;
                Quad<U8>::split(pre[j], in_shift, eI[q], eQ[q], oI[q], oQ[q]);   // (I0,Q0) (I1,Q1) (I2,Q2) (I3,Q3)
            }
        }
        if (chunk + 1 < last) fetch(chunk + 1);           // in flight during the whole chunk
        __syncthreads();

        const bool live = chunk >= first;                 // warm-up chunk produces no output
        static_for<1, L + 1>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int R = dc_R(s);
            constexpr int NOUT = C >> s;
            constexpr int MODE = dc_mode(L, FC, s);
            int yI[R], yQ[R];
            const bool act = tid < NOUT / R;
            if (act) {
                if constexpr (s == 1) {
                    const uint32_t* oI = lds + dc_off(1), *oQ = oI + dc_arr(1), *eI = oQ + dc_arr(1), *eQ = eI + dc_arr(1);
                    stage_pk16_r8<64, MODE, PRE>(oI, oQ, eI, eQ, tid, yI, yQ);
                } else {
                    const int* oI = reinterpret_cast<const int*>(lds + dc_off(s));
                    const int* oQ = oI + dc_arr(s), *eI = oQ + dc_arr(s), *eQ = eI + dc_arr(s);
                    stage_i32<64, MODE, R, dc_m24(s, PRE)>(oI, oQ, eI, eQ, tid, yI, yQ);
                }
                if constexpr (s < L) {
                    int* oI = reinterpret_cast<int*>(lds + dc_off(s + 1));
                    int* oQ = oI + dc_arr(s + 1), *eI = oQ + dc_arr(s + 1), *eQ = eI + dc_arr(s + 1);
                    put_i32<R>(oI, oQ, eI, eQ, tid, yI, yQ);
                } else if (live) {
                    const long base = chunk * NOUT + (long)R * tid;
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if (base + r < n_out) out[base + r] = pack_iq(yI[r] >> post, yQ[r] >> post);
                }
            }
            __syncthreads();
        });

        // ---- carry: last 32 entries of every array become the next chunk's history
        static_for<1, L + 1>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int HD = s == 1 ? HIST / 2 : HIST;              // history dwords per array
            constexpr int ND = dc_arr(s) - HD;                        // chunk dwords per array
            uint32_t* a = lds + dc_off(s);
            for (int i = tid; i < 4 * HD; i += NT) {
                const int arr = i / HD, e = i % HD;
                a[arr * dc_arr(s) + e] = a[arr * dc_arr(s) + ND + e];
            }
        });
        __syncthreads();
    }
  }   // segments
}

} // namespace sdrx
