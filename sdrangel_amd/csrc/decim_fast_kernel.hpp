// FAST path of the Decimators chain: one WAVE = one private pipeline, no workgroup barrier (NW = 1, long launches; the
// NW = 4 flavour for short launches is described at the kernel).
//
// Each 64-lane workgroup (a single wavefront) owns `spw` consecutive sub-chunks of 1024 input
// samples and walks them in order through all L half-band stages, every stage's history carried
// in its own 11 KB LDS slice; 4 warm-up sub-chunks (4096 >= 62*(2^L-1) samples) in front of the
// segment make that state exact.  With no s_barrier anywhere, the 12-14 resident waves of a CU
// interleave freely on the SIMDs, and every stage keeps all 64 lanes busy:
//     stage 1,2,3 : packed-int16 arms, v_dot2c_i32_i16, 8 / 4 / 2 outputs per lane (I and Q)
//     stage 4     : int32 arms, lane = (output pair, component)
//     stage 5     : int32 arms, lane = (output pair, component, half of the taps)  + DPP add
//     stage 6     : int32 arms, lane = (output pair, component, quarter of the taps) + DPP adds
//
// Exactness: stage 1 is exact for ANY int16 input.  Stages 2 and 3 read their inputs as int16;
// that is exact iff every stage-1 and stage-2 output fits int16, which the kernel CHECKS (OR of
// y + 0x8000 must stay below 2^16).  For 8/12-bit device data honouring its contract the worst-case bound (3.49x per
// stage) guarantees it; otherwise the wave raises the overflow flag of every 4096-sample chunk
// from the failing sub-chunk to the end of its segment and the EXACT kernel (decim_kernel.hpp,
// int32 arms throughout, launched right behind on the same stream) recomputes exactly those
// chunks.  With both outputs < 2^15 the static bounds also make v_mad_i32_i24 exact in stages 4-6.
#pragma once
#include "hb_common.hpp"
#include "hb_mfma.hpp"
#include "decim_kernel.hpp"

namespace sdrx {

constexpr int DF_SUB = 1024;          // input samples per sub-chunk (one wave iteration)
constexpr int DF_WARM = 4;            // warm-up sub-chunks
constexpr int DF_CHUNK = 4096;        // flag granularity == DC_CHUNK of the EXACT kernel

// generic packed-int16 stage, R outputs (I and Q) per lane; see stage_pk16_r8 for the derivation
// The packed tap pair of output r and window dword d depends on e = r - 2d only (pk_coef): DF_NCOEF values, a third of
// them non-zero.  DF_SGPR_COEF = 1 holds them in SGPRs (filled once per kernel) instead of 32-bit literals in every v_dot2c.
#ifndef DF_SGPR_COEF
#define DF_SGPR_COEF 1
#endif
constexpr int DF_NCOEF = 48, DF_COEF_BIAS = 40;
struct DfCoef { uint32_t v[DF_NCOEF]; };
template<int ORDER, int MODE> __device__ __forceinline__ void df_coef_fill(DfCoef& c)
{
    static_for<0, DF_NCOEF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = i - DF_COEF_BIAS;                  // r - 2d; with d = 0: r = e
        constexpr int lo = hb_tap<ORDER>(e + 32), hi = hb_tap<ORDER>(e + 31);
        constexpr uint32_t val = pk16(MODE != MODE_CEN ? -lo : lo, hi);
        uint32_t x = val;
        if constexpr (val != 0) asm volatile("s_mov_b32 %0, %1" : "=s"(x) : "i"(val));
        c.v[i] = x;
    });
}

template<int ORDER, int MODE, int SHL, int R>
__device__ __forceinline__ void stage_pk16(const uint32_t* __restrict__ oI, const uint32_t* __restrict__ oQ,
                                           const uint32_t* __restrict__ eI, const uint32_t* __restrict__ eQ,
                                           int t, int (&yI)[R], int (&yQ)[R], const DfCoef& ctab)
{
    constexpr int P = hb_pairs<ORDER>();
    constexpr int CD = P - 1;
    constexpr int NW = (R + 32) / 2;                 // window dwords: int16 i holds o[k0-32+i]
    constexpr int EB = (32 - CD - 1) / 2, NE = R / 2 + 1;
    uint32_t wI[NW], wQ[NW], vI[NE], vQ[NE];
    const int b = (R / 2) * t;
    if constexpr (R == 8) {
#pragma unroll
        for (int q = 0; q < NW / 4; q++) {
            uint4 a = reinterpret_cast<const uint4*>(oI + b)[q], c = reinterpret_cast<const uint4*>(oQ + b)[q];
            wI[4*q] = a.x; wI[4*q+1] = a.y; wI[4*q+2] = a.z; wI[4*q+3] = a.w;
            wQ[4*q] = c.x; wQ[4*q+1] = c.y; wQ[4*q+2] = c.z; wQ[4*q+3] = c.w;
        }
    } else if constexpr (R == 4) {
#pragma unroll
        for (int q = 0; q < NW / 2; q++) {
            uint2 a = reinterpret_cast<const uint2*>(oI + b)[q], c = reinterpret_cast<const uint2*>(oQ + b)[q];
            wI[2*q] = a.x; wI[2*q+1] = a.y; wQ[2*q] = c.x; wQ[2*q+1] = c.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NW; q++) { wI[q] = oI[b + q]; wQ[q] = oQ[b + q]; }
    }
    if constexpr (R == 8) { ld_centre5<EB>(eI + b, vI); ld_centre5<EB>(eQ + b, vQ); }
    else if constexpr (R == 4 && EB % 2 == 0) {
        // b = 2t: 8-byte aligned; three dwords as b64 + b32
        const uint2 a = *reinterpret_cast<const uint2*>(eI + b + EB), c = *reinterpret_cast<const uint2*>(eQ + b + EB);
        vI[0] = a.x; vI[1] = a.y; vI[2] = eI[b + EB + 2];
        vQ[0] = c.x; vQ[1] = c.y; vQ[2] = eQ[b + EB + 2];
    } else {
#pragma unroll
        for (int q = 0; q < NE; q++) { vI[q] = eI[b + EB + q]; vQ[q] = eQ[b + EB + q]; }
    }

    static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int dd = (r + 1) >> 1, hf = (r + 1) & 1;
        // the accumulator starts as the centre tap (one SDWA shift) instead of 0 + a seventeenth dot2
        int aI, aQ;
        if constexpr (MODE == MODE_CEN) {
            aI = centre_shl<hf>(vI[dd]); aQ = centre_shl<hf>(vQ[dd]);
        } else {
            // centre sample: inf: k odd -> (-im, re), k even -> (im, -re); sup: negated
            constexpr bool neg_first = ((r & 1) == 1) == (MODE == MODE_INF);
            const int cq = centre_shl<hf>(vQ[dd]), ci = centre_shl<hf>(vI[dd]);
            aI = neg_first ? (int)(0u - (uint32_t)cq) : cq;
            aQ = neg_first ? ci : (int)(0u - (uint32_t)ci);
        }
        static_for<0, NW>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            constexpr uint32_t cf = pk_coef<ORDER, MODE>(r, d);
            if constexpr (cf != 0) {
                if constexpr (DF_SGPR_COEF) { const uint32_t cs = ctab.v[r - 2 * d + DF_COEF_BIAS]; aI = dot2(wI[d], cs, aI); aQ = dot2(wQ[d], cs, aQ); }
                else { aI = dot2(wI[d], cf, aI); aQ = dot2(wQ[d], cf, aQ); }
            }
        });
        yI[r] = (int)((uint32_t)aI << SHL) >> (HB_SHIFT - 1);
        yQ[r] = (int)((uint32_t)aQ << SHL) >> (HB_SHIFT - 1);
    });
}

// R outputs -> next stage's PACKED int16 arms (dword 16 = first of the chunk)
template<int R>
__device__ __forceinline__ void put_pk16(uint32_t* __restrict__ oI, uint32_t* __restrict__ oQ,
                                         uint32_t* __restrict__ eI, uint32_t* __restrict__ eQ,
                                         int t, const int (&yI)[R], const int (&yQ)[R])
{
    static_assert(R == 8 || R == 4, "packed output needs at least two samples per arm");
    if constexpr (R == 8) {
        const int p = HIST / 2 + 2 * t;
        *reinterpret_cast<uint2*>(eI + p) = make_uint2(pack_iq(yI[0], yI[2]), pack_iq(yI[4], yI[6]));
        *reinterpret_cast<uint2*>(oI + p) = make_uint2(pack_iq(yI[1], yI[3]), pack_iq(yI[5], yI[7]));
        *reinterpret_cast<uint2*>(eQ + p) = make_uint2(pack_iq(yQ[0], yQ[2]), pack_iq(yQ[4], yQ[6]));
        *reinterpret_cast<uint2*>(oQ + p) = make_uint2(pack_iq(yQ[1], yQ[3]), pack_iq(yQ[5], yQ[7]));
    } else {
        const int p = HIST / 2 + t;
        eI[p] = pack_iq(yI[0], yI[2]); oI[p] = pack_iq(yI[1], yI[3]);
        eQ[p] = pack_iq(yQ[0], yQ[2]); oQ[p] = pack_iq(yQ[1], yQ[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// int32-arm stage for the low-rate tail: lane = (output pair p, component, tap slice h).
// Two outputs k = 2p, 2p+1 of ONE component per lane, taps [h*16/SPLIT, (h+1)*16/SPLIT) of the 16
// coefficient pairs, partial sums combined over the SPLIT neighbouring lanes.  Every lane of a
// (p, component) group ends up holding the finished outputs.
// ---------------------------------------------------------------------------------------------
// acc + d * c as ONE v_mad_i32_i24 (4 cycles).  Left to itself the compiler turns the chain into v_mul_i32_i24 plus one
// v_add3_u32 per two products (4 + 2 cycles per tap pair).  KS: the coefficient is a compile-time constant -> SGPR operand.
template<bool KS>
__device__ __forceinline__ int mad24(int acc, int d, int c)
{
    int r;
    if constexpr (KS) asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(d), "s"(c), "v"(acc));
    else asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(d), "v"(c), "v"(acc));
    return r;
}

template<int MODE, int SPLIT, bool M24>
__device__ __forceinline__ void stage_i32_split(const int* __restrict__ oI, const int* __restrict__ oQ,
                                                const int* __restrict__ eI, const int* __restrict__ eQ,
                                                int lane, int (&y)[2], int& p_out, int& comp_out)
{
    constexpr int P = 16, TAPS = 32, CD = 15, PP = P / SPLIT;
    const int h = lane % SPLIT, comp = (lane / SPLIT) & 1, p = lane / (2 * SPLIT);
    const int* o = comp ? oQ : oI;
    // a[j] = o[k - i], b[j] = o[k - 31 + i] for i = h*PP + ii; indices relative to entry 32 + 2p
    const int abase = HIST + 2 * p - (h * PP + PP - 1);          // o index of a for (r = 0, ii = PP-1)
    const int bbase = HIST + 2 * p - (TAPS - 1) + h * PP;         // o index of b for (r = 0, ii = 0)
    int a[PP + 1], b[PP + 1];
#pragma unroll
    for (int q = 0; q < PP + 1; q++) { a[q] = o[abase + q]; b[q] = o[bbase + q]; }
    int acc[2] = { 0, 0 };
    static_for<0, 2>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        static_for<0, PP>([&](auto ic) {
            constexpr int ii = decltype(ic)::value;
            // coefficient index i = h*PP + ii is lane dependent: fetch c[i] from a tiny per-lane table
            const int av = a[PP - 1 - ii + r], bv = b[ii + r];
            int cf;
            if constexpr (SPLIT == 1) cf = hb_c<64>(ii);
            else if constexpr (SPLIT == 2) cf = h ? hb_c<64>(PP + ii) : hb_c<64>(ii);
            else cf = h == 0 ? hb_c<64>(ii) : h == 1 ? hb_c<64>(PP + ii) : h == 2 ? hb_c<64>(2 * PP + ii) : hb_c<64>(3 * PP + ii);
            if constexpr (MODE == MODE_CEN) {
                acc[r] = M24 ? mad24<SPLIT == 1>(acc[r], (int)((uint32_t)av + (uint32_t)bv), cf) : mac<false>(acc[r], (int)((uint32_t)av + (uint32_t)bv), cf);
            } else {
                // s(m) = (-1)^(m+1), m = k - i, k = 2p + r, i = h*PP + ii with PP even: parity of r - ii
                constexpr int sg = (((r - ii) & 1) == 0) ? -1 : 1;
                acc[r] = M24 ? mad24<SPLIT == 1>(acc[r], (int)((uint32_t)av - (uint32_t)bv), sg * cf) : mac<false>(acc[r], (int)((uint32_t)av - (uint32_t)bv), sg * cf);
            }
        });
    });
    // partner lanes sit inside one quad: DPP quad_perm (a VALU operand modifier), not ds_bpermute (an LDS round trip)
    if constexpr (SPLIT >= 2) { acc[0] += __builtin_amdgcn_mov_dpp(acc[0], 0xB1, 0xf, 0xf, true); acc[1] += __builtin_amdgcn_mov_dpp(acc[1], 0xB1, 0xf, 0xf, true); }
    if constexpr (SPLIT >= 4) { acc[0] += __builtin_amdgcn_mov_dpp(acc[0], 0x4E, 0xf, 0xf, true); acc[1] += __builtin_amdgcn_mov_dpp(acc[1], 0x4E, 0xf, 0xf, true); }
    // centre tap e[k - 15]: own component for centre mode, the other one (signed) for inf/sup
    const int* e = (MODE == MODE_CEN) ? (comp ? eQ : eI) : (comp ? eI : eQ);
    const int cbase = HIST + 2 * p - CD;
    static_for<0, 2>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const uint32_t c = (uint32_t)e[cbase + r] << (HB_SHIFT - 1);
        uint32_t u;
        if constexpr (MODE == MODE_CEN) u = (uint32_t)acc[r] + c;
        else {
            // I: -eQ when neg_first else +eQ;  Q: +eI when neg_first else -eI
            constexpr bool neg_first = ((r & 1) == 1) == (MODE == MODE_INF);
            const bool sub = neg_first != (comp != 0);
            u = sub ? (uint32_t)acc[r] - c : (uint32_t)acc[r] + c;
        }
        y[r] = (int)u >> (HB_SHIFT - 1);
    });
    p_out = p; comp_out = comp;
}

// LDS layout of one wave (dwords)
__host__ __device__ constexpr bool df_in16(int s) { return s <= 3; }          // stage s reads packed int16 arms
__host__ __device__ constexpr int df_arr(int s, int S = DF_SUB)
{
    // int32 (tail) arrays are only ever touched with 4-byte accesses by lanes that alternate between the I and
    // the Q array: an ODD array pitch puts the two on opposite bank parities (SQ_LDS_BANK_CONFLICT was 24 % of
    // the LDS cycles with the natural 96/64/48-dword pitches, all multiples of 16 banks)
    return df_in16(s) ? (HIST / 2 + (S >> (s + 1))) : (HIST + (S >> s) + 1);
}
__host__ __device__ constexpr int df_off(int s, int S = DF_SUB) { int o = 0; for (int u = 1; u < s; u++) o += 4 * df_arr(u, S); return o; }
__host__ __device__ constexpr int df_lds_dwords(int L, int S = DF_SUB) { return df_off(L + 1, S); }
// The same layout with the pitch the MATRIX-CORE TAIL wants (MXT: single-wave matrix-core flavour, stages 4..6 as one MFMA tile): the
// int32 arrays then have an even pitch (their windows are read as aligned 16-byte vectors); the dot2 tail keeps the odd one.
// PAD (matrix-core engine): the packed arrays of stages 1..3 get 16 more dwords, so that their pitch is 32 mod 64 dwords: a tile reads
// the I and the Q array in ONE wave instruction (columns alternate I / Q) and 128 bytes between the two keep the ds_read_b128 conflict-free.
template<bool MXT, bool PAD = false> struct DfLay {
    static __host__ __device__ constexpr int arr(int s, int S = DF_SUB) { return df_in16(s) ? (HIST / 2 + (S >> (s + 1)) + (PAD ? 16 : 0)) : (HIST + (S >> s) + (MXT ? 0 : 1)); }
    static __host__ __device__ constexpr int off(int s, int S = DF_SUB) { int o = 0; for (int u = 1; u < s; u++) o += 4 * arr(u, S); return o; }
    static __host__ __device__ constexpr int total(int L, int S = DF_SUB) { return off(L + 1, S); }
};
constexpr uint32_t HBM_BIAS4 = 0x00808080u;       // XORed into every int32 odd-arm entry the matrix-core tail reads (bytes 0..2 as signed)

// NW = waves per workgroup.  NW = 1: the wave-private pipeline described at the top (sub-chunks of 1024 samples, 4 warm-up
// sub-chunks per segment).  NW = 4: the same lane work on sub-chunks of 4096 samples with a real barrier behind every stage and ONE
// warm-up sub-chunk per segment -- a quarter of the warm-up per wave, for calls too short to give every SIMD several
// single-wave segments of a useful length (the host picks, sdrx_decim.hip).
// MX = true (the default engine): the stages that read packed int16 arms (1..3) evaluate their odd-arm FIR on the matrix
// cores (hb_mfma.hpp: v_mfma_i32_16x16x64_i8, exact int32 sums) instead of v_dot2c; same LDS layout, same overflow guard,
// same int32 tail stages.  The odd arms those stages read carry 0x0080 XORed into every int16 (the primitive's byte bias).
template<int L, int FC, int PRE, bool U8, int NW, bool MX>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(MX ? 3 : 1)))
void decim_fast_kernel(const DecimJobs jobs, int spw, int post, int in_shift)
{
    typedef typename Quad<U8>::T QT;
    const DecimJob& job = jobs.j[blockIdx.y];              // blockIdx.y = device stream; scalar loads from the kernarg segment
    const QT* __restrict__ hist = static_cast<const QT*>(job.hist);
    const QT* __restrict__ in = static_cast<const QT*>(job.in);
    uint32_t* __restrict__ out = job.out;
    uint32_t* __restrict__ ovf_flags = job.flags;          // one per DF_CHUNK-sample chunk of this call
    const long n_in = job.n_in;
    const int n_sub = job.n_units;
    constexpr int S = DF_SUB * NW, NT = 64 * NW, WARM = DF_WARM / NW, LPT = S / 4 / NT;   // 4 uint4 per lane per sub-chunk
    static_assert(NW == 1 || NW == 2 || NW == 4, "warm-up = 4096 samples = a whole number of sub-chunks (2 measured: never the best)");
    constexpr bool MXT = MX && NW == 1 && L >= 4;              // stages 4..L as ONE matrix-core tile per iteration (needs the skewed loop)
    typedef DfLay<MXT, MX> DL;
    __shared__ __attribute__((aligned(16))) uint32_t lds[DL::total(L, S)];
    const int lane = threadIdx.x;
    const long first = (long)blockIdx.x * spw;
    if (first >= n_sub) return;
    long last = first + spw; if (last > n_sub) last = n_sub;
    const long n_in4 = n_in >> 2, n_out = n_in >> L;

    for (int i = lane; i < DL::total(L, S); i += NT) lds[i] = 0;
    constexpr int MXS = MX ? (L < 3 ? L : 3) : 0;             // stages 1..MXS run on the matrix cores
    if constexpr (MX) {
        __syncthreads();
        static_for<1, MXS + 1>([&](auto sc) {                  // a zero sample of a biased odd arm is 0x0080
            constexpr int s = decltype(sc)::value;
            for (int i = lane; i < 2 * DL::arr(s, S); i += NT) lds[DL::off(s, S) + i] = HBM_BIAS2;
        });
    }

    DfCoef ctab_cen, ctab_rot;
    if constexpr (DF_SGPR_COEF && !MX) { df_coef_fill<64, MODE_CEN>(ctab_cen); df_coef_fill<64, MODE_INF>(ctab_rot); }
    // matrix-core operands: plain taps for centre stages, alternating-sign taps for inf/sup stages (the rotation of the odd arm
    // folded into the taps: exact modulo 2^32).  Only the sets the chain's first three stages use are built.
    constexpr bool NEED_CEN = MXT || (MX && (dc_mode(L, FC, 1) == MODE_CEN || (MXS >= 2 && dc_mode(L, FC, 2) == MODE_CEN) || (MXS >= 3 && dc_mode(L, FC, 3) == MODE_CEN)));
    constexpr bool NEED_ROT = MX && (dc_mode(L, FC, 1) != MODE_CEN || (MXS >= 2 && dc_mode(L, FC, 2) != MODE_CEN) || (MXS >= 3 && dc_mode(L, FC, 3) != MODE_CEN));
    HbMfmaTaps<64, false> taps_cen; HbMfmaTaps<64, true> taps_rot;
    const int wl = lane & 63, n16 = wl & 15, g4 = wl >> 4, comp = n16 & 1, bn = n16 >> 1;      // tile column n: block n / 2 of component n & 1
    const int wv = MX ? __builtin_amdgcn_readfirstlane(lane >> 6) : 0;
    if constexpr (NEED_CEN) taps_cen.init(wl);
    if constexpr (NEED_ROT) taps_rot.init(wl);
    // ---- matrix-core tail (MXT): stages 4..L of one iteration form ONE pair of tiles.  A tail stage keeps its odd arm as TWO packed
    // int16 planes -- low halves (biased by 0x8000: u = s + 32768) and high halves of the int32 samples -- so that the int16 primitive
    // of hb_mfma.hpp and the operands already resident for stages 1..3 serve it: sum h x = 65536 sum h xh + sum h s + 32768 sum h, exact
    // modulo 2^32 for ANY int32 input like the reference's accumulator (the dot2 tail needs 24-bit differences).  Columns 0..7 = stage 4
    // (4 blocks x I, Q), 8..11 = stage 5, 12..13 = stage 6; the stages work on different sub-chunks (skewed loop), so they are independent.
    // The rotation of an inf/sup stage's odd arm is applied by its PRODUCER (int32: -x is exact modulo 2^32): every column uses the plain taps.
    // Layout of a tail stage (same 4 (32 + n) dwords as four int32 arms): oI.lo oI.hi oQ.lo oQ.hi (16 + n / 2 dwords each) eI eQ (32 + n each).
    int t_lo = 0, t_pp = 0, t_co = 0, t_no = 0, t_npp = 0, t_ne = 0, t_blk = 0, t_stage = 4;
    uint32_t t_me = 0, t_mo = 0, t_ng = 0;
    bool t_act = false, t_next = false;
    if constexpr (MXT) {
        __syncthreads();
        static_for<4, L + 1>([&](auto sc) {                    // zero samples of the biased planes
            constexpr int s = decltype(sc)::value;
            constexpr int pp = HIST / 2 + (S >> (s + 1));
            for (int i = lane; i < 4 * pp; i += NT) lds[DL::off(s, S) + i] = ((i / pp) & 1) ? HBM_BIAS2 : (HBM_BIAS2 | 0x80008000u);
        });
        const int col = n16;
        t_stage = col < 8 ? 4 : col < 12 ? 5 : 6;
        t_act = t_stage <= L && col < 14;
        if (!t_act) t_stage = 4;                                // idle columns mirror a valid one (reads only)
        t_blk = !t_act ? 0 : t_stage == 4 ? (col >> 1) : t_stage == 5 ? ((col - 8) >> 1) : 0;
        const int tc = col & 1;
        auto pick = [&](int a4, int a5, int a6) { return t_stage == 4 ? a4 : t_stage == 5 ? a5 : a6; };
        constexpr int M4 = dc_mode(L, FC, 4), M5 = L >= 5 ? dc_mode(L, FC, 5) : MODE_CEN, M6 = L >= 6 ? dc_mode(L, FC, 6) : MODE_CEN;
        const int tmode = pick(M4, M5, M6);
        const int tn = pick(S >> 4, S >> 5, S >> 6), nn = tn >> 1;              // entries per arm and sub-chunk: this stage, the next one
        const int toff = pick(DL::off(4, S), DL::off(5, S), DL::off(6, S)), noff = pick(DL::off(5, S), DL::off(6, S), DL::off(7, S));
        t_pp = HIST / 2 + tn / 2; t_npp = HIST / 2 + nn / 2;
        const int tec = tmode == MODE_CEN ? tc : 1 - tc;        // inf/sup: the centre tap comes from the other component
        t_lo = toff + 2 * tc * t_pp + 8 * t_blk + 4 * g4;       // window entry 0 of the block = int16 index 16 blk of the low plane
        t_co = toff + 4 * t_pp + tec * (HIST + tn) + 16 * t_blk + 4 * g4 + 16;   // centre taps e[k - 15]: int32 entries 16 blk + 4 g + 17 + i
        t_next = t_act && t_stage < L;
        t_no = noff + 2 * tc * t_npp + HIST / 2 + 4 * t_blk + g4;               // the lane's two odd outputs: one packed dword per plane
        t_ne = noff + 4 * t_npp + tc * (HIST + nn) + HIST + 8 * t_blk + 2 * g4; // its two even outputs: two int32
        // centre-tap sign for even / odd outputs (0: +, ~0: -): inf: k even -> (+im, -re), k odd -> (-im, +re); sup: negated
        const bool neg_even = tmode != MODE_CEN && ((tc == 0) != (tmode == MODE_INF));
        t_me = tmode == MODE_CEN ? 0u : (neg_even ? ~0u : 0u);
        t_mo = tmode == MODE_CEN ? 0u : (neg_even ? 0u : ~0u);
        // the consumer's rotation of the odd arm, applied here: entry index even -> negated
        const int nmode = pick(M5, M6, MODE_CEN);
        t_ng = (t_next && nmode != MODE_CEN) ? ~0u : 0u;
    }

    QT pre[LPT];
    // A sub-chunk is either wholly history (sub < 0) or wholly input: the source is chosen with a wave-uniform (scalar)
    // branch and the only per-lane test is a 32-bit compare against the quads left in the stream.
    auto fetch = [&](long sub) {
        const QT* __restrict__ src;
        long left;
        if (sub < 0) { src = hist + (sub + WARM) * (S / 4); left = S / 4; }
        else { src = in + sub * (S / 4); left = n_in4 - sub * (S / 4); }
        const int lim = left > S / 4 ? S / 4 : (int)left;
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            const int q = j * NT + lane;
            pre[j] = q < lim ? src[q] : Quad<U8>::zero();
        }
    };
    fetch(first - WARM);
    bool bad = false;
    uint32_t ovf_or = 0;                                       // OR of (y + 0x8000) over every int16-stored output so far
    __syncthreads();

    // raw sub-chunk (already in registers) -> the four packed arms of stage 1
    auto split_in = [&]() {
        uint32_t* oI = lds + DL::off(1, S), *oQ = oI + DL::arr(1, S), *eI = oQ + DL::arr(1, S), *eQ = eI + DL::arr(1, S);
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            const int q = HIST / 2 + j * NT + lane;
            uint32_t a, b, c, d;
            Quad<U8>::split(pre[j], in_shift, a, b, c, d);
            eI[q] = a; eQ[q] = b;
            if constexpr (MX) { oI[q] = c ^ HBM_BIAS2; oQ[q] = d ^ HBM_BIAS2; } else { oI[q] = c; oQ[q] = d; }
        }
    };
    // stage s over sub-chunk `sub` (arrays of stage s -> arrays of stage s + 1, or the output when s == L and `live`)
    auto do_stage = [&](auto sc, const long sub, const bool live) {
        constexpr int s = decltype(sc)::value;
        constexpr int MODE = dc_mode(L, FC, s);
        constexpr int NOUT = S >> s;
        const uint32_t* iI = lds + DL::off(s, S), *iQ = iI + DL::arr(s, S), *jI = iQ + DL::arr(s, S), *jQ = jI + DL::arr(s, S);   // oI,oQ,eI,eQ
        uint32_t* nI = lds + DL::off(s + 1, S);                // next stage: oI, oQ, eI, eQ
        if constexpr (MX && s <= 3) {
            // ---- the stage on the matrix cores.  A tile = 8 blocks of 16 consecutive outputs of I (columns 0..7) and of Q
            // (columns 8..15); lane (n, g) ends up with outputs 4g .. 4g+3 of block 8 t + (n & 7) of component n >> 3.
            constexpr int SHL = (s == 1 ? PRE : 0);
            constexpr int TPW = 8 >> s;                                // tiles per wave and sub-chunk: 4, 2, 1
            constexpr int arr = DL::arr(s, S), arr2 = DL::arr(s + 1, S);
            constexpr bool NEXT16 = s < L && s + 1 <= 3;              // the outputs are re-read as packed int16 (by an MFMA stage)
            const int ec = MODE == MODE_CEN ? comp : 1 - comp;         // inf/sup: the centre tap comes from the other component
            // sign of the centre tap for even / odd outputs: inf: k even -> (+im, -re), k odd -> (-im, +re); sup: negated
            const int m_even = MODE == MODE_CEN ? 2048 : ((comp == 0) == (MODE == MODE_INF) ? 2048 : -2048);
            const int m_odd = MODE == MODE_CEN ? 2048 : -m_even;
            const uint32_t* ob = iI + comp * arr + 8 * bn + 4 * g4;    // window entry 0 of block bn: int16 index 16 blk
            const uint32_t* eb = jI + ec * arr + 8 * bn + 2 * g4 + 8;  // centre taps e[k - 15]: int16 entries 16 blk + 4 g + 17 + i
            constexpr int BIASV = MODE == MODE_CEN ? HbMfmaTaps<64, false>::BIAS : HbMfmaTaps<64, true>::BIAS;
            const v4i bias = { BIASV, BIASV, BIASV, BIASV };
            // every load of the stage's tiles first, then the MFMAs, then the epilogues: the stores of one tile must not sit
            // between the loads of the next (same LDS array as far as the compiler can tell)
            // (two tiles at a time: four in flight cost 160 VGPRs and a wave per SIMD)
            constexpr int TG = TPW < 2 ? TPW : 2;
            static_for<0, TPW / TG>([&](auto gc) {
            constexpr int g0 = decltype(gc)::value * TG;
            v4i b0[TG], b1[TG], S4[TG]; uint2 c01[TG]; uint32_t c2[TG];
            static_for<0, TG>([&](auto tc) {
                constexpr int tt = decltype(tc)::value;
                const int t = wv * TPW + g0 + tt;
                b0[tt] = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(ob + 64 * t, 16));
                b1[tt] = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(ob + 64 * t + 16, 16));
                c01[tt] = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(eb + 64 * t, 8));
                c2[tt] = eb[64 * t + 2];
            });
            static_for<0, TG>([&](auto tc) {
                constexpr int tt = decltype(tc)::value;
                if constexpr (MODE == MODE_CEN) S4[tt] = taps_cen.tile(b0[tt], b1[tt], bias); else S4[tt] = taps_rot.tile(b0[tt], b1[tt], bias);
            });
            static_for<0, TG>([&](auto tc) {
                constexpr int tt = decltype(tc)::value;
                const int t = wv * TPW + g0 + tt;
                const int e[4] = { (int)c01[tt].x >> 16, (int)(int16_t)c01[tt].y, (int)c01[tt].y >> 16, (int)(int16_t)c2[tt] };
                int y[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // acc = S +- (e << 11);  y = (acc << SHL) >> 11 with the 32-bit wrap of the reference's accumulator
                    const int acc = __mul24(e[i], (i & 1) ? m_odd : m_even) + S4[tt][i];
                    y[i] = (int)((uint32_t)acc << SHL) >> (HB_SHIFT - 1);
                }
                const int blk = 8 * t + bn;
                if constexpr (s < L) {
                    if constexpr (NEXT16) {
#pragma unroll
                        for (int i = 0; i < 4; i++) ovf_or |= (uint32_t)y[i] + 0x8000u;
                        const int p = HIST / 2 + 4 * blk + g4;
                        nI[comp * arr2 + p] = pack_iq(y[1], y[3]) ^ HBM_BIAS2;               // odd arm of the next stage
                        nI[(2 + comp) * arr2 + p] = pack_iq(y[0], y[2]);                     // even arm
                    } else {
                        if constexpr (MXT) {
                            // matrix-core tail: the odd outputs go to stage 4's two packed planes (low halves biased by 0x8000, both by the
                            // primitive's 0x0080), stage 4's rotation (if any) applied here: even entry index -> negated; even outputs: int32
                            constexpr bool ROT4 = dc_mode(L, FC, 4) != MODE_CEN;
                            constexpr int pp4 = HIST / 2 + (S >> 5), n4 = S >> 4;
                            const uint32_t y1 = (uint32_t)(ROT4 ? -y[1] : y[1]), y3 = (uint32_t)y[3];
                            const int q = 2 * comp * pp4 + HIST / 2 + 4 * blk + g4;
                            nI[q] = __builtin_amdgcn_perm(y3, y1, 0x05040100u) ^ (HBM_BIAS2 | 0x80008000u);
                            nI[q + pp4] = __builtin_amdgcn_perm(y3, y1, 0x07060302u) ^ HBM_BIAS2;
                            int* de = reinterpret_cast<int*>(nI) + 4 * pp4 + comp * (HIST + n4) + HIST + 8 * blk + 2 * g4;
                            de[0] = y[0]; de[1] = y[2];
                        } else {
                            int* d = reinterpret_cast<int*>(nI);                             // stage 4: int32 arms oI, oQ, eI, eQ
                            const int p = HIST + 8 * blk + 2 * g4;
                            d[comp * arr2 + p] = y[1]; d[comp * arr2 + p + 1] = y[3];
                            d[(2 + comp) * arr2 + p] = y[0]; d[(2 + comp) * arr2 + p + 1] = y[2];
                        }
                    }
                } else {
                    // last stage: the partner lane (n ^ 8, same row of 16) holds the other component of the same four outputs;
                    // the I lane stores outputs 0, 1 and the Q lane outputs 2, 3 after one DPP exchange (row_ror:8)
                    const uint32_t p01 = pack_iq(y[0] >> post, y[1] >> post), p23 = pack_iq(y[2] >> post, y[3] >> post);
                    const uint32_t keep = comp ? p23 : p01, snd = comp ? p01 : p23;
                    const uint32_t rcv = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd, 0xB1, 0xf, 0xf, true);
                    const uint32_t rep = comp ? rcv : keep, imp = comp ? keep : rcv;
                    if (live) {
                        const long base = sub * NOUT + 16 * blk + 4 * g4 + 2 * comp;
                        if (base < n_out)     out[base]     = __builtin_amdgcn_perm(imp, rep, 0x05040100u);
                        if (base + 1 < n_out) out[base + 1] = __builtin_amdgcn_perm(imp, rep, 0x07060302u);
                    }
                }
            });
            });
        } else if constexpr (s <= 3) {
            constexpr int R = 16 >> s;                     // 8, 4, 2
            int yI[R], yQ[R];
            stage_pk16<64, MODE, (s == 1 ? PRE : 0), R>(iI, iQ, jI, jQ, lane, yI, yQ, MODE == MODE_CEN ? ctab_cen : ctab_rot);
            if constexpr (s < L) {
                if constexpr (s + 1 <= 3) {
                    // these outputs are re-read as int16: y fits iff (y + 0x8000) has no bit above 15.  add + or
                    // are full-rate VALU ops on gfx950, v_max/v_min are half rate (profiles/r01_valu_issue_rates.txt)
#pragma unroll
                    for (int r = 0; r < R; r++)
                        ovf_or |= ((uint32_t)yI[r] + 0x8000u) | ((uint32_t)yQ[r] + 0x8000u);
                    put_pk16<R>(nI, nI + DL::arr(s + 1, S), nI + 2 * DL::arr(s + 1, S), nI + 3 * DL::arr(s + 1, S), lane, yI, yQ);
                } else {
                    int* d = reinterpret_cast<int*>(nI);
                    put_i32<R>(d, d + DL::arr(s + 1, S), d + 2 * DL::arr(s + 1, S), d + 3 * DL::arr(s + 1, S), lane, yI, yQ);
                }
            } else if (live) {
                const long base = sub * NOUT + (long)R * lane;
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (base + r < n_out) out[base + r] = pack_iq(yI[r] >> post, yQ[r] >> post);
            }
        } else {
            constexpr int SPLIT = s == 4 ? 1 : s == 5 ? 2 : 4;
            int y[2], p, comp;
            stage_i32_split<MODE, SPLIT, true>(reinterpret_cast<const int*>(iI), reinterpret_cast<const int*>(iQ),
                                               reinterpret_cast<const int*>(jI), reinterpret_cast<const int*>(jQ),
                                               lane, y, p, comp);
            if constexpr (s < L) {
                int* d = reinterpret_cast<int*>(nI);       // oI, oQ, eI, eQ of the next stage
                int* od = d + (comp ? DL::arr(s + 1, S) : 0);
                int* ed = d + (comp ? 3 * DL::arr(s + 1, S) : 2 * DL::arr(s + 1, S));
                ed[HIST + p] = y[0]; od[HIST + p] = y[1];
            } else {
                // partner component sits SPLIT lanes up; lanes with comp == 0 and slice 0 store
                // lane + SPLIT: quad_perm [1,2,3,3] / [2,3,3,3] inside a quad, row_shl:4 across quads (only lanes whose partner exists store)
                constexpr int DN = SPLIT == 1 ? 0xF9 : SPLIT == 2 ? 0xFE : 0x104;
                const int q0 = __builtin_amdgcn_mov_dpp(y[0], DN, 0xf, 0xf, true), q1 = __builtin_amdgcn_mov_dpp(y[1], DN, 0xf, 0xf, true);
                if (live && comp == 0 && (lane % SPLIT) == 0) {
                    const long base = sub * NOUT + 2 * p;
                    if (base < n_out)     out[base]     = pack_iq(y[0] >> post, q0 >> post);
                    if (base + 1 < n_out) out[base + 1] = pack_iq(y[1] >> post, q1 >> post);
                }
            }
        }
    };

    if constexpr (MX && NW == 1) {
        // ---- SKEWED pipeline (single-wave workgroups, matrix-core engine): in iteration `it` stage s works on sub-chunk it - (s - 1),
        // highest stage first.  A stage reads only what the PREVIOUS iteration wrote, so nothing inside an iteration waits for
        // anything else inside it: no fence between the stages (one per iteration), the LDS reads of a stage overlap the MFMAs and
        // the epilogue of the one before.  L - 1 extra iterations drain the pipe.  Carry: the tails of all arrays are read at the
        // top of the iteration (nothing has been overwritten yet); an array's head is rewritten right after its consumer stage has
        // read it -- stage 1's array before the next raw sub-chunk is split into it.
        const auto cbar = [] { asm volatile("" ::: "memory"); };            // compiler-only: LDS operations of one wave execute in order
        for (long it = first - WARM; it < last + (L - 1); ++it) {
            uint32_t keep[L][2];
            static_for<1, L + 1>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const uint32_t* a = lds + DL::off(s, S);
                if constexpr (MXT && s >= 4) {
                    // matrix-core tail layout: four packed planes (16 + n / 2 dwords) then two int32 even arms (32 + n)
                    constexpr int n = S >> s, pp = HIST / 2 + n / 2;
                    keep[s - 1][0] = a[(lane / 16) * pp + n / 2 + (lane % 16)];
                    keep[s - 1][1] = a[4 * pp + (lane / 32) * (HIST + n) + n + (lane % 32)];
                } else {
                    constexpr int HD = df_in16(s) ? HIST / 2 : HIST;
                    constexpr int ND = df_in16(s) ? (S >> (s + 1)) : (S >> s);
                    constexpr int PER = 4 * HD / NT;
#pragma unroll
                    for (int q = 0; q < PER; q++) { const int i = q * NT + lane; keep[s - 1][q] = a[(i / HD) * DL::arr(s, S) + ND + (i % HD)]; }
                }
            });
            cbar();
            auto put_head = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int HD = df_in16(s) ? HIST / 2 : HIST;
                constexpr int PER = 4 * HD / NT;
                uint32_t* a = lds + DL::off(s, S);
#pragma unroll
                for (int q = 0; q < PER; q++) { const int i = q * NT + lane; a[(i / HD) * DL::arr(s, S) + (i % HD)] = keep[s - 1][q]; }
            };
            put_head(std::integral_constant<int, 1>{});                         // stage 1 read this array in the previous iteration
            cbar();
            if (it < last) {
                split_in();
                if constexpr (!MXT) { if (it + 1 < last) fetch(it + 1); }
            }
            cbar();
            if constexpr (MXT) {
                // ---- stages 4..L: one pair of tiles (low planes, high planes), the int16 primitive with the resident operands
                const long subS = it - (t_stage - 1);                          // the sub-chunk this lane's stage works on
                const bool on = t_act && subS >= first - WARM && subS < last;
                const v4i l0 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(lds + t_lo, 16));
                const v4i l1 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(lds + t_lo + 16, 16));
                const v4i h0 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(lds + t_lo + t_pp, 16));
                const v4i h1 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(lds + t_lo + t_pp + 16, 16));
                const v4i cq = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(lds + t_co, 16));
                const uint32_t c4 = lds[t_co + 4];
                cbar();
                // every tail array has been read: its head may move on.  Planes: 4 x 16 dwords, even arms: 2 x 32 dwords per stage.
                static_for<4, L + 1>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    constexpr int n = S >> s, pp = HIST / 2 + n / 2;
                    uint32_t* a = lds + DL::off(s, S);
                    a[(lane / 16) * pp + (lane % 16)] = keep[s - 1][0];
                    a[4 * pp + (lane / 32) * (HIST + n) + (lane % 32)] = keep[s - 1][1];
                });
                cbar();
                constexpr int BV = HbMfmaTaps<64, false>::BIAS;
                const v4i biasv = { BV, BV, BV, BV };
                const v4i SL = taps_cen.tile(l0, l1, biasv), SH = taps_cen.tile(h0, h1, biasv);
                const uint32_t ce[4] = { (uint32_t)cq[1], (uint32_t)cq[2], (uint32_t)cq[3], c4 };
                int y[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // x = 65536 xh + s + 32768:  S = (SH << 16) + SL + 32768 sum h (modulo 2^32)
                    const uint32_t S4 = ((uint32_t)SH[i] << 16) + (uint32_t)SL[i] + (uint32_t)(32768 * hb_tap_sum<64>());
                    const uint32_t m = (i & 1) ? t_mo : t_me;                   // acc = S +- (e << 11), 32-bit wrap
                    const uint32_t acc = S4 + (((ce[i] << (HB_SHIFT - 1)) ^ m) - m);
                    y[i] = (int)acc >> (HB_SHIFT - 1);
                }
                if (on && t_next) {
                    const uint32_t y1 = ((uint32_t)y[1] ^ t_ng) - t_ng, y3 = (uint32_t)y[3];   // the consumer's rotation: even entry index negated
                    lds[t_no] = __builtin_amdgcn_perm(y3, y1, 0x05040100u) ^ (HBM_BIAS2 | 0x80008000u);
                    lds[t_no + t_npp] = __builtin_amdgcn_perm(y3, y1, 0x07060302u) ^ HBM_BIAS2;
                    int* d = reinterpret_cast<int*>(lds);
                    *reinterpret_cast<int2*>(d + t_ne) = make_int2(y[0], y[2]);
                }
                {
                    // last stage: the partner lane n ^ 1 holds the other component; the I lane stores outputs 0, 1 and the Q lane 2, 3
                    const int tc = n16 & 1;
                    const uint32_t p01 = pack_iq(y[0] >> post, y[1] >> post), p23 = pack_iq(y[2] >> post, y[3] >> post);
                    const uint32_t keepv = tc ? p23 : p01, snd = tc ? p01 : p23;
                    const uint32_t rcv = (uint32_t)__builtin_amdgcn_mov_dpp((int)snd, 0xB1, 0xf, 0xf, true);
                    const uint32_t rep = tc ? rcv : keepv, imp = tc ? keepv : rcv;
                    // the prefetch of the next raw sub-chunk goes out HERE: its 16 registers are free while the tile above runs -- and BEFORE the
                    // output stores: the compiler drains vmcnt in front of the prefetch (its registers may still be pending on the drain path),
                    // which behind the stores was a wait for their acknowledgement in every iteration
                    const uint32_t w0 = __builtin_amdgcn_perm(imp, rep, 0x05040100u), w1 = __builtin_amdgcn_perm(imp, rep, 0x07060302u);
                    cbar();
                    if (it + 1 < last) fetch(it + 1);
                    cbar();
                    if (on && t_stage == L && subS >= first) {
                        const long base = subS * (S >> L) + 16 * t_blk + 4 * g4 + 2 * tc;
                        if (base < n_out)     out[base]     = w0;
                        if (base + 1 < n_out) out[base + 1] = w1;
                    }
                }
                cbar();
            }
            static_for<0, (MXT ? 3 : L)>([&](auto ic) {
                constexpr int s = (MXT ? 3 : L) - decltype(ic)::value;
                const long sub = it - (s - 1);
                if (sub >= first - WARM && sub < last) do_stage(std::integral_constant<int, s>{}, sub, sub >= first);
                cbar();
                if constexpr (s >= 2) put_head(std::integral_constant<int, s>{});
                cbar();
            });
            if constexpr (L >= 2) { if (!bad && __any((ovf_or >> 16) != 0)) bad = true; }
            const long subL = it - (L - 1);
            // conservative: `bad` may already hold an overflow of a LATER sub-chunk (stage 1 is L - 1 sub-chunks ahead)
            if (subL >= first && subL < last && lane == 0 && ((subL + 1) % (DF_CHUNK / S) == 0 || subL + 1 == last))
                ovf_flags[subL / (DF_CHUNK / S)] = bad ? 1u : 0u;
            __syncthreads();                                                    // single wave: a fence, no s_barrier
        }
        return;
    }

    for (long sub = first - WARM; sub < last; ++sub) {
        split_in();
        if (sub + 1 < last) fetch(sub + 1);
        __syncthreads();                                       // single-wave workgroup: a fence, no s_barrier

        const bool live = sub >= first;
        static_for<1, L + 1>([&](auto sc) {
            do_stage(sc, sub, live);
            __syncthreads();
        });

        // overflow bookkeeping (wave-uniform): any int16-stored output out of range so far?
        if constexpr (L >= 2) {
            const bool mine = (ovf_or >> 16) != 0;
            if constexpr (NW == 1) { if (!bad && __any(mine)) bad = true; }
            else { if (__syncthreads_or(mine ? 1 : 0)) bad = true; }
        }
        if (live && lane == 0 && ((sub + 1) % (DF_CHUNK / S) == 0 || sub + 1 == last))
            ovf_flags[sub / (DF_CHUNK / S)] = bad ? 1u : 0u;

        // carry: the last HD dwords of every array become the next sub-chunk's history.  Every stage's tail is read first
        // (the regions overlap when ND < HD), then everything is written: one LDS round trip for all stages, not one each.
        uint32_t keep[L][2];
        static_for<1, L + 1>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int HD = df_in16(s) ? HIST / 2 : HIST;
            constexpr int ND = df_in16(s) ? (S >> (s + 1)) : (S >> s);                // payload dwords per array
            constexpr int TOT = 4 * HD, PER = (TOT + NT - 1) / NT;
            static_assert(TOT % 64 == 0, "four arrays x 16 or 32 history dwords: a whole number of wave-wide accesses");
            const uint32_t* a = lds + DL::off(s, S);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int i = q * NT + lane;
                if (NW == 1 || i < TOT) keep[s - 1][q] = a[(i / HD) * DL::arr(s, S) + ND + (i % HD)];
            }
        });
        __syncthreads();
        static_for<1, L + 1>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int HD = df_in16(s) ? HIST / 2 : HIST;
            constexpr int TOT = 4 * HD, PER = (TOT + NT - 1) / NT;
            uint32_t* a = lds + DL::off(s, S);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int i = q * NT + lane;
                if (NW == 1 || i < TOT) a[(i / HD) * DL::arr(s, S) + (i % HD)] = keep[s - 1][q];
            }
        });
        __syncthreads();
    }
}

} // namespace sdrx
