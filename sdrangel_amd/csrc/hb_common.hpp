// Half-band stage device functions shared by the Decimators chain kernel and the
// DownChannelizer tree kernel.  gfx950 (CDNA4) only: wave64, v_dot2c_i32_i16, v_mad_i32_i24,
// v_perm_b32, LDS staged sliding windows.  These are the dot2 (vector-ALU) forms of the stage; since round 3 the stages
// whose input is int16 run by default on the matrix cores instead (hb_mfma.hpp: the odd-arm FIR as a banded-Toeplitz
// i8 MFMA contraction, bit-exact) and the functions below are the second gfx950 implementation (SDRX_*_ENGINE=valu).
//
// Math (reference: IntHalfbandFilterEO::doFIR,
// sdrbase/dsp/inthalfbandfiltereo.h:832-870).  For a stage of order N (P = N/4 coefficient
// pairs, S = hbShift = 12) with (rotated) input x[n], output k uses M = 2k+1:
//     acc  = sum_{i<P} c[i] (x[M-2i] + x[M-(N-2)+2i]) + (x[M-(N/2-1)] << 11);   y[k] = acc >> 11
// Split x into its polyphase arms  o[m] = x[2m+1] (odd)  and  e[m] = x[2m] (even):
//     y[k] = ( sum_{j<2P} h[j] o[k-j]  +  2048 e[k-(P-1)] ) >> 11,   h = {c[0..P-1], c[P-1..0]}
// i.e. a 2P-tap FIR on the odd arm plus one delayed even-arm sample.  Both arms live in LDS as
// planar arrays (oI, oQ, eI, eQ), each with a 32-entry history in front of the chunk, so that a
// lane computing R consecutive outputs reads ONE contiguous, 16-byte aligned window per array.
//
// Rotations of the inf/sup (lower/upper half) modes, x[n] = in[n] * (+-j)^(n+1)
// (inthalfbandfiltereo.h:626-692), never move data in the int32 flavour: on the odd arm they are
// the sign pattern (-1)^(m+1) (same for inf and sup), folded into the compile-time tap
// constants; on the even arm (centre tap only) they pick I or Q and a sign per output parity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace sdrx {

enum { MODE_CEN = 0, MODE_INF = 1, MODE_SUP = 2 };   // == SDRX_MODE_CENTER / LOWER / UPPER

constexpr int HB_SHIFT = 12;
constexpr int HIST = 32;           // history entries kept in front of every polyphase array

template<int ORDER> __host__ __device__ constexpr int hb_pairs() { return ORDER / 4; }

// HBFIRFilterTraits<64|48>::hbCoeffs  (hbfiltertraits.cpp:136-154, :85-99): (int32)(c * 4096)
template<int ORDER> __host__ __device__ constexpr int hb_c(int i)
{
    if (ORDER == 64) {
        constexpr int c[16] = { -1, 2, -5, 8, -12, 17, -25, 35, -47, 64, -86, 117, -164, 244, -424, 1300 };
        return c[i];
    } else {
        constexpr int c[12] = { -4, 7, -12, 19, -31, 48, -71, 103, -152, 236, -419, 1299 };
        return c[i];
    }
}
// impulse response of the odd arm, j in [0, 2P); 0 elsewhere
template<int ORDER> __host__ __device__ constexpr int hb_tap(int j)
{
    constexpr int P = hb_pairs<ORDER>();
    return (j < 0 || j >= 2 * P) ? 0 : (j < P ? hb_c<ORDER>(j) : hb_c<ORDER>(2 * P - 1 - j));
}
// sum |h| + 2048: worst-case (L1) gain numerator of one stage, /2048
template<int ORDER> __host__ __device__ constexpr long hb_l1()
{
    long s = 0;
    for (int j = 0; j < 2 * hb_pairs<ORDER>(); j++) { int t = hb_tap<ORDER>(j); s += t < 0 ? -t : t; }
    return s + 2048;
}

template<int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot2(uint32_t a, uint32_t coef, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, coef), acc, false);
}

// sext(int16 half of v) << 11 in ONE full-rate instruction (SDWA operand select + sign extension): the centre tap
// `x << (hbShift - 1)` of a stage whose even arm is packed int16.  Starting the accumulator with it replaces the
// `v_mov 0` + `v_dot2c (2048, 0)` pair that a dot2-only chain needs (v_dot2c accumulates in place).
template<int HALF>
__device__ __forceinline__ int centre_shl(uint32_t v)
{
    int r;
    if constexpr (HALF)
        asm("v_lshlrev_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "s"(HB_SHIFT - 1), "v"(v));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "s"(HB_SHIFT - 1), "v"(v));
    return r;
}

__host__ __device__ constexpr uint32_t pk16(int lo, int hi)
{
    return (uint32_t)(uint16_t)(int16_t)lo | ((uint32_t)(uint16_t)(int16_t)hi << 16);
}

// Packed tap pair for output r (0..7) and window dword d (0..19) of stage_pk16_r8.
// Window int16 i holds o[k0-32+i]; o[k-j] = w16[r+32-j]  ->  dword d: lo j = r+32-2d, hi j = r+31-2d.
// lo half = even odd-arm index m -> rotation sign -1, hi half -> +1 (inf and sup alike).
template<int ORDER, int MODE> __host__ __device__ constexpr uint32_t pk_coef(int r, int d)
{
    int lo = hb_tap<ORDER>(r + 32 - 2 * d), hi = hb_tap<ORDER>(r + 31 - 2 * d);
    if (MODE != MODE_CEN) lo = -lo;
    return pk16(lo, hi);
}

// ---- raw input quads: 4 consecutive complex samples as one vector load, de-interleaved into the four
// packed-int16 polyphase dwords  eI=(I0,I2)  eQ=(Q0,Q2)  oI=(I1,I3)  oQ=(Q1,Q3)
//   S16: the reference's Sample stream (int16 I,Q): 16 bytes per quad
//   U8 : DecimatorsU input (quint8 I,Q, value = byte - Shift; decimatorsu.h:218-230): 8 bytes per quad
template<bool U8> struct Quad;
template<> struct Quad<false> {
    typedef uint4 T;
    static __device__ __forceinline__ T zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ void split(const T v, int, uint32_t& eI, uint32_t& eQ, uint32_t& oI, uint32_t& oQ)
    {
        eI = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u); eQ = __builtin_amdgcn_perm(v.z, v.x, 0x07060302u);
        oI = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u); oQ = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u);
    }
};
template<> struct Quad<true> {
    typedef uint2 T;
    // a zero SAMPLE is the byte `shift`; padding beyond the input only feeds outputs that are never stored
    static __device__ __forceinline__ T zero() { return make_uint2(0, 0); }
    static __device__ __forceinline__ void split(const T v, int shift, uint32_t& eI, uint32_t& eQ, uint32_t& oI, uint32_t& oQ)
    {
        // bytes of v.x: I0 Q0 I1 Q1, of v.y: I2 Q2 I3 Q3; selector 0x0c yields a zero byte (zero extension)
        typedef short v2 __attribute__((ext_vector_type(2)));
        const v2 sh = { (short)shift, (short)shift };
        auto sub = [&](uint32_t x) { return __builtin_bit_cast(uint32_t, (v2)(__builtin_bit_cast(v2, x) - sh)); };
        eI = sub(__builtin_amdgcn_perm(v.y, v.x, 0x0c040c00u)); eQ = sub(__builtin_amdgcn_perm(v.y, v.x, 0x0c050c01u));
        oI = sub(__builtin_amdgcn_perm(v.y, v.x, 0x0c060c02u)); oQ = sub(__builtin_amdgcn_perm(v.y, v.x, 0x0c070c03u));
    }
};

// five centre-tap dwords p[EB .. EB+4] of a lane whose p is 16-byte aligned (p = arm + 4t): as wide aligned
// reads -- five separate ds_read_b32 with a lane stride of 16 bytes are 4-way bank conflicts each
template<int EB>
__device__ __forceinline__ void ld_centre5(const uint32_t* __restrict__ p, uint32_t (&v)[5])
{
    if constexpr (EB % 4 == 0) {
        const uint4 a = *reinterpret_cast<const uint4*>(p + EB);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = p[EB + 4];
    } else {
        static_assert(EB % 4 == 2, "centre window starts on an 8-byte boundary");
        const uint4 a = *reinterpret_cast<const uint4*>(p + EB - 2), b = *reinterpret_cast<const uint4*>(p + EB + 2);
        v[0] = a.z; v[1] = a.w; v[2] = b.x; v[3] = b.y; v[4] = b.z;
    }
}

// ---------------------------------------------------------------------------------------------
// Stage on PACKED int16 arms (two int16 per dword).  Used where the stage input is int16 by
// construction: stage 1 of every Decimators chain (raw samples; the `<< pre` of
// decimation_shifts is applied to the accumulator instead, which is the same thing modulo 2^32)
// and every DownChannelizer stage (Sample storage is int16).  8 outputs per lane, 16.5
// v_dot2c_i32_i16 per output and component instead of 16 adds + 16 multiplies.
//   oI,oQ,eI,eQ : LDS dword arrays, entry 0 = history[-32]; chunk-relative sample m at int16 index 32+m
//   t           : lane's output block, outputs k0 = 8t .. 8t+7 (chunk relative)
//   SHL         : left shift applied to the accumulator (decimation_shifts::preK)
// ---------------------------------------------------------------------------------------------
template<int ORDER, int MODE, int SHL>
__device__ __forceinline__ void stage_pk16_r8(const uint32_t* __restrict__ oI, const uint32_t* __restrict__ oQ,
                                              const uint32_t* __restrict__ eI, const uint32_t* __restrict__ eQ,
                                              int t, int (&yI)[8], int (&yQ)[8])
{
    constexpr int P = hb_pairs<ORDER>();
    constexpr int CD = P - 1;                    // centre tap: e[k - CD]
    uint32_t wI[20], wQ[20];
    {
        const uint4* pI = reinterpret_cast<const uint4*>(oI + 4 * t);
        const uint4* pQ = reinterpret_cast<const uint4*>(oQ + 4 * t);
#pragma unroll
        for (int q = 0; q < 5; q++) {
            uint4 a = pI[q], b = pQ[q];
            wI[4*q] = a.x; wI[4*q+1] = a.y; wI[4*q+2] = a.z; wI[4*q+3] = a.w;
            wQ[4*q] = b.x; wQ[4*q+1] = b.y; wQ[4*q+2] = b.z; wQ[4*q+3] = b.w;
        }
    }
    // even arm: e[k0+r-CD] sits at int16 index 32+k0+r-CD; 32-CD is odd, so the first one needed
    // is the HIGH half of dword EB and output r uses int16 offset r+1 of a 5-dword window.
    constexpr int EB = (32 - CD - 1) / 2;
    uint32_t vI[5], vQ[5];
    ld_centre5<EB>(eI + 4 * t, vI);
    ld_centre5<EB>(eQ + 4 * t, vQ);

    static_for<0, 8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        int aI = 0, aQ = 0;
        static_for<0, 20>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            constexpr uint32_t cf = pk_coef<ORDER, MODE>(r, d);
            if constexpr (cf != 0) { aI = dot2(wI[d], cf, aI); aQ = dot2(wQ[d], cf, aQ); }
        });
        constexpr int dd = (r + 1) >> 1, hf = (r + 1) & 1;
        constexpr uint32_t cp = hf ? pk16(0, 2048) : pk16(2048, 0);
        constexpr uint32_t cn = hf ? pk16(0, -2048) : pk16(-2048, 0);
        if constexpr (MODE == MODE_CEN) {
            aI = dot2(vI[dd], cp, aI); aQ = dot2(vQ[dd], cp, aQ);
        } else {
            // centre sample n = 2(k-CD'): inf: k odd -> (-im, re), k even -> (im, -re); sup: negated
            constexpr bool neg_first = ((r & 1) == 1) == (MODE == MODE_INF);
            aI = dot2(vQ[dd], neg_first ? cn : cp, aI);
            aQ = dot2(vI[dd], neg_first ? cp : cn, aQ);
        }
        yI[r] = (int)((uint32_t)aI << SHL) >> (HB_SHIFT - 1);
        yQ[r] = (int)((uint32_t)aQ << SHL) >> (HB_SHIFT - 1);
    });
}

// ---------------------------------------------------------------------------------------------
// Stage on int32 arms (one sample per dword) -- the exact, wrap-around int32 flavour of
// Decimators stages 2..6.  R outputs per lane.  M24: the static bound on |a-b| is < 2^23, so the
// full-rate v_mad_i32_i24 is exact; otherwise the 32-bit multiply.
//   arrays: entry 0 = history[-32]; chunk-relative sample m at index 32+m;  k0 = R*t
// ---------------------------------------------------------------------------------------------
template<int R> struct VecLoad;
template<> struct VecLoad<8> { typedef uint4 T; static constexpr int W = 4; };
template<> struct VecLoad<4> { typedef uint4 T; static constexpr int W = 4; };
template<> struct VecLoad<2> { typedef uint2 T; static constexpr int W = 2; };

template<int R, int N>
__device__ __forceinline__ void lds_window(const int* __restrict__ base, int (&w)[N])
{
    // base is aligned to R dwords (16 B for R >= 4, 8 B for R = 2); N is a multiple of the vector width
    typedef typename VecLoad<R>::T V;
    constexpr int W = VecLoad<R>::W;
    static_assert(N % W == 0, "window must be a whole number of vectors");
    const V* p = reinterpret_cast<const V*>(base);
#pragma unroll
    for (int q = 0; q < N / W; q++) {
        V v = p[q];
        if constexpr (W == 4) { w[4*q] = v.x; w[4*q+1] = v.y; w[4*q+2] = v.z; w[4*q+3] = v.w; }
        else { w[2*q] = v.x; w[2*q+1] = v.y; }
    }
}

template<bool M24>
__device__ __forceinline__ int mac(int acc, int d, int c)
{
    if constexpr (M24) return acc + __mul24(d, c);
    else return (int)((uint32_t)acc + (uint32_t)d * (uint32_t)c);
}

template<int ORDER, int MODE, int R, bool M24>
__device__ __forceinline__ void stage_i32(const int* __restrict__ oI, const int* __restrict__ oQ,
                                          const int* __restrict__ eI, const int* __restrict__ eQ,
                                          int t, int (&yI)[R], int (&yQ)[R])
{
    constexpr int P = hb_pairs<ORDER>();
    constexpr int TAPS = 2 * P, CD = P - 1;
    const int k0 = R * t;
    int wI[R + 32], wQ[R + 32];          // w[i] = o[k0-32+i]
    lds_window<R>(oI + k0, wI);
    lds_window<R>(oQ + k0, wQ);
    // e[k0+r-CD] at index 32+k0+r-CD; aligned window from index k0 + EA, EA = (32-CD) rounded down to R
    constexpr int EA = ((32 - CD) / R) * R, EO = (32 - CD) - EA;
    constexpr int EN = ((EO + R + VecLoad<R>::W - 1) / VecLoad<R>::W) * VecLoad<R>::W;
    int vI[EN], vQ[EN];
    lds_window<R>(eI + k0 + EA, vI);
    lds_window<R>(eQ + k0 + EA, vQ);

    static_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        int aI = 0, aQ = 0;
        static_for<0, P>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int ia = r + 32 - i, ib = r + 32 - (TAPS - 1) + i;      // o[k-i], o[k-(TAPS-1)+i]
            if constexpr (MODE == MODE_CEN) {
                aI = mac<M24>(aI, (int)((uint32_t)wI[ia] + (uint32_t)wI[ib]), hb_c<ORDER>(i));
                aQ = mac<M24>(aQ, (int)((uint32_t)wQ[ia] + (uint32_t)wQ[ib]), hb_c<ORDER>(i));
            } else {
                // s(m) = (-1)^(m+1) with m = k-i (k0 even): partner has the opposite sign
                constexpr int sg = (((r - i) & 1) == 0) ? -1 : 1;
                aI = mac<M24>(aI, (int)((uint32_t)wI[ia] - (uint32_t)wI[ib]), sg * hb_c<ORDER>(i));
                aQ = mac<M24>(aQ, (int)((uint32_t)wQ[ia] - (uint32_t)wQ[ib]), sg * hb_c<ORDER>(i));
            }
        });
        const uint32_t cI = (uint32_t)vI[EO + r] << (HB_SHIFT - 1);
        const uint32_t cQ = (uint32_t)vQ[EO + r] << (HB_SHIFT - 1);
        uint32_t uI, uQ;
        if constexpr (MODE == MODE_CEN) { uI = (uint32_t)aI + cI; uQ = (uint32_t)aQ + cQ; }
        else {
            constexpr bool neg_first = ((r & 1) == 1) == (MODE == MODE_INF);
            if constexpr (neg_first) { uI = (uint32_t)aI - cQ; uQ = (uint32_t)aQ + cI; }
            else                     { uI = (uint32_t)aI + cQ; uQ = (uint32_t)aQ - cI; }
        }
        yI[r] = (int)uI >> (HB_SHIFT - 1);
        yQ[r] = (int)uQ >> (HB_SHIFT - 1);
    });
}

// write R consecutive outputs (k0 = R*t) into the next stage's int32 arms
template<int R>
__device__ __forceinline__ void put_i32(int* __restrict__ oI, int* __restrict__ oQ,
                                        int* __restrict__ eI, int* __restrict__ eQ,
                                        int t, const int (&yI)[R], const int (&yQ)[R])
{
    const int p = HIST + (R / 2) * t;
    if constexpr (R == 8) {
        *reinterpret_cast<int4*>(eI + p) = make_int4(yI[0], yI[2], yI[4], yI[6]);
        *reinterpret_cast<int4*>(oI + p) = make_int4(yI[1], yI[3], yI[5], yI[7]);
        *reinterpret_cast<int4*>(eQ + p) = make_int4(yQ[0], yQ[2], yQ[4], yQ[6]);
        *reinterpret_cast<int4*>(oQ + p) = make_int4(yQ[1], yQ[3], yQ[5], yQ[7]);
    } else if constexpr (R == 4) {
        *reinterpret_cast<int2*>(eI + p) = make_int2(yI[0], yI[2]);
        *reinterpret_cast<int2*>(oI + p) = make_int2(yI[1], yI[3]);
        *reinterpret_cast<int2*>(eQ + p) = make_int2(yQ[0], yQ[2]);
        *reinterpret_cast<int2*>(oQ + p) = make_int2(yQ[1], yQ[3]);
    } else {
        eI[p] = yI[0]; oI[p] = yI[1]; eQ[p] = yQ[0]; oQ[p] = yQ[1];
    }
}

__device__ __forceinline__ uint32_t pack_iq(int re, int im)
{
    // Sample::setReal((FixReal) v): keep the low 16 bits of each
    return __builtin_amdgcn_perm((uint32_t)im, (uint32_t)re, 0x05040100u);
}

} // namespace sdrx
