// DownChannelizer bank as a tree of half-band stages -- ONE generic gfx950 kernel, launched once
// per "pass" (reference: DownChannelizer::feed, sdrbase/dsp/downchannelizer.cpp:50-91, and
// IntHalfbandFilterEO<qint32,qint32,48>::workDecimate{Center,LowerHalf,UpperHalf}(Sample*),
// inthalfbandfiltereo.h:37-63,158-206,357-405 with doFIR(Sample*) :792-830).
//
// The N channels' stage strings form a trie; every distinct prefix (= node) is ONE order-48
// half-band stage evaluated ONCE.  The host planner (sdrx_chan.hip) cuts the trie into passes of
// at most 4 levels by default (TK_MAX_LEVELS is the structural limit): a pass reads a stream (the raw
// input, or a node stream a previous pass wrote to global memory), walks it in chunks of 4096 samples
// with all stage histories carried in LDS, and writes channel outputs and/or deeper node streams.
// `warm` warm-up chunks (warm * 4096 >= 46*(2^levels - 1): one chunk up to 6 levels) in front of every
// time segment make the carried LDS state exact.
//
// Everything here is int16 by construction (Sample storage), so every stage runs on packed int16
// polyphase arms with v_dot2c_i32_i16 (stage_pk16_r8).  The int16-wrapping negation of the
// lower/upper-half rotations ((FixReal) -sample->imag(), :164) matters on the odd arm
// (-(-32768) stays -32768): producers therefore store an extra, explicitly wrap-negated
// "alternating sign" copy of the odd arm for L/U children.  On the even arm (centre tap only) the
// sign is folded into the tap: a wrap there changes acc by 2^27, i.e. y by exactly 2^16, which
// the int16 store discards.
#pragma once
#include "hb_common.hpp"
#include "hb_mfma.hpp"

namespace sdrx {

constexpr int TK_CHUNK = 4096;
constexpr int TK_THREADS = 256;
constexpr int TK_MAX_LEVELS = 10;
constexpr int TK_DEFAULT_LEVELS = 4;         // what the planner uses unless told otherwise: measured best for 32, 128 and 256 channels
                                             // (profiles/r02_tree_plan_sweep.txt); up to 6 levels need one warm-up chunk
constexpr int TK_HIST = 2 * TK_CHUNK;        // samples of stream history kept between feeds: (warm + 1) chunks, this for warm = 1

// One table entry = one half-band stage, or a FUSED lower/upper sibling pair: the lower- and the upper-half
// child of a node rotate the odd arm identically (j^(n+1) = (-j)^(n+1) for odd n) and only differ in the sign
// of the centre tap, so 24 of their 25 taps are one shared sum.  The pair costs 12.5 + 2 dot2 per output and
// component instead of 2 x 13.5, and reads the parent's window once.
struct TkOut {                  // 12 dwords: where one stage's outputs go + its centre taps
    int outE_I, outE_Q;         // own output arms (-1: none): even
    int outO_I, outO_Q;         //   odd, plain (for a centre child)
    int outA_I, outA_Q;         //   odd, alternating wrap-negated (for lower/upper children)
    int sink;                   // head of this stage's sink list (index into the sink table), -1: none
    int present;                // 0: this half of the entry is unused
    uint32_t cIe, cIo, cQe, cQo;// packed centre taps for even / odd output index
};
constexpr int TK_NODE_DW = 32;
struct TkNode {                 // 32 dwords
    int oddI, oddQ;             // LDS dword offsets: odd arm it reads (parent's plain or alt copy)
    int cenI, cenQ;             // even-arm arrays feeding the I / Q accumulators (swapped for L/U)
    TkOut a;                    // the stage itself (the LOWER child when fused)
    TkOut b;                    // the UPPER sibling when fused (present = 1)
    int mode_a;                 // SDRX_MODE_* of `a` (the MFMA path derives the centre-tap signs from it)
    int pad[3];
};
static_assert(sizeof(TkOut) == 48 && sizeof(TkNode) == TK_NODE_DW * 4, "node table layout");

struct TkLevel {
    int node_base, n_nodes, jobs_log2, nout;     // nout = outputs per node per chunk; jobs per node = nout >> r_log2
    int arr_base, arr_cnt;                       // arrays PRODUCED by this level's stages (relative to the subtree's array list)
    int r_log2;                                  // outputs per job: 8, 4 or 2
    int in_len;                                  // dwords of every array this level READS (its parents' arms)
    int mfma;                                    // 1: the level runs on the matrix cores (hb_mfma.hpp): nout >= 256, whole jobs per entry
    int mjob_base, n_mjobs;                      // its jobs in the group's TkMJob table (a job = 16 blocks of 16 outputs of one entry, I and Q = two MFMA tiles)
    uint32_t xm;                                 // XORed into the odd-arm dwords this level PRODUCES: HBM_BIAS2 if the next level is an MFMA level
    // The arrays of one level are allocated back to back with one length (sdrx_chan.hip), and so are their history slots (16 dwords each,
    // in array order): the history walk computes its addresses from these five numbers instead of reading a per-array table from LDS
    // (a dependent LDS round trip in front of every copy).  Everything a level needs sits in this one 64-byte record = one scalar load.
    int prev_off, prev_arr_cnt;                  // the arrays this level READS (its parents' arms; level 1: the root arms): first window, count;
                                                 // their length is in_len, their slots end where this level's begin
    int arr_off, arr_len;                        // the arrays this level PRODUCES: first window (LDS dword offset), length of each
};
static_assert(sizeof(TkLevel) == 64, "one s_load_dwordx16 per level");

// One matrix-core job, everything resolved by the planner to LDS BYTE addresses of the job's first element (the lane adds its
// share): wave-uniform, fetched with three wide scalar loads.  o[0] / o[1]: a centre stage uses o[0]; a lower/upper pair has the
// lower child in o[0] and the upper one in o[1] (either may be absent: flags = 0, sink = -1).  Absent arm arrays of a present
// child point at a scratch slot, so the epilogue has no branch per array.
struct TkMOut { int E_I, E_Q, O_I, O_Q, A_I, A_Q; int sink; int flags; };     // flags: 1 = even arms, 2 = plain odd arms, 4 = alternating odd arms
struct TkMJob {
    int bI, bQ;                 // window entry 0 of block 16 tb of the odd arm feeding I / Q
    int cI, cQ;                 // dword holding even-arm entry 16 (16 tb) + 20 (centre taps of the job's first block)
    int mode;                   // 0: centre stage, else lower/upper pair
    int out0;                   // first output of the job inside the chunk: 256 tb
    int fast, kinds;            // fast = 1: a lower/upper pair whose two children are inner nodes with even arms + ONE odd-arm kind (its address in
                                // O_I / O_Q) and no sink: stores-only epilogue; kinds bit 0 / 1: child 0 / 1 wants the alternating-sign copy
    TkMOut o[2];
    int pad2[8];
};
static_assert(sizeof(TkMJob) == 128, "job table layout");

struct TkSubtree {
    int n_levels;
    int warm;                   // warm-up chunks in front of a segment
    int n_nodes;                // all levels
    int node_base;              // first node (global index) -- levels index relative to the table
    int n_arrays, array_base;   // all arrays: [root arrays][level-1 arrays][level-2 arrays]...
    int root_arr_cnt;           // the first root_arr_cnt arrays are the root arms
    int lds_dwords;             // two arm regions + history store + node table copy (+ 256 B scratch for the MFMA jobs' absent arms)
    int sink_base, n_sinks;     // this subtree's sinks are one contiguous run of the group's sink table
    int sink_tab;               // LDS dword offset of the copy of that run (TK_SINK_DW dwords each)
    int node_tab;               // LDS dword offset of the node table copy
    int store_base;             // LDS dword offset of the history slots: 16 dwords per array, in array order (root arms first)
    int root_off, root_len;     // the root arms: first window, length of each
    int rootE_I, rootE_Q, rootO_I, rootO_Q, rootA_I, rootA_Q;   // root arms (-1: none)
    uint32_t root_xm;           // XORed into the root odd arms (HBM_BIAS2 if level 1 is an MFMA level)
    int dbg;                    // timing experiments only (SDRX_CHAN_DBG, results are WRONG when set): 1 skip MFMA jobs, 2 skip the
                                // history walks, 4 skip the root fill (a bit 16, skip the MFMA epilogues, gave DESIGN 4.3a its split and was
                                // removed: its branch sat between the MFMAs and the epilogues of every pair)
    TkLevel lv[TK_MAX_LEVELS];
};

// One polyphase array.  Its window [off, off+len) = 16 dwords of history + the chunk's payload lives in one of
// two LDS regions that alternate by tree level (level l's arrays are dead once level l+1 has consumed them, so
// level l+2 reuses the space: 61 KB -> 40 KB for the cfg-3 raw pass, 2 -> 4 workgroups per CU); the 16-dword
// history survives in a persistent slot `store`: saved when the consumer level is done, restored in front of
// the window before the producer writes the next chunk.
struct TkArray { int off, len, store, bias; };     // bias = 1: an odd arm an MFMA level reads: its zero history is HBM_BIAS2

struct TkStream {               // per feed, per input stream of a pass
    const uint32_t* hist;       // hist_len samples: absolute positions [t_old - hist_len, t_old)
    const uint32_t* in;         // new samples: absolute positions [t_old, t_new)
    long t_old, t_new;
    long c_first, c_last;       // absolute chunk range to (re)compute
    int cps;                    // chunks per segment
    int subtree;
    long hist_len;              // (warm + 1) * TK_CHUNK
};

constexpr int TK_SINK_DW = 8;
struct TkSink {                 // where a node's outputs go in global memory
    uint32_t* ptr0;             // element for absolute output index 0 (= buffer - base: never dereferenced outside [lo, hi))
    long lo, hi;                // store only absolute output indices in [lo, hi)
    int shift;                  // 0: raw node stream; n > 0: channel end, value / 2^n (toward zero)
    int next;                   // next sink of the same node, -1: end of list
};

static_assert(sizeof(TkSink) == TK_SINK_DW * 4, "sink table layout");

__device__ __forceinline__ int div_pow2_trunc(int v, int n)
{
    // s.m_real /= (1 << n)  (downchannelizer.cpp:80): C division of the promoted int16
    return (v + ((v >> 31) & ((1 << n) - 1))) >> n;
}

// MX = true: levels flagged `mfma` by the planner run on the matrix cores (the default engine); MX = false is the
// all-VALU kernel of rounds 1-2 (SDRX_CHAN_ENGINE=valu), kept as the second opinion the tests compare with.
template<bool MX>
__global__ __launch_bounds__(TK_THREADS, 4)
void tree_kernel(const TkSubtree* __restrict__ subtrees, const TkNode* __restrict__ nodes,
                 const TkArray* __restrict__ arrays, const TkStream* __restrict__ streams,
                 const TkSink* __restrict__ sinks, const TkMJob* __restrict__ mjobs)
{
    constexpr int C = TK_CHUNK, NT = TK_THREADS, LPT = C / 4 / NT;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const TkStream sp = streams[blockIdx.y];
    const long first = sp.c_first + (long)blockIdx.x * sp.cps;
    if (first > sp.c_last) return;
    long last = first + sp.cps - 1; if (last > sp.c_last) last = sp.c_last;
    // (The subtree descriptor stays in global/constant memory: the compiler keeps its fields in SGPRs; an LDS copy of it
    // turned every `st.` access into an LDS load and tripled the kernel time.)
    const TkSubtree& st = subtrees[sp.subtree];
    const int tid = threadIdx.x;

    for (int i = tid; i < st.lds_dwords; i += NT) lds[i] = 0;
    __syncthreads();
    {   // node table -> LDS
        const uint32_t* src = reinterpret_cast<const uint32_t*>(nodes + st.node_base);
        for (int i = tid; i < st.n_nodes * TK_NODE_DW; i += NT) lds[st.node_tab + i] = src[i];
        // this feed's sink descriptors (pointers, ranges): read by every job of the sink levels -- from LDS, not as a
        // dependent global load in front of the stores
        const uint32_t* sk = reinterpret_cast<const uint32_t*>(sinks + st.sink_base);
        for (int i = tid; i < st.n_sinks * TK_SINK_DW; i += NT) lds[st.sink_tab + i] = sk[i];
        if constexpr (MX) {
            // odd arms an MFMA level reads carry 0x0080 in every int16 (hb_mfma.hpp): so does their zero history
            for (int i = tid; i < st.n_arrays * 16; i += NT) {
                const TkArray a = arrays[st.array_base + (i >> 4)];
                if (a.bias) lds[a.store + (i & 15)] = HBM_BIAS2;
            }
        }
    }
    // matrix-core operands: the order-48 taps as a banded Toeplitz block, built once per wave
    typedef HbMfmaTaps<48, false> Taps;
    Taps taps;
    const int lane = tid & 63, n16 = lane & 15, g4 = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (MX) taps.init(lane);
    const uint32_t root_xm = MX ? st.root_xm : 0u;
    // timing by elimination (TkSubtree::dbg) is compiled in only by `make EXTRA=-DSDRX_TK_DBG=1` (tools/dbgsweep_chan.sh does that on the GPU
    // box): even never-taken, its branches split the basic blocks of the hot loops (the one between the MFMAs and the epilogues cost 5 %)
#ifndef SDRX_TK_DBG
#define SDRX_TK_DBG 0
#endif
    const int dbg = SDRX_TK_DBG ? st.dbg : 0;
    // loop bounds and table offsets the level loop needs at every turn: pinned in registers (opaque to the compiler, which otherwise
    // re-loads them from the descriptor inside the loops -- a scalar-cache round trip in front of every level and every history walk)
    int n_levels = st.n_levels, store_base = st.store_base, root_off = st.root_off, root_len = st.root_len;
    asm volatile("" : "+s"(n_levels), "+s"(store_base), "+s"(root_off), "+s"(root_len));
    // the same for the root fill at the top of every chunk (eight dependent scalar loads before the first LDS write otherwise)
    int rE_I = st.rootE_I, rE_Q = st.rootE_Q, rO_I = st.rootO_I, rO_Q = st.rootO_Q, rA_I = st.rootA_I, rA_Q = st.rootA_Q, root_cnt16 = st.root_arr_cnt * 16;
    asm volatile("" : "+s"(rE_I), "+s"(rE_Q), "+s"(rO_I), "+s"(rO_Q), "+s"(rA_I), "+s"(rA_Q), "+s"(root_cnt16));

    uint4 pre[LPT];
    auto fetch = [&](long chunk) {
        const long c0 = chunk * C;
        if (c0 >= sp.t_old && c0 + C <= sp.t_new) {               // the whole chunk lies in this feed (wave-uniform): no per-lane tests
            // GLOBAL loads, not flat ones (the pointer comes out of a table, so the compiler cannot tell): a flat load also counts on
            // lgkmcnt, and then the first LDS wait of the level loop waits for the whole prefetch
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            typedef const u4v __attribute__((address_space(1))) gq4;
            gq4* src = (gq4*)reinterpret_cast<const u4v*>(sp.in + (c0 - sp.t_old));
#pragma unroll
            for (int j = 0; j < LPT; j++) { const u4v v = src[j * NT + tid]; pre[j] = make_uint4(v[0], v[1], v[2], v[3]); }
            return;
        }
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            const long p0 = chunk * C + 4 * (j * NT + tid);       // absolute position of the 4 samples
            if (p0 >= sp.t_old && p0 + 4 <= sp.t_new) {
                pre[j] = *reinterpret_cast<const uint4*>(sp.in + (p0 - sp.t_old));
            } else {
                uint32_t v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const long p = p0 + e;
                    v[e] = p < sp.t_old ? sp.hist[p - (sp.t_old - sp.hist_len)] : (p < sp.t_new ? sp.in[p - sp.t_old] : 0u);
                }
                pre[j] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
        // (boundary chunks only) nothing of this path stays in flight: loads pending in its scratch registers make the compiler drain
        // vmcnt in front of the level loop on EVERY path, which exposes the whole prefetch of the fast path above
        __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0) only
    };
    fetch(first - st.warm);
    __syncthreads();

    for (long chunk = first - st.warm; chunk <= last; ++chunk) {
        // ---- stream samples -> root arms
#pragma unroll
        for (int j = 0; j < LPT; j++) {
            if (dbg & 4) break;
            const int q = HIST / 2 + j * NT + tid;
            const uint4 v = pre[j];
            const uint32_t oI = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u);
            const uint32_t oQ = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u);
            lds[rE_I + q] = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u);
            lds[rE_Q + q] = __builtin_amdgcn_perm(v.z, v.x, 0x07060302u);
            if (rO_I >= 0) { lds[rO_I + q] = oI ^ root_xm; lds[rO_Q + q] = oQ ^ root_xm; }
            if (rA_I >= 0) {
                // odd-arm index m even (low half) -> wrap-negated, m odd -> as is
                typedef unsigned short us2r __attribute__((ext_vector_type(2)));
                const us2r sg = { 0xffffu, 1u };                                   // one packed multiply by (-1, +1): the low half wraps
                lds[rA_I + q] = __builtin_bit_cast(uint32_t, (us2r)(__builtin_bit_cast(us2r, oI) * sg)) ^ root_xm;
                lds[rA_Q + q] = __builtin_bit_cast(uint32_t, (us2r)(__builtin_bit_cast(us2r, oQ) * sg)) ^ root_xm;
            }
        }
        for (int i = tid; i < root_cnt16; i += NT)                             // history in front of the root windows
            lds[root_off + (i >> 4) * root_len + (i & 15)] = lds[store_base + i];
        if (chunk < last) fetch(chunk + 1);
        __syncthreads();

        const bool live = chunk >= first;
        for (int l = 0; l < n_levels; l++) {
            // the level record: one s_load_dwordx16 (field by field the compiler fetches it in three dependent pieces)
            typedef int s16i __attribute__((ext_vector_type(16)));
            const s16i rec = *(const s16i __attribute__((address_space(4)))*)reinterpret_cast<const int*>(&st.lv[l]);
            TkLevel lv;
            lv.node_base = rec[0]; lv.n_nodes = rec[1]; lv.jobs_log2 = rec[2]; lv.nout = rec[3]; lv.arr_base = rec[4]; lv.arr_cnt = rec[5];
            lv.r_log2 = rec[6]; lv.in_len = rec[7]; lv.mfma = rec[8]; lv.mjob_base = rec[9]; lv.n_mjobs = rec[10]; lv.xm = (uint32_t)rec[11];
            lv.prev_off = rec[12]; lv.prev_arr_cnt = rec[13]; lv.arr_off = rec[14]; lv.arr_len = rec[15];
            const int njobs = lv.n_nodes << lv.jobs_log2;
            {   // ONE pass over the level's arrays, before the jobs: (1) history of the arrays this level PRODUCES goes in
                // front of their windows, (2) the arrays this level READS are complete and only read from here on, so their
                // last 16 dwords are kept for the next chunk now (this used to be a second dependent LDS round trip behind
                // the jobs of every level).
                const int n_restore = lv.arr_cnt * 16, n_all = (dbg & 2) ? 0 : n_restore + lv.prev_arr_cnt * 16;
                const int slot0 = store_base + 16 * lv.arr_base;                  // this level's slots; its parents' end right there
                for (int i = tid; i < n_all; i += NT) {
                    const int k = i - n_restore;                                  // i < n_restore: slot -> window head; else: window tail -> slot
                    const int src = k < 0 ? slot0 + i : lv.prev_off + (k >> 4) * lv.in_len + lv.in_len - 16 + (k & 15);
                    const int dst = k < 0 ? lv.arr_off + (i >> 4) * lv.arr_len + (i & 15) : slot0 - lv.prev_arr_cnt * 16 + k;
                    lds[dst] = lds[src];
                }
            }
            if (MX && lv.mfma) {
                // ---- the level on the matrix cores (hb_mfma.hpp).  A job = 16 blocks of 16 consecutive outputs of ONE table entry,
                // I and Q: two tiles, column n = block 16 tb + n; lane (n, g) ends up with outputs 4g .. 4g+3 of its block for both
                // components = one packed dword for the even and one for the odd arm of the children.  Each wave takes a contiguous
                // run of the level's jobs, two at a time: both jobs' loads, then 20 MFMAs, then the epilogues -- one long basic
                // block (the job descriptors are wave-uniform scalars; absent arm arrays point at a scratch slot).
                typedef int s8i __attribute__((ext_vector_type(8)));
                typedef const s8i __attribute__((address_space(4))) cs8;
                const int per = (lv.n_mjobs + NT / 64 - 1) / (NT / 64);
                const int t0 = wv * per, t1 = t0 + per < lv.n_mjobs ? t0 + per : lv.n_mjobs;
                const v4i bias = { Taps::BIAS, Taps::BIAS, Taps::BIAS, Taps::BIAS };
                const uint32_t xm = lv.xm;
                const char* ldsb = reinterpret_cast<const char*>(lds);
                char* ldsw = reinterpret_cast<char*>(lds);
                const int wl = 32 * n16 + 16 * g4, cl = 32 * n16 + 8 * g4, pl = 16 * n16 + 4 * g4;      // the lane's byte offsets
                struct JobIn { v4i bI0, bI1, bQ0, bQ1; uint2 cI01, cQ01; uint32_t cI2, cQ2; };
                auto load = [&](const s8i h, JobIn& r) {
                    r.bI0 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(ldsb + h[0] + wl, 16));
                    r.bI1 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(ldsb + h[0] + wl + 64, 16));
                    r.bQ0 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(ldsb + h[1] + wl, 16));
                    r.bQ1 = *reinterpret_cast<const v4i*>(__builtin_assume_aligned(ldsb + h[1] + wl + 64, 16));
                    // centre taps e[k - 11], k = 16 blk + 4 g + i: int16 entries 16 blk + 4 g + 21 + i of the even arm that feeds the
                    // I accumulator (the parent's eI, or its eQ below a lower/upper stage) and of the one feeding Q
                    r.cI01 = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(ldsb + h[2] + cl, 8));
                    r.cI2 = *reinterpret_cast<const uint32_t*>(ldsb + h[2] + cl + 8);
                    r.cQ01 = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(ldsb + h[3] + cl, 8));
                    r.cQ2 = *reinterpret_cast<const uint32_t*>(ldsb + h[3] + cl + 8);
                };
                // one stage's outputs, PACKED: eI = (y0, y2), oI = (y1, y3) of the I component, eQ / oQ of Q (int16 halves, wrapped:
                // Sample storage, inthalfbandfiltereo.h:828-829): arms of the children (LDS), node streams / channel ends (global memory)
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                auto padd = [](uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b))); };
                auto psub = [](uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b))); };
                auto emit = [&](const uint32_t eI, const uint32_t oI, const uint32_t eQ, const uint32_t oQ, const s8i o, const long abs0) {
                    if (o[7]) {
                        *reinterpret_cast<uint32_t*>(ldsw + o[0] + pl) = eI;
                        *reinterpret_cast<uint32_t*>(ldsw + o[1] + pl) = eQ;
                        if (o[7] & 2) {
                            *reinterpret_cast<uint32_t*>(ldsw + o[2] + pl) = oI ^ xm;
                            *reinterpret_cast<uint32_t*>(ldsw + o[3] + pl) = oQ ^ xm;
                        }
                        if (o[7] & 4) {
                            // alternating-sign copy for lower/upper children: entry 8 blk + 2 g is even -> wrap-negated (one packed
                            // multiply by (-1, +1): the low half wraps like (FixReal) -x)
                            const us2 sg = { 0xffffu, 1u };
                            *reinterpret_cast<uint32_t*>(ldsw + o[4] + pl) = __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, oI) * sg)) ^ xm;
                            *reinterpret_cast<uint32_t*>(ldsw + o[5] + pl) = __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, oQ) * sg)) ^ xm;
                        }
                    }
                    if (live) {
                        for (int si = o[6]; si >= 0; ) {
                            const s8i sr = *(cs8*)reinterpret_cast<const int*>(sinks + si);
                            const long ptr0 = ((long)sr[1] << 32) | (uint32_t)sr[0];
                            const long lo = ((long)sr[3] << 32) | (uint32_t)sr[2], hi = ((long)sr[5] << 32) | (uint32_t)sr[4];
                            const int shift = sr[6];
                            uint32_t w[4];
                            if (shift) {
                                w[0] = pack_iq(div_pow2_trunc((int)(int16_t)eI, shift), div_pow2_trunc((int)(int16_t)eQ, shift));
                                w[1] = pack_iq(div_pow2_trunc((int)(int16_t)oI, shift), div_pow2_trunc((int)(int16_t)oQ, shift));
                                w[2] = pack_iq(div_pow2_trunc((int)eI >> 16, shift), div_pow2_trunc((int)eQ >> 16, shift));
                                w[3] = pack_iq(div_pow2_trunc((int)oI >> 16, shift), div_pow2_trunc((int)oQ >> 16, shift));
                            } else {
                                w[0] = __builtin_amdgcn_perm(eQ, eI, 0x05040100u); w[1] = __builtin_amdgcn_perm(oQ, oI, 0x05040100u);
                                w[2] = __builtin_amdgcn_perm(eQ, eI, 0x07060302u); w[3] = __builtin_amdgcn_perm(oQ, oI, 0x07060302u);
                            }
                            typedef uint32_t __attribute__((address_space(1))) gu32;
                            gu32* dst = (gu32*)(reinterpret_cast<uint32_t*>(ptr0) + abs0);
                            const long rel = abs0 - lo, span = hi - lo;
                            if (rel >= 0 && rel + 4 <= span) {
                                // ONE 16-byte store (dword alignment is all a global store needs; four assignments came out as dwordx3 + dword): 2.5-3.5 %
                                typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
                                *(u4a __attribute__((address_space(1)))*)dst = u4a{ w[0], w[1], w[2], w[3] };
                            }
                            else if (rel > -4 && rel < span) {
#pragma unroll
                                for (int i = 0; i < 4; i++) if (rel + i >= 0 && rel + i < span) dst[i] = w[i];
                            }
                            si = sr[7];
                        }
                    }
                };
                auto finish = [&](const JobIn& r, const v4i SI, const v4i SQ, const s8i h, const s8i oa, const s8i ob) {
                    // (S +- (e << 11)) >> 11 == (S >> 11) +- e (|S| < 2^28 for any int16 data), and the int16 store keeps the sum
                    // modulo 2^16: pack the shifted sums and the centre taps first, then add / subtract two outputs per instruction
                    // (lo, hi) int16 halves = (a >> 11, b >> 11): the second shift writes its low half straight into the high half of
                    // the first's result (SDWA, one instruction instead of a shift and a v_perm; a and b are VALU results, no MFMA hazard)
                    auto shpack = [](int a, int b) {
                        uint32_t r = (uint32_t)(a >> (HB_SHIFT - 1));
                        asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "s"(HB_SHIFT - 1), "v"(b));
                        return r;
                    };
                    const uint32_t sI02 = shpack(SI[0], SI[2]), sI13 = shpack(SI[1], SI[3]);
                    const uint32_t sQ02 = shpack(SQ[0], SQ[2]), sQ13 = shpack(SQ[1], SQ[3]);
                    // centre taps e0..e3 = int16 entries 1, 2, 3, 4 of the three dwords read: (e0, e2) and (e1, e3)
                    const uint32_t cI02 = __builtin_amdgcn_perm(r.cI01.y, r.cI01.x, 0x07060302u), cI13 = __builtin_amdgcn_perm(r.cI2, r.cI01.y, 0x05040100u);
                    const uint32_t cQ02 = __builtin_amdgcn_perm(r.cQ01.y, r.cQ01.x, 0x07060302u), cQ13 = __builtin_amdgcn_perm(r.cQ2, r.cQ01.y, 0x05040100u);
                    if (h[6]) {
                        // the common inner job (planner: TkMJob::fast): eight stores, no branch -- the generic path below spends a dozen
                        // scalar branches per job on presence flags and sink lists, each a bubble in a wave that has little else in flight
                        typedef unsigned short us2f __attribute__((ext_vector_type(2)));
                        auto alt = [&](uint32_t v, uint32_t m) { return __builtin_bit_cast(uint32_t, (us2f)(__builtin_bit_cast(us2f, v) * __builtin_bit_cast(us2f, m))) ^ xm; };
                        const uint32_t ma = (h[7] & 1) ? 0x0001ffffu : 0x00010001u, mb = (h[7] & 2) ? 0x0001ffffu : 0x00010001u;   // (-1, +1) or (+1, +1)
                        auto st4 = [&](const s8i o, uint32_t eI, uint32_t eQ, uint32_t oI, uint32_t oQ, uint32_t m) {
                            *reinterpret_cast<uint32_t*>(ldsw + o[0] + pl) = eI;
                            *reinterpret_cast<uint32_t*>(ldsw + o[1] + pl) = eQ;
                            *reinterpret_cast<uint32_t*>(ldsw + o[2] + pl) = alt(oI, m);
                            *reinterpret_cast<uint32_t*>(ldsw + o[3] + pl) = alt(oQ, m);
                        };
                        st4(oa, padd(sI02, cI02), psub(sQ02, cQ02), psub(sI13, cI13), padd(sQ13, cQ13), ma);
                        st4(ob, psub(sI02, cI02), padd(sQ02, cQ02), padd(sI13, cI13), psub(sQ13, cQ13), mb);
                        return;
                    }
                    const long abs0 = chunk * lv.nout + h[5] + 16 * n16 + 4 * g4;
                    if (h[4] == 0) {
                        emit(padd(sI02, cI02), padd(sI13, cI13), padd(sQ02, cQ02), padd(sQ13, cQ13), oa, abs0);
                    } else {
                        // lower half: k even -> (+im, -re), k odd -> (-im, +re); upper half: the negation (inthalfbandfiltereo.h:158-206,
                        // 357-405; cI / cQ already come from the OTHER component's even arm)
                        if (oa[7] | (oa[6] >= 0)) emit(padd(sI02, cI02), psub(sI13, cI13), psub(sQ02, cQ02), padd(sQ13, cQ13), oa, abs0);
                        if (ob[7] | (ob[6] >= 0)) emit(psub(sI02, cI02), padd(sI13, cI13), padd(sQ02, cQ02), psub(sQ13, cQ13), ob, abs0);
                    }
                };
                int tt = (dbg & 1) ? t1 : t0;
                for (; tt + 1 < t1; tt += 2) {
                    cs8* d0 = (cs8*)reinterpret_cast<const int*>(mjobs + lv.mjob_base + tt);
                    cs8* d1 = (cs8*)reinterpret_cast<const int*>(mjobs + lv.mjob_base + tt + 1);
                    const s8i h0 = d0[0], a0 = d0[1], b0 = d0[2], h1 = d1[0], a1 = d1[1], b1 = d1[2];
                    JobIn r0, r1;
                    __builtin_amdgcn_s_setprio(2);
                    load(h0, r0); load(h1, r1);
                    __builtin_amdgcn_sched_barrier(0);                             // every LDS read of the pair is in flight before the first MFMA
                    const v4i SI0 = taps.tile(r0.bI0, r0.bI1, bias), SQ0 = taps.tile(r0.bQ0, r0.bQ1, bias);
                    const v4i SI1 = taps.tile(r1.bI0, r1.bI1, bias), SQ1 = taps.tile(r1.bQ0, r1.bQ1, bias);
                    // (the wave asks for issue priority while it puts its operand loads and MFMAs out: the matrix pipe then works under the
                    // other waves' epilogues -- 1.3 % at 1 Gi samples, A/B on one box with tools/ab_libs.sh)
                    __builtin_amdgcn_s_setprio(0);
                    finish(r0, SI0, SQ0, h0, a0, b0);
                    finish(r1, SI1, SQ1, h1, a1, b1);
                }
                if (tt < t1) {                                                     // odd job count: the last one on its own
                    cs8* d0 = (cs8*)reinterpret_cast<const int*>(mjobs + lv.mjob_base + tt);
                    const s8i h0 = d0[0], a0 = d0[1], b0 = d0[2];
                    JobIn r0;
                    load(h0, r0);
                    const v4i SI0 = taps.tile(r0.bI0, r0.bI1, bias), SQ0 = taps.tile(r0.bQ0, r0.bQ1, bias);
                    finish(r0, SI0, SQ0, h0, a0, b0);
                }
                __syncthreads();
                continue;
            }
            // One job = (table entry, R consecutive outputs).  R = 8 while that still gives every lane a job; narrow levels
            // (few entries, short chunks: the bottom of every subtree) drop to R = 4 or 2 so that the lanes stay busy --
            // a level costs one job time, and a job's time is proportional to R.
            auto job = [&](auto Rc, const int j) {
                constexpr int R = decltype(Rc)::value;
                constexpr int NW = (R + 32) / 2;                                   // window dwords: int16 i holds o[k0 - 32 + i]
                constexpr int NE = R / 2 + 1;
                const int ni = lv.node_base + (j >> lv.jobs_log2);
                const int t = j & ((1 << lv.jobs_log2) - 1);
                const uint4* nt = reinterpret_cast<const uint4*>(lds + st.node_tab + ni * TK_NODE_DW);
                const uint4 n0 = nt[0];
                // --- shared part: R outputs x (I,Q) of the 24-tap odd-arm sum, packed int16 arms
                int sI[R], sQ[R];
                uint32_t vI[NE], vQ[NE];
                {
                    const uint32_t* oI = lds + (int)n0.x + (R / 2) * t, *oQ = lds + (int)n0.y + (R / 2) * t;
                    const uint32_t* cI = lds + (int)n0.z + (R / 2) * t, *cQ = lds + (int)n0.w + (R / 2) * t;
                    uint32_t wI[NW], wQ[NW];
                    constexpr int EB = (32 - (hb_pairs<48>() - 1) - 1) / 2;      // 10
                    if constexpr (R == 8) {
                        const uint4* pI = reinterpret_cast<const uint4*>(oI);
                        const uint4* pQ = reinterpret_cast<const uint4*>(oQ);
#pragma unroll
                        for (int q = 0; q < 5; q++) {
                            uint4 a = pI[q], b = pQ[q];
                            wI[4*q] = a.x; wI[4*q+1] = a.y; wI[4*q+2] = a.z; wI[4*q+3] = a.w;
                            wQ[4*q] = b.x; wQ[4*q+1] = b.y; wQ[4*q+2] = b.z; wQ[4*q+3] = b.w;
                        }
                        ld_centre5<EB>(cI, vI);
                        ld_centre5<EB>(cQ, vQ);
                    } else if constexpr (R == 4) {
                        const uint2* pI = reinterpret_cast<const uint2*>(oI);
                        const uint2* pQ = reinterpret_cast<const uint2*>(oQ);
#pragma unroll
                        for (int q = 0; q < NW / 2; q++) {
                            uint2 a = pI[q], b = pQ[q];
                            wI[2*q] = a.x; wI[2*q+1] = a.y; wQ[2*q] = b.x; wQ[2*q+1] = b.y;
                        }
                        const uint2 a = *reinterpret_cast<const uint2*>(cI + EB), b = *reinterpret_cast<const uint2*>(cQ + EB);
                        vI[0] = a.x; vI[1] = a.y; vI[2] = cI[EB + 2]; vQ[0] = b.x; vQ[1] = b.y; vQ[2] = cQ[EB + 2];
                    } else {
#pragma unroll
                        for (int q = 0; q < NW; q++) { wI[q] = oI[q]; wQ[q] = oQ[q]; }
#pragma unroll
                        for (int q = 0; q < NE; q++) { vI[q] = cI[EB + q]; vQ[q] = cQ[EB + q]; }
                    }
                    static_for<0, R>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        int aI = 0, aQ = 0;
                        static_for<0, NW>([&](auto dc) {
                            constexpr int d = decltype(dc)::value;
                            constexpr uint32_t cf = pk_coef<48, MODE_CEN>(r, d);
                            if constexpr (cf != 0) { aI = dot2(wI[d], cf, aI); aQ = dot2(wQ[d], cf, aQ); }
                        });
                        sI[r] = aI; sQ[r] = aQ;
                    });
                }
                const long abs0 = chunk * lv.nout + R * t;
                // --- per stage of the entry: centre tap, int16 store, own arms, sinks
                auto emit = [&](const uint4 o0, const uint4 o1, const uint4 oc) {
                    int yI[R], yQ[R];
                    static_for<0, R>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        constexpr int dd = (r + 1) >> 1;
                        const int aI = dot2(vI[dd], (r & 1) ? oc.y : oc.x, sI[r]);
                        const int aQ = dot2(vQ[dd], (r & 1) ? oc.w : oc.z, sQ[r]);
                        // Sample::setReal (:828) keeps the low 16 bits: every store below (arms, node streams) takes just those,
                        // so the sign extension is left to the one consumer that needs the value (the channel end's division)
                        yI[r] = aI >> (HB_SHIFT - 1);
                        yQ[r] = aQ >> (HB_SHIFT - 1);
                    });
                    auto alt = [](uint32_t v) { return ((0u - v) & 0xffffu) | (v & 0xffff0000u); };   // arm entry m even (low half): wrap-negated
                    if ((int)o0.x >= 0) {                                          // own arms for the children
                        if constexpr (R == 8) {
                            const int p = HIST / 2 + 2 * t;
                            const uint32_t e0I = pack_iq(yI[0], yI[2]), e1I = pack_iq(yI[4], yI[6]);
                            const uint32_t e0Q = pack_iq(yQ[0], yQ[2]), e1Q = pack_iq(yQ[4], yQ[6]);
                            *reinterpret_cast<uint2*>(lds + (int)o0.x + p) = make_uint2(e0I, e1I);
                            *reinterpret_cast<uint2*>(lds + (int)o0.y + p) = make_uint2(e0Q, e1Q);
                            const uint32_t o0I = pack_iq(yI[1], yI[3]), o1I = pack_iq(yI[5], yI[7]);
                            const uint32_t o0Q = pack_iq(yQ[1], yQ[3]), o1Q = pack_iq(yQ[5], yQ[7]);
                            if ((int)o0.z >= 0) {
                                *reinterpret_cast<uint2*>(lds + (int)o0.z + p) = make_uint2(o0I, o1I);
                                *reinterpret_cast<uint2*>(lds + (int)o0.w + p) = make_uint2(o0Q, o1Q);
                            }
                            if ((int)o1.x >= 0) {
                                *reinterpret_cast<uint2*>(lds + (int)o1.x + p) = make_uint2(alt(o0I), alt(o1I));
                                *reinterpret_cast<uint2*>(lds + (int)o1.y + p) = make_uint2(alt(o0Q), alt(o1Q));
                            }
                        } else if constexpr (R == 4) {
                            const int p = HIST / 2 + t;
                            lds[(int)o0.x + p] = pack_iq(yI[0], yI[2]);
                            lds[(int)o0.y + p] = pack_iq(yQ[0], yQ[2]);
                            const uint32_t oI1 = pack_iq(yI[1], yI[3]), oQ1 = pack_iq(yQ[1], yQ[3]);
                            if ((int)o0.z >= 0) { lds[(int)o0.z + p] = oI1; lds[(int)o0.w + p] = oQ1; }
                            if ((int)o1.x >= 0) { lds[(int)o1.x + p] = alt(oI1); lds[(int)o1.y + p] = alt(oQ1); }
                        } else {
                            // outputs 2t (even -> E'[t]) and 2t + 1 (odd -> O'[t]): one int16 per arm
                            const int h = HIST + t;                                // int16 index inside the array
                            reinterpret_cast<uint16_t*>(lds + (int)o0.x)[h] = (uint16_t)yI[0];
                            reinterpret_cast<uint16_t*>(lds + (int)o0.y)[h] = (uint16_t)yQ[0];
                            if ((int)o0.z >= 0) {
                                reinterpret_cast<uint16_t*>(lds + (int)o0.z)[h] = (uint16_t)yI[1];
                                reinterpret_cast<uint16_t*>(lds + (int)o0.w)[h] = (uint16_t)yQ[1];
                            }
                            if ((int)o1.x >= 0) {
                                const bool neg = (t & 1) == 0;                    // entry m = t even: wrap-negated
                                reinterpret_cast<uint16_t*>(lds + (int)o1.x)[h] = (uint16_t)(neg ? 0 - yI[1] : yI[1]);
                                reinterpret_cast<uint16_t*>(lds + (int)o1.y)[h] = (uint16_t)(neg ? 0 - yQ[1] : yQ[1]);
                            }
                        }
                    }
                    if (live) {                                                    // channel ends / node streams
                        for (int si = (int)o1.z; si >= 0; ) {
                            TkSink sk;
                            {
                                const uint2* q = reinterpret_cast<const uint2*>(lds + st.sink_tab + (si - st.sink_base) * TK_SINK_DW);
                                uint2 w[TK_SINK_DW / 2];
#pragma unroll
                                for (int u = 0; u < TK_SINK_DW / 2; u++) w[u] = q[u];
                                __builtin_memcpy(&sk, w, sizeof sk);
                            }
                            const long rel = abs0 - sk.lo, span = sk.hi - sk.lo;
                            // the sink pointers come out of a table: tell the compiler they are global memory (flat_store otherwise)
                            typedef uint32_t __attribute__((address_space(1))) gu32;
                            gu32* dst = (gu32*)(sk.ptr0 + abs0);
                            if (rel >= 0 && rel + R <= span) {                     // whole job in range: no per-sample guards
#pragma unroll
                                for (int r = 0; r < R; r++)
                                    dst[r] = sk.shift ? pack_iq(div_pow2_trunc((int)(int16_t)yI[r], sk.shift), div_pow2_trunc((int)(int16_t)yQ[r], sk.shift))
                                                      : pack_iq(yI[r], yQ[r]);
                            } else if (rel > -R && rel < span) {
#pragma unroll
                                for (int r = 0; r < R; r++)
                                    if (rel + r >= 0 && rel + r < span)
                                        dst[r] = sk.shift ? pack_iq(div_pow2_trunc((int)(int16_t)yI[r], sk.shift), div_pow2_trunc((int)(int16_t)yQ[r], sk.shift))
                                                          : pack_iq(yI[r], yQ[r]);
                            }
                            si = sk.next;
                        }
                    }
                };
                emit(nt[1], nt[2], nt[3]);
                if ((int)nt[5].w != 0) emit(nt[4], nt[5], nt[6]);                  // fused upper sibling
            };
            if (lv.r_log2 == 3)      for (int j = tid; j < njobs; j += NT) job(std::integral_constant<int, 8>{}, j);
            else if (lv.r_log2 == 2) for (int j = tid; j < njobs; j += NT) job(std::integral_constant<int, 4>{}, j);
            else                     for (int j = tid; j < njobs; j += NT) job(std::integral_constant<int, 2>{}, j);
            __syncthreads();
        }
    }
}

// new history = last hist_len samples of (old history ++ new samples); one block row per stream
struct TkHistJob { const uint32_t* old_hist; const uint32_t* in; uint32_t* new_hist; long n_new; long hist_len; };

__global__ void tree_hist_kernel(const TkHistJob* __restrict__ jobs)
{
    const TkHistJob jb = jobs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= jb.hist_len) return;
    const long src = (long)i + jb.n_new - jb.hist_len;
    jb.new_hist[i] = src >= 0 ? jb.in[src] : jb.old_hist[i + jb.n_new];
}

} // namespace sdrx
