// libsdrx.so: sdrx_decim_* -- drop-in for Decimators<qint32,qint16,16,InputBits>
// (sdrbase/dsp/decimators.h:279-341).  Host logic + kernel dispatch; kernels in decim_kernel.hpp.
#include "sdrx_common.hpp"
#include "decim_kernel.hpp"
#include "decim_fast_kernel.hpp"
#include <cstring>
#include <vector>
#include <utility>
#include <cstdlib>
#include <new>

namespace sdrx {

// decimate1 (decimators.h:344-355): (int16)(x << pre1), elementwise
__global__ void decim1_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, long n, int pre)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint32_t v = in[i];
        const int re = (int)(int16_t)(v & 0xffffu), im = (int)(int16_t)(v >> 16);
        out[i] = pack_iq((int)((uint32_t)re << pre), (int)((uint32_t)im << pre));
    }
}

// DecimatorsU::decimate1 (decimatorsu.h:218-230): (int16)((byte - Shift) << pre1); one complex sample = one uint16
__global__ void decim1_u8_kernel(const uint16_t* __restrict__ in, uint32_t* __restrict__ out, long n, int pre, int shift)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint32_t v = in[i];
        const int re = (int)(v & 0xffu) - shift, im = (int)(v >> 8) - shift;
        out[i] = pack_iq((int)((uint32_t)re << pre), (int)((uint32_t)im << pre));
    }
}

// new history = last `hist_dw` dwords of (old history ++ consumed input), all counted in dwords; blockIdx.y = stream
struct HistJob { const uint32_t* old_hist; const uint32_t* in; uint32_t* new_hist; long n_in_dw; };
struct HistJobs { HistJob j[DJ_MAX]; };
__global__ void hist_update_kernel(const HistJobs jobs, int hist_dw)
{
    const HistJob& job = jobs.j[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hist_dw) return;
    const long src = (long)i + job.n_in_dw - hist_dw;
    job.new_hist[i] = src >= 0 ? job.in[src] : job.old_hist[i + job.n_in_dw];
}

// ---- explicit stage states: the six IntHalfbandFilterEO members of ONE Decimators object (m_decimator2 .. m_decimator64,
// decimators.h:326-333).  Every decimateK_x of the object runs its cascade on the same six filters, so after a change of
// K / fcPos the new cascade starts from what each stage saw last.  The parallel kernels derive the filter state from the
// last 4096 input samples of THEIR variant; this serial walk (one lane: a variant change is a rare event) is the bridge:
// it turns an input history into the six rings (`zero_init`, no output) and it runs the first 4096 samples after a change
// from explicit rings (in/out), after which the input history alone determines the state again (62 * 63 < 4096).
// Ring layout: per stage 64 (re, im) int32 pairs, oldest first, as stored by storeSample32 (i.e. AFTER the rotation).
constexpr int SG_STAGES = 6, SG_RING = 64, SG_STAGE_DW = SG_RING * 2, SG_DW = SG_STAGES * SG_STAGE_DW;

struct SerialJob {
    const void* in; long n_in; uint32_t* out; int32_t* rings;
    int L, pre, post, u8, in_shift, zero_init;
    int mode[SG_STAGES];
};

__global__ void decim_serial_kernel(const SerialJob jb)
{
    __shared__ int ring[SG_STAGES][2][SG_RING];
    __shared__ unsigned cnt[SG_STAGES];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int s = 0; s < jb.L; s++) {
        for (int k = 0; k < SG_RING; k++) {
            ring[s][0][k] = jb.zero_init ? 0 : jb.rings[s * SG_STAGE_DW + 2 * k];
            ring[s][1][k] = jb.zero_init ? 0 : jb.rings[s * SG_STAGE_DW + 2 * k + 1];
        }
        cnt[s] = 0;                                         // next store -> slot 0 (the oldest), rotation phase 0
    }
    long k_out = 0;
    for (long i = 0; i < jb.n_in; i++) {
        int re, im;
        if (jb.u8) {
            const uint32_t v = static_cast<const uint16_t*>(jb.in)[i];
            re = (int)((uint32_t)((int)(v & 0xffu) - jb.in_shift) << jb.pre); im = (int)((uint32_t)((int)(v >> 8) - jb.in_shift) << jb.pre);
        } else {
            const uint32_t v = static_cast<const uint32_t*>(jb.in)[i];
            re = (int)((uint32_t)(int)(int16_t)(v & 0xffffu) << jb.pre); im = (int)((uint32_t)(int)(int16_t)(v >> 16) << jb.pre);
        }
        int s = 0;
        for (; s < jb.L; s++) {
            const unsigned n = cnt[s];
            int xr = re, xi = im;
            const int mode = jb.mode[s];
            if (mode) {                                     // inf: j^(n+1), sup: (-j)^(n+1)  (inthalfbandfiltereo.h:626-692)
                const unsigned ph = n & 3u;
                const int nr = (int)(0u - (uint32_t)re), ni = (int)(0u - (uint32_t)im);
                if (ph == 1) { xr = nr; xi = ni; }
                else if (ph != 3) {
                    if ((ph == 0) == (mode == 1)) { xr = ni; xi = re; }        // inf ph 0 / sup ph 2: (-y, x)
                    else { xr = im; xi = nr; }                                 // inf ph 2 / sup ph 0: (y, -x)
                }
            }
            const unsigned M = n & 63u;
            ring[s][0][M] = xr; ring[s][1][M] = xi;
            cnt[s] = n + 1;
            if (!(n & 1u)) break;
            uint32_t ar = 0, ai = 0;
            for (int t = 0; t < 16; t++) {
                const unsigned a = (M - 2u * (unsigned)t) & 63u, b = (M - 62u + 2u * (unsigned)t) & 63u;
                const uint32_t c = (uint32_t)hb_c<64>(t);
                ar += ((uint32_t)ring[s][0][a] + (uint32_t)ring[s][0][b]) * c;
                ai += ((uint32_t)ring[s][1][a] + (uint32_t)ring[s][1][b]) * c;
            }
            const unsigned mc = (M - 31u) & 63u;
            ar += (uint32_t)ring[s][0][mc] << (HB_SHIFT - 1);
            ai += (uint32_t)ring[s][1][mc] << (HB_SHIFT - 1);
            re = (int)ar >> (HB_SHIFT - 1); im = (int)ai >> (HB_SHIFT - 1);
        }
        if (s == jb.L && jb.out) jb.out[k_out++] = pack_iq(re >> jb.post, im >> jb.post);
    }
    for (int s = 0; s < jb.L; s++)
        for (int k = 0; k < SG_RING; k++) {
            const unsigned src = (cnt[s] + (unsigned)k) & 63u;               // cnt & 63 = the oldest entry
            jb.rings[s * SG_STAGE_DW + 2 * k] = ring[s][0][src];
            jb.rings[s * SG_STAGE_DW + 2 * k + 1] = ring[s][1][src];
        }
}

typedef void (*chain_fn)(const DecimJobs, int, int, int);
typedef void (*fast_fn)(const DecimJobs, int, int, int);

struct ChainEntry { chain_fn fn; fast_fn fast; const char* name; const char* fast_name; int lds; int fast_lds; fast_fn fast4; const char* fast4_name; int fast4_lds;
                    fast_fn fast_mx; fast_fn fast4_mx; };      // the FAST kernels with stages 1-3 on the matrix cores (the default engine)

template<int L, int FC, int PRE, bool U8> static ChainEntry entry()
{
    static char name[64], fname[64], f4name[64];
    snprintf(name, sizeof name, "decim_chain_kernel<%d,%d,%d,%d>", L, FC, PRE, (int)U8);
    snprintf(fname, sizeof fname, "decim_fast_kernel<%d,%d,%d,%d,1>", L, FC, PRE, (int)U8);
    snprintf(f4name, sizeof f4name, "decim_fast_kernel<%d,%d,%d,%d,4>", L, FC, PRE, (int)U8);
    return ChainEntry{ &decim_chain_kernel<L, FC, PRE, U8>, &decim_fast_kernel<L, FC, PRE, U8, 1, false>, name, fname,
                       dc_lds_dwords(L) * 4, df_lds_dwords(L) * 4,
                       &decim_fast_kernel<L, FC, PRE, U8, 4, false>, f4name, df_lds_dwords(L, 4 * DF_SUB) * 4,
                       &decim_fast_kernel<L, FC, PRE, U8, 1, true>, &decim_fast_kernel<L, FC, PRE, U8, 4, true> };
}

// decimation_shifts<16,InputBits> (decimators.h:25-185)
static void shifts(int bits, int log2, int* pre, int* post)
{
    static const int pre12[7] = { 4, 3, 2, 1, 0, 0, 0 }, post12[7] = { 0, 0, 0, 0, 0, 1, 2 };
    static const int pre8[7]  = { 8, 7, 6, 5, 4, 3, 2 };
    if (bits == 12) { *pre = pre12[log2]; *post = post12[log2]; }
    else if (bits == 8) { *pre = pre8[log2]; *post = 0; }
    else { *pre = 0; *post = log2; }
}

template<int L, int FC> static bool pick_pre(int pre, bool u8, ChainEntry* e)
{
    // the only `pre` values decimation_shifts<16,{8,12,16}> produce for this L; DecimatorsU is <16,8> only
    constexpr int p12[7] = { 4, 3, 2, 1, 0, 0, 0 }, p8[7] = { 8, 7, 6, 5, 4, 3, 2 };
    if (u8) { if (pre != p8[L]) return false; *e = entry<L, FC, p8[L], true>(); return true; }
    if (pre == 0)      { *e = entry<L, FC, 0, false>(); return true; }
    if (pre == p12[L]) { *e = entry<L, FC, p12[L], false>(); return true; }
    if (pre == p8[L])  { *e = entry<L, FC, p8[L], false>(); return true; }
    return false;
}
template<int L> static bool pick_fc(int fc, int pre, bool u8, ChainEntry* e)
{
    switch (fc) {
    case 0: return pick_pre<L, 0>(pre, u8, e);
    case 1: return pick_pre<L, 1>(pre, u8, e);
    default: return pick_pre<L, 2>(pre, u8, e);
    }
}
static bool pick(int L, int fc, int pre, bool u8, ChainEntry* e)
{
    switch (L) {
    case 1: return pick_fc<1>(fc, pre, u8, e);
    case 2: return pick_fc<2>(fc, pre, u8, e);
    case 3: return pick_fc<3>(fc, pre, u8, e);
    case 4: return pick_fc<4>(fc, pre, u8, e);
    case 5: return pick_fc<5>(fc, pre, u8, e);
    case 6: return pick_fc<6>(fc, pre, u8, e);
    }
    return false;
}

} // namespace sdrx

using namespace sdrx;

struct sdrx_decim {
    int device = 0, log2 = 0, fcpos = 2, bits = 12, pre = 0, post = 0, group = 2;
    int cus = 256;
    bool u8 = false;              // DecimatorsU flavour: quint8 I/Q input, value = byte - in_shift
    int in_shift = 0;
    int bps = 4;                  // bytes per complex input sample
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint32_t* d_hist[2] = { nullptr, nullptr };
    int cur = 0;
    DevBuf d_in, d_out, d_flags;
    // after sdrx_decim_load_stages: explicit rings, current while `since_load` < DC_CHUNK samples have been processed
    int32_t* d_rings = nullptr;
    bool rings_live = false;
    long since_load = 0;
    int path = 0;                 // 0 auto (FAST + flagged EXACT), 1 exact only, 2 fast only (debug: no fallback)
    ChainEntry k{ nullptr, nullptr, "", "", 0, 0, nullptr, "", 0, nullptr, nullptr };
    bool mfma = true;             // half-band engine of the FAST kernel: matrix cores unless SDRX_DECIM_ENGINE=valu (read at create)
    char last_name[96] = "";
    int last_grid = 0, last_block = 0, last_lds = 0;
    EventTimer timer;
    // pinned, double-buffered host path (sdrx_decim_ring_*): the caller's receive buffers ARE slots of this ring
    struct Ring {
        int n_slots = 0, flush = 1;
        long slot_elems = 0, slot_out = 0;                          // input elements (int16 or bytes) and outputs per FULL slot
        char* h_in = nullptr; char* h_out = nullptr;                // pinned
        char* d_in = nullptr; char* d_out = nullptr;                // device mirrors, same slot layout
        hipStream_t s_in = nullptr, s_out = nullptr;                // copy streams: H2D of run k+1 overlaps kernel k and D2H k-1
        std::vector<hipEvent_t> ev_in, ev_k;                        // per slot: its run's H2D done / kernels done
        std::vector<hipEvent_t> ev;                                 // per slot; a run of slots flushed together completes on its last slot's event
        std::vector<int> state, ev_of;                              // state: 0 free, 1 acquired, 2 submitted (not yet flushed), 3 in flight
        std::vector<long> n_elems, n_out;
        long head = 0, tail = 0, flushed = 0;                       // next to acquire / next to retire / next to flush (monotonic counters)
    } ring;
};

static int group_int16(int log2, int fcpos)
{
    // `pos +=` strides of decimateK_* (decimators.h:348, 467, 545, 687, 833, 1099, 1605, 2614, 2664, 2742, 2862, 3080, 3492)
    if (log2 == 0) return 2;
    if (log2 <= 2) return 4 << log2;
    return fcpos == SDRX_FC_CEN ? (2 << log2) : (4 << log2);
}

// choose chunks-per-segment: minimise (cps + 1 warm-up) * waves of segments over `slots` resident workgroups
static int choose_cps(long n_chunks, int slots)
{
    const char* env = getenv("SDRX_DECIM_CPS");
    if (env && atoi(env) > 0) return atoi(env);
    long best = 1; double best_cost = 1e300;
    for (long cps = 1; cps <= n_chunks && cps <= 4096; cps = cps < 16 ? cps + 1 : cps + cps / 8) {
        const long segs = (n_chunks + cps - 1) / cps;
        const long rounds = (segs + slots - 1) / slots;
        const double cost = (double)(cps + 1) * (double)rounds;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = cps; }
    }
    return (int)best;
}

static void stage_modes(int log2, int fcpos, int* mode)
{
    // decimateK_inf: Inf, Sup, ..., Sup, Cen ; _sup: Sup, Inf, ..., Inf, Cen ; K = 2: single ; K = 4: pair (decimators.h:463-2584)
    for (int s = 0; s < SG_STAGES; s++) {
        int m = 0;
        if (s < log2 && fcpos != SDRX_FC_CEN) {
            const int first = fcpos == SDRX_FC_INF ? 1 : 2, other = 3 - first;
            m = s == 0 ? first : (log2 >= 3 && s == log2 - 1) ? 0 : other;
        }
        mode[s] = m;
    }
}

static int launch_serial(sdrx_decim* h, hipStream_t stream, const void* d_in, long n, uint32_t* d_out, int32_t* d_rings, bool zero_init)
{
    SerialJob jb;
    jb.in = d_in; jb.n_in = n; jb.out = d_out; jb.rings = d_rings;
    jb.L = h->log2; jb.pre = h->pre; jb.post = h->post; jb.u8 = h->u8 ? 1 : 0; jb.in_shift = h->in_shift; jb.zero_init = zero_init ? 1 : 0;
    stage_modes(h->log2, h->fcpos, jb.mode);
    hipLaunchKernelGGL(decim_serial_kernel, dim3(1), dim3(64), 0, stream, jb);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

// A handle that was given explicit stage states runs its first DC_CHUNK samples through the serial walk (outputs + updated
// rings + its input history), then the parallel kernels take over.  Returns how many samples were consumed here.
static int run_transition(sdrx_decim* x, hipStream_t stream, const void* d_iq, long n_cplx, int16_t* d_out, long* consumed)
{
    *consumed = 0;
    if (!x->rings_live || n_cplx <= 0 || x->log2 == 0) return SDRX_OK;
    long left = DC_CHUNK - x->since_load;
    left = (left + 127) / 128 * 128;                       // keeps the rest of the call on whole groups and 16-byte aligned
    const long m = n_cplx < left ? n_cplx : left;
    int rc = launch_serial(x, stream, d_iq, m, reinterpret_cast<uint32_t*>(d_out), x->d_rings, false); if (rc) return rc;
    const int hist_dw = DC_CHUNK * x->bps / 4;
    HistJobs hj;
    std::memset(&hj, 0, sizeof hj);
    hj.j[0].old_hist = x->d_hist[x->cur]; hj.j[0].in = static_cast<const uint32_t*>(d_iq);
    hj.j[0].new_hist = x->d_hist[x->cur ^ 1]; hj.j[0].n_in_dw = m * x->bps / 4;
    hipLaunchKernelGGL(hist_update_kernel, dim3((unsigned)((hist_dw + 255) / 256), 1u), dim3(256), 0, stream, hj, hist_dw);
    SDRX_HIP(hipGetLastError());
    x->cur ^= 1;
    x->since_load += m;
    if (x->since_load >= DC_CHUNK) x->rings_live = false;
    *consumed = m;
    return SDRX_OK;
}

// Waves per workgroup of the FAST kernel for a launch of `total_in` samples (all streams of the batch).  A single-wave
// segment carries 4096 samples of warm-up; a call of a few M samples cannot give every SIMD several of them at a useful
// length.  Four waves on one segment share ONE warm-up chunk and a quarter of the serial path, at the price of a barrier
// behind every stage: 1 Mi samples 67 vs 36 GS/s, 4 Mi 172 vs 128, 10 M (BASELINE cfg 2's own size) 207 vs 177, 16 Mi 252 vs
// 257, 64 Mi 344 vs 392, 1 Gi 452 vs 553 (profiles/r02_decim_nw_sweep.txt).
static int fast_nw(const sdrx_decim* h, long total_in)
{
    const char* env = getenv("SDRX_DECIM_NW");
    if (env && (atoi(env) == 1 || atoi(env) == 4)) return atoi(env);
    // matrix-core engine (round 3 sweep, four-wave vs single-wave; the single-wave flavour has all six stages on the matrix cores):
    // 4 Mi 170 vs 129 GS/s, 10 M 254 vs 202, 16 Mi 339 vs 323, 32 Mi 448 vs 450, 64 Mi 491 vs 539, 128 Mi 484 vs 525
    if (h->mfma) return total_in <= 28L * 1024 * 1024 ? 4 : 1;
    return total_in <= 12L * 1024 * 1024 ? 4 : 1;
}

// One launch (per kernel) for n <= DJ_MAX streams of one configuration: hs[i] consumes n_cplx[i] whole-group samples
// at d_iq[i] into d_out[i].  Everything is queued on hs[0]'s stream; timing and last_launch are kept on hs[0].
static int launch_batch(sdrx_decim* const* hs, int n, const void* const* d_iq_in, const long* n_cplx_in, int16_t* const* d_out_in)
{
    sdrx_decim* h = hs[0];
    const void* d_iq[DJ_MAX]; long n_cplx[DJ_MAX]; int16_t* d_out[DJ_MAX];
    for (int i = 0; i < n; i++) {
        d_iq[i] = d_iq_in[i]; n_cplx[i] = n_cplx_in[i]; d_out[i] = d_out_in[i];
        long used = 0;
        int rc = run_transition(hs[i], h->stream, d_iq[i], n_cplx[i], d_out[i], &used); if (rc) return rc;
        if (used) {
            d_iq[i] = static_cast<const char*>(d_iq[i]) + used * hs[i]->bps;
            d_out[i] += 2 * (used >> hs[i]->log2);
            n_cplx[i] -= used;
        }
    }
    long total = 0, longest = 0;
    for (int i = 0; i < n; i++) { total += n_cplx[i]; if (n_cplx[i] > longest) longest = n_cplx[i]; }
    if (total <= 0) return SDRX_OK;
    if (h->log2 == 0) {
        for (int i = 0; i < n; i++) {
            if (n_cplx[i] <= 0) continue;
            const int block = 256;
            long grid = (n_cplx[i] + block - 1) / block; if (grid > 4096) grid = 4096;
            if (h->u8)
                hipLaunchKernelGGL(decim1_u8_kernel, dim3((unsigned)grid), dim3(block), 0, h->stream,
                                   static_cast<const uint16_t*>(d_iq[i]), reinterpret_cast<uint32_t*>(d_out[i]), n_cplx[i], h->pre, h->in_shift);
            else
                hipLaunchKernelGGL(decim1_kernel, dim3((unsigned)grid), dim3(block), 0, h->stream,
                                   static_cast<const uint32_t*>(d_iq[i]), reinterpret_cast<uint32_t*>(d_out[i]), n_cplx[i], h->pre);
            SDRX_HIP(hipGetLastError());
            snprintf(h->last_name, sizeof h->last_name, "decim1_kernel");
            h->last_grid = (int)grid; h->last_block = block; h->last_lds = 0;
        }
        return SDRX_OK;
    }
    const long max_chunks = (longest + DC_CHUNK - 1) / DC_CHUNK;
    if (max_chunks > 0x7fffffffL / 4) { set_error("input too long for one call"); return SDRX_EINVAL; }
    DecimJobs jobs;
    std::memset(&jobs, 0, sizeof jobs);
    for (int i = 0; i < n; i++) {
        DecimJob& j = jobs.j[i];
        j.hist = hs[i]->d_hist[hs[i]->cur]; j.in = d_iq[i]; j.out = reinterpret_cast<uint32_t*>(d_out[i]);
        j.flags = nullptr; j.n_in = n_cplx[i] > 0 ? n_cplx[i] : 0; j.n_units = 0;
    }
    bool have_flags = false;
    int trc = h->timer.begin(h->stream); if (trc) return trc;
    if (h->path != 1) {
        // FAST: one wave per segment of `spw` sub-chunks (multiple of 4 = one flag chunk), 4 warm-up sub-chunks -- or, for
        // calls too short for that (fast_nw), four waves per segment on sub-chunks of 4096 samples with one warm-up sub-chunk
        long total_in = 0;
        for (int i = 0; i < n; i++) total_in += jobs.j[i].n_in;
        const int nw = fast_nw(h, total_in);
        const long sub_len = (long)DF_SUB * nw;
        long tot_sub = 0, max_sub = 0;
        for (int i = 0; i < n; i++) {
            const long chunks = (jobs.j[i].n_in + DC_CHUNK - 1) / DC_CHUNK;
            trc = hs[i]->d_flags.reserve((size_t)(chunks > 0 ? chunks : 1) * 4); if (trc) return trc;
            jobs.j[i].flags = static_cast<uint32_t*>(hs[i]->d_flags.p);
            const long ns = (jobs.j[i].n_in + sub_len - 1) / sub_len;
            jobs.j[i].n_units = (int)ns; tot_sub += ns; if (ns > max_sub) max_sub = ns;
        }
        have_flags = true;
        long spw;
        if (nw == 1) {
            // segment length: 32 sub-chunks (12.5 % warm-up) measured best once that still gives >= 8 waves
            // per CU (sweep in profiles/r01_decim_sweep.txt); shorter inputs trade warm-up against fill.  A batch is
            // sized by the sub-chunks of ALL its streams: that is what fills the chip.
            const long slots = (long)h->cus * 8;
            spw = 32; double best = 1e300;
            const char* env = getenv("SDRX_DECIM_SPW");
            if (env && atoi(env) >= 4) spw = (atoi(env) / 4) * 4;
            else if (h->mfma && tot_sub >= 32 * slots) {
                // skewed matrix-core kernel: a segment costs spw + 4 warm-up + (L - 1) drain iterations, so long launches want longer
                // segments (about 4096 of them): 256 Mi samples 0.434 (32) / 0.414 (64) / 0.401 ms (88); 1 Gi 1.42 (64) / 1.38 (128) / 1.41 ms (256)
                spw = ((tot_sub / 4096 + 3) / 4) * 4;
                if (spw < 32) spw = 32;
                if (spw > 128) spw = 128;
            }
            else if (tot_sub >= 8 * 32 * slots) spw = 64;          // long launches: halve the warm-up share (512 Mi samples: 429 vs 418 GS/s, 1 Gi: 494 vs 471)
            else if (tot_sub < 32 * slots) for (long c = 4; c <= 32; c += 4) {
                long segs = 0;
                for (int i = 0; i < n; i++) segs += (jobs.j[i].n_units + c - 1) / c;
                const long rounds = (segs + slots - 1) / slots;
                const double cost = (double)(c + DF_WARM + (h->mfma ? h->log2 - 1 : 0)) * (double)rounds;
                if (cost < best - 1e-9) { best = cost; spw = c; }
            }
            if (spw > max_sub) spw = ((max_sub + 3) / 4) * 4;
        } else {
            // 4-wave workgroups: 3 per CU (45 KB of LDS each), one warm-up sub-chunk per segment
            const long slots = (long)h->cus * 3;
            const int warm = DF_WARM / nw;
            spw = 8; double best = 1e300;
            const char* env = getenv("SDRX_DECIM_SPW");
            if (env && atoi(env) >= 1) spw = atoi(env);
            else for (long c = 1; c <= 32; c++) {
                long segs = 0;
                for (int i = 0; i < n; i++) segs += (jobs.j[i].n_units + c - 1) / c;
                const long rounds = (segs + slots - 1) / slots;
                const double cost = (double)(c + warm) * (double)rounds;
                if (cost < best - 1e-9) { best = cost; spw = c; }
            }
            if (spw > max_sub) spw = max_sub;
        }
        const long segs = (max_sub + spw - 1) / spw;
        hipLaunchKernelGGL(h->mfma ? (nw == 1 ? h->k.fast_mx : h->k.fast4_mx) : (nw == 1 ? h->k.fast : h->k.fast4),
                           dim3((unsigned)segs, (unsigned)n), dim3(64 * nw), 0, h->stream, jobs, (int)spw, h->post, h->in_shift);
        SDRX_HIP(hipGetLastError());
        snprintf(h->last_name, sizeof h->last_name, "%s%s", nw == 1 ? h->k.fast_name : h->k.fast4_name, h->mfma ? "+mfma" : "");
        h->last_grid = (int)(segs * n); h->last_block = 64 * nw; h->last_lds = nw == 1 ? h->k.fast_lds : h->k.fast4_lds;
    }
    if (h->path != 2) {
        for (int i = 0; i < n; i++) jobs.j[i].n_units = (int)((jobs.j[i].n_in + DC_CHUNK - 1) / DC_CHUNK);
        const int per = h->cus * 3 / n > 0 ? h->cus * 3 / n : 1;
        const int cps = choose_cps(max_chunks, per);
        long segs = (max_chunks + cps - 1) / cps;
        if (have_flags) {                                      // fallback run: grid-stride scan of the flags
            const long cap = (h->cus + n - 1) / n;
            if (segs > cap) segs = cap;
        }
        hipLaunchKernelGGL(h->k.fn, dim3((unsigned)segs, (unsigned)n), dim3(DC_THREADS), 0, h->stream, jobs, cps, h->post, h->in_shift);
        SDRX_HIP(hipGetLastError());
        if (h->path == 1) {
            snprintf(h->last_name, sizeof h->last_name, "%s", h->k.name);
            h->last_grid = (int)(segs * n); h->last_block = DC_THREADS; h->last_lds = h->k.lds;
        }
    }
    trc = h->timer.end(h->stream); if (trc) return trc;
    const int hist_dw = DC_CHUNK * h->bps / 4;
    HistJobs hj;
    std::memset(&hj, 0, sizeof hj);
    for (int i = 0; i < n; i++) {
        hj.j[i].old_hist = hs[i]->d_hist[hs[i]->cur]; hj.j[i].in = static_cast<const uint32_t*>(d_iq[i]);
        hj.j[i].new_hist = hs[i]->d_hist[hs[i]->cur ^ 1]; hj.j[i].n_in_dw = jobs.j[i].n_in * h->bps / 4;
    }
    hipLaunchKernelGGL(hist_update_kernel, dim3((unsigned)((hist_dw + 255) / 256), (unsigned)n), dim3(256), 0, h->stream, hj, hist_dw);
    SDRX_HIP(hipGetLastError());
    for (int i = 0; i < n; i++) hs[i]->cur ^= 1;
    return SDRX_OK;
}

static int launch(sdrx_decim* h, const void* d_iq, long n_cplx, int16_t* d_out)
{
    // n_cplx: whole groups only (caller truncated)
    if (n_cplx <= 0) return SDRX_OK;
    return launch_batch(&h, 1, &d_iq, &n_cplx, &d_out);
}

/* ---- pinned double-buffered host path ------------------------------------------------------------------------- */
static void ring_free(sdrx_decim* h)
{
    sdrx_decim::Ring& r = h->ring;
    for (hipEvent_t e : r.ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r.ev_in) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r.ev_k) if (e) (void)hipEventDestroy(e);
    if (r.s_in) { (void)hipStreamSynchronize(r.s_in); (void)hipStreamDestroy(r.s_in); }
    if (r.s_out) { (void)hipStreamSynchronize(r.s_out); (void)hipStreamDestroy(r.s_out); }
    if (r.h_in) (void)hipHostFree(r.h_in);
    if (r.h_out) (void)hipHostFree(r.h_out);
    if (r.d_in) (void)hipFree(r.d_in);
    if (r.d_out) (void)hipFree(r.d_out);
    r = sdrx_decim::Ring();
}

static int ring_flush(sdrx_decim* h)
{
    sdrx_decim::Ring& r = h->ring;
    const size_t esz = h->u8 ? 1 : 2;
    while (r.flushed < r.head && r.state[(size_t)(r.flushed % r.n_slots)] == 2) {
        // longest run of submitted FULL slots that is contiguous in the ring (no wrap): one copy in, one launch, one copy out.
        // A short slot (fewer elements than the slot holds: its own tail-drop rule applies) travels alone.
        const long s0 = r.flushed % r.n_slots;
        const bool full0 = r.n_elems[(size_t)s0] == r.slot_elems;
        long run = 1;
        if (full0)
            while (r.flushed + run < r.head && s0 + run < r.n_slots && r.state[(size_t)(s0 + run)] == 2 && r.n_elems[(size_t)(s0 + run)] == r.slot_elems) run++;
        const long elems = full0 ? run * r.slot_elems : r.n_elems[(size_t)s0];
        const long groups = elems / h->group;
        const long n_cplx = groups * (h->group / 2), n_out = n_cplx >> h->log2;
        const size_t in_off = (size_t)s0 * (size_t)r.slot_elems * esz, out_off = (size_t)s0 * (size_t)r.slot_out * 4;
        // three streams, chained by events: copy in -> kernels (the handle's stream, where the filter state lives) -> copy out
        const size_t last = (size_t)(s0 + run - 1);
        if (elems) SDRX_HIP(hipMemcpyAsync(r.d_in + in_off, r.h_in + in_off, (size_t)elems * esz, hipMemcpyHostToDevice, r.s_in));
        SDRX_HIP(hipEventRecord(r.ev_in[last], r.s_in));
        SDRX_HIP(hipStreamWaitEvent(h->stream, r.ev_in[last], 0));
        int rc = launch(h, r.d_in + in_off, n_cplx, reinterpret_cast<int16_t*>(r.d_out + out_off)); if (rc) return rc;
        SDRX_HIP(hipEventRecord(r.ev_k[last], h->stream));
        SDRX_HIP(hipStreamWaitEvent(r.s_out, r.ev_k[last], 0));
        if (n_out) SDRX_HIP(hipMemcpyAsync(r.h_out + out_off, r.d_out + out_off, (size_t)n_out * 4, hipMemcpyDeviceToHost, r.s_out));
        SDRX_HIP(hipEventRecord(r.ev[last], r.s_out));
        for (long k = 0; k < run; k++) {
            const size_t sl = (size_t)(s0 + k);
            r.state[sl] = 3; r.ev_of[sl] = (int)(s0 + run - 1);
            r.n_out[sl] = full0 ? r.slot_out : n_out;
        }
        r.flushed += run;
    }
    return SDRX_OK;
}

extern "C" {

int sdrx_decim_ring_create(sdrx_decim_t* h, int32_t slot_elems, int32_t n_slots, int32_t flush_slots)
{
    if (!h || n_slots < 2 || slot_elems <= 0) { set_error("sdrx_decim_ring_create: need a handle, >= 2 slots, a positive slot size"); return SDRX_EINVAL; }
    if (slot_elems % h->group) { set_error("sdrx_decim_ring_create: the slot size must be a whole number of decimation groups (sdrx_decim_group_int16)"); return SDRX_EINVAL; }
    if ((size_t)slot_elems * (h->u8 ? 1 : 2) % 16) { set_error("sdrx_decim_ring_create: slot bytes must be a multiple of 16"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    ring_free(h);
    sdrx_decim::Ring& r = h->ring;
    r.n_slots = n_slots; r.flush = flush_slots < 1 ? 1 : (flush_slots > n_slots - 1 ? n_slots - 1 : flush_slots);
    r.slot_elems = slot_elems; r.slot_out = (slot_elems / 2) >> h->log2;
    const size_t in_bytes = (size_t)n_slots * (size_t)slot_elems * (h->u8 ? 1 : 2);
    const size_t out_bytes = (size_t)n_slots * (size_t)(r.slot_out > 0 ? r.slot_out : 1) * 4;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&r.h_in), in_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&r.h_out), out_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r.d_in), in_bytes + 64);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r.d_out), out_bytes + 64);
    if (e != hipSuccess) { ring_free(h); return hip_fail(e, "sdrx_decim_ring_create", __FILE__, __LINE__); }
    r.ev.assign((size_t)n_slots, nullptr); r.ev_in.assign((size_t)n_slots, nullptr); r.ev_k.assign((size_t)n_slots, nullptr);
    r.state.assign((size_t)n_slots, 0); r.ev_of.assign((size_t)n_slots, 0);
    r.n_elems.assign((size_t)n_slots, 0); r.n_out.assign((size_t)n_slots, 0);
    for (auto* v : { &r.ev, &r.ev_in, &r.ev_k })
        for (auto& ev : *v) { e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (e != hipSuccess) { ring_free(h); return hip_fail(e, "hipEventCreate", __FILE__, __LINE__); } }
    e = hipStreamCreateWithFlags(&r.s_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r.s_out, hipStreamNonBlocking);
    if (e != hipSuccess) { ring_free(h); return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    return SDRX_OK;
}

int sdrx_decim_ring_destroy(sdrx_decim_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    ring_free(h);
    return SDRX_OK;
}

void* sdrx_decim_ring_acquire(sdrx_decim_t* h)
{
    if (!h || !h->ring.n_slots) { set_error("sdrx_decim_ring_acquire: no ring"); return nullptr; }
    sdrx_decim::Ring& r = h->ring;
    if (r.head - r.tail >= r.n_slots) { set_error("sdrx_decim_ring_acquire: ring full -- retire the oldest slot first"); return nullptr; }
    const size_t sl = (size_t)(r.head % r.n_slots);
    r.state[sl] = 1;                                                   // acquiring twice without a submit hands out the same slot
    return r.h_in + sl * (size_t)r.slot_elems * (h->u8 ? 1 : 2);
}

int sdrx_decim_ring_submit(sdrx_decim_t* h, int32_t n_elems)
{
    if (!h || !h->ring.n_slots) { set_error("sdrx_decim_ring_submit: no ring"); return SDRX_ESTATE; }
    sdrx_decim::Ring& r = h->ring;
    const size_t sl = (size_t)(r.head % r.n_slots);
    if (r.state[sl] != 1) { set_error("sdrx_decim_ring_submit: no acquired slot"); return SDRX_ESTATE; }
    if (n_elems < 0 || n_elems > r.slot_elems) { set_error("sdrx_decim_ring_submit: more elements than the slot holds"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    r.n_elems[sl] = n_elems; r.state[sl] = 2; r.head++;
    // launch once `flush` slots are waiting, when the run reaches the end of the ring, or when the slot is short
    if (r.head - r.flushed >= r.flush || r.head % r.n_slots == 0 || n_elems != r.slot_elems) return ring_flush(h);
    return SDRX_OK;
}

int sdrx_decim_ring_retire(sdrx_decim_t* h, const int16_t** out_iq, int32_t* n_out_cplx)
{
    if (!h || !h->ring.n_slots || !out_iq || !n_out_cplx) { set_error("sdrx_decim_ring_retire: bad argument"); return SDRX_EINVAL; }
    sdrx_decim::Ring& r = h->ring;
    if (r.tail >= r.head) { set_error("sdrx_decim_ring_retire: nothing submitted"); return SDRX_ESTATE; }
    SDRX_HIP(hipSetDevice(h->device));
    const size_t sl = (size_t)(r.tail % r.n_slots);
    if (r.state[sl] == 2) { int rc = ring_flush(h); if (rc) return rc; }
    SDRX_HIP(hipEventSynchronize(r.ev[(size_t)r.ev_of[sl]]));
    *out_iq = reinterpret_cast<const int16_t*>(r.h_out + sl * (size_t)r.slot_out * 4);
    *n_out_cplx = (int32_t)r.n_out[sl];
    r.state[sl] = 0; r.tail++;
    return SDRX_OK;
}

} // extern "C"

extern "C" {

int sdrx_decim_group_int16(int log2_decim, int fcpos) { return group_int16(log2_decim, fcpos); }

static int create_common(sdrx_decim_t** out, int device, int log2_decim, int fcpos, int input_bits, bool u8, int shift);

int sdrx_decim_create(sdrx_decim_t** out, int device, int log2_decim, int fcpos, int input_bits)
{
    return create_common(out, device, log2_decim, fcpos, input_bits, false, 0);
}

int sdrx_decim_create_u8(sdrx_decim_t** out, int device, int log2_decim, int fcpos, int shift)
{
    if (shift < 0 || shift > 255) { set_error("sdrx_decim_create_u8: shift 0..255"); return SDRX_EINVAL; }
    return create_common(out, device, log2_decim, fcpos, 8, true, shift);
}

static int create_common(sdrx_decim_t** out, int device, int log2_decim, int fcpos, int input_bits, bool u8, int shift)
{
    if (!out) { set_error("sdrx_decim_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    if (log2_decim < 0 || log2_decim > 6 || fcpos < 0 || fcpos > 2 ||
        (input_bits != 8 && input_bits != 12 && input_bits != 16)) {
        set_error("sdrx_decim_create: log2 0..6, fcpos 0..2, input_bits 8|12|16");
        return SDRX_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_decim* h = new (std::nothrow) sdrx_decim;
    if (!h) return SDRX_ENOMEM;
    h->device = device; h->log2 = log2_decim; h->fcpos = fcpos; h->bits = input_bits;
    h->u8 = u8; h->in_shift = shift; h->bps = u8 ? 2 : 4;
    shifts(input_bits, log2_decim, &h->pre, &h->post);
    h->group = group_int16(log2_decim, fcpos);
    h->cus = device_cu_count(device);
    { const char* pe = getenv("SDRX_DECIM_PATH"); h->path = !pe ? 0 : !strcmp(pe, "exact") ? 1 : !strcmp(pe, "fast") ? 2 : 0; }
    { const char* pe = getenv("SDRX_DECIM_ENGINE"); h->mfma = !(pe && !strcmp(pe, "valu")); }
    if (log2_decim > 0 && !pick(log2_decim, fcpos, h->pre, u8, &h->k)) {
        delete h; set_error("sdrx_decim_create: no kernel for this configuration"); return SDRX_EINVAL;
    }
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    h->stream = h->own_stream;
    for (int i = 0; i < 2; i++) {
        e = hipMalloc(reinterpret_cast<void**>(&h->d_hist[i]), (size_t)DC_CHUNK * h->bps);
        if (e != hipSuccess) { sdrx_decim_destroy(h); return hip_fail(e, "hipMalloc(hist)", __FILE__, __LINE__); }
    }
    *out = h;
    return sdrx_decim_reset(h);
}

int sdrx_decim_destroy(sdrx_decim_t* h)
{
    if (!h) return SDRX_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); }
    for (int i = 0; i < 2; i++) if (h->d_hist[i]) (void)hipFree(h->d_hist[i]);
    if (h->d_rings) (void)hipFree(h->d_rings);
    h->d_in.release(); h->d_out.release(); h->d_flags.release(); h->timer.release();
    ring_free(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SDRX_OK;
}

int sdrx_decim_reset(sdrx_decim_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    // a zero SAMPLE: all-zero bytes for int16 input, the byte `in_shift` for the unsigned 8-bit flavour
    SDRX_HIP(hipMemsetAsync(h->d_hist[h->cur], h->u8 ? h->in_shift : 0, (size_t)DC_CHUNK * h->bps, h->stream));
    h->rings_live = false; h->since_load = 0;
    return SDRX_OK;
}

int sdrx_decim_set_stream(sdrx_decim_t* h, void* hip_stream)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));            // order pending work before switching
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return SDRX_OK;
}

int sdrx_decim_sync(sdrx_decim_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_decim_process_u8(sdrx_decim_t* h, const uint8_t* iq, int32_t n_uint8, int16_t* out_iq, int32_t* n_out_cplx)
{
    if (!h || !h->u8) { set_error("sdrx_decim_process_u8: handle was not made by sdrx_decim_create_u8"); return SDRX_ESTATE; }
    if (n_uint8 < 0 || (n_uint8 > 0 && (!iq || !out_iq))) { set_error("sdrx_decim_process_u8: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    const int64_t groups = n_uint8 / h->group;
    const int64_t n_cplx = groups * (h->group / 2);
    const int64_t n_out = n_cplx >> h->log2;
    if (n_out_cplx) *n_out_cplx = (int32_t)n_out;
    if (n_cplx == 0) return SDRX_OK;
    int rc = h->d_in.reserve((size_t)n_cplx * 2); if (rc) return rc;
    rc = h->d_out.reserve((size_t)n_out * 4); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(h->d_in.p, iq, (size_t)n_cplx * 2, hipMemcpyHostToDevice, h->stream));
    rc = launch(h, h->d_in.p, (long)n_cplx, static_cast<int16_t*>(h->d_out.p));
    if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(out_iq, h->d_out.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_decim_process_dev_u8(sdrx_decim_t* h, const uint8_t* d_iq, int64_t n_uint8, int16_t* d_out_iq, int64_t* n_out_cplx)
{
    if (!h || !h->u8) { set_error("sdrx_decim_process_dev_u8: handle was not made by sdrx_decim_create_u8"); return SDRX_ESTATE; }
    if (n_uint8 < 0 || (n_uint8 > 0 && (!d_iq || !d_out_iq))) { set_error("sdrx_decim_process_dev_u8: bad argument"); return SDRX_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(d_iq) & 7u) || (reinterpret_cast<uintptr_t>(d_out_iq) & 3u)) {
        set_error("sdrx_decim_process_dev_u8: d_iq must be 8-byte aligned"); return SDRX_EINVAL;
    }
    SDRX_HIP(hipSetDevice(h->device));
    const int64_t groups = n_uint8 / h->group;
    const int64_t n_cplx = groups * (h->group / 2);
    if (n_out_cplx) *n_out_cplx = n_cplx >> h->log2;
    return launch(h, d_iq, (long)n_cplx, d_out_iq);
}

int sdrx_decim_process_dev(sdrx_decim_t* h, const int16_t* d_iq, int64_t n_int16, int16_t* d_out_iq, int64_t* n_out_cplx)
{
    if (h && h->u8) { set_error("sdrx_decim_process_dev: handle takes unsigned 8-bit input, use the _u8 call"); return SDRX_ESTATE; }
    if (!h || n_int16 < 0 || (n_int16 > 0 && (!d_iq || !d_out_iq))) { set_error("sdrx_decim_process_dev: bad argument"); return SDRX_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15u) || (reinterpret_cast<uintptr_t>(d_out_iq) & 3u)) {
        set_error("sdrx_decim_process_dev: d_iq must be 16-byte aligned"); return SDRX_EINVAL;
    }
    SDRX_HIP(hipSetDevice(h->device));
    const int64_t groups = n_int16 / h->group;            // trailing partial group dropped (decimators.h:3492)
    const int64_t n_cplx = groups * (h->group / 2);
    if (n_out_cplx) *n_out_cplx = n_cplx >> h->log2;
    return launch(h, d_iq, (long)n_cplx, d_out_iq);
}

int sdrx_decim_process_dev_batch(sdrx_decim_t* const* handles, int32_t n_handles, const void* const* d_iq, const int64_t* n_int16,
                                 int16_t* const* d_out_iq, int64_t* n_out_cplx)
{
    if (!handles || n_handles <= 0 || !d_iq || !n_int16 || !d_out_iq) { set_error("sdrx_decim_process_dev_batch: bad argument"); return SDRX_EINVAL; }
    sdrx_decim* h0 = handles[0];
    if (!h0) { set_error("sdrx_decim_process_dev_batch: null handle"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h0->device));
    for (int i = 0; i < n_handles; i++) {
        sdrx_decim* h = handles[i];
        if (!h || h->u8 != h0->u8 || h->device != h0->device || h->log2 != h0->log2 || h->fcpos != h0->fcpos || h->bits != h0->bits ||
            h->in_shift != h0->in_shift || h->path != h0->path) {
            set_error("sdrx_decim_process_dev_batch: all handles must share one configuration and device"); return SDRX_EINVAL;
        }
        for (int k = 0; k < i; k++) if (handles[k] == h) { set_error("sdrx_decim_process_dev_batch: a handle appears twice"); return SDRX_EINVAL; }
        if (n_int16[i] < 0 || (n_int16[i] > 0 && (!d_iq[i] || !d_out_iq[i]))) { set_error("sdrx_decim_process_dev_batch: bad stream argument"); return SDRX_EINVAL; }
        if ((reinterpret_cast<uintptr_t>(d_iq[i]) & (h0->u8 ? 7u : 15u)) || (reinterpret_cast<uintptr_t>(d_out_iq[i]) & 3u)) {
            set_error("sdrx_decim_process_dev_batch: d_iq must be 16-byte aligned (8 for the u8 flavour)"); return SDRX_EINVAL;
        }
        if (h->stream != h0->stream) {                       // the batch runs on handles[0]'s stream: join it once
            SDRX_HIP(hipStreamSynchronize(h->stream));
            h->stream = h0->stream;
        }
    }
    for (int base = 0; base < n_handles; base += DJ_MAX) {
        const int n = n_handles - base < DJ_MAX ? n_handles - base : DJ_MAX;
        const void* in[DJ_MAX]; long nc[DJ_MAX]; int16_t* out[DJ_MAX];
        for (int i = 0; i < n; i++) {
            const int64_t groups = n_int16[base + i] / h0->group;          // trailing partial group dropped, per stream
            nc[i] = (long)(groups * (h0->group / 2));
            in[i] = d_iq[base + i]; out[i] = d_out_iq[base + i];
            if (n_out_cplx) n_out_cplx[base + i] = nc[i] >> h0->log2;
        }
        sdrx_decim* first = handles[base];
        if (base) {                                          // later sub-batches: keep timing / last_launch on handles[0]
            std::swap(first->timer, h0->timer);
        }
        const int rc = launch_batch(handles + base, n, in, nc, out);
        if (base) {
            std::swap(first->timer, h0->timer);
            snprintf(h0->last_name, sizeof h0->last_name, "%s", first->last_name);
        }
        if (rc) return rc;
    }
    return SDRX_OK;
}

int sdrx_decim_process(sdrx_decim_t* h, const int16_t* iq, int32_t n_int16, int16_t* out_iq, int32_t* n_out_cplx)
{
    if (h && h->u8) { set_error("sdrx_decim_process: handle takes unsigned 8-bit input, use the _u8 call"); return SDRX_ESTATE; }
    if (!h || n_int16 < 0 || (n_int16 > 0 && (!iq || !out_iq))) { set_error("sdrx_decim_process: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    const int64_t groups = n_int16 / h->group;
    const int64_t n_cplx = groups * (h->group / 2);
    const int64_t n_out = n_cplx >> h->log2;
    if (n_out_cplx) *n_out_cplx = (int32_t)n_out;
    if (n_cplx == 0) return SDRX_OK;
    int rc = h->d_in.reserve((size_t)n_cplx * 4); if (rc) return rc;
    rc = h->d_out.reserve((size_t)n_out * 4); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(h->d_in.p, iq, (size_t)n_cplx * 4, hipMemcpyHostToDevice, h->stream));
    rc = launch(h, h->d_in.p, (long)n_cplx, static_cast<int16_t*>(h->d_out.p));
    if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(out_iq, h->d_out.p, (size_t)n_out * 4, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int64_t sdrx_decim_state_bytes(const sdrx_decim_t* h) { return (int64_t)DC_CHUNK * (h ? h->bps : 4); }

int sdrx_decim_get_state(sdrx_decim_t* h, void* host_buf)
{
    if (!h || !host_buf) return SDRX_EINVAL;
    if (h->rings_live) { set_error("sdrx_decim_get_state: the handle still runs on loaded stage states (first 4096 samples after sdrx_decim_load_stages); save those with sdrx_decim_save_stages"); return SDRX_ESTATE; }
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemcpyAsync(host_buf, h->d_hist[h->cur], (size_t)DC_CHUNK * h->bps, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_decim_set_state(sdrx_decim_t* h, const void* host_buf)
{
    if (!h || !host_buf) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemcpyAsync(h->d_hist[h->cur], host_buf, (size_t)DC_CHUNK * h->bps, hipMemcpyHostToDevice, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    h->rings_live = false;
    return SDRX_OK;
}

int sdrx_decim_set_timing(sdrx_decim_t* h, int enabled)
{
    if (!h) return SDRX_EINVAL;
    h->timer.enabled = enabled != 0;
    return SDRX_OK;
}

int sdrx_decim_get_timing(sdrx_decim_t* h, double* total_ms, int64_t* launches, int reset)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    int rc = h->timer.collect(h->stream); if (rc) return rc;
    if (total_ms) *total_ms = h->timer.total_ms;
    if (launches) *launches = h->timer.count;
    if (reset) { h->timer.total_ms = 0; h->timer.count = 0; }
    return SDRX_OK;
}

int sdrx_decim_last_launch(const sdrx_decim_t* h, char* kernel_name, int name_cap, int* grid, int* block, int* lds_bytes)
{
    if (!h) return SDRX_EINVAL;
    if (kernel_name && name_cap > 0) snprintf(kernel_name, (size_t)name_cap, "%s", h->last_name);
    if (grid) *grid = h->last_grid;
    if (block) *block = h->last_block;
    if (lds_bytes) *lds_bytes = h->last_lds;
    return SDRX_OK;
}

} // extern "C"

/* ---- one Decimators object, several decimateK_x: the shared six stage states ------------------------------------- */
struct sdrx_decim_stages { int device = 0; int32_t* d_rings = nullptr; };

extern "C" {

int sdrx_decim_stages_create(sdrx_decim_stages_t** out, int device)
{
    if (!out) { set_error("sdrx_decim_stages_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_decim_stages* s = new (std::nothrow) sdrx_decim_stages;
    if (!s) return SDRX_ENOMEM;
    s->device = device;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_rings), SG_DW * 4);
    if (e == hipSuccess) e = hipMemset(s->d_rings, 0, SG_DW * 4);          // freshly constructed filters: all-zero rings
    if (e == hipSuccess) e = hipDeviceSynchronize();                       // the null-stream memset is not ordered against the handles' non-blocking streams
    if (e != hipSuccess) { if (s->d_rings) (void)hipFree(s->d_rings); delete s; return hip_fail(e, "sdrx_decim_stages_create", __FILE__, __LINE__); }
    *out = s;
    return SDRX_OK;
}

int sdrx_decim_stages_destroy(sdrx_decim_stages_t* s)
{
    if (!s) return SDRX_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->d_rings) (void)hipFree(s->d_rings);
    delete s;
    return SDRX_OK;
}

int sdrx_decim_save_stages(sdrx_decim_t* h, sdrx_decim_stages_t* s)
{
    if (!h || !s || h->device != s->device) { set_error("sdrx_decim_save_stages: bad argument (handle and stage set must live on one device)"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    if (h->log2 == 0) return SDRX_OK;                                       // decimate1 touches no filter
    if (h->rings_live) {
        SDRX_HIP(hipMemcpyAsync(s->d_rings, h->d_rings, (size_t)h->log2 * SG_STAGE_DW * 4, hipMemcpyDeviceToDevice, h->stream));
    } else {
        // steady state: the rings are a function of the last DC_CHUNK input samples (zero state in front of them is exact)
        int rc = launch_serial(h, h->stream, h->d_hist[h->cur], DC_CHUNK, nullptr, s->d_rings, true); if (rc) return rc;
    }
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_decim_load_stages(sdrx_decim_t* h, const sdrx_decim_stages_t* s)
{
    if (!h || !s || h->device != s->device) { set_error("sdrx_decim_load_stages: bad argument (handle and stage set must live on one device)"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    if (h->log2 == 0) return SDRX_OK;
    SDRX_HIP(hipStreamSynchronize(h->stream));
    if (!h->d_rings) SDRX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_rings), SG_DW * 4));
    SDRX_HIP(hipMemcpyAsync(h->d_rings, s->d_rings, SG_DW * 4, hipMemcpyDeviceToDevice, h->stream));
    h->rings_live = true; h->since_load = 0;
    return SDRX_OK;
}

} // extern "C"
