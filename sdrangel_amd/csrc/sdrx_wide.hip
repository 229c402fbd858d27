// libsdrx.so: the 24-bit sample flavour of the integer half-band path (the reference built with SDR_RX_SAMPLE_24BIT:
// dsptypes.h:24-34 FixReal = qint32, Sample = 8 bytes; decimators.h:326-333 and downchannelizer.h:78-81
// IntHalfbandFilterEO<qint64,qint64,N>; decimation_shifts<24,InputBits>, decimators.h:62-185).
//   sdrx_decim24_*      Decimators<qint32, qint16, 24, {8,12,16}>::decimate{1..64}_{cen,inf,sup}   int16 in, Sample{int32,int32} out
//   sdrx_chan24_bank_*  N DownChannelizer stage chains (order 48) on Sample{int32,int32}, final `/= (1 << n)`
// Accumulators reach 2^38 in this build, so the packed-int16 dot2 kernels of the 16-bit flavour do not carry over: this is a
// plain, exact 64-bit implementation -- one generic kernel, runtime stage modes, no sharing of stages between channels --
// offered for completeness of the build switch, not tuned.
// Semantics kept (all pinned against the reference's 24-bit build, oracle/ref_shim24.cpp): int64 pair sums and products,
// the centre tap `((int32_t) x) << 11` as an INT shift that wraps at 32 bits (inthalfbandfiltereo.h:818-827,858-867),
// `acc >> 11` narrowed to int32, rotations with int32 negation, whole groups / dropped tail for the decimators, carried
// phase (no drop) and C division toward zero for the channelizer.
#include "sdrx_common.hpp"
#include "hb_common.hpp"
#include <new>
#include <vector>
#include <cstring>

using namespace sdrx;

extern "C" int sdrx_chan_plan(int32_t in_rate, int32_t req_rate, int32_t req_fc, uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs);

namespace {

constexpr int W_CHUNK = 2048;          // input samples of a pass per chunk (absolute multiples: rotation phase = relative index & 3)
constexpr int W_WARM = 2;              // warm-up chunks: 4096 >= 62 * 63 (six order-64 stages) >= 46 * 63
constexpr int W_HIST = W_CHUNK * W_WARM;
constexpr int W_NT = 256;
constexpr int W_MAXS = 6;              // stages per pass
constexpr int W_H = 64;                // history entries in front of every stage array (62 needed)

struct WJob {                          // one stream through one pass
    const int32_t* hist;               // W_HIST samples (int32 pairs; int16 pairs when in16) in front of t_old
    const void* in;                    // new samples [t_old, t_new)
    int32_t* out;                      // element 0 <-> absolute output index o_base
    long t_old, t_new, o_base;
    long c_first, c_last;              // absolute chunk range to compute
    int n_stages, in16, pre, post, div_log2, cps;
    int mode[W_MAXS];
};

__host__ __device__ constexpr int w_arr(int s) { return W_H + (W_CHUNK >> (s - 1)); }       // entries of stage s's input array
__host__ __device__ constexpr int w_off(int s) { int o = 0; for (int u = 1; u < s; u++) o += 2 * w_arr(u); return o; }
__host__ __device__ constexpr int w_lds() { return w_off(W_MAXS + 1); }

// rotated sample of a stage input at relative index j (phase = j & 3; chunks start on absolute multiples of 4 per stage)
__device__ __forceinline__ void w_rot(int mode, int j, int re, int im, int& xr, int& xi)
{
    xr = re; xi = im;
    if (mode == 0) return;
    const int ph = j & 3;
    auto ng = [](int v) { return (int)(0u - (uint32_t)v); };
    if (mode == 1) {            // lower half: j^(n+1): (-y, x) (-x, -y) (y, -x) (x, y)
        if (ph == 0) { xr = ng(im); xi = re; } else if (ph == 1) { xr = ng(re); xi = ng(im); } else if (ph == 2) { xr = im; xi = ng(re); }
    } else {                    // upper half: (-j)^(n+1): (y, -x) (-x, -y) (-y, x) (x, y)
        if (ph == 0) { xr = im; xi = ng(re); } else if (ph == 1) { xr = ng(re); xi = ng(im); } else if (ph == 2) { xr = ng(im); xi = re; }
    }
}

template<int ORDER>
__global__ __launch_bounds__(W_NT)
void wide_pass_kernel(const WJob* __restrict__ jobs)
{
    constexpr int P = ORDER / 4;
    __shared__ int lds[w_lds()];
    const WJob jb = jobs[blockIdx.y];
    const int tid = threadIdx.x;
    const long first = jb.c_first + (long)blockIdx.x * jb.cps;
    if (first > jb.c_last) return;
    long last = first + jb.cps - 1; if (last > jb.c_last) last = jb.c_last;
    const int L = jb.n_stages;
    for (int i = tid; i < w_lds(); i += W_NT) lds[i] = 0;
    __syncthreads();
    const long o_lo = jb.t_old >> L, o_hi = jb.t_new >> L;
    for (long chunk = first - W_WARM; chunk <= last; ++chunk) {
        // ---- input chunk -> stage-1 array (entry W_H + j <-> relative sample j)
        {
            int* aI = lds + w_off(1), *aQ = aI + w_arr(1);
            for (int j = tid; j < W_CHUNK; j += W_NT) {
                const long p = chunk * W_CHUNK + j;
                int re = 0, im = 0;
                if (p >= jb.t_old - W_HIST && p < jb.t_new && p >= 0) {
                    const bool h = p < jb.t_old;
                    const long idx = h ? p - (jb.t_old - W_HIST) : p - jb.t_old;
                    if (jb.in16) {
                        const uint32_t v = h ? reinterpret_cast<const uint32_t*>(jb.hist)[idx] : static_cast<const uint32_t*>(jb.in)[idx];
                        re = (int)((uint32_t)(int)(int16_t)(v & 0xffffu) << jb.pre); im = (int)((uint32_t)(int)(int16_t)(v >> 16) << jb.pre);
                    } else {
                        const int2 v = h ? reinterpret_cast<const int2*>(jb.hist)[idx] : static_cast<const int2*>(jb.in)[idx];
                        re = v.x; im = v.y;
                    }
                }
                aI[W_H + j] = re; aQ[W_H + j] = im;
            }
        }
        __syncthreads();
        const bool live = chunk >= first;
        for (int s = 1; s <= L; s++) {
            const int nout = W_CHUNK >> s, mode = jb.mode[s - 1];
            const int* iI = lds + w_off(s), *iQ = iI + w_arr(s);
            int* oI = lds + w_off(s + 1), *oQ = oI + w_arr(s + 1);
            for (int k = tid; k < nout; k += W_NT) {
                const int M = 2 * k + 1;
                long aR = 0, aIm = 0;
#pragma unroll 4
                for (int i = 0; i < P; i++) {
                    const int ja = M - 2 * i, jbb = M - (ORDER - 2) + 2 * i;
                    int ar, ai, br, bi;
                    w_rot(mode, ja, iI[W_H + ja], iQ[W_H + ja], ar, ai);
                    w_rot(mode, jbb, iI[W_H + jbb], iQ[W_H + jbb], br, bi);
                    const long c = hb_c<ORDER>(i);
                    aR += ((long)ar + (long)br) * c;
                    aIm += ((long)ai + (long)bi) * c;
                }
                const int jc = M - (ORDER / 2 - 1);
                int cr, ci;
                w_rot(mode, jc, iI[W_H + jc], iQ[W_H + jc], cr, ci);
                aR += (long)(int)((uint32_t)cr << (HB_SHIFT - 1));          // ((int32_t) x) << 11: wraps at 32 bits, then widens
                aIm += (long)(int)((uint32_t)ci << (HB_SHIFT - 1));
                const int yr = (int)(uint32_t)(unsigned long)(aR >> (HB_SHIFT - 1));
                const int yi = (int)(uint32_t)(unsigned long)(aIm >> (HB_SHIFT - 1));
                if (s < L) { oI[W_H + k] = yr; oQ[W_H + k] = yi; }
                else if (live) {
                    const long ao = chunk * nout + k;                      // absolute output index
                    if (ao >= o_lo && ao < o_hi) {
                        int vr, vi;
                        if (jb.div_log2 >= 0) { const int d = 1 << jb.div_log2; vr = yr / d; vi = yi / d; }   // s.m_real /= (1 << n)
                        else { vr = yr >> jb.post; vi = yi >> jb.post; }
                        reinterpret_cast<int2*>(jb.out)[ao - jb.o_base] = make_int2(vr, vi);
                    }
                }
            }
            __syncthreads();
        }
        // ---- carry: the last W_H entries of every stage array become the next chunk's history
        for (int s = 1; s <= L; s++) {
            int* a = lds + w_off(s);
            const int n = W_CHUNK >> (s - 1);
            int keep = 0;
            const int arr = tid / W_H, e = tid % W_H;                       // 2 arrays x 64 entries: threads 0..127
            if (tid < 2 * W_H) keep = a[arr * w_arr(s) + n + e];
            __syncthreads();
            if (tid < 2 * W_H) a[arr * w_arr(s) + e] = keep;
        }
        __syncthreads();
    }
}

// new history = last `hist_elems` elements (4- or 8-byte) of (old history ++ new input); blockIdx.y = stream
struct WHistJob { const void* old_hist; const void* in; void* new_hist; long n_new; int elem8; };
__global__ void wide_hist_kernel(const WHistJob* __restrict__ jobs)
{
    const WHistJob jb = jobs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W_HIST) return;
    const long src = (long)i + jb.n_new - W_HIST;
    if (jb.elem8) static_cast<int2*>(jb.new_hist)[i] = src >= 0 ? static_cast<const int2*>(jb.in)[src] : static_cast<const int2*>(jb.old_hist)[i + jb.n_new];
    else static_cast<uint32_t*>(jb.new_hist)[i] = src >= 0 ? static_cast<const uint32_t*>(jb.in)[src] : static_cast<const uint32_t*>(jb.old_hist)[i + jb.n_new];
}

// decimate1: Sample(x << pre1) (decimators.h decimate1), no filter
__global__ void wide_shift_kernel(const int16_t* __restrict__ in, int32_t* __restrict__ out, long n, int pre)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)((uint32_t)(int32_t)in[i] << pre);
}

// one chain = a list of passes of <= 6 stages; pass p reads stream p (p = 0: the raw input) and writes stream p + 1
struct WPassState { int n_stages = 0; int mode[W_MAXS] = { 0 }; void* hist[2] = { nullptr, nullptr }; int cur = 0; DevBuf out; };
struct WChain {
    int n = 0; uint8_t modes[32] = { 0 };
    int32_t out_rate = 0, ofs = 0;
    std::vector<WPassState> pass;                   // pass[0] uses the engine's shared raw history unless the chain owns it
    long last_n = 0;
};

struct WEngine {
    int device = 0, order = 64, cus = 256;
    hipStream_t stream = nullptr;
    long T = 0;                                      // raw samples consumed since reset
    bool in16 = false; int pre = 0, post = 0;
    void* raw_hist[2] = { nullptr, nullptr }; int raw_cur = 0;
    std::vector<WChain> chains;
    DevBuf d_in;
    // job tables: pinned + device
    void* h_tab = nullptr; void* d_tab = nullptr; size_t tab_cap = 0; hipEvent_t tab_ev = nullptr;
};

int w_alloc_hist(void** p, size_t bytes, hipStream_t s)
{
    SDRX_HIP(hipMalloc(p, bytes));
    SDRX_HIP(hipMemsetAsync(*p, 0, bytes, s));
    return SDRX_OK;
}

void w_free(WEngine* e)
{
    for (int i = 0; i < 2; i++) if (e->raw_hist[i]) (void)hipFree(e->raw_hist[i]);
    for (auto& c : e->chains) for (auto& p : c.pass) { for (int i = 0; i < 2; i++) if (p.hist[i]) (void)hipFree(p.hist[i]); p.out.release(); }
    e->d_in.release();
    if (e->h_tab) (void)hipHostFree(e->h_tab);
    if (e->d_tab) (void)hipFree(e->d_tab);
    if (e->tab_ev) (void)hipEventDestroy(e->tab_ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
}

int w_init(WEngine* e, int device)
{
    e->device = device; e->cus = device_cu_count(device);
    SDRX_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    SDRX_HIP(hipEventCreateWithFlags(&e->tab_ev, hipEventDisableTiming));
    e->tab_cap = 64 * 1024;
    SDRX_HIP(hipHostMalloc(&e->h_tab, e->tab_cap, hipHostMallocDefault));
    SDRX_HIP(hipMalloc(&e->d_tab, e->tab_cap));
    const size_t raw_bytes = (size_t)W_HIST * (e->in16 ? 4 : 8);
    for (int i = 0; i < 2; i++) { int rc = w_alloc_hist(&e->raw_hist[i], raw_bytes, e->stream); if (rc) return rc; }
    return SDRX_OK;
}

int w_add_chain(WEngine* e, int n, const uint8_t* modes)
{
    WChain c; c.n = n; if (n) std::memcpy(c.modes, modes, (size_t)n);
    for (int base = 0; base < n; base += W_MAXS) {
        WPassState p; p.n_stages = n - base < W_MAXS ? n - base : W_MAXS;
        for (int i = 0; i < p.n_stages; i++) p.mode[i] = modes[base + i];
        if (base > 0) for (int i = 0; i < 2; i++) { int rc = w_alloc_hist(&p.hist[i], (size_t)W_HIST * 8, e->stream); if (rc) return rc; }
        c.pass.push_back(std::move(p));
    }
    e->chains.push_back(std::move(c));
    return SDRX_OK;
}

int w_reset(WEngine* e)
{
    SDRX_HIP(hipStreamSynchronize(e->stream));
    e->T = 0;
    SDRX_HIP(hipMemsetAsync(e->raw_hist[e->raw_cur], 0, (size_t)W_HIST * (e->in16 ? 4 : 8), e->stream));
    for (auto& c : e->chains) { c.last_n = 0; for (auto& p : c.pass) if (p.hist[p.cur]) SDRX_HIP(hipMemsetAsync(p.hist[p.cur], 0, (size_t)W_HIST * 8, e->stream)); }
    return SDRX_OK;
}

// feed n raw samples (device pointer) through every chain; outputs land in each chain's last pass buffer
int w_feed(WEngine* e, const void* d_in, long n)
{
    if (n <= 0) return SDRX_OK;
    const long T0 = e->T, T1 = e->T + n;
    size_t max_pass = 0;
    for (auto& c : e->chains) max_pass = c.pass.size() > max_pass ? c.pass.size() : max_pass;
    const size_t nch = e->chains.size();
    SDRX_HIP(hipEventSynchronize(e->tab_ev));
    if ((nch * (max_pass ? max_pass : 1)) * (sizeof(WJob) + sizeof(WHistJob)) + sizeof(WHistJob) > e->tab_cap) { set_error("too many chains for the job table"); return SDRX_EINVAL; }
    char* hp = static_cast<char*>(e->h_tab); char* dp = static_cast<char*>(e->d_tab);
    size_t used = 0;
    struct Launch { size_t off; int count; long max_segs; bool hist; };
    std::vector<Launch> launches;
    for (size_t p = 0; p < max_pass; p++) {
        WJob* jobs = reinterpret_cast<WJob*>(hp + used);
        int cnt = 0; long max_segs = 0;
        std::vector<WHistJob> hj;
        for (auto& c : e->chains) {
            if (p >= c.pass.size()) continue;
            WPassState& ps = c.pass[p];
            const int depth_in = (int)p * W_MAXS;                      // stages in front of this pass
            const long t_old = T0 >> depth_in, t_new = T1 >> depth_in;
            const long o_old = t_old >> ps.n_stages, o_new = t_new >> ps.n_stages;
            int rc = ps.out.reserve((size_t)(o_new - o_old + 1) * 8); if (rc) return rc;
            WJob& j = jobs[cnt++];
            std::memset(&j, 0, sizeof j);
            j.hist = static_cast<const int32_t*>(p == 0 ? e->raw_hist[e->raw_cur] : ps.hist[ps.cur]);
            j.in = p == 0 ? d_in : c.pass[p - 1].out.p;
            j.out = static_cast<int32_t*>(ps.out.p);
            j.t_old = t_old; j.t_new = t_new; j.o_base = o_old;
            j.c_first = t_old / W_CHUNK; j.c_last = t_new > t_old ? (t_new - 1) / W_CHUNK : j.c_first - 1;
            j.n_stages = ps.n_stages; j.in16 = (p == 0 && e->in16) ? 1 : 0; j.pre = e->pre;
            const bool lastp = p + 1 == c.pass.size();
            j.post = lastp ? e->post : 0;
            j.div_log2 = (lastp && !e->in16) ? c.n : -1;               // channelizer: /(1 << n) on the final output; decimators: >> post
            if (!lastp) j.div_log2 = -1;
            for (int i = 0; i < ps.n_stages; i++) j.mode[i] = ps.mode[i];
            const long chunks = j.c_last - j.c_first + 1;
            // segments: enough to fill the chip (4 per CU over all chains), but at least 8 chunks each where the feed has
            // them -- every segment spends W_WARM chunks re-creating its filter state
            const long target = (long)e->cus * 4 / (long)(nch ? nch : 1) + 1;
            long cps = (chunks + target - 1) / target;
            if (cps < 8) cps = chunks < 8 ? chunks : 8;
            if (cps < 1) cps = 1;
            if (cps > 64) cps = 64;
            j.cps = (int)cps;
            const long segs = chunks > 0 ? (chunks + cps - 1) / cps : 0;
            if (segs > max_segs) max_segs = segs;
            if (lastp) c.last_n = o_new - o_old;
            if (p > 0) hj.push_back(WHistJob{ ps.hist[ps.cur], j.in, ps.hist[ps.cur ^ 1], t_new - t_old, 1 });
        }
        launches.push_back(Launch{ used, cnt, max_segs, false });
        used += (size_t)cnt * sizeof(WJob);
        if (p == 0) hj.push_back(WHistJob{ e->raw_hist[e->raw_cur], d_in, e->raw_hist[e->raw_cur ^ 1], n, e->in16 ? 0 : 1 });
        if (!hj.empty()) {
            std::memcpy(hp + used, hj.data(), hj.size() * sizeof(WHistJob));
            launches.push_back(Launch{ used, (int)hj.size(), 0, true });
            used += hj.size() * sizeof(WHistJob);
        }
    }
    SDRX_HIP(hipMemcpyAsync(dp, hp, used, hipMemcpyHostToDevice, e->stream));
    SDRX_HIP(hipEventRecord(e->tab_ev, e->stream));
    for (const Launch& l : launches) {
        if (l.count == 0) continue;
        if (l.hist) hipLaunchKernelGGL(wide_hist_kernel, dim3(W_HIST / 256, (unsigned)l.count), dim3(256), 0, e->stream, reinterpret_cast<const WHistJob*>(dp + l.off));
        else if (l.max_segs > 0) {
            if (e->order == 64) hipLaunchKernelGGL(wide_pass_kernel<64>, dim3((unsigned)l.max_segs, (unsigned)l.count), dim3(W_NT), 0, e->stream, reinterpret_cast<const WJob*>(dp + l.off));
            else hipLaunchKernelGGL(wide_pass_kernel<48>, dim3((unsigned)l.max_segs, (unsigned)l.count), dim3(W_NT), 0, e->stream, reinterpret_cast<const WJob*>(dp + l.off));
        }
        SDRX_HIP(hipGetLastError());
    }
    e->raw_cur ^= 1;
    for (auto& c : e->chains) for (size_t p = 1; p < c.pass.size(); p++) c.pass[p].cur ^= 1;
    e->T = T1;
    return SDRX_OK;
}

} // namespace

struct sdrx_decim24 { WEngine e; int log2 = 0, fcpos = 2, bits = 12, group = 2; DevBuf d_out; };
struct sdrx_chan24_bank { WEngine e; int32_t in_rate = 0; };

extern "C" {

int sdrx_decim24_create(sdrx_decim24_t** out, int device, int log2_decim, int fcpos, int input_bits)
{
    if (!out) { set_error("sdrx_decim24_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    if (log2_decim < 0 || log2_decim > 6 || fcpos < 0 || fcpos > 2 || (input_bits != 8 && input_bits != 12 && input_bits != 16)) {
        set_error("sdrx_decim24_create: log2 0..6, fcpos 0..2, input_bits 8|12|16"); return SDRX_EINVAL;
    }
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_decim24* h = new (std::nothrow) sdrx_decim24;
    if (!h) return SDRX_ENOMEM;
    h->log2 = log2_decim; h->fcpos = fcpos; h->bits = input_bits;
    h->group = sdrx_decim_group_int16(log2_decim, fcpos);
    h->e.order = 64; h->e.in16 = true;
    h->e.pre = (input_bits == 16 ? 8 : input_bits == 12 ? 12 : 16) - log2_decim;       // decimation_shifts<24, InputBits>
    h->e.post = 0;
    rc = w_init(&h->e, device);
    if (!rc) {
        uint8_t modes[6];
        for (int s = 0; s < log2_decim; s++) {                                          // stage modes of decimateK_{inf,sup,cen}
            int m = 0;
            if (fcpos != SDRX_FC_CEN) {
                const int first = fcpos == SDRX_FC_INF ? 1 : 2, other = 3 - first;
                m = s == 0 ? first : (log2_decim >= 3 && s == log2_decim - 1) ? 0 : other;
            }
            modes[s] = (uint8_t)m;
        }
        rc = w_add_chain(&h->e, log2_decim, modes);
    }
    if (rc) { w_free(&h->e); delete h; return rc; }
    SDRX_HIP(hipStreamSynchronize(h->e.stream));
    *out = h;
    return SDRX_OK;
}

int sdrx_decim24_destroy(sdrx_decim24_t* h)
{
    if (!h) return SDRX_OK;
    (void)hipSetDevice(h->e.device);
    if (h->e.stream) (void)hipStreamSynchronize(h->e.stream);
    h->d_out.release();
    w_free(&h->e);
    delete h;
    return SDRX_OK;
}

int sdrx_decim24_reset(sdrx_decim24_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->e.device));
    return w_reset(&h->e);
}

int sdrx_decim24_process(sdrx_decim24_t* h, const int16_t* iq, int32_t n_int16, int32_t* out_iq, int32_t* n_out_cplx)
{
    if (!h || n_int16 < 0 || (n_int16 > 0 && (!iq || !out_iq))) { set_error("sdrx_decim24_process: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->e.device));
    const long groups = n_int16 / h->group;                  // whole groups only, the tail is dropped (decimators.h:3492)
    const long n_cplx = groups * (h->group / 2), n_out = n_cplx >> h->log2;
    if (n_out_cplx) *n_out_cplx = (int32_t)n_out;
    if (n_cplx == 0) return SDRX_OK;
    int rc = h->e.d_in.reserve((size_t)n_cplx * 4); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(h->e.d_in.p, iq, (size_t)n_cplx * 4, hipMemcpyHostToDevice, h->e.stream));
    if (h->log2 == 0) {
        rc = h->d_out.reserve((size_t)n_cplx * 8); if (rc) return rc;
        hipLaunchKernelGGL(wide_shift_kernel, dim3((unsigned)((2 * n_cplx + 255) / 256)), dim3(256), 0, h->e.stream,
                           static_cast<const int16_t*>(h->e.d_in.p), static_cast<int32_t*>(h->d_out.p), 2 * n_cplx, h->e.pre);
        SDRX_HIP(hipGetLastError());
        SDRX_HIP(hipMemcpyAsync(out_iq, h->d_out.p, (size_t)n_cplx * 8, hipMemcpyDeviceToHost, h->e.stream));
        SDRX_HIP(hipStreamSynchronize(h->e.stream));
        return SDRX_OK;
    }
    rc = w_feed(&h->e, h->e.d_in.p, n_cplx); if (rc) return rc;
    WChain& c = h->e.chains[0];
    SDRX_HIP(hipMemcpyAsync(out_iq, c.pass.back().out.p, (size_t)n_out * 8, hipMemcpyDeviceToHost, h->e.stream));
    SDRX_HIP(hipStreamSynchronize(h->e.stream));
    return SDRX_OK;
}

// device-resident flavour: d_iq = n_cplx int16 pairs (whole groups are the caller's business: n_cplx is taken as is),
// d_out = room for n_cplx >> log2 {int32, int32}; asynchronous on the handle's stream, sdrx_decim24_sync waits
int sdrx_decim24_process_dev(sdrx_decim24_t* h, const void* d_iq, int64_t n_cplx, void* d_out, int64_t* n_out_cplx)
{
    if (!h || n_cplx < 0 || (n_cplx > 0 && (!d_iq || !d_out))) { set_error("sdrx_decim24_process_dev: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->e.device));
    const long n_out = ((h->e.T + n_cplx) >> h->log2) - (h->e.T >> h->log2);
    if (n_out_cplx) *n_out_cplx = n_out;
    if (n_cplx == 0) return SDRX_OK;
    if (h->log2 == 0) {
        hipLaunchKernelGGL(wide_shift_kernel, dim3((unsigned)((2 * n_cplx + 255) / 256)), dim3(256), 0, h->e.stream,
                           static_cast<const int16_t*>(d_iq), static_cast<int32_t*>(d_out), (long)(2 * n_cplx), h->e.pre);
        SDRX_HIP(hipGetLastError());
        return SDRX_OK;
    }
    int rc = w_feed(&h->e, d_iq, (long)n_cplx); if (rc) return rc;
    if (n_out > 0) SDRX_HIP(hipMemcpyAsync(d_out, h->e.chains[0].pass.back().out.p, (size_t)n_out * 8, hipMemcpyDeviceToDevice, h->e.stream));
    return SDRX_OK;
}

int sdrx_decim24_sync(sdrx_decim24_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->e.device));
    SDRX_HIP(hipStreamSynchronize(h->e.stream));
    return SDRX_OK;
}

int sdrx_chan24_bank_create(sdrx_chan24_bank_t** out, int device, int32_t in_rate, int32_t n_ch, const int32_t* req_rate, const int32_t* req_fc)
{
    if (!out || n_ch <= 0 || !req_rate || !req_fc || in_rate <= 0) { set_error("sdrx_chan24_bank_create: bad argument"); return SDRX_EINVAL; }
    *out = nullptr;
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_chan24_bank* b = new (std::nothrow) sdrx_chan24_bank;
    if (!b) return SDRX_ENOMEM;
    b->in_rate = in_rate; b->e.order = 48; b->e.in16 = false;
    rc = w_init(&b->e, device);
    for (int c = 0; c < n_ch && !rc; c++) {
        uint8_t modes[32]; int32_t orate = 0, ofs = 0;
        const int n = sdrx_chan_plan(in_rate, req_rate[c], req_fc[c], modes, &orate, &ofs);     // the float bisection (downchannelizer.cpp:250-287)
        rc = w_add_chain(&b->e, n, modes);
        if (!rc) { b->e.chains.back().out_rate = orate; b->e.chains.back().ofs = ofs; }
    }
    if (rc) { w_free(&b->e); delete b; return rc; }
    SDRX_HIP(hipStreamSynchronize(b->e.stream));
    *out = b;
    return SDRX_OK;
}

int sdrx_chan24_bank_destroy(sdrx_chan24_bank_t* b)
{
    if (!b) return SDRX_OK;
    (void)hipSetDevice(b->e.device);
    if (b->e.stream) (void)hipStreamSynchronize(b->e.stream);
    w_free(&b->e);
    delete b;
    return SDRX_OK;
}

int sdrx_chan24_bank_reset(sdrx_chan24_bank_t* b)
{
    if (!b) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(b->e.device));
    return w_reset(&b->e);
}

int sdrx_chan24_bank_info(const sdrx_chan24_bank_t* b, int32_t c, int32_t* n_stages, uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs)
{
    if (!b || c < 0 || c >= (int32_t)b->e.chains.size()) { set_error("sdrx_chan24_bank_info: bad channel"); return SDRX_EINVAL; }
    const WChain& ch = b->e.chains[(size_t)c];
    if (n_stages) *n_stages = ch.n;
    if (modes) std::memcpy(modes, ch.modes, (size_t)ch.n);
    if (out_rate) *out_rate = ch.out_rate;
    if (residual_ofs) *residual_ofs = ch.ofs;
    return SDRX_OK;
}

int sdrx_chan24_bank_feed(sdrx_chan24_bank_t* b, const int32_t* iq, int64_t n_cplx)
{
    if (!b || n_cplx < 0 || (n_cplx > 0 && !iq)) { set_error("sdrx_chan24_bank_feed: bad argument"); return SDRX_EINVAL; }
    if (n_cplx == 0) { for (auto& c : b->e.chains) c.last_n = 0; return SDRX_OK; }
    SDRX_HIP(hipSetDevice(b->e.device));
    SDRX_HIP(hipStreamSynchronize(b->e.stream));
    int rc = b->e.d_in.reserve((size_t)n_cplx * 8); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(b->e.d_in.p, iq, (size_t)n_cplx * 8, hipMemcpyHostToDevice, b->e.stream));
    for (auto& c : b->e.chains) if (c.n == 0) c.last_n = n_cplx;               // no stage: the input goes straight through (downchannelizer.cpp:57-60)
    return w_feed(&b->e, b->e.d_in.p, (long)n_cplx);
}

// device-resident flavour: d_iq = n_cplx {int32, int32}; the outputs stay on the device (sdrx_chan24_bank_out_dev) until
// the next feed; asynchronous on the bank's stream, sdrx_chan24_bank_sync waits
int sdrx_chan24_bank_feed_dev(sdrx_chan24_bank_t* b, const void* d_iq, int64_t n_cplx)
{
    if (!b || n_cplx < 0 || (n_cplx > 0 && !d_iq)) { set_error("sdrx_chan24_bank_feed_dev: bad argument"); return SDRX_EINVAL; }
    if (n_cplx == 0) { for (auto& c : b->e.chains) c.last_n = 0; return SDRX_OK; }
    SDRX_HIP(hipSetDevice(b->e.device));
    for (auto& c : b->e.chains) if (c.n == 0) {
        int rc = b->e.d_in.reserve((size_t)n_cplx * 8); if (rc) return rc;
        SDRX_HIP(hipMemcpyAsync(b->e.d_in.p, d_iq, (size_t)n_cplx * 8, hipMemcpyDeviceToDevice, b->e.stream));
        break;
    }
    for (auto& c : b->e.chains) if (c.n == 0) c.last_n = n_cplx;
    return w_feed(&b->e, d_iq, (long)n_cplx);
}

int sdrx_chan24_bank_out_dev(sdrx_chan24_bank_t* b, int32_t c, const void** d_out, int64_t* n_cplx)
{
    if (!b || c < 0 || c >= (int32_t)b->e.chains.size() || !d_out || !n_cplx) { set_error("sdrx_chan24_bank_out_dev: bad argument"); return SDRX_EINVAL; }
    WChain& ch = b->e.chains[(size_t)c];
    *d_out = ch.n == 0 ? b->e.d_in.p : ch.pass.back().out.p;
    *n_cplx = ch.last_n;
    return SDRX_OK;
}

int sdrx_chan24_bank_sync(sdrx_chan24_bank_t* b)
{
    if (!b) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(b->e.device));
    SDRX_HIP(hipStreamSynchronize(b->e.stream));
    return SDRX_OK;
}

int64_t sdrx_chan24_bank_read(sdrx_chan24_bank_t* b, int32_t c, int32_t* out_iq, int64_t cap)
{
    if (!b || c < 0 || c >= (int32_t)b->e.chains.size() || cap < 0 || (cap > 0 && !out_iq)) { set_error("sdrx_chan24_bank_read: bad argument"); return SDRX_EINVAL; }
    if (hipSetDevice(b->e.device) != hipSuccess) return SDRX_EHIP;
    WChain& ch = b->e.chains[(size_t)c];
    const int64_t n = ch.last_n < cap ? ch.last_n : cap;
    if (n <= 0) return 0;
    const void* src = ch.n == 0 ? b->e.d_in.p : ch.pass.back().out.p;
    hipError_t e = hipMemcpyAsync(out_iq, src, (size_t)n * 8, hipMemcpyDeviceToHost, b->e.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->e.stream);
    if (e != hipSuccess) return hip_fail(e, "sdrx_chan24_bank_read", __FILE__, __LINE__);
    return n;
}

} // extern "C"
