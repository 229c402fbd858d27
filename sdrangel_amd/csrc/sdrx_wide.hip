// libsdrx.so: the 24-bit sample flavour of the integer half-band path (the reference built with SDR_RX_SAMPLE_24BIT:
// dsptypes.h:24-34 FixReal = qint32, Sample = 8 bytes; decimators.h:326-333 and downchannelizer.h:78-81
// IntHalfbandFilterEO<qint64,qint64,N>; decimation_shifts<24,InputBits>, decimators.h:62-185).
//   sdrx_decim24_*      Decimators<qint32, qint16, 24, {8,12,16}>::decimate{1..64}_{cen,inf,sup}   int16 in, Sample{int32,int32} out
//   sdrx_chan24_bank_*  N DownChannelizer stage chains (order 48) on Sample{int32,int32}, final `/= (1 << n)`
// Accumulators reach 2^38 in this build, so the packed-int16 dot2 kernels of the 16-bit flavour do not carry over: this is a
// plain, exact 64-bit implementation -- one generic kernel, runtime stage modes, the channels' common stages evaluated once --
// offered for completeness of the build switch, not tuned like the 16-bit path.
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

struct WJob {                          // one segment of the stage trie: <= 6 stages on one stream
    const int32_t* hist;               // W_HIST samples (int32 pairs; int16 pairs when in16) in front of t_old
    const void* in;                    // new samples [t_old, t_new)
    int32_t* out;                      // the segment's output as a stream for deeper segments (nullptr: none); element 0 <-> o_base
    int32_t* out2;                     // the same samples as a channel / decimator result: >> post, or / (1 << div_log2) (nullptr: none)
    long t_old, t_new, o_base;
    long c_first, c_last;              // absolute chunk range to compute
    int n_stages, in16, pre, post, div_log2, cps, warm;
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
    for (long chunk = first - jb.warm; chunk <= last; ++chunk) {
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
                // arrays hold ROTATED stage inputs (what storeSample keeps): each array has one consumer stage, so the
                // rotation is applied once where the sample is produced, not at each of the 2P + 1 reads of the FIR
                int xr, xi;
                w_rot(jb.mode[0], j, re, im, xr, xi);
                aI[W_H + j] = xr; aQ[W_H + j] = xi;
            }
        }
        __syncthreads();
        const bool live = chunk >= first;
        for (int s = 1; s <= L; s++) {
            const int nout = W_CHUNK >> s;
            const int* iI = lds + w_off(s), *iQ = iI + w_arr(s);
            int* oI = lds + w_off(s + 1), *oQ = oI + w_arr(s + 1);
            for (int k = tid; k < nout; k += W_NT) {
                const int M = 2 * k + 1;
                long aR = 0, aIm = 0;
#pragma unroll 4
                for (int i = 0; i < P; i++) {
                    const int ja = M - 2 * i, jbb = M - (ORDER - 2) + 2 * i;
                    const long c = hb_c<ORDER>(i);
                    aR += ((long)iI[W_H + ja] + (long)iI[W_H + jbb]) * c;
                    aIm += ((long)iQ[W_H + ja] + (long)iQ[W_H + jbb]) * c;
                }
                const int jc = M - (ORDER / 2 - 1);
                aR += (long)(int)((uint32_t)iI[W_H + jc] << (HB_SHIFT - 1));   // ((int32_t) x) << 11: wraps at 32 bits, then widens
                aIm += (long)(int)((uint32_t)iQ[W_H + jc] << (HB_SHIFT - 1));
                const int yr = (int)(uint32_t)(unsigned long)(aR >> (HB_SHIFT - 1));
                const int yi = (int)(uint32_t)(unsigned long)(aIm >> (HB_SHIFT - 1));
                if (s < L) { int xr, xi; w_rot(jb.mode[s], k, yr, yi, xr, xi); oI[W_H + k] = xr; oQ[W_H + k] = xi; }
                else if (live) {
                    const long ao = chunk * nout + k;                      // absolute output index
                    if (ao >= o_lo && ao < o_hi) {
                        if (jb.out) reinterpret_cast<int2*>(jb.out)[ao - jb.o_base] = make_int2(yr, yi);
                        if (jb.out2) {
                            int vr, vi;
                            if (jb.div_log2 >= 0) { const int d = 1 << jb.div_log2; vr = yr / d; vi = yi / d; }   // s.m_real /= (1 << n)
                            else { vr = yr >> jb.post; vi = yi >> jb.post; }
                            reinterpret_cast<int2*>(jb.out2)[ao - jb.o_base] = make_int2(vr, vi);
                        }
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

// The channels' stage strings form a trie (as in the 16-bit bank); it is cut into SEGMENTS: maximal runs of <= 6 stages without
// a branch and without a channel ending inside.  A segment is one kernel job; segments whose input is ready run in the same
// launch ("round" = number of segments between the raw stream and the job), blockIdx.y = job.  A segment's output is kept as a
// stream (new samples + W_HIST samples of history) when deeper segments read it, and/or as channel result (divided by 2^n).
struct WStream { DevBuf out; void* hist[2] = { nullptr, nullptr }; int cur = 0; int depth = 0; bool feeds = false; };
struct WSeg { int in_stream = 0, out_stream = -1; int n_stages = 0; int mode[W_MAXS] = { 0 }; int round = 1; int depth_out = 0;
              bool ends = false; DevBuf res; long last_n = 0; };
struct WTrieNode { int child[3] = { -1, -1, -1 }; int depth = 0; bool ends = false; int seg = -1; };
struct WChain { int n = 0; uint8_t modes[32] = { 0 }; int32_t out_rate = 0, ofs = 0; int seg = -1; };

struct WEngine {
    int device = 0, order = 64, cus = 256;
    hipStream_t stream = nullptr;
    long T = 0;                                      // raw samples consumed since reset
    bool in16 = false; int pre = 0, post = 0;
    std::vector<WChain> chains;
    std::vector<WStream> streams;                    // [0] = the raw input
    std::vector<WSeg> segs;
    int max_round = 0;
    long pass_n = 0;                                 // what the last feed was (pass-through channels)
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
    for (auto& st : e->streams) { for (int i = 0; i < 2; i++) if (st.hist[i]) (void)hipFree(st.hist[i]); st.out.release(); }
    for (auto& sg : e->segs) sg.res.release();
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
    return SDRX_OK;
}

void w_add_chain(WEngine* e, int n, const uint8_t* modes)
{
    WChain c; c.n = n; if (n) std::memcpy(c.modes, modes, (size_t)n);
    e->chains.push_back(c);
}

// trie -> segments -> streams; allocates the histories
int w_plan(WEngine* e)
{
    std::vector<WTrieNode> trie(1);
    std::vector<int> end_node(e->chains.size(), 0);
    for (size_t c = 0; c < e->chains.size(); c++) {
        int id = 0;
        for (int s = 0; s < e->chains[c].n; s++) {
            const int m = e->chains[c].modes[s];
            if (trie[(size_t)id].child[m] < 0) { WTrieNode nn; nn.depth = s + 1; trie.push_back(nn); trie[(size_t)id].child[m] = (int)trie.size() - 1; }
            id = trie[(size_t)id].child[m];
        }
        if (e->chains[c].n) trie[(size_t)id].ends = true;
        end_node[c] = id;
    }
    e->streams.clear(); e->segs.clear(); e->max_round = 0;
    { WStream raw; raw.depth = 0; raw.feeds = true; e->streams.push_back(std::move(raw)); }
    struct Todo { int node, stream, round; };
    std::vector<Todo> todo{ Todo{ 0, 0, 0 } };
    for (size_t ti = 0; ti < todo.size(); ti++) {
        const Todo t = todo[ti];
        for (int m = 0; m < 3; m++) {
            int id = trie[(size_t)t.node].child[m];
            if (id < 0) continue;
            WSeg sg; sg.in_stream = t.stream; sg.round = t.round + 1; sg.n_stages = 0;
            int mode = m;
            for (;;) {
                sg.mode[sg.n_stages++] = mode;
                int kids = 0, only = -1, only_mode = 0;
                for (int k = 0; k < 3; k++) if (trie[(size_t)id].child[k] >= 0) { kids++; only = trie[(size_t)id].child[k]; only_mode = k; }
                if (kids != 1 || trie[(size_t)id].ends || sg.n_stages == W_MAXS) break;
                id = only; mode = only_mode;
            }
            sg.depth_out = trie[(size_t)id].depth;
            sg.ends = trie[(size_t)id].ends;
            bool has_kids = false;
            for (int k = 0; k < 3; k++) has_kids = has_kids || trie[(size_t)id].child[k] >= 0;
            if (has_kids) {
                WStream st; st.depth = sg.depth_out; st.feeds = true;
                sg.out_stream = (int)e->streams.size();
                e->streams.push_back(std::move(st));
                todo.push_back(Todo{ id, sg.out_stream, sg.round });
            }
            trie[(size_t)id].seg = (int)e->segs.size();
            if (sg.round > e->max_round) e->max_round = sg.round;
            e->segs.push_back(std::move(sg));
        }
    }
    for (size_t c = 0; c < e->chains.size(); c++) e->chains[c].seg = e->chains[c].n ? trie[(size_t)end_node[c]].seg : -1;
    for (size_t i = 0; i < e->streams.size(); i++) {
        const size_t bytes = (size_t)W_HIST * ((i == 0 && e->in16) ? 4 : 8);
        for (int k = 0; k < 2; k++) { int rc = w_alloc_hist(&e->streams[i].hist[k], bytes, e->stream); if (rc) return rc; }
    }
    const size_t need = (e->segs.size() + 1) * sizeof(WJob) + (e->streams.size() + 1) * sizeof(WHistJob) + 256;
    e->tab_cap = need < 64 * 1024 ? 64 * 1024 : need;
    SDRX_HIP(hipHostMalloc(&e->h_tab, e->tab_cap, hipHostMallocDefault));
    SDRX_HIP(hipMalloc(&e->d_tab, e->tab_cap));
    return SDRX_OK;
}

int w_reset(WEngine* e)
{
    SDRX_HIP(hipStreamSynchronize(e->stream));
    e->T = 0;
    for (size_t i = 0; i < e->streams.size(); i++)
        SDRX_HIP(hipMemsetAsync(e->streams[i].hist[e->streams[i].cur], 0, (size_t)W_HIST * ((i == 0 && e->in16) ? 4 : 8), e->stream));
    for (auto& sg : e->segs) sg.last_n = 0;
    e->pass_n = 0;
    return SDRX_OK;
}

// feed n raw samples (device pointer) through every segment; results land in the segments' `res` buffers
int w_feed(WEngine* e, const void* d_in, long n)
{
    if (n <= 0) return SDRX_OK;
    const long T0 = e->T, T1 = e->T + n;
    SDRX_HIP(hipEventSynchronize(e->tab_ev));
    char* hp = static_cast<char*>(e->h_tab); char* dp = static_cast<char*>(e->d_tab);
    size_t used = 0;
    struct Launch { size_t off; int count; long max_segs; bool hist; };
    std::vector<Launch> launches;
    for (int r = 1; r <= e->max_round; r++) {
        WJob* jobs = reinterpret_cast<WJob*>(hp + used);
        int cnt = 0; long max_segs = 0, njobs = 0;
        for (auto& sg : e->segs) if (sg.round == r) njobs++;
        for (auto& sg : e->segs) {
            if (sg.round != r) continue;
            WStream& in = e->streams[(size_t)sg.in_stream];
            const long t_old = T0 >> in.depth, t_new = T1 >> in.depth;
            const long o_old = t_old >> sg.n_stages, o_new = t_new >> sg.n_stages;
            WJob& j = jobs[cnt++];
            std::memset(&j, 0, sizeof j);
            j.hist = static_cast<const int32_t*>(in.hist[in.cur]);
            j.in = sg.in_stream == 0 ? d_in : in.out.p;
            if (sg.out_stream >= 0) {
                WStream& os = e->streams[(size_t)sg.out_stream];
                int rc = os.out.reserve((size_t)(o_new - o_old + 1) * 8); if (rc) return rc;
                j.out = static_cast<int32_t*>(os.out.p);
            }
            if (sg.ends) {
                int rc = sg.res.reserve((size_t)(o_new - o_old + 1) * 8); if (rc) return rc;
                j.out2 = static_cast<int32_t*>(sg.res.p);
                sg.last_n = o_new - o_old;
            }
            j.t_old = t_old; j.t_new = t_new; j.o_base = o_old;
            j.c_first = t_old / W_CHUNK; j.c_last = t_new > t_old ? (t_new - 1) / W_CHUNK : j.c_first - 1;
            j.n_stages = sg.n_stages; j.in16 = (sg.in_stream == 0 && e->in16) ? 1 : 0; j.pre = e->pre;
            j.post = e->post;
            j.div_log2 = e->in16 ? -1 : sg.depth_out;                  // channelizer: / (1 << n) on a channel's result; decimators: >> post
            for (int i = 0; i < sg.n_stages; i++) j.mode[i] = sg.mode[i];
            j.warm = (int)(((long)(e->order - 2) * ((1L << sg.n_stages) - 1) + W_CHUNK - 1) / W_CHUNK);
            if (j.warm < 1) j.warm = 1;
            const long chunks = j.c_last - j.c_first + 1;
            // segments of the time axis: enough to fill the chip (4 per CU over the jobs of this launch), but at least 8 chunks
            // each where the feed has them -- every one spends `warm` chunks re-creating its filter state
            const long target = (long)e->cus * 4 / (njobs ? njobs : 1) + 1;
            long cps = (chunks + target - 1) / target;
            if (cps < 8) cps = chunks < 8 ? chunks : 8;
            if (cps < 1) cps = 1;
            if (cps > 64) cps = 64;
            j.cps = (int)cps;
            const long sgs = chunks > 0 ? (chunks + cps - 1) / cps : 0;
            if (sgs > max_segs) max_segs = sgs;
        }
        launches.push_back(Launch{ used, cnt, max_segs, false });
        used += (size_t)cnt * sizeof(WJob);
    }
    {
        WHistJob* hj = reinterpret_cast<WHistJob*>(hp + used);
        int cnt = 0;
        for (size_t i = 0; i < e->streams.size(); i++) {
            WStream& st = e->streams[i];
            const long nn = (T1 >> st.depth) - (T0 >> st.depth);
            hj[cnt++] = WHistJob{ st.hist[st.cur], i == 0 ? d_in : st.out.p, st.hist[st.cur ^ 1], nn, (i == 0 && e->in16) ? 0 : 1 };
        }
        launches.push_back(Launch{ used, cnt, 0, true });
        used += (size_t)cnt * sizeof(WHistJob);
    }
    if (used > e->tab_cap) { set_error("sdrx_wide: job table overflow"); return SDRX_EINVAL; }
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
    for (auto& st : e->streams) st.cur ^= 1;
    e->T = T1;
    e->pass_n = n;
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
        w_add_chain(&h->e, log2_decim, modes);
        rc = w_plan(&h->e);
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
    SDRX_HIP(hipMemcpyAsync(out_iq, h->e.segs[(size_t)h->e.chains[0].seg].res.p, (size_t)n_out * 8, hipMemcpyDeviceToHost, h->e.stream));
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
    if (n_out > 0) SDRX_HIP(hipMemcpyAsync(d_out, h->e.segs[(size_t)h->e.chains[0].seg].res.p, (size_t)n_out * 8, hipMemcpyDeviceToDevice, h->e.stream));
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
        w_add_chain(&b->e, n, modes);
        b->e.chains.back().out_rate = orate; b->e.chains.back().ofs = ofs;
    }
    if (!rc) rc = w_plan(&b->e);
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
    if (n_cplx == 0) { for (auto& sg : b->e.segs) sg.last_n = 0; b->e.pass_n = 0; return SDRX_OK; }
    SDRX_HIP(hipSetDevice(b->e.device));
    SDRX_HIP(hipStreamSynchronize(b->e.stream));
    int rc = b->e.d_in.reserve((size_t)n_cplx * 8); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(b->e.d_in.p, iq, (size_t)n_cplx * 8, hipMemcpyHostToDevice, b->e.stream));
    b->e.pass_n = n_cplx;                                                      // no stage: the input goes straight through (downchannelizer.cpp:57-60)
    return w_feed(&b->e, b->e.d_in.p, (long)n_cplx);
}

// device-resident flavour: d_iq = n_cplx {int32, int32}; the outputs stay on the device (sdrx_chan24_bank_out_dev) until
// the next feed; asynchronous on the bank's stream, sdrx_chan24_bank_sync waits
int sdrx_chan24_bank_feed_dev(sdrx_chan24_bank_t* b, const void* d_iq, int64_t n_cplx)
{
    if (!b || n_cplx < 0 || (n_cplx > 0 && !d_iq)) { set_error("sdrx_chan24_bank_feed_dev: bad argument"); return SDRX_EINVAL; }
    if (n_cplx == 0) { for (auto& sg : b->e.segs) sg.last_n = 0; b->e.pass_n = 0; return SDRX_OK; }
    SDRX_HIP(hipSetDevice(b->e.device));
    for (auto& c : b->e.chains) if (c.n == 0) {                                // a pass-through channel hands out a copy of the input
        int rc = b->e.d_in.reserve((size_t)n_cplx * 8); if (rc) return rc;
        SDRX_HIP(hipMemcpyAsync(b->e.d_in.p, d_iq, (size_t)n_cplx * 8, hipMemcpyDeviceToDevice, b->e.stream));
        break;
    }
    b->e.pass_n = n_cplx;
    return w_feed(&b->e, d_iq, (long)n_cplx);
}

int sdrx_chan24_bank_out_dev(sdrx_chan24_bank_t* b, int32_t c, const void** d_out, int64_t* n_cplx)
{
    if (!b || c < 0 || c >= (int32_t)b->e.chains.size() || !d_out || !n_cplx) { set_error("sdrx_chan24_bank_out_dev: bad argument"); return SDRX_EINVAL; }
    WChain& ch = b->e.chains[(size_t)c];
    *d_out = ch.n == 0 ? b->e.d_in.p : b->e.segs[(size_t)ch.seg].res.p;
    *n_cplx = ch.n == 0 ? b->e.pass_n : b->e.segs[(size_t)ch.seg].last_n;
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
    const int64_t have = ch.n == 0 ? b->e.pass_n : b->e.segs[(size_t)ch.seg].last_n;
    const int64_t n = have < cap ? have : cap;
    if (n <= 0) return 0;
    const void* src = ch.n == 0 ? b->e.d_in.p : b->e.segs[(size_t)ch.seg].res.p;
    hipError_t e = hipMemcpyAsync(out_iq, src, (size_t)n * 8, hipMemcpyDeviceToHost, b->e.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->e.stream);
    if (e != hipSuccess) return hip_fail(e, "sdrx_chan24_bank_read", __FILE__, __LINE__);
    return n;
}

} // extern "C"
