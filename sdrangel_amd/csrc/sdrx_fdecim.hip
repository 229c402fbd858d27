// libsdrx.so: sdrx_fdecim_* -- drop-in for the float half-band decimators DecimatorsFI (decimatorsfi.h, AirspyHF
// thread member), DecimatorsFF (decimatorsff.h) and DecimatorsIF<qint16,InputBits> (decimatorsif.h).
// Host logic + kernel dispatch; kernels in fdecim_kernel.hpp.
#include "sdrx_common.hpp"
#include "fdecim_kernel.hpp"
#include <cstring>
#include <cstdlib>
#include <new>
#include <algorithm>

using namespace sdrx;

namespace {

typedef void (*fd_chain_fn)(const float*, float*, const void*, void*, long, long, int, int, int, int, float);

template<int IN> fd_chain_fn chain_for(int ns)
{
    switch (ns) {
    case 1: return &fdecim_chain_kernel<1, IN>;
    case 2: return &fdecim_chain_kernel<2, IN>;
    case 3: return &fdecim_chain_kernel<3, IN>;
    case 4: return &fdecim_chain_kernel<4, IN>;
    case 5: return &fdecim_chain_kernel<5, IN>;
    case 6: return &fdecim_chain_kernel<6, IN>;
    }
    return nullptr;
}

// elements (floats or int16) consumed per loop iteration of decimateK_x (decimatorsfi.cpp:23-779 `pos +=` strides)
int fd_group(int log2, int fcpos)
{
    if (log2 == 0) return 2;
    if (log2 == 1) return fcpos == SDRX_FC_CEN ? 4 : 8;
    return 2 << log2;
}

int choose_cps(long n_chunks, int slots, int warm)
{
    long best = 1; double best_cost = 1e300;
    for (long cps = 1; cps <= n_chunks && cps <= 4096; cps = cps < 16 ? cps + 1 : cps + cps / 8) {
        const long segs = (n_chunks + cps - 1) / cps;
        const long rounds = (segs + slots - 1) / slots;
        const double cost = (double)(cps + warm) * (double)rounds;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = cps; }
    }
    return (int)best;
}

} // namespace

struct sdrx_fdecim {
    int device = 0, log2 = 0, fcpos = 2, in_kind = 0, out_kind = 0, bits = 16;
    int ns = 0;                   // half-band stages after the front end
    int fe = FD_FE_ID;            // front end
    int pre_per_group = 1;        // pre-samples (or, without a filter, outputs) per group
    int group = 2;                // input elements per group
    int in_elem = 4, out_cplx_bytes = 4;
    float scale = 1.0f;
    int cus = 256;
    fd_chain_fn chain = nullptr;
    // chains longer than 3 stages run as two passes (stages 1..ns1, then ns1+1..ns through a float intermediate in HBM):
    // every stage is the same myDecimate(), so decimate64 == decimate8 of decimate8, bit for bit, and in a 3-stage pass
    // all 256 lanes have work in every stage (4 / 2 / 1 outputs), where stages 4..6 of a single pass idle 1/2 .. 7/8 of them
    int ns1 = 0, ns2 = 0;
    fd_chain_fn chain2 = nullptr;
    DevBuf d_mid;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // the carried state: the six filters' rings as arm histories (FD_STATE floats per stage, cascade order), double-buffered:
    // a call's first workgroup reads one copy while its last one writes the other
    float* d_state[2] = { nullptr, nullptr };
    int cur = 0;
    DevBuf d_in, d_out;
    char last_name[96] = "";
    int last_grid = 0, last_block = 0, last_lds = 0;
    EventTimer timer;
};

static int launch(sdrx_fdecim* h, const void* d_in, long n_groups, void* d_out, long* n_out_p)
{
    const long n_pre = n_groups * h->pre_per_group;
    const long n_out = h->ns ? n_pre >> h->ns : n_pre;
    if (n_out_p) *n_out_p = n_out;
    if (n_groups <= 0) return SDRX_OK;
    int trc = h->timer.begin(h->stream); if (trc) return trc;
    if (h->ns == 0) {
        const int block = 256;
        long grid = (n_out + block - 1) / block; if (grid > 8192) grid = 8192;
        if (h->in_kind == 0) hipLaunchKernelGGL(fd_pointwise_kernel<0>, dim3((unsigned)grid), dim3(block), 0, h->stream, d_in, d_out, n_out, h->fe, h->out_kind, h->scale, h->log2 == 0);
        else                 hipLaunchKernelGGL(fd_pointwise_kernel<1>, dim3((unsigned)grid), dim3(block), 0, h->stream, d_in, d_out, n_out, h->fe, h->out_kind, h->scale, h->log2 == 0);
        SDRX_HIP(hipGetLastError());
        snprintf(h->last_name, sizeof h->last_name, "fd_pointwise_kernel<%d>", h->in_kind);
        h->last_grid = (int)grid; h->last_block = block; h->last_lds = 0;
        return h->timer.end(h->stream);
    }
    auto run_pass = [&](fd_chain_fn fn, int ns, int stage0, const void* in, void* out, long np, int fe, int out_kind, float scale, int* grid_out) -> int {
        const long n_chunks = (np + FD_CHUNK - 1) / FD_CHUNK;
        if (n_chunks > 0x7fffffffL / 4) { set_error("sdrx_fdecim: input too long for one call"); return SDRX_EINVAL; }
        const int warm = fd_warm_chunks(ns);
        const int lds = fd_lds_floats(ns) * 4;
        const int wg_per_cu = std::max(1, std::min(8, (160 * 1024) / std::max(lds, 1)));
        const int cps = std::max(choose_cps(n_chunks, h->cus * wg_per_cu, warm), warm);      // later segments warm up inside the call
        const long segs = (n_chunks + cps - 1) / cps;
        hipLaunchKernelGGL(fn, dim3((unsigned)segs), dim3(FD_THREADS), 0, h->stream, h->d_state[h->cur] + stage0 * FD_STATE, h->d_state[h->cur ^ 1] + stage0 * FD_STATE,
                           in, out, np, np >> ns, (int)n_chunks, cps, fe, out_kind, scale);
        SDRX_HIP(hipGetLastError());
        if (grid_out) *grid_out = (int)segs;
        return SDRX_OK;
    };
    int rc;
    if (h->ns2 == 0) {
        rc = run_pass(h->chain, h->ns, 0, d_in, d_out, n_pre, h->fe, h->out_kind, h->scale, &h->last_grid); if (rc) return rc;
        snprintf(h->last_name, sizeof h->last_name, "fdecim_chain_kernel<%d,%d>", h->ns, h->in_kind);
        h->last_lds = fd_lds_floats(h->ns) * 4;
    } else {
        const long n_mid = n_pre >> h->ns1;
        rc = h->d_mid.reserve((size_t)std::max<long>(n_mid, 1) * sizeof(float2)); if (rc) return rc;
        rc = run_pass(h->chain, h->ns1, 0, d_in, h->d_mid.p, n_pre, h->fe, 1, 1.0f, &h->last_grid); if (rc) return rc;
        rc = run_pass(h->chain2, h->ns2, h->ns1, h->d_mid.p, d_out, n_mid, FD_FE_ID, h->out_kind, h->scale, nullptr); if (rc) return rc;
        snprintf(h->last_name, sizeof h->last_name, "fdecim_chain_kernel<%d,%d> + <%d,0>", h->ns1, h->in_kind, h->ns2);
        h->last_lds = fd_lds_floats(h->ns1) * 4;
    }
    h->last_block = FD_THREADS;
    trc = h->timer.end(h->stream); if (trc) return trc;
    h->cur ^= 1;                                           // (a handle only ever reads the stages of its own cascade)
    return SDRX_OK;
}

extern "C" {

int32_t sdrx_fdecim_group(int log2_decim, int fcpos) { return fd_group(log2_decim, fcpos); }

int sdrx_fdecim_create(sdrx_fdecim_t** out, int device, int log2_decim, int fcpos, int in_kind, int out_kind, int input_bits)
{
    if (!out) { set_error("sdrx_fdecim_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    if (log2_decim < 0 || log2_decim > 6 || fcpos < 0 || fcpos > 2 || in_kind < 0 || in_kind > 1 || out_kind < 0 || out_kind > 1 ||
        (in_kind == 1 && input_bits != 8 && input_bits != 12 && input_bits != 16) || (in_kind == 1 && out_kind == 0)) {
        set_error("sdrx_fdecim_create: log2 0..6, fcpos 0..2, (float in, int16|float out) or (int16 in with input_bits 8|12|16, float out)");
        return SDRX_EINVAL;
    }
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_fdecim* h = new (std::nothrow) sdrx_fdecim;
    if (!h) return SDRX_ENOMEM;
    h->device = device; h->log2 = log2_decim; h->fcpos = fcpos; h->in_kind = in_kind; h->out_kind = out_kind; h->bits = input_bits;
    const int L = log2_decim;
    h->group = fd_group(L, fcpos);
    h->in_elem = in_kind == 0 ? 4 : 2;
    h->out_cplx_bytes = out_kind == 0 ? 4 : 8;
    if (fcpos == SDRX_FC_CEN || L == 0) { h->fe = FD_FE_ID; h->ns = L; h->pre_per_group = h->group / 2; }
    else if (L == 1) { h->fe = fcpos == SDRX_FC_INF ? FD_FE_INF2 : FD_FE_SUP2; h->ns = 0; h->pre_per_group = 2; }
    else {
        h->fe = fcpos == SDRX_FC_INF ? FD_FE_INF4 : (L <= 3 ? FD_FE_SUP4_A : FD_FE_SUP4_B);
        h->ns = L - 2; h->pre_per_group = h->group / 8;
    }
    // DecimatorsIF: decimation_scale<InputBits>::scaleIn (decimatorsif.cpp)
    h->scale = in_kind == 1 ? (input_bits == 8 ? (float)(1.0 / 128.0) : input_bits == 12 ? (float)(1.0 / 2048.0) : (float)(1.0 / 32768.0)) : 1.0f;
    h->cus = device_cu_count(device);
    { const char* sp = getenv("SDRX_FDECIM_SPLIT"); if (h->ns > 3 && !(sp && atoi(sp) == 0)) { h->ns1 = 3; h->ns2 = h->ns - 3; } else { h->ns1 = h->ns; h->ns2 = 0; } }
    if (h->ns) h->chain = in_kind == 0 ? chain_for<0>(h->ns1) : chain_for<1>(h->ns1);
    if (h->ns2) h->chain2 = chain_for<0>(h->ns2);
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    h->stream = h->own_stream;
    for (int i = 0; i < 2; i++) {
        e = hipMalloc(reinterpret_cast<void**>(&h->d_state[i]), (size_t)6 * FD_STATE * sizeof(float));
        if (e != hipSuccess) { sdrx_fdecim_destroy(h); return hip_fail(e, "hipMalloc(state)", __FILE__, __LINE__); }
    }
    *out = h;
    return sdrx_fdecim_reset(h);
}

int sdrx_fdecim_destroy(sdrx_fdecim_t* h)
{
    if (!h) return SDRX_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    for (int i = 0; i < 2; i++) if (h->d_state[i]) (void)hipFree(h->d_state[i]);
    h->d_in.release(); h->d_out.release(); h->d_mid.release(); h->timer.release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SDRX_OK;
}

int sdrx_fdecim_reset(sdrx_fdecim_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemsetAsync(h->d_state[h->cur], 0, (size_t)6 * FD_STATE * sizeof(float), h->stream));
    return SDRX_OK;
}

int sdrx_fdecim_set_stream(sdrx_fdecim_t* h, void* hip_stream)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return SDRX_OK;
}

int sdrx_fdecim_sync(sdrx_fdecim_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_fdecim_process_dev(sdrx_fdecim_t* h, const void* d_in, int64_t n_elems, void* d_out, int64_t* n_out_cplx)
{
    if (!h || n_elems < 0 || (n_elems > 0 && (!d_in || !d_out))) { set_error("sdrx_fdecim_process_dev: bad argument"); return SDRX_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(d_in) & 15u) || (reinterpret_cast<uintptr_t>(d_out) & 7u)) {
        set_error("sdrx_fdecim_process_dev: d_in must be 16-byte, d_out 8-byte aligned"); return SDRX_EINVAL;
    }
    SDRX_HIP(hipSetDevice(h->device));
    long n_out = 0;
    const int rc = launch(h, d_in, (long)(n_elems / h->group), d_out, &n_out);      // trailing partial group dropped
    if (n_out_cplx) *n_out_cplx = n_out;
    return rc;
}

int sdrx_fdecim_process(sdrx_fdecim_t* h, const void* in, int32_t n_elems, void* out, int32_t* n_out_cplx)
{
    if (!h || n_elems < 0 || (n_elems > 0 && (!in || !out))) { set_error("sdrx_fdecim_process: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    const int64_t groups = n_elems / h->group;
    const int64_t n_in_elems = groups * h->group;
    const int64_t n_pre = groups * h->pre_per_group;
    const int64_t n_out = h->ns ? n_pre >> h->ns : n_pre;
    if (n_out_cplx) *n_out_cplx = (int32_t)n_out;
    if (groups == 0) return SDRX_OK;
    int rc = h->d_in.reserve((size_t)n_in_elems * h->in_elem); if (rc) return rc;
    rc = h->d_out.reserve((size_t)n_out * h->out_cplx_bytes); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(h->d_in.p, in, (size_t)n_in_elems * h->in_elem, hipMemcpyHostToDevice, h->stream));
    rc = launch(h, h->d_in.p, (long)groups, h->d_out.p, nullptr); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(out, h->d_out.p, (size_t)n_out * h->out_cplx_bytes, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_fdecim_set_timing(sdrx_fdecim_t* h, int enabled)
{
    if (!h) return SDRX_EINVAL;
    h->timer.enabled = enabled != 0;
    return SDRX_OK;
}

int sdrx_fdecim_get_timing(sdrx_fdecim_t* h, double* total_ms, int64_t* launches, int reset)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    int rc = h->timer.collect(h->stream); if (rc) return rc;
    if (total_ms) *total_ms = h->timer.total_ms;
    if (launches) *launches = h->timer.count;
    if (reset) { h->timer.total_ms = 0; h->timer.count = 0; }
    return SDRX_OK;
}

int sdrx_fdecim_last_launch(const sdrx_fdecim_t* h, char* kernel_name, int name_cap, int* grid, int* block, int* lds_bytes)
{
    if (!h) return SDRX_EINVAL;
    if (kernel_name && name_cap > 0) snprintf(kernel_name, (size_t)name_cap, "%s", h->last_name);
    if (grid) *grid = h->last_grid;
    if (block) *block = h->last_block;
    if (lds_bytes) *lds_bytes = h->last_lds;
    return SDRX_OK;
}

} // extern "C"

extern "C" {

/* checkpoint of the carried state: the cascade's filter rings (6 x FD_STATE floats; stages beyond the cascade are zero) */
int64_t sdrx_fdecim_state_bytes(const sdrx_fdecim_t*) { return (int64_t)6 * FD_STATE * sizeof(float); }

int sdrx_fdecim_get_state(sdrx_fdecim_t* h, void* host_buf)
{
    if (!h || !host_buf) { set_error("sdrx_fdecim_get_state: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    std::memset(host_buf, 0, (size_t)6 * FD_STATE * sizeof(float));
    if (h->ns) SDRX_HIP(hipMemcpyAsync(host_buf, h->d_state[h->cur], (size_t)h->ns * FD_STATE * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_fdecim_set_state(sdrx_fdecim_t* h, const void* host_buf)
{
    if (!h || !host_buf) { set_error("sdrx_fdecim_set_state: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemcpyAsync(h->d_state[h->cur], host_buf, (size_t)6 * FD_STATE * sizeof(float), hipMemcpyHostToDevice, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

} // extern "C"

/* ---- one DecimatorsFI / FF / IF object, several decimateK_x: the shared six filters -------------------------------- */
struct sdrx_fdecim_stages { int device = 0; float* d_state = nullptr; };

extern "C" {

int sdrx_fdecim_stages_create(sdrx_fdecim_stages_t** out, int device)
{
    if (!out) { set_error("sdrx_fdecim_stages_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_fdecim_stages* s = new (std::nothrow) sdrx_fdecim_stages;
    if (!s) return SDRX_ENOMEM;
    s->device = device;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_state), (size_t)6 * FD_STATE * sizeof(float));
    if (e == hipSuccess) e = hipMemset(s->d_state, 0, (size_t)6 * FD_STATE * sizeof(float));
    if (e == hipSuccess) e = hipDeviceSynchronize();       // the null-stream memset is not ordered against the handles' non-blocking streams
    if (e != hipSuccess) { if (s->d_state) (void)hipFree(s->d_state); delete s; return hip_fail(e, "sdrx_fdecim_stages_create", __FILE__, __LINE__); }
    *out = s;
    return SDRX_OK;
}

int sdrx_fdecim_stages_destroy(sdrx_fdecim_stages_t* s)
{
    if (!s) return SDRX_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->d_state) (void)hipFree(s->d_state);
    delete s;
    return SDRX_OK;
}

int sdrx_fdecim_save_stages(sdrx_fdecim_t* h, sdrx_fdecim_stages_t* s)
{
    if (!h || !s || h->device != s->device) { set_error("sdrx_fdecim_save_stages: bad argument (handle and stage set must live on one device)"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    if (h->ns) SDRX_HIP(hipMemcpyAsync(s->d_state, h->d_state[h->cur], (size_t)h->ns * FD_STATE * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_fdecim_load_stages(sdrx_fdecim_t* h, const sdrx_fdecim_stages_t* s)
{
    if (!h || !s || h->device != s->device) { set_error("sdrx_fdecim_load_stages: bad argument (handle and stage set must live on one device)"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemcpyAsync(h->d_state[h->cur], s->d_state, (size_t)6 * FD_STATE * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return SDRX_OK;
}

} // extern "C"
