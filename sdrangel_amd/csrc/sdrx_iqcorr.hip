// libsdrx.so: sdrx_dccorr_* -- the DC offset correction DSPDeviceSourceEngine::work applies to every FIFO span before the
// sinks see it (dspdevicesourceengine.cpp:339-343,375-379 -> iqCorrections(begin, end, false), :175-181,255-259):
//     m_iBeta(re); m_qBeta(im);  re -= (int32) m_iBeta;  im -= (int32) m_qBeta;
// with m_iBeta / m_qBeta = MovingAverageUtil<int32_t, int64_t, 1024> (util/movingaverage.h): total of the last 1024
// samples (fewer while filling up) / 1024, C++ truncating division.  Integer sums are associative, so the sliding total
// comes from a prefix sum instead of the reference's sample-by-sample walk, bit for bit:
//     avg[n] = trunc(sum(x[n-1023 .. n]) / 1024)  (x = 0 before the stream starts),   y[n] = (int16)(x[n] - avg[n]).
// sdrx_dccorr_* is the m_dcOffsetCorrection-only case.  The I/Q imbalance branch (:217-253, IMBALANCE_INT undefined) is a
// chain of float/double moving averages with a division and a square root per sample: serial and not associative, so it
// cannot be spread along time bit-exactly.  sdrx_iqimb_* (end of this file) offers it the only exact way: one lane per
// device stream walks its stream sample by sample (the reference's statement order, strict IEEE), many streams side by side.
#include "sdrx_common.hpp"
#include <new>
#include <vector>

using namespace sdrx;

namespace {

constexpr int DC_WIN = 1024;            // MovingAverageUtil<.., 1024>
constexpr int DC_TILE = 3072;           // outputs per workgroup
constexpr int DC_EXT = DC_TILE + DC_WIN;// samples a workgroup scans: position 0 <-> output 0 minus 1024
constexpr int DC_NT = 256;
constexpr int DC_PER = DC_EXT / DC_NT;  // 16 consecutive samples per lane

__device__ __forceinline__ int trunc_div_1024(int v) { return (v + ((v >> 31) & 1023)) >> 10; }

// hist: the 1024 samples in front of this call's first one (only the last 1023 matter), packed Samples
__global__ __launch_bounds__(DC_NT)
void dccorr_kernel(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, long n)
{
    __shared__ int pI[DC_EXT + 1], pQ[DC_EXT + 1];          // exclusive prefix sums: p[i] = sum of ext samples [0, i)
    __shared__ int wI[DC_NT], wQ[DC_NT];
    const int tid = threadIdx.x;
    const long t0 = (long)blockIdx.x * DC_TILE;             // first output of this tile; ext sample e <-> absolute t0 - 1024 + e
    int sI[DC_PER], sQ[DC_PER];
    int accI = 0, accQ = 0;
#pragma unroll
    for (int j = 0; j < DC_PER; j++) {
        const long p = t0 - DC_WIN + tid * DC_PER + j;
        const uint32_t v = p < 0 ? hist[p + DC_WIN] : (p < n ? in[p] : 0u);
        sI[j] = (int)(int16_t)(v & 0xffffu); sQ[j] = (int)(int16_t)(v >> 16);
        accI += sI[j]; accQ += sQ[j];
    }
    wI[tid] = accI; wQ[tid] = accQ;
    __syncthreads();
    // exclusive scan of the 256 lane totals (|sum| < 4096 * 32768 = 2^27: int32 is exact)
    for (int d = 1; d < DC_NT; d <<= 1) {
        const int a = tid >= d ? wI[tid - d] : 0, b = tid >= d ? wQ[tid - d] : 0;
        __syncthreads();
        wI[tid] += a; wQ[tid] += b;
        __syncthreads();
    }
    int runI = wI[tid] - accI, runQ = wQ[tid] - accQ;
#pragma unroll
    for (int j = 0; j < DC_PER; j++) {
        pI[tid * DC_PER + j] = runI; pQ[tid * DC_PER + j] = runQ;
        runI += sI[j]; runQ += sQ[j];
    }
    if (tid == DC_NT - 1) { pI[DC_EXT] = runI; pQ[DC_EXT] = runQ; }
    __syncthreads();
    for (int k = tid; k < DC_TILE; k += DC_NT) {
        const long g = t0 + k;
        if (g >= n) break;
        const int e = k + DC_WIN;                            // ext index of output k; window = ext samples [e - 1023, e]
        const uint32_t v = in[g];
        const int re = (int)(int16_t)(v & 0xffffu), im = (int)(int16_t)(v >> 16);
        const int yI = re - trunc_div_1024(pI[e + 1] - pI[e - 1023]);
        const int yQ = im - trunc_div_1024(pQ[e + 1] - pQ[e - 1023]);
        out[g] = ((uint32_t)yI & 0xffffu) | ((uint32_t)yQ << 16);
    }
}

// new history = last 1024 samples of (old history ++ input)
__global__ void dccorr_hist_kernel(const uint32_t* __restrict__ old_hist, const uint32_t* __restrict__ in, uint32_t* __restrict__ new_hist, long n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DC_WIN) return;
    const long src = (long)i + n - DC_WIN;
    new_hist[i] = src >= 0 ? in[src] : old_hist[i + n];
}

} // namespace

struct sdrx_dccorr {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint32_t* d_hist[2] = { nullptr, nullptr };
    int cur = 0;
    DevBuf d_in, d_out;
};

static int launch(sdrx_dccorr* h, const void* d_in, void* d_out, long n)
{
    if (n <= 0) return SDRX_OK;
    const long tiles = (n + DC_TILE - 1) / DC_TILE;
    hipLaunchKernelGGL(dccorr_kernel, dim3((unsigned)tiles), dim3(DC_NT), 0, h->stream,
                       h->d_hist[h->cur], static_cast<const uint32_t*>(d_in), static_cast<uint32_t*>(d_out), n);
    SDRX_HIP(hipGetLastError());
    hipLaunchKernelGGL(dccorr_hist_kernel, dim3(DC_WIN / 256), dim3(256), 0, h->stream,
                       h->d_hist[h->cur], static_cast<const uint32_t*>(d_in), h->d_hist[h->cur ^ 1], n);
    SDRX_HIP(hipGetLastError());
    h->cur ^= 1;
    return SDRX_OK;
}

extern "C" {

int sdrx_dccorr_create(sdrx_dccorr_t** out, int device)
{
    if (!out) { set_error("sdrx_dccorr_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_dccorr* h = new (std::nothrow) sdrx_dccorr;
    if (!h) return SDRX_ENOMEM;
    h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    h->stream = h->own_stream;
    for (int i = 0; i < 2; i++) {
        e = hipMalloc(reinterpret_cast<void**>(&h->d_hist[i]), DC_WIN * 4);
        if (e != hipSuccess) { sdrx_dccorr_destroy(h); return hip_fail(e, "hipMalloc(hist)", __FILE__, __LINE__); }
    }
    *out = h;
    return sdrx_dccorr_reset(h);
}

int sdrx_dccorr_destroy(sdrx_dccorr_t* h)
{
    if (!h) return SDRX_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    for (int i = 0; i < 2; i++) if (h->d_hist[i]) (void)hipFree(h->d_hist[i]);
    h->d_in.release(); h->d_out.release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SDRX_OK;
}

int sdrx_dccorr_reset(sdrx_dccorr_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemsetAsync(h->d_hist[h->cur], 0, DC_WIN * 4, h->stream));     // == freshly constructed MovingAverageUtil members
    return SDRX_OK;
}

int sdrx_dccorr_set_stream(sdrx_dccorr_t* h, void* hip_stream)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return SDRX_OK;
}

int sdrx_dccorr_sync(sdrx_dccorr_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_dccorr_process_dev(sdrx_dccorr_t* h, const int16_t* d_iq, int16_t* d_out_iq, int64_t n_cplx)
{
    if (!h || n_cplx < 0 || (n_cplx > 0 && (!d_iq || !d_out_iq))) { set_error("sdrx_dccorr_process_dev: bad argument"); return SDRX_EINVAL; }
    if (d_iq == d_out_iq) { set_error("sdrx_dccorr_process_dev: not in place (a tile reads its neighbour's input)"); return SDRX_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(d_iq) & 3u) || (reinterpret_cast<uintptr_t>(d_out_iq) & 3u)) { set_error("sdrx_dccorr_process_dev: 4-byte alignment"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    return launch(h, d_iq, d_out_iq, (long)n_cplx);
}

int sdrx_dccorr_process(sdrx_dccorr_t* h, int16_t* iq, int64_t n_cplx)
{
    if (!h || n_cplx < 0 || (n_cplx > 0 && !iq)) { set_error("sdrx_dccorr_process: bad argument"); return SDRX_EINVAL; }
    if (n_cplx == 0) return SDRX_OK;
    SDRX_HIP(hipSetDevice(h->device));
    int rc = h->d_in.reserve((size_t)n_cplx * 4); if (rc) return rc;
    rc = h->d_out.reserve((size_t)n_cplx * 4); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(h->d_in.p, iq, (size_t)n_cplx * 4, hipMemcpyHostToDevice, h->stream));
    rc = launch(h, h->d_in.p, h->d_out.p, (long)n_cplx); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(iq, h->d_out.p, (size_t)n_cplx * 4, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

} // extern "C"


/* =====================================================================================================================
 * sdrx_iqimb_* -- DSPDeviceSourceEngine::iqCorrections(begin, end, imbalanceCorrection = true), float flavour
 * (dspdevicesourceengine.cpp:175-181, 217-253; members dspdevicesourceengine.h:106-107, 120-125).
 * One workgroup per device stream: lane 0 runs the recurrence (its eight moving-average rings live in LDS), the other
 * lanes move the samples.  ~MS/s per stream; the parallelism is across streams (many FileSource device sets).
 * ===================================================================================================================== */
namespace {

struct IqImbState {                       // == the eight MovingAverageUtil members, freshly constructed when zero
    int32_t bI[1024], bQ[1024];           // m_iBeta / m_qBeta rings
    float   sII[128], sIQ[128], sII2[128], sQQ2[128];
    double  sPhi[128], sAmp[128];
    long long totI, totQ;
    double  tII, tIQ, tII2, tQQ2, tPhi, tAmp;
    int nB, idxB, nP, idxP, nPhi, idxPhi, nAmp, idxAmp;
};
static_assert(sizeof(IqImbState) % 8 == 0, "state is copied as dwords");

struct IqImbJob { const uint32_t* in; uint32_t* out; long n; };

__global__ __launch_bounds__(64)
void iqimb_kernel(IqImbState* __restrict__ states, const IqImbJob* __restrict__ jobs)
{
    __shared__ __attribute__((aligned(8))) IqImbState st;
    __shared__ uint32_t io[64];
    const int lane = threadIdx.x;
    IqImbState* gs = states + blockIdx.x;
    const IqImbJob jb = jobs[blockIdx.x];
    for (int i = lane; i < (int)(sizeof(IqImbState) / 4); i += 64) reinterpret_cast<uint32_t*>(&st)[i] = reinterpret_cast<const uint32_t*>(gs)[i];
    __syncthreads();
    for (long base = 0; base < jb.n; base += 64) {
        const long m = jb.n - base < 64 ? jb.n - base : 64;
        if (lane < m) io[lane] = jb.in[base + lane];
        __syncthreads();
        if (lane == 0) {
            for (int k = 0; k < (int)m; k++) {
                const uint32_t v = io[k];
                const int re = (int)(int16_t)(v & 0xffffu), im = (int)(int16_t)(v >> 16);
                // m_iBeta(re); m_qBeta(im)   (MovingAverageUtil<int32_t, int64_t, 1024>::operator(), movingaverage.h:42-56)
                if (st.nB < 1024) { st.bI[st.nB] = re; st.bQ[st.nB] = im; st.nB++; st.totI += re; st.totQ += im; }
                else {
                    st.totI += re - st.bI[st.idxB]; st.totQ += im - st.bQ[st.idxB];
                    st.bI[st.idxB] = re; st.bQ[st.idxB] = im; st.idxB = (st.idxB + 1) & 1023;
                }
                const float xi = (float)(re - (int)(st.totI / 1024)) / 32768.0f;
                const float xq = (float)(im - (int)(st.totQ / 1024)) / 32768.0f;
                const float pII = xi * xi, pIQ = xi * xq;
                const bool fillP = st.nP < 128;
                const int ip = fillP ? st.nP : st.idxP;
                if (fillP) { st.tII += (double)pII; st.tIQ += (double)pIQ; }
                else { st.tII += (double)(pII - st.sII[ip]); st.tIQ += (double)(pIQ - st.sIQ[ip]); }
                st.sII[ip] = pII; st.sIQ[ip] = pIQ;
                if (st.tII / 128.0 != 0.0) {
                    const double phi = (st.tIQ / 128.0) / (st.tII / 128.0);
                    if (st.nPhi < 128) { st.sPhi[st.nPhi++] = phi; st.tPhi += phi; }
                    else { st.tPhi += phi - st.sPhi[st.idxPhi]; st.sPhi[st.idxPhi] = phi; st.idxPhi = (st.idxPhi + 1) & 127; }
                }
                const float yq = (float)((double)xq - (st.tPhi / 128.0) * (double)xi);
                const float pII2 = xi * xi, pQQ2 = yq * yq;
                if (fillP) { st.tII2 += (double)pII2; st.tQQ2 += (double)pQQ2; st.nP++; }
                else { st.tII2 += (double)(pII2 - st.sII2[ip]); st.tQQ2 += (double)(pQQ2 - st.sQQ2[ip]); st.idxP = (st.idxP + 1) & 127; }
                st.sII2[ip] = pII2; st.sQQ2[ip] = pQQ2;
                if (st.tQQ2 / 128.0 != 0.0) {
                    const double a = __builtin_sqrt((st.tII2 / 128.0) / (st.tQQ2 / 128.0));
                    if (st.nAmp < 128) { st.sAmp[st.nAmp++] = a; st.tAmp += a; }
                    else { st.tAmp += a - st.sAmp[st.idxAmp]; st.sAmp[st.idxAmp] = a; st.idxAmp = (st.idxAmp + 1) & 127; }
                }
                const float zq = (float)((st.tAmp / 128.0) * (double)yq);
                const int yr = (int)(xi * 32768.0f), yi2 = (int)(zq * 32768.0f);        // float -> int (truncation), then the low 16 bits
                io[k] = ((uint32_t)yr & 0xffffu) | ((uint32_t)yi2 << 16);
            }
        }
        __syncthreads();
        if (lane < m) jb.out[base + lane] = io[lane];
        __syncthreads();
    }
    for (int i = lane; i < (int)(sizeof(IqImbState) / 4); i += 64) reinterpret_cast<uint32_t*>(gs)[i] = reinterpret_cast<const uint32_t*>(&st)[i];
}

} // namespace

struct sdrx_iqimb {
    int device = 0, n_streams = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    IqImbState* d_state = nullptr;
    IqImbJob* d_jobs = nullptr;
    IqImbJob* h_jobs = nullptr;               // pinned
    hipEvent_t jobs_ev = nullptr;
    std::vector<DevBuf> stage;
};

extern "C" {

int sdrx_iqimb_create(sdrx_iqimb_t** out, int device, int32_t n_streams)
{
    if (!out || n_streams <= 0) { set_error("sdrx_iqimb_create: bad argument"); return SDRX_EINVAL; }
    *out = nullptr;
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_iqimb* h = new (std::nothrow) sdrx_iqimb;
    if (!h) return SDRX_ENOMEM;
    h->device = device; h->n_streams = n_streams; h->stage.resize((size_t)n_streams);
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) { h->stream = h->own_stream; e = hipMalloc(reinterpret_cast<void**>(&h->d_state), sizeof(IqImbState) * (size_t)n_streams); }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_jobs), sizeof(IqImbJob) * (size_t)n_streams);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->h_jobs), sizeof(IqImbJob) * (size_t)n_streams, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->jobs_ev, hipEventDisableTiming);
    if (e != hipSuccess) { sdrx_iqimb_destroy(h); return hip_fail(e, "sdrx_iqimb_create", __FILE__, __LINE__); }
    *out = h;
    return sdrx_iqimb_reset(h);
}

int sdrx_iqimb_destroy(sdrx_iqimb_t* h)
{
    if (!h) return SDRX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_jobs) (void)hipFree(h->d_jobs);
    if (h->h_jobs) (void)hipHostFree(h->h_jobs);
    if (h->jobs_ev) (void)hipEventDestroy(h->jobs_ev);
    for (auto& b : h->stage) b.release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SDRX_OK;
}

int sdrx_iqimb_reset(sdrx_iqimb_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipMemsetAsync(h->d_state, 0, sizeof(IqImbState) * (size_t)h->n_streams, h->stream));   // == freshly constructed members
    return SDRX_OK;
}

int sdrx_iqimb_set_stream(sdrx_iqimb_t* h, void* hip_stream)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return SDRX_OK;
}

int sdrx_iqimb_sync(sdrx_iqimb_t* h)
{
    if (!h) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

int sdrx_iqimb_process_dev(sdrx_iqimb_t* h, const int16_t* const* d_iq, int16_t* const* d_out_iq, const int64_t* n_cplx)
{
    if (!h || !d_iq || !d_out_iq || !n_cplx) { set_error("sdrx_iqimb_process_dev: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    SDRX_HIP(hipEventSynchronize(h->jobs_ev));                 // the previous call's table has been uploaded
    for (int s = 0; s < h->n_streams; s++) {
        if (n_cplx[s] < 0 || (n_cplx[s] > 0 && (!d_iq[s] || !d_out_iq[s])) || (reinterpret_cast<uintptr_t>(d_iq[s]) & 3u) || (reinterpret_cast<uintptr_t>(d_out_iq[s]) & 3u)) {
            set_error("sdrx_iqimb_process_dev: bad stream argument (4-byte aligned device pointers)"); return SDRX_EINVAL;
        }
        h->h_jobs[s] = IqImbJob{ reinterpret_cast<const uint32_t*>(d_iq[s]), reinterpret_cast<uint32_t*>(d_out_iq[s]), (long)n_cplx[s] };
    }
    SDRX_HIP(hipMemcpyAsync(h->d_jobs, h->h_jobs, sizeof(IqImbJob) * (size_t)h->n_streams, hipMemcpyHostToDevice, h->stream));
    SDRX_HIP(hipEventRecord(h->jobs_ev, h->stream));
    hipLaunchKernelGGL(iqimb_kernel, dim3((unsigned)h->n_streams), dim3(64), 0, h->stream, h->d_state, h->d_jobs);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

int sdrx_iqimb_process(sdrx_iqimb_t* h, int16_t* const* iq, const int64_t* n_cplx)
{
    if (!h || !iq || !n_cplx) { set_error("sdrx_iqimb_process: bad argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(h->device));
    std::vector<const int16_t*> din((size_t)h->n_streams); std::vector<int16_t*> dout((size_t)h->n_streams);
    for (int s = 0; s < h->n_streams; s++) {
        if (n_cplx[s] < 0 || (n_cplx[s] > 0 && !iq[s])) { set_error("sdrx_iqimb_process: bad stream argument"); return SDRX_EINVAL; }
        int rc = h->stage[(size_t)s].reserve((size_t)(n_cplx[s] > 0 ? n_cplx[s] : 1) * 4); if (rc) return rc;
        if (n_cplx[s]) SDRX_HIP(hipMemcpyAsync(h->stage[(size_t)s].p, iq[s], (size_t)n_cplx[s] * 4, hipMemcpyHostToDevice, h->stream));
        din[(size_t)s] = static_cast<const int16_t*>(h->stage[(size_t)s].p); dout[(size_t)s] = static_cast<int16_t*>(h->stage[(size_t)s].p);
    }
    int rc = sdrx_iqimb_process_dev(h, din.data(), dout.data(), n_cplx); if (rc) return rc;   // in place on the device: a lane reads a sample before it overwrites it
    for (int s = 0; s < h->n_streams; s++)
        if (n_cplx[s]) SDRX_HIP(hipMemcpyAsync(iq[s], h->stage[(size_t)s].p, (size_t)n_cplx[s] * 4, hipMemcpyDeviceToHost, h->stream));
    SDRX_HIP(hipStreamSynchronize(h->stream));
    return SDRX_OK;
}

} // extern "C"
