// libsdrx.so: error plumbing, device queries, SampleSinkFifo mirror.
#include "sdrx_common.hpp"
#include <mutex>
#include <vector>
#include <cstring>

namespace sdrx {

static thread_local std::string g_last_error;

void set_error(const std::string& s) { g_last_error = s; }

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
    (void)hipGetLastError();   // clear sticky error
    return e == hipErrorOutOfMemory ? SDRX_ENOMEM : SDRX_EHIP;
}

int check_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available: libsdrx has no CPU fallback");
        return SDRX_ENODEV;
    }
    if (device < 0 || device >= n) {
        set_error("device index out of range");
        return SDRX_ENODEV;
    }
    return SDRX_OK;
}

int device_cu_count(int device)
{
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cu <= 0) {
        (void)hipGetLastError();
        cu = 256;
    }
    return cu;
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return SDRX_OK;
    size_t want = cap ? cap : 4096;
    while (want < bytes) want *= 2;
    void* np = nullptr;
    SDRX_HIP(hipMalloc(&np, want));
    if (p) (void)hipFree(p);
    p = np; cap = want;
    return SDRX_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
}

int EventTimer::begin(hipStream_t s)
{
    if (!enabled) return SDRX_OK;
    if (used + 2 > ev.size()) {
        hipEvent_t a, b;
        SDRX_HIP(hipEventCreate(&a)); SDRX_HIP(hipEventCreate(&b));
        ev.push_back(a); ev.push_back(b);
    }
    SDRX_HIP(hipEventRecord(ev[used], s));
    return SDRX_OK;
}

int EventTimer::end(hipStream_t s)
{
    if (!enabled) return SDRX_OK;
    SDRX_HIP(hipEventRecord(ev[used + 1], s));
    used += 2;
    return SDRX_OK;
}

int EventTimer::collect(hipStream_t s)
{
    SDRX_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i + 1 < used; i += 2) {
        float ms = 0;
        SDRX_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        total_ms += ms; count++;
    }
    used = 0;
    return SDRX_OK;
}

void EventTimer::release()
{
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear(); used = 0;
}

} // namespace sdrx

extern "C" {

const char* sdrx_version(void) { return "sdrx 0.2 (gfx950)"; }
const char* sdrx_last_error(void) { return sdrx::g_last_error.c_str(); }

int sdrx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

/* ------------------------------------------------------------------ .sdriq header (filerecord.cpp:129-148) */
int sdrx_sdriq_parse_header(const uint8_t* bytes, uint64_t n_bytes, sdrx_sdriq_header* out)
{
    if (!bytes || !out || n_bytes < SDRX_SDRIQ_HEADER_BYTES) { sdrx::set_error("sdrx_sdriq_parse_header: need 24 bytes"); return SDRX_EINVAL; }
    std::memcpy(&out->sample_rate, bytes, 4);
    std::memcpy(&out->center_frequency, bytes + 4, 8);
    std::memcpy(&out->start_timestamp, bytes + 12, 8);
    std::memcpy(&out->sample_size, bytes + 20, 4);
    if (out->sample_size != 16 && out->sample_size != 24) out->sample_size = 16;   // "assume 16 bits if garbage (old I/Q file)"
    return SDRX_OK;
}

int sdrx_sdriq_write_header(uint8_t* bytes24, const sdrx_sdriq_header* hdr)
{
    if (!bytes24 || !hdr) return SDRX_EINVAL;
    std::memcpy(bytes24, &hdr->sample_rate, 4);
    std::memcpy(bytes24 + 4, &hdr->center_frequency, 8);
    std::memcpy(bytes24 + 12, &hdr->start_timestamp, 8);
    std::memcpy(bytes24 + 20, &hdr->sample_size, 4);
    return SDRX_OK;
}

/* ------------------------------------------------------------------ measured HBM read ceiling
 * the roofline denominator is quoted twice -- the 8 TB/s datasheet figure and what a read-only
 * streaming kernel (sum of int32) reaches on this very box.  Diagnostic only; not on the sample path.    */
namespace {
__global__ void __launch_bounds__(256) sdrx_stream_sum_kernel(const uint4* __restrict__ p, size_t n_vec, uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {       // four independent 16-byte loads in flight per lane
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < n_vec; i += stride) { const uint4 a = p[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 0x9e3779b9u) atomicAdd(sink, acc);            // keeps the loads alive, practically never taken
}
}

int sdrx_measure_hbm_read(int device, uint64_t n_bytes, int32_t reps, double* gb_per_s)
{
    if (!gb_per_s || n_bytes < (1u << 20) || reps < 1) { sdrx::set_error("sdrx_measure_hbm_read: bad argument"); return SDRX_EINVAL; }
    int rc = sdrx::check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    void* buf = nullptr; uint32_t* sink = nullptr;
    SDRX_HIP(hipMalloc(&buf, n_bytes));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&sink), 4);
    if (e != hipSuccess) { (void)hipFree(buf); return sdrx::hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double best_ms = 1e30;
    e = hipMemset(buf, 1, n_bytes);
    if (e == hipSuccess) e = hipMemset(sink, 0, 4);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const size_t n_vec = n_bytes / 16;
    for (unsigned blocks_per_cu : { 4u, 8u, 16u, 32u }) {     // the best grid shape counts: this is a ceiling, not a product kernel
        for (int r = 0; r < reps + 1 && e == hipSuccess; r++) {  // first launch of a shape is a warm-up
            e = hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(sdrx_stream_sum_kernel, dim3(256 * blocks_per_cu), dim3(256), 0, nullptr, static_cast<const uint4*>(buf), n_vec, sink);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e == hipSuccess && r > 0 && ms < best_ms) best_ms = ms;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(sink);
    if (e != hipSuccess) return sdrx::hip_fail(e, "sdrx_measure_hbm_read", __FILE__, __LINE__);
    *gb_per_s = (double)(n_vec * 16) / (best_ms * 1e-3) / 1e9;
    return SDRX_OK;
}

/* ------------------------------------------------------------------ SampleSinkFifo mirror
 * Same observable contract as sdrbase/dsp/samplesinkfifo.cpp:70-231: a writer may add at most
 * size-fill samples (the rest is dropped and counted), readers see up to two contiguous spans,
 * commit advances the head.  The Qt signal dataReady() becomes a callback fired after every
 * write that leaves the FIFO non-empty (samplesinkfifo.cpp:107-108).                           */
struct sdrx_fifo {
    std::mutex mtx;
    std::vector<uint32_t> data;      // one packed Sample per element
    uint32_t size = 0, fill = 0, head = 0, tail = 0;
    uint64_t dropped = 0;
    sdrx_fifo_data_ready_cb cb = nullptr;
    void* user = nullptr;
};

int sdrx_fifo_create(sdrx_fifo_t** out, uint32_t size_samples)
{
    if (!out) { sdrx::set_error("sdrx_fifo_create: null out"); return SDRX_EINVAL; }
    sdrx_fifo* f = new (std::nothrow) sdrx_fifo;
    if (!f) return SDRX_ENOMEM;
    try { f->data.resize(size_samples); } catch (...) { delete f; sdrx::set_error("SampleSinkFifo: out of memory"); return SDRX_ENOMEM; }
    f->size = size_samples;
    *out = f;
    return SDRX_OK;
}

int sdrx_fifo_destroy(sdrx_fifo_t* f) { delete f; return SDRX_OK; }

int sdrx_fifo_set_size(sdrx_fifo_t* f, uint32_t size_samples)
{
    if (!f) return SDRX_EINVAL;
    std::lock_guard<std::mutex> g(f->mtx);
    f->size = f->fill = f->head = f->tail = 0;           // create(): everything restarts empty
    try { f->data.resize(size_samples); } catch (...) { sdrx::set_error("SampleSinkFifo: out of memory"); return SDRX_ENOMEM; }
    f->size = size_samples;
    return SDRX_OK;
}

uint32_t sdrx_fifo_size(sdrx_fifo_t* f) { std::lock_guard<std::mutex> g(f->mtx); return f->size; }
uint32_t sdrx_fifo_fill(sdrx_fifo_t* f) { std::lock_guard<std::mutex> g(f->mtx); return f->fill; }
uint64_t sdrx_fifo_dropped(sdrx_fifo_t* f) { std::lock_guard<std::mutex> g(f->mtx); return f->dropped; }

void sdrx_fifo_on_data_ready(sdrx_fifo_t* f, sdrx_fifo_data_ready_cb cb, void* user)
{
    std::lock_guard<std::mutex> g(f->mtx);
    f->cb = cb; f->user = user;
}

uint32_t sdrx_fifo_write(sdrx_fifo_t* f, const int16_t* iq, uint32_t count)
{
    sdrx_fifo_data_ready_cb cb = nullptr; void* user = nullptr;
    uint32_t total;
    {
        std::lock_guard<std::mutex> g(f->mtx);
        const uint32_t room = f->size - f->fill;
        total = count < room ? count : room;
        f->dropped += count - total;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(iq);
        uint32_t left = total;
        while (left > 0) {
            uint32_t run = f->size - f->tail; if (run > left) run = left;
            std::memcpy(&f->data[f->tail], src, (size_t)run * 4);
            f->tail = (f->tail + run) % f->size;
            f->fill += run; src += run; left -= run;
        }
        if (f->fill > 0) { cb = f->cb; user = f->user; }
    }
    if (cb) cb(user);                                     // outside the lock, like a queued signal
    return total;
}

uint32_t sdrx_fifo_write_bytes(sdrx_fifo_t* f, const uint8_t* data, uint32_t count_bytes)
{
    return sdrx_fifo_write(f, reinterpret_cast<const int16_t*>(data), count_bytes / 4);   // count /= sizeof(Sample)
}

uint32_t sdrx_fifo_read(sdrx_fifo_t* f, int16_t* iq, uint32_t count)
{
    std::lock_guard<std::mutex> g(f->mtx);
    const uint32_t total = count < f->fill ? count : f->fill;
    uint32_t* dst = reinterpret_cast<uint32_t*>(iq);
    uint32_t left = total;
    while (left > 0) {
        uint32_t run = f->size - f->head; if (run > left) run = left;
        std::memcpy(dst, &f->data[f->head], (size_t)run * 4);
        f->head = (f->head + run) % f->size;
        f->fill -= run; dst += run; left -= run;
    }
    return total;
}

uint32_t sdrx_fifo_read_begin(sdrx_fifo_t* f, uint32_t count, const int16_t** part1, uint32_t* n1,
                              const int16_t** part2, uint32_t* n2)
{
    std::lock_guard<std::mutex> g(f->mtx);
    const uint32_t total = count < f->fill ? count : f->fill;
    *part1 = *part2 = nullptr; *n1 = *n2 = 0;
    if (total > 0) {
        uint32_t run = f->size - f->head; if (run > total) run = total;
        *part1 = reinterpret_cast<const int16_t*>(&f->data[f->head]); *n1 = run;
        if (total > run) { *part2 = reinterpret_cast<const int16_t*>(&f->data[0]); *n2 = total - run; }
    }
    return total;
}

uint32_t sdrx_fifo_read_commit(sdrx_fifo_t* f, uint32_t count)
{
    std::lock_guard<std::mutex> g(f->mtx);
    if (count > f->fill) count = f->fill;                 // "cannot commit more than available samples"
    if (f->size) f->head = (f->head + count) % f->size;
    f->fill -= count;
    return count;
}

} // extern "C"
