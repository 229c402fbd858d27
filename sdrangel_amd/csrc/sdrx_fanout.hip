// libsdrx.so: sdrx_fanout_* -- one source stream staged on one GPU handed to several GPUs by peer copies (xGMI on an
// 8 x MI355X node).  SURVEY 8e / north_star: "one stream per GPU ... xGMI only for fan-out (no collective on the per-sample
// path)".  This is the optional staging piece: a FileSource stream that was uploaded once can feed the channelizer banks of
// several GPUs (e.g. 8 x 128 channels of the same stream) without crossing PCIe again.  Point-to-point hipMemcpyPeerAsync, one
// stream and one event per destination; nothing here reduces or gathers.
#include "sdrx_common.hpp"
#include <new>
#include <vector>

using namespace sdrx;

struct sdrx_fanout {
    int src_device = 0;
    int64_t cap = 0;
    struct Dst { int device = 0; void* buf = nullptr; hipStream_t stream = nullptr; hipEvent_t done = nullptr; };
    std::vector<Dst> dst;
    hipEvent_t ready = nullptr;                 // recorded on the producer's stream (source device)
};

extern "C" {

int sdrx_fanout_destroy(sdrx_fanout_t* f)
{
    if (!f) return SDRX_OK;
    for (auto& d : f->dst) {
        (void)hipSetDevice(d.device);
        if (d.stream) { (void)hipStreamSynchronize(d.stream); (void)hipStreamDestroy(d.stream); }
        if (d.done) (void)hipEventDestroy(d.done);
        if (d.buf) (void)hipFree(d.buf);
    }
    (void)hipSetDevice(f->src_device);
    if (f->ready) (void)hipEventDestroy(f->ready);
    delete f;
    return SDRX_OK;
}

int sdrx_fanout_create(sdrx_fanout_t** out, int src_device, int32_t n_dst, const int32_t* dst_devices, int64_t max_bytes)
{
    if (!out || n_dst <= 0 || !dst_devices || max_bytes <= 0) { set_error("sdrx_fanout_create: bad argument"); return SDRX_EINVAL; }
    *out = nullptr;
    int rc = check_device(src_device); if (rc) return rc;
    for (int i = 0; i < n_dst; i++) { rc = check_device(dst_devices[i]); if (rc) return rc; }
    sdrx_fanout* f = new (std::nothrow) sdrx_fanout;
    if (!f) return SDRX_ENOMEM;
    f->src_device = src_device; f->cap = max_bytes;
    f->dst.resize((size_t)n_dst);
    hipError_t e = hipSetDevice(src_device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ready, hipEventDisableTiming);
    for (int i = 0; i < n_dst && e == hipSuccess; i++) {
        sdrx_fanout::Dst& d = f->dst[(size_t)i];
        d.device = dst_devices[i];
        if (d.device != src_device) {
            int can = 0;
            e = hipDeviceCanAccessPeer(&can, d.device, src_device);
            if (e == hipSuccess && can) {
                e = hipSetDevice(d.device);
                if (e == hipSuccess) { e = hipDeviceEnablePeerAccess(src_device, 0); if (e == hipErrorPeerAccessAlreadyEnabled) { e = hipSuccess; (void)hipGetLastError(); } }
            }                                                   // no peer access: hipMemcpyPeerAsync still works, staged by the runtime
        }
        if (e == hipSuccess) e = hipSetDevice(d.device);
        if (e == hipSuccess) e = hipMalloc(&d.buf, (size_t)max_bytes);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.done, hipEventDisableTiming);
    }
    if (e != hipSuccess) { const int r = hip_fail(e, "sdrx_fanout_create", __FILE__, __LINE__); sdrx_fanout_destroy(f); return r; }
    *out = f;
    return SDRX_OK;
}

int sdrx_fanout_send(sdrx_fanout_t* f, const void* d_src, int64_t bytes, void* producer_stream)
{
    if (!f || bytes < 0 || bytes > f->cap || (bytes > 0 && !d_src)) { set_error("sdrx_fanout_send: bad argument (more bytes than the buffers hold?)"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(f->src_device));
    // the copies start when what the producer queued so far (the upload, a decimator, ...) is done
    SDRX_HIP(hipEventRecord(f->ready, static_cast<hipStream_t>(producer_stream)));
    for (auto& d : f->dst) {
        SDRX_HIP(hipSetDevice(d.device));
        SDRX_HIP(hipStreamWaitEvent(d.stream, f->ready, 0));
        if (bytes) SDRX_HIP(hipMemcpyPeerAsync(d.buf, d.device, d_src, f->src_device, (size_t)bytes, d.stream));
        SDRX_HIP(hipEventRecord(d.done, d.stream));
    }
    SDRX_HIP(hipSetDevice(f->src_device));
    return SDRX_OK;
}

void* sdrx_fanout_buffer(sdrx_fanout_t* f, int32_t i)
{
    if (!f || i < 0 || i >= (int32_t)f->dst.size()) { set_error("sdrx_fanout_buffer: bad destination"); return nullptr; }
    return f->dst[(size_t)i].buf;
}

int sdrx_fanout_wait(sdrx_fanout_t* f, int32_t i)
{
    if (!f || i < 0 || i >= (int32_t)f->dst.size()) { set_error("sdrx_fanout_wait: bad destination"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(f->dst[(size_t)i].device));
    SDRX_HIP(hipEventSynchronize(f->dst[(size_t)i].done));
    return SDRX_OK;
}

int sdrx_fanout_stream_wait(sdrx_fanout_t* f, int32_t i, void* consumer_stream)
{
    if (!f || i < 0 || i >= (int32_t)f->dst.size() || !consumer_stream) { set_error("sdrx_fanout_stream_wait: bad argument (a real stream object is needed)"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(f->dst[(size_t)i].device));
    SDRX_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), f->dst[(size_t)i].done, 0));
    return SDRX_OK;
}

} // extern "C"
