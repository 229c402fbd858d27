// host-side helpers shared by the C-ABI translation units of libsdrx.so
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstdio>
#include "../../include/sdrx.h"

namespace sdrx {

void set_error(const std::string& s);
int  hip_fail(hipError_t e, const char* what, const char* file, int line);
int  check_device(int device);           // SDRX_OK or SDRX_ENODEV (sets error)
int  device_cu_count(int device);

#define SDRX_HIP(call)                                                                 \
    do { hipError_t e_ = (call);                                                       \
         if (e_ != hipSuccess) return ::sdrx::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);           // SDRX_OK / SDRX_ENOMEM
    void release();
};

// pairs of HIP events bracketing kernel launches on a stream
struct EventTimer {
    bool enabled = false;
    std::vector<hipEvent_t> ev;          // start0, stop0, start1, stop1, ...
    size_t used = 0;
    double total_ms = 0; long count = 0;
    int begin(hipStream_t s);            // records a start event (no-op when disabled)
    int end(hipStream_t s);
    int collect(hipStream_t s);          // sync + fold the recorded pairs into total_ms / count
    void release();
};

} // namespace sdrx
