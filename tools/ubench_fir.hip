// Stand-alone timing of the back-end's schedule + FIR kernels on a cfg-4 shaped bank (256 channels, 64 Mi-sample feed):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -o tools/ubench_fir tools/ubench_fir.hip
// Prints kernel times from HIP events; used to tune be_fir_kernel / be_schedule_kernel without the Python bench around them.
#include "../sdrangel_amd/csrc/backend_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace sdrx;

int main(int argc, char** argv)
{
    const int n_ch = argc > 1 ? atoi(argv[1]) : 256;
    const int base_n = argc > 2 ? atoi(argv[2]) : 65536;
    const int nt = 72;
    std::vector<BeChan> hc((size_t)n_ch); std::vector<BeBufs> hb((size_t)n_ch);
    std::vector<float> taps((size_t)16 * nt);
    for (auto& t : taps) t = (float)(rand() / (double)RAND_MAX) - 0.5f;
    float* d_taps; CK(hipMalloc((void**)&d_taps, taps.size() * 4)); CK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
    long n_max = 0;
    for (int c = 0; c < n_ch; c++) {
        const bool wide = (c % 8) == 7;                      // a few channels one stage shallower: twice the rate, step 2.5
        const long n_in = wide ? 2L * base_n : base_n;
        n_max = n_in > n_max ? n_in : n_max;
        BeChan& s = hc[(size_t)c]; memset(&s, 0, sizeof s);
        s.step = wide ? 2.5f : 1.25f; s.ntaps = nt; s.phase_steps = 16; s.taps_off = 0; s.filt_mode = 2; s.discri = 1; s.half = 512;
        BeBufs& b = hb[(size_t)c]; memset(&b, 0, sizeof b);
        std::vector<float> x((size_t)(BE_HIST + n_in) * 2);
        for (auto& v : x) v = (float)(rand() / (double)RAND_MAX) - 0.5f;
        CK(hipMalloc((void**)&b.mixed, x.size() * 4)); CK(hipMemcpy(b.mixed, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc((void**)&b.res, (size_t)(n_in + 2048) * 8));
        CK(hipMalloc((void**)&b.cplx_out, (size_t)(n_in + 2048) * 8));
        b.n_in = n_in; b.sched_stride = n_ch;
    }
    uint2* d_sched; CK(hipMalloc((void**)&d_sched, (size_t)(n_max + 1024) * n_ch * sizeof(uint2)));
    for (int c = 0; c < n_ch; c++) hb[(size_t)c].sched = d_sched + c;
    BeChan* d_ch; BeBufs* d_b;
    CK(hipMalloc((void**)&d_ch, hc.size() * sizeof(BeChan))); CK(hipMalloc((void**)&d_b, hb.size() * sizeof(BeBufs)));
    CK(hipMemcpy(d_b, hb.data(), hb.size() * sizeof(BeBufs), hipMemcpyHostToDevice));
    std::vector<int> perm((size_t)n_ch); for (int c = 0; c < n_ch; c++) perm[(size_t)c] = c;
    int* d_perm; CK(hipMalloc((void**)&d_perm, perm.size() * 4)); CK(hipMemcpy(d_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemcpy(d_ch, hc.data(), hc.size() * sizeof(BeChan), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(be_schedule_kernel, dim3((n_ch + 63) / 64), dim3(64), 0, 0, d_ch, d_b, d_perm, n_ch);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("be_schedule_kernel %.3f ms\n", ms);
    }
    std::vector<BeChan> back(hc.size());
    CK(hipMemcpy(back.data(), d_ch, hc.size() * sizeof(BeChan), hipMemcpyDeviceToHost));
    long tot = 0; for (auto& s : back) tot += s.n_res;
    printf("outputs per feed: %ld (%.1f M taps)\n", tot, tot * (double)nt / 1e6);
    const dim3 grid((unsigned)((base_n + 2 + BE_FIR_TO - 1) / BE_FIR_TO), (unsigned)((n_ch + BE_FIR_TC - 1) / BE_FIR_TC));
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(be_fir_kernel, grid, dim3(256), 0, 0, d_ch, d_b, d_taps, d_perm, n_ch);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("be_fir_kernel grid %u x %u: %.3f ms  (%.1f G MAC-pairs/s)\n", grid.x, grid.y, ms, tot * (double)nt / ms / 1e6);
    }
    // checksum so that a change of the kernel that alters results shows here
    std::vector<float> r((size_t)2 * back[0].n_res);
    CK(hipMemcpy(r.data(), hb[0].res, r.size() * 4, hipMemcpyDeviceToHost));
    double acc = 0; for (size_t i = 0; i < r.size(); i++) acc += r[i] * (double)((i % 7) + 1);
    printf("checksum ch0: %.9g\n", acc);
    // host check of a few channels: same taps order, separate mul/add (-ffp-contract=off on the host side too)
    std::vector<uint2> sch((size_t)(n_max + 1024) * n_ch);
    CK(hipMemcpy(sch.data(), d_sched, sch.size() * sizeof(uint2), hipMemcpyDeviceToHost));
    long bad = 0, seen = 0;
    for (int c : { 0, 7, n_ch - 1 }) {
        const long n_in = hb[(size_t)c].n_in;
        std::vector<float> x((size_t)(BE_HIST + n_in) * 2), got((size_t)2 * back[(size_t)c].n_res);
        CK(hipMemcpy(x.data(), hb[(size_t)c].mixed, x.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), hb[(size_t)c].res, got.size() * 4, hipMemcpyDeviceToHost));
        for (long o = 0; o < back[(size_t)c].n_res; o++) {
            const uint2 e = sch[(size_t)o * n_ch + c];
            float d; memcpy(&d, &e.y, 4);
            int ph = (int)floorf(d * 16.0f); if (ph < 0) ph = 0;
            volatile float ra = 0, ia = 0;
            for (int i = 0; i < nt; i++) {
                const float t = taps[(size_t)ph * nt + i];
                const volatile float pr = t * x[2 * (size_t)(BE_HIST + (long)e.x - i)], pi = t * x[2 * (size_t)(BE_HIST + (long)e.x - i) + 1];
                ra = ra + pr; ia = ia + pi;
            }
            seen++;
            if (ra != got[2 * (size_t)o] || ia != got[2 * (size_t)o + 1]) { if (bad < 5) printf("mismatch c %d o %ld: %g %g vs %g %g\n", c, o, (float)ra, (float)ia, got[2 * (size_t)o], got[2 * (size_t)o + 1]); bad++; }
        }
    }
    printf("host check: %ld outputs, %ld mismatches\n", seen, bad);
    return 0;
}
