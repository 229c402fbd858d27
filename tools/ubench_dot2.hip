// dot2c operand-kind microbenchmark: literal vs SGPR vs VGPR coefficient, 8 independent chains, 4 waves/SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template<int KIND>
__global__ __launch_bounds__(256) void k(const int* __restrict__ in, int* __restrict__ out, int iters, int sc0, int sc1)
{
    int a[8], x = in[threadIdx.x], y = in[threadIdx.x + 256];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = in[threadIdx.x + 512 + i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_dot2c_i32_i16 %0, 0x0514fe58, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 1) asm volatile("v_dot2c_i32_i16 %0, %2, %1" : "+v"(a[i]) : "v"(x), "s"(sc0));
                else if (KIND == 2) asm volatile("v_dot2c_i32_i16 %0, %2, %1" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 3) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(sc1));
                else if (KIND == 4) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(sc0));
                else if (KIND == 5) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(sc0));
                else if (KIND == 6) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
                else if (KIND == 7) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 8) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 9) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(a[i]));
                else if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 11) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
                else if (KIND == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
                else if (KIND == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 15) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 16) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 17) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(x));
                else if (KIND == 18) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 19) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template<int KIND> void run(const char* name, int* din, int* dout, int cus)
{
    const int iters = 2000, grid = cus * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, din, dout, 10, 0x0514fe58, 0x00170023);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, din, dout, iters, 0x0514fe58, 0x00170023);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.3f ms  %6.3f ns/wave-instr/SIMD\n", name, ms, ms * 1e6 / (4.0 * iters * 16 * 8));
}
int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int* din; int* dout; hipMalloc(&din, 4096 * 4); hipMalloc(&dout, cus * 4 * 256 * 4);
    std::vector<int> h(4096); for (int i = 0; i < 4096; i++) h[i] = (i * 2654435761u) >> 7;
    hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<9>("v_bfe_i32 (ref full rate?)", din, dout, cus);
    run<0>("v_dot2c literal", din, dout, cus);
    run<1>("v_dot2c sgpr", din, dout, cus);
    run<2>("v_dot2c vgpr", din, dout, cus);
    run<3>("v_dot2 vop3p sgpr", din, dout, cus);
    run<4>("v_mad_i32_i24 sgpr", din, dout, cus);
    run<18>("v_mul_i32_i24 vgpr", din, dout, cus);
    run<5>("v_add_u32 sgpr", din, dout, cus);
    run<6>("v_add_u32 literal", din, dout, cus);
    run<12>("v_sub_u32 vgpr", din, dout, cus);
    run<7>("v_max_i32", din, dout, cus);
    run<8>("v_max3_i32", din, dout, cus);
    run<10>("v_mov_b32", din, dout, cus);
    run<11>("v_lshlrev_b32", din, dout, cus);
    run<13>("v_cndmask_b32", din, dout, cus);
    run<14>("v_and_b32", din, dout, cus);
    run<19>("v_or_b32", din, dout, cus);
    run<15>("v_cvt_pk_i16_i32", din, dout, cus);
    run<16>("v_pk_sub_i16", din, dout, cus);
    run<17>("v_alignbit_b32", din, dout, cus);
    return 0;
}
