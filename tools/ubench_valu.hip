// Micro-benchmark: issue rate of the integer VALU instructions the half-band kernels can be built
// from, on gfx950.  One 256-thread block per CU x 4 (4 waves per SIMD), 8 independent accumulator
// chains per lane, N iterations; reports cycles per wave-instruction per SIMD (2.0 = full rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short v2s __attribute__((ext_vector_type(2)));

#define CHAIN8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template<int KIND>
__global__ __launch_bounds__(256) void k(const int* __restrict__ in, int* __restrict__ out, int iters)
{
    int a[8], x = in[threadIdx.x], y = in[threadIdx.x + 256];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = in[threadIdx.x + 512 + i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, x), __builtin_bit_cast(v2s, y), a[i], false);
                else if (KIND == 1) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 4) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 5) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 6) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 7) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 9) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 10) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 11) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 12) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 13) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(long long*)&a[i & 6]) : "v"(x), "v"(y) : "vcc");
                else if (KIND == 15) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 16) asm volatile("v_ashrrev_i32 %0, 11, %0" : "+v"(a[i]));
                else if (KIND == 17) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 18) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 19) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 21) asm volatile("v_lshlrev_b32_sdwa %0, %1, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "s"(11));
                else if (KIND == 22) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 23) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(a[i]) : "v"(x));
                else if (KIND == 24) asm volatile("v_ashrrev_i32_sdwa %0, %1, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "s"(3));
                else if (KIND == 25) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 26) asm volatile("v_lshlrev_b32 %0, 11, %0" : "+v"(a[i]));
                else if (KIND == 27) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 28) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 29) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else if (KIND == 30) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(1300));
                else if (KIND == 31) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 20) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(long long*)&a[i & 6]) : "v"(*(long long*)&a[(i+2)&6]), "v"(*(long long*)&a[(i+4)&6]));
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template<int KIND> double run(const char* name, int* din, int* dout, int cus, double ghz_hint)
{
    const int iters = 2000, grid = cus * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, din, dout, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, din, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 blocks/CU x 4 waves/block / 4 SIMDs = 4 waves, each iters*16*8 instr
    const double instr_per_simd = 4.0 * iters * 16 * 8;
    const double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-18s %8.3f ms  %6.3f ns/wave-instr/SIMD  (~%.2f cycles @%.1f GHz)\n", name, ms, ns_per, ns_per * ghz_hint, ghz_hint);
    return ns_per;
}

int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int* din; int* dout;
    hipMalloc(&din, 4096 * 4); hipMalloc(&dout, cus * 4 * 256 * 4);
    std::vector<int> h(4096); for (int i = 0; i < 4096; i++) h[i] = (i * 2654435761u) >> 7;
    hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("CUs %d\n", cus);
    const double g = 2.4;
    run<3>("v_add_u32", din, dout, cus, g);
    run<0>("sdot2 (builtin)", din, dout, cus, g);
    run<13>("v_dot2c_i32_i16", din, dout, cus, g);
    run<9>("v_dot2_i32_i16", din, dout, cus, g);
    run<1>("v_mad_i32_i24", din, dout, cus, g);
    run<10>("v_mad_u32_u24", din, dout, cus, g);
    run<2>("v_mul_lo_u32", din, dout, cus, g);
    run<14>("v_mad_u64_u32", din, dout, cus, g);
    run<4>("v_pk_mad_i16", din, dout, cus, g);
    run<12>("v_pk_mul_lo_u16", din, dout, cus, g);
    run<5>("v_pk_add_i16", din, dout, cus, g);
    run<6>("v_mad_i32_i16", din, dout, cus, g);
    run<7>("v_dot4_i32_i8", din, dout, cus, g);
    run<8>("v_perm_b32", din, dout, cus, g);
    run<11>("v_lshl_add_u32", din, dout, cus, g);
    run<15>("v_add3_u32", din, dout, cus, g);
    run<16>("v_ashrrev_i32", din, dout, cus, g);
    run<17>("v_pk_fma_f16", din, dout, cus, g);
    run<18>("v_dot2_f32_f16", din, dout, cus, g);
    run<19>("v_fma_f32", din, dout, cus, g);
    run<20>("v_pk_fma_f32", din, dout, cus, g);
    run<21>("v_lshlrev_b32_sdwa", din, dout, cus, g);
    run<26>("v_lshlrev_b32", din, dout, cus, g);
    run<22>("v_pack_b32_f16", din, dout, cus, g);
    run<23>("v_mov_b32_sdwa", din, dout, cus, g);
    run<24>("v_ashrrev_sdwa", din, dout, cus, g);
    run<25>("v_add_u32_sdwa", din, dout, cus, g);
    run<27>("v_mul_i32_i24", din, dout, cus, g);
    run<28>("v_and_b32", din, dout, cus, g);
    run<29>("v_cvt_pk_i16_i32", din, dout, cus, g);
    run<30>("v_mad_i32_i24 sgpr", din, dout, cus, g);
    run<31>("v_bfi_b32", din, dout, cus, g);
    run<32>("v_and_or_b32", din, dout, cus, g);
    return 0;
}
