// Check + rate of the i8 matrix-core half-band primitive (sdrangel_amd/csrc/hb_mfma.hpp) on gfx950.
//   1. exactness: random int16 odd arms (full range, incl. -32768 / 32767 runs) -> S = sum_j h_j o[k-j] for every output
//      of a tile, orders 48 and 64, plain and alternating-sign taps, against a host loop (int64, wrapped to int32);
//   2. rate: every SIMD loops over tiles read from LDS (2 x ds_read_b128 + 5 or 6 MFMA + the limb combine).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sdrangel_amd/csrc tools/ubench_hb_i8.hip -o tools/ubench_hb_i8
#include "hb_mfma.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

using namespace sdrx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int NBLK = 64;                      // blocks of 16 outputs per array = 4 tiles
constexpr int ARR = HIST + 16 * NBLK + 32;    // int16 entries (slack behind: the second K-step reads up to entry 16 blk - T + 63)

template<int ORDER, bool ALT>
__global__ __launch_bounds__(64) void check_kernel(const int16_t* __restrict__ arm, int* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[ARR / 2];
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    for (int i = lane; i < ARR / 2; i += 64) lds[i] = reinterpret_cast<const uint32_t*>(arm)[i] ^ HBM_BIAS2;
    HbMfmaTaps<ORDER, ALT> taps; taps.init(lane);
    __syncthreads();
    const int B = HbMfmaTaps<ORDER, ALT>::BIAS;
    const v4i bias = { B, B, B, B };
    for (int t = 0; t < NBLK / 16; t++) {
        const int blk = 16 * t + n;
        const int idx = HIST - ORDER / 2 + 16 * blk + 8 * g;          // int16 index of the lane's 8 entries, K-step 0
        const v4i b0 = *reinterpret_cast<const v4i*>(lds + idx / 2);
        const v4i b1 = *reinterpret_cast<const v4i*>(lds + idx / 2 + 16);
        const v4i S = taps.tile(b0, b1, bias);
#pragma unroll
        for (int i = 0; i < 4; i++) out[16 * blk + 4 * g + i] = S[i];
    }
}

// VAR 0: LDS reads + MFMAs + limb combine (the primitive); 1: operands in registers (no LDS); 2: as 1, MFMAs only (the three
// accumulators are XORed raw); 3: LDS reads only
template<int ORDER, int VAR>
__global__ __launch_bounds__(256) void rate_kernel(int* __restrict__ out, int reps)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (uint32_t)i * 2654435761u;
    HbMfmaTaps<ORDER, false> taps; taps.init(lane);
    __syncthreads();
    const v4i bias = { 7, 7, 7, 7 };
    v4i acc = { 0, 0, 0, 0 };
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int idx = 1024 * wv + 8 * (16 * t + n) + 4 * g + (r & 3) * 4;
            v4i b0, b1;
            if constexpr (VAR == 0 || VAR == 3) {
                b0 = *reinterpret_cast<const v4i*>(lds + (idx & 4092));
                b1 = *reinterpret_cast<const v4i*>(lds + ((idx + 16) & 4092));
            } else { b0 = acc + t; b1 = acc - t; }
            if constexpr (VAR == 3) { acc ^= b0; acc += b1; }
            else if constexpr (VAR == 2) {
                const v4i z = { 0, 0, 0, 0 };
                v4i P3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(taps.p3[0], b0, bias, 0, 0, 0);
                v4i P2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(taps.p2[0], b0, z, 0, 0, 0);
                v4i P1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(taps.p1[0], b0, z, 0, 0, 0);
                P3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(taps.p3[1], b1, P3, 0, 0, 0);
                P2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(taps.p2[1], b1, P2, 0, 0, 0);
                acc ^= P3; acc ^= P2; acc ^= P1;
            } else {
                const v4i S = taps.tile(b0, b1, bias);
                acc ^= S;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template<int ORDER, bool ALT>
static int check(unsigned seed)
{
    std::vector<int16_t> arm(ARR);
    srand(seed);
    for (int i = 0; i < ARR; i++) {
        const int r = rand() % 16;
        arm[i] = r == 0 ? -32768 : r == 1 ? 32767 : (int16_t)(rand() & 0xffff);
    }
    for (int i = 200; i < 280; i++) arm[i] = -32768;                 // worst-case runs
    for (int i = 400; i < 480; i++) arm[i] = (i & 1) ? 32767 : -32768;
    int16_t* d_arm; int* d_out;
    CK(hipMalloc(&d_arm, ARR * 2)); CK(hipMalloc(&d_out, 16 * NBLK * 4));
    CK(hipMemcpy(d_arm, arm.data(), ARR * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((check_kernel<ORDER, ALT>), dim3(1), dim3(64), 0, 0, d_arm, d_out);
    CK(hipDeviceSynchronize());
    std::vector<int> got(16 * NBLK);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int k = 0; k < 16 * NBLK; k++) {
        long s = 0;
        for (int j = 0; j < ORDER / 2; j++) {
            const int idx = HIST + k - j;                            // o[k - j]
            long x = arm[idx];
            if (ALT && (idx & 1) == 0) x = -x;
            s += (long)hb_tap<ORDER>(j) * x;
        }
        if ((int)s != got[k]) { if (bad < 5) printf("  order %d alt %d: k %d want %ld got %d\n", ORDER, (int)ALT, k, s, got[k]); bad++; }
    }
    printf("order %d alt %d: %d outputs, %d mismatches\n", ORDER, (int)ALT, 16 * NBLK, bad);
    CK(hipFree(d_arm)); CK(hipFree(d_out));
    return bad;
}

template<int ORDER, int VAR>
static void rate()
{
    int* d_out; const int blocks = 256 * 8, reps = 2000;
    CK(hipMalloc(&d_out, blocks * 256 * 4));
    hipLaunchKernelGGL((rate_kernel<ORDER, VAR>), dim3(blocks), dim3(256), 0, 0, d_out, 10);
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((rate_kernel<ORDER, VAR>), dim3(blocks), dim3(256), 0, 0, d_out, reps);
    CK(hipDeviceSynchronize());
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double tiles = (double)blocks * 4 * reps * 8;
    printf("order %d var %d: %.3f ms, %.3g tiles/s = %.3g outputs/s; per SIMD (1024 SIMDs) %.1f ns per tile (= %.0f cycles at 2.1 GHz)\n",
           ORDER, VAR, s * 1e3, tiles / s, tiles / s * 256, s / (tiles / 1024) * 1e9, s / (tiles / 1024) * 2.1e9);
    CK(hipFree(d_out));
}

int main()
{
    int bad = 0;
    bad += check<48, false>(1); bad += check<48, true>(2); bad += check<64, false>(3); bad += check<64, true>(4);
    rate<48, 0>(); rate<64, 0>(); rate<48, 1>(); rate<48, 2>(); rate<48, 3>(); rate<64, 1>();
    printf(bad ? "FAILED\n" : "ALL EXACT\n");
    return bad ? 1 : 0;
}
