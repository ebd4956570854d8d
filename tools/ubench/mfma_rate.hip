// Microbenchmark: issue rate of bf16 MFMAs on every SIMD of the chip (development tool).
//   hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BIG>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * (threadIdx.x % 7 + j)); b[j] = (__bf16)(seed * (threadIdx.x % 5 + 2 * j)); }
    typedef typename std::conditional<BIG, f32x16, f32x4>::type acc_t;
    acc_t acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < (BIG ? 16 : 4); ++j) acc[i][j] = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (BIG) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("" : "+a"(acc[i]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < (BIG ? 16 : 4); ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool BIG>
void run(const char* name, int blocks, float seed) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, BIG><<<blocks, 256>>>(out, cyc, 1000, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, BIG><<<blocks, 256>>>(out, cyc, iters, seed);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * NACC;
    const double flop = nm * (BIG ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2) * 4 * blocks;
    printf("%-28s blocks %4d seed %.1f: %.2f cycles/MFMA (s_memtime), %.1f TFLOP/s, eff clock %.2f GHz\n", name, blocks, seed, c / nm,
           flop / ms / 1e9, c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (float seed : {0.f, 0.37f}) {
        run<8, false>("16x16x32 8 acc, 1 wave/SIMD", 256, seed);
        run<8, false>("16x16x32 8 acc, 2 wave/SIMD", 512, seed);
        run<2, false>("16x16x32 2 acc, 1 wave/SIMD", 256, seed);
        run<3, false>("16x16x32 3 acc, 1 wave/SIMD", 256, seed);
        run<4, false>("16x16x32 4 acc, 1 wave/SIMD", 256, seed);
        run<1, false>("16x16x32 1 acc, 1 wave/SIMD", 256, seed);
        run<4, false>("16x16x32 4 acc, 2 wave/SIMD", 512, seed);
        run<4, true>("32x32x16 4 acc, 1 wave/SIMD", 256, seed);
        run<4, true>("32x32x16 4 acc, 2 wave/SIMD", 512, seed);
    }
    return 0;
}
