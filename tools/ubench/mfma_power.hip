// Microbenchmark: what the chip sustains on v_mfma_f32_32x32x16_f16 ALONE -- no LDS, no memory, no vector work: four
// independent accumulator chains per wave, one or two waves per SIMD, every CU, ~50 ms per launch, with random or zero
// operands.  The fp16 dense peak of MI355X_MICROARCH.md (2.5 PFLOP/s) is 256 CUs x 4 SIMDs x one instruction per 32
// cycles at 2.4 GHz; this prints the rate and the clock it implies when the matrix pipe never idles.
// hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void k(float* out, int iters, int random) {
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = random ? (_Float16)(((int)(h >> 20) - 2048) * (1.f / 1024.f)) : (_Float16)0.f;
        h = h * 1664525u + 1013904223u;
        b[i] = random ? (_Float16)(((int)(h >> 20) - 2048) * (1.f / 1024.f)) : (_Float16)0.f;
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static void run(int threads, int random) {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int waves_per_simd = threads / 256;
    const int iters = 300000 / waves_per_simd;
    k<<<256, threads>>>(out, iters / 10, random); (void)hipDeviceSynchronize();
    float best = 1e30f, last = 0.f;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        k<<<256, threads>>>(out, iters, random);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&last, e0, e1);
        if (last < best) best = last;
    }
    const double mfma_per_simd = 12.0 * iters * waves_per_simd;
    const double flops = mfma_per_simd * 1024 * 2.0 * 32 * 32 * 16;
    printf("%d wave(s) per SIMD, %s operands: best %.2f ms, last (warm) %.2f ms -> %.0f / %.0f TFLOP/s = %.3f / %.3f of 2 500; implied clock %.2f GHz (32 cycles per instruction)\n",
           waves_per_simd, random ? "random" : "zero", best, last, flops / best * 1e-9, flops / last * 1e-9, flops / best * 1e-9 / 2500, flops / last * 1e-9 / 2500,
           mfma_per_simd * 32 / (last * 1e-3) * 1e-9);
    (void)hipFree(out);
}
int main() { run(256, 1); run(512, 1); run(256, 0); run(512, 0); run(256, 1); return 0; }
