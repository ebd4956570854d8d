// Microbenchmark: the rate of v_mfma_f32_32x32x16_f16 (random operands, one wave per SIMD, every CU) with OTHER work
// in its shadow, to see what the power budget charges for it: per matrix instruction L x ds_read_b128 (1 KiB per wave,
// conflict-free, random LDS contents) and V x v_fma_f32 (random operands).  tools/ubench/mfma_power.hip is the L = V = 0 row.
// hipcc -O3 --offload-arch=gfx950 mfma_power_mix.hip -o mfma_power_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int L, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ f32x4 lds[4096];                               // 64 KiB
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int i = threadIdx.x; i < 4096; i += 256) {
        f32x4 v;
        for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; v[e] = (float)((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 20)); }
        lds[i] = v;
    }
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.f / 1024.f));
        h = h * 1664525u + 1013904223u; b[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.f / 1024.f));
    }
    float f0 = a[0], f1 = b[0], f2 = a[1], f3 = b[1], m0 = 1.0001f + (float)a[2] * 1e-4f, m1 = 0.9999f;
    __syncthreads();
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 r0 = {}, r1 = {};
    const int lane16 = (threadIdx.x & 63) + 64 * (threadIdx.x >> 6);     // this wave's 1 KiB window, moved every read
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            f32x16& c = (u & 3) == 0 ? c0 : (u & 3) == 1 ? c1 : (u & 3) == 2 ? c2 : c3;
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const f32x4 t = lds[(lane16 + 256 * ((it * 12 + u + 5 * l) & 15)) & 4095];
                if (l & 1) r1 += t; else r0 = t;              // (the add keeps the second read alive; L = 1: no vector work at all)
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                // ONE instruction each, four independent chains
                if ((v & 3) == 0) f0 = __builtin_fmaf(f0, m0, m1);
                else if ((v & 3) == 1) f1 = __builtin_fmaf(f1, m1, m0);
                else if ((v & 3) == 2) f2 = __builtin_fmaf(f2, m0, m1);
                else f3 = __builtin_fmaf(f3, m1, m0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(a), "+v"(b), "+v"(r0), "+v"(r1));
    }
    float s = f0 + f1 + f2 + f3 + r0[0] + r1[1];
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int L, int V>
static void run() {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 200000;
    k<L, V><<<256, 256>>>(out, iters / 10); (void)hipDeviceSynchronize();
    float ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        k<L, V><<<256, 256>>>(out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfma_per_simd = 12.0 * iters;
    const double flops = mfma_per_simd * 1024 * 2.0 * 32 * 32 * 16;
    printf("per matrix instruction %d ds_read_b128 + %d v_fma_f32: %.2f ms -> %.0f TFLOP/s = %.3f of 2 500; %.1f ns per matrix instruction (13.3 at 2.4 GHz, 19.9 at 1.61 GHz)\n",
           L, V, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500, ms * 1e6 / mfma_per_simd);
    (void)hipFree(out);
}
int main() {
    run<0, 0>(); run<1, 0>(); run<0, 2>(); run<0, 3>(); run<1, 2>(); run<1, 3>(); run<0, 4>(); run<0, 0>();
    return 0;
}
