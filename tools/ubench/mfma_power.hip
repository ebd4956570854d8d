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
    // random: 1 = the SAME random operand pair for every instruction (only the accumulators change), 2 = eight different
    // random pairs in rotation (the operand buses toggle on every instruction, as they do in a real product), 0 = zeros
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    f16x8 a[8], b[8];
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            a[j][i] = random ? (_Float16)(((int)(h >> 20) - 2048) * (1.f / 1024.f)) : (_Float16)0.f;
            h = h * 1664525u + 1013904223u;
            b[j][i] = random ? (_Float16)(((int)(h >> 20) - 2048) * (1.f / 1024.f)) : (_Float16)0.f;
        }
    if (random != 2)
        for (int j = 1; j < 8; ++j) { a[j] = a[0]; b[j] = b[0]; }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    // one asm block per trip: hipcc otherwise shuffles the sixteen operand registers around between the instructions
#define MF(c, x, y) "v_mfma_f32_32x32x16_f16 %" #c ", %" #x ", %" #y ", %" #c "\n\t"
    for (int it = 0; it < iters; ++it)
        asm volatile(MF(0, 4, 15) MF(1, 5, 16) MF(2, 6, 17) MF(3, 7, 18) MF(0, 8, 19) MF(1, 9, 12) MF(2, 10, 13) MF(3, 11, 14)
                     MF(0, 4, 17) MF(1, 5, 18) MF(2, 6, 19) MF(3, 7, 12)
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                       "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
#undef MF
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
           waves_per_simd, random == 2 ? "eight random pairs in rotation as" : random ? "one random pair as" : "zero", best, last, flops / best * 1e-9, flops / last * 1e-9, flops / best * 1e-9 / 2500, flops / last * 1e-9 / 2500,
           mfma_per_simd * 32 / (last * 1e-3) * 1e-9);
    (void)hipFree(out);
}
int main() { run(256, 1); run(256, 2); run(512, 2); run(256, 0); run(512, 0); run(256, 2); return 0; }
