// Microbenchmark: v_mfma_f32_32x32x16_f16 accumulating onto NA accumulators in rotation (dependent distance NA), lone wave
// per SIMD.  hipcc -O3 --offload-arch=gfx950 mfma32_chain.hip -o mfma32_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(acc) "v_mfma_f32_32x32x16_f16 a[" #acc ":" #acc "+15], v[100:103], v[104:107], a[" #acc ":" #acc "+15]\n"
#define CLOBA "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
template <int NA>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, int iters) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (NA == 1) asm volatile(MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) ::: CLOBA);
        if constexpr (NA == 2) asm volatile(MF(0) MF(16) MF(0) MF(16) MF(0) MF(16) MF(0) MF(16) MF(0) MF(16) MF(0) MF(16) ::: CLOBA);
        if constexpr (NA == 3) asm volatile(MF(0) MF(16) MF(32) MF(0) MF(16) MF(32) MF(0) MF(16) MF(32) MF(0) MF(16) MF(32) ::: CLOBA);
        if constexpr (NA == 4) asm volatile(MF(0) MF(16) MF(32) MF(48) MF(0) MF(16) MF(32) MF(48) MF(0) MF(16) MF(32) MF(48) ::: CLOBA);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NA> void run() {
    unsigned long long* cyc; (void)hipMalloc(&cyc, 256 * 8);
    k<NA><<<256, 256>>>(cyc, 1000); (void)hipDeviceSynchronize();
    k<NA><<<256, 256>>>(cyc, 20000); (void)hipDeviceSynchronize();
    unsigned long long c0; (void)hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d accumulator(s) in rotation: %.1f cycles per matrix instruction\n", NA, (double)c0 / 20000 / 12);
    (void)hipFree(cyc);
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); return 0; }
