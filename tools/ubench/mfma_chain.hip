// Microbenchmark: v_mfma_f32_16x16x32_bf16 issue rate against the distance between dependent accumulations
// (hand-written asm loop, one wave per SIMD on every CU).  hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

#define M(i) "v_mfma_f32_16x16x32_bf16 a[" #i "*4:" #i "*4+3], %0, %1, a[" #i "*4:" #i "*4+3]\n"
#define CL "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"

template <int DIST>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, int iters, float seed) {
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * (threadIdx.x % 7 + j)); b[j] = (__bf16)(seed * (threadIdx.x % 5 + 2 * j)); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        // 8 MFMAs per iteration; DIST = number of distinct accumulators used round-robin
        if constexpr (DIST == 1) asm volatile(M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) :: "v"(a), "v"(b) : CL);
        if constexpr (DIST == 2) asm volatile(M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) :: "v"(a), "v"(b) : CL);
        if constexpr (DIST == 3) asm volatile(M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) :: "v"(a), "v"(b) : CL);
        if constexpr (DIST == 4) asm volatile(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) :: "v"(a), "v"(b) : CL);
        if constexpr (DIST == 8) asm volatile(M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) :: "v"(a), "v"(b) : CL);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DIST>
void run(int blocks, float seed) {
    unsigned long long* cyc;
    (void)hipMalloc(&cyc, blocks * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<DIST><<<blocks, 256>>>(cyc, 1000, seed);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<DIST><<<blocks, 256>>>(cyc, iters, seed);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * (DIST == 3 ? 9 : 8);
    printf("distance %d, %d waves/SIMD, seed %.1f: %.2f cycles/MFMA, %.0f TFLOP/s, clock %.2f GHz\n", DIST, blocks / 256, seed, c / nm,
           nm * 16384.0 * 4 * blocks / ms / 1e9, c / (ms * 1e6));
    (void)hipFree(cyc);
}

int main() {
    for (float seed : {0.f, 0.37f}) {
        run<1>(256, seed); run<2>(256, seed); run<3>(256, seed); run<4>(256, seed); run<8>(256, seed);
        run<1>(512, seed); run<2>(512, seed); run<4>(512, seed); run<8>(512, seed);
    }
    return 0;
}
