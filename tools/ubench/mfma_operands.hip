// Microbenchmark: does v_mfma_f32_16x16x32_f16 keep its 16-cycle rate when every instruction reads DIFFERENT operand
// registers (the Gram pattern: A fixed for a few instructions, B from a large register-resident block)?
// hipcc -O3 --offload-arch=gfx950 mfma_operands.hip -o mfma_operands
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(acc, a, b) "v_mfma_f32_16x16x32_f16 a[" #acc ":" #acc "+3], v[" #a ":" #a "+3], v[" #b ":" #b "+3], a[" #acc ":" #acc "+3]\n"
#define MB(acc, a, b) "v_mfma_f32_16x16x32_bf16 a[" #acc ":" #acc "+3], v[" #a ":" #a "+3], v[" #b ":" #b "+3], a[" #acc ":" #acc "+3]\n"
#define M32(acc, a, b) "v_mfma_f32_32x32x16_f16 a[" #acc ":" #acc "+15], v[" #a ":" #a "+3], v[" #b ":" #b "+3], a[" #acc ":" #acc "+15]\n"

// MODE 0: same A, same B.  1: same A, B varies over 12 register quads.  2: A and B vary.  3: like 1, bf16.  4: 32x32x16, B varies
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, int iters) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0)
            asm volatile(MF(0, 100, 20) MF(4, 100, 20) MF(8, 100, 20) MF(12, 100, 20) MF(0, 100, 20) MF(4, 100, 20) MF(8, 100, 20) MF(12, 100, 20)
                         MF(0, 100, 20) MF(4, 100, 20) MF(8, 100, 20) MF(12, 100, 20) ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
        if constexpr (MODE == 1)
            asm volatile(MF(0, 100, 20) MF(4, 100, 24) MF(8, 100, 28) MF(12, 100, 32) MF(0, 100, 36) MF(4, 100, 40) MF(8, 100, 44) MF(12, 100, 48)
                         MF(0, 100, 52) MF(4, 100, 56) MF(8, 100, 60) MF(12, 100, 64) ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
        if constexpr (MODE == 2)
            asm volatile(MF(0, 100, 20) MF(4, 104, 24) MF(8, 108, 28) MF(12, 112, 32) MF(0, 116, 36) MF(4, 120, 40) MF(8, 124, 44) MF(12, 128, 48)
                         MF(0, 132, 52) MF(4, 136, 56) MF(8, 140, 60) MF(12, 144, 64) ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
        if constexpr (MODE == 3)
            asm volatile(MB(0, 100, 20) MB(4, 100, 24) MB(8, 100, 28) MB(12, 100, 32) MB(0, 100, 36) MB(4, 100, 40) MB(8, 100, 44) MB(12, 100, 48)
                         MB(0, 100, 52) MB(4, 100, 56) MB(8, 100, 60) MB(12, 100, 64) ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
        if constexpr (MODE == 4)
            asm volatile(M32(0, 100, 20) M32(16, 100, 24) M32(0, 104, 20) M32(16, 104, 24) M32(0, 100, 28) M32(16, 100, 32) ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
                         "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* what) {
    unsigned long long* cyc;
    const int blocks = 256, iters = 20000;
    (void)hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(cyc, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(cyc, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * (MODE == 4 ? 6 : 12);
    printf("%-44s %.2f counter ticks/MFMA, %.2f ns/MFMA\n", what, c / nm, ms * 1e6 / nm);
    (void)hipFree(cyc);
}

int main() {
    run<0>("16x16x32 f16, same A, same B");
    run<1>("16x16x32 f16, same A, B varies");
    run<2>("16x16x32 f16, A and B vary");
    run<3>("16x16x32 bf16, same A, B varies");
    run<4>("32x32x16 f16, operands vary");
    return 0;
}
