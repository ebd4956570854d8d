// Microbenchmark: do matrix instructions and vector instructions overlap on a SIMD -- within one wave, and across the
// 1, 2 or 3 waves of a SIMD?  Per loop iteration a wave issues 12 v_mfma_f32_16x16x32_f16 (192 cycles of matrix work)
// and / or 48 v_pk_fma_f32 (192 cycles of vector issue), blocked or interleaved 1 : 4.
// hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(acc, b) "v_mfma_f32_16x16x32_f16 a[" #acc ":" #acc "+3], v[100:103], v[" #b ":" #b "+3], a[" #acc ":" #acc "+3]\n"
#define VF(d) "v_pk_fma_f32 v[" #d ":" #d "+1], v[60:61], v[62:63], v[" #d ":" #d "+1]\n"
#ifdef PLAIN_FMA
#undef VF
#define VF(d) "v_fma_f32 v" #d ", v60, v62, v" #d "\n"
#endif
#ifdef INT_ADD
#undef VF
#define VF(d) "v_add_u32 v" #d ", v60, v" #d "\n"
#endif
#define V4 VF(40) VF(42) VF(44) VF(46)
#define V4B VF(48) VF(50) VF(52) VF(54)
#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"

// MODE 0: matrix only.  1: vector only.  2: 12 matrix, then 48 vector.  3: interleaved 1 : 4.
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* cyc, int iters) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2)
            asm volatile(MF(0, 20) MF(4, 24) MF(8, 28) MF(12, 32) MF(0, 36) MF(4, 20) MF(8, 24) MF(12, 28) MF(0, 32) MF(4, 36) MF(8, 20) MF(12, 24) ::: CLOB);
        if constexpr (MODE == 1 || MODE == 2)
            asm volatile(V4 V4B V4 V4B V4 V4B V4 V4B V4 V4B V4 V4B ::: CLOB);
        if constexpr (MODE == 3)
            asm volatile(MF(0, 20) V4 MF(4, 24) V4B MF(8, 28) V4 MF(12, 32) V4B MF(0, 36) V4 MF(4, 20) V4B MF(8, 24) V4 MF(12, 28) V4B
                         MF(0, 32) V4 MF(4, 36) V4B MF(8, 20) V4 MF(12, 24) V4B ::: CLOB);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* what, int waves_per_simd) {
    unsigned long long* cyc;
    const int blocks = 256, iters = 20000;
    (void)hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256 * waves_per_simd>>>(cyc, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256 * waves_per_simd>>>(cyc, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %d wave(s)/SIMD: %8.1f ns per iteration of one wave, %8.1f ns per iteration and SIMD-wave\n", what, waves_per_simd,
           ms * 1e6 / iters, ms * 1e6 / iters / waves_per_simd);
    (void)hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 3; ++w) {
        run<0>("12 matrix", w);
        run<1>("48 vector (packed fma)", w);
        run<2>("12 matrix then 48 vector", w);
        run<3>("interleaved 1 matrix : 4 vector", w);
    }
    return 0;
}
