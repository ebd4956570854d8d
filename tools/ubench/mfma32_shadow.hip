// Microbenchmark: how many other instructions issue in the shadow of a v_mfma_f32_32x32x16_f16 on a lone wave per SIMD?
// Per iteration a wave issues 8 matrix instructions (2 accumulators, alternating) with K fillers behind each:
//   MODE 0: v_fma_f32   1: s_add_u32 pairs   2: ds_read_b128 (one per matrix instruction, K ignored above 1)   3: mix (1 ds_read_b128 + (K-1) v_fma)
// hipcc -O3 --offload-arch=gfx950 mfma32_shadow.hip -o mfma32_shadow
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(acc) "v_mfma_f32_32x32x16_f16 a[" #acc ":" #acc "+15], v[100:103], v[104:107], a[" #acc ":" #acc "+15]\n"
#define VF(d) "v_fma_f32 v" #d ", v60, v62, v" #d "\n"
#define SF(d) "s_add_u32 s" #d ", s" #d ", 1\n"
#define DR(d) "ds_read_b128 v[" #d ":" #d "+3], v70\n"
#define CLOBV "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v80","v81","v82","v83","v84","v85","v86","v87"
#define CLOBA "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"
#define CLOBS "s40","s41","s42","s43","s44","s45","s46","s47"

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, int iters) {
    __shared__ float lds[8192];
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    asm volatile("v_lshlrev_b32 v70, 4, %0" ::"v"(threadIdx.x & 63) : "v70");
    unsigned sreg[4] = {1, 2, 3, 4};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define ONE(i, a) if constexpr (K > i) { \
        if constexpr (MODE == 0) asm volatile("v_fma_f32 v%0, v60, v62, v%0" :: "n"(a) : CLOBV); \
        if constexpr (MODE == 4) asm volatile("v_pk_fma_f32 v[%0:%0+1], v[60:61], v[62:63], v[%0:%0+1]" :: "n"(40 + 2 * (i)) : CLOBV); \
        if constexpr (MODE == 5) asm volatile("v_accvgpr_read_b32 v%0, a%1" :: "n"(a), "n"(32 + (i)) : CLOBV); \
        if constexpr (MODE == 6) asm volatile("v_add_f64 v[%0:%0+1], v[60:61], v[%0:%0+1]" :: "n"(40 + 2 * (i)) : CLOBV); \
        if constexpr (MODE == 7) asm volatile("v_cvt_pkrtz_f16_f32 v%0, v60, v62" :: "n"(a) : CLOBV); \
        if constexpr (MODE == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg[(i) & 3])); \
        if constexpr (MODE == 8) asm volatile("ds_write_b128 v70, v[80:83] offset:%0" :: "n"(4096 * ((i) & 1)) : "memory"); \
        if constexpr (MODE == 9) asm volatile("v_add_u32 v%0, v60, v%0" :: "n"(a) : CLOBV); }
#define FILL(a, b, c, d, e, f, g, h) ONE(0, a) ONE(1, b) ONE(2, c) ONE(3, d) ONE(4, e) ONE(5, f) ONE(6, g) ONE(7, h)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile(MF(0) ::: CLOBA);
            FILL(40, 41, 42, 43, 44, 45, 46, 47)
            asm volatile(MF(16) ::: CLOBA);
            FILL(48, 49, 50, 51, 52, 53, 54, 55)
        }
        if constexpr (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0 + (sreg[0] + sreg[1] + sreg[2] + sreg[3] == 12345u ? 1 : 0);
}

template <int MODE, int K>
void run(const char* what) {
    unsigned long long* cyc;
    const int blocks = 256, iters = 20000;
    (void)hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, K><<<blocks, 256>>>(cyc, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE, K><<<blocks, 256>>>(cyc, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c0; (void)hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s K=%d: %7.1f ns and %7.1f cycles per matrix instruction\n", what, K, ms * 1e6 / iters / 8, (double)c0 / iters / 8);
    (void)hipFree(cyc);
}

int main() {
    run<0, 0>("bare 32x32x16");
    run<0, 3>("v_fma_f32"); run<0, 4>("v_fma_f32"); run<0, 6>("v_fma_f32");
    run<4, 2>("v_pk_fma_f32"); run<4, 3>("v_pk_fma_f32"); run<4, 4>("v_pk_fma_f32"); run<4, 6>("v_pk_fma_f32");
    run<5, 4>("v_accvgpr_read"); run<5, 8>("v_accvgpr_read");
    run<6, 4>("v_add_f64"); run<6, 8>("v_add_f64");
    run<7, 4>("v_cvt_pk_f16_f32 (pkrtz)"); run<7, 8>("v_cvt_pk_f16_f32 (pkrtz)");
    run<1, 4>("s_add_u32"); run<1, 8>("s_add_u32");
    run<8, 2>("ds_write_b128"); run<8, 4>("ds_write_b128");
    run<9, 4>("v_add_u32"); run<9, 8>("v_add_u32");
    return 0;
}
