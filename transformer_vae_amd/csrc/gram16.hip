// D x D second moments of the centred points on the 16-bit matrix cores:  C = X^T X  for X = the rows [row0, row0 + n) of a
// split-fp16 image (hi + lo = s x, mmd.hip stage B), three fp16 products per fp32 product (hi*hi + hi*lo + lo*hi: 2^-22 of a
// product dropped), fp32 accumulation.  This is the closed-form part Z'^T Z' / Y'^T Y' of the Q-form MMD gradient
// (DESIGN 4.1g): until round 3 it ran on the fp32-input matrix instruction of the projection GEMM, 16x slower per product,
// from the fp32 latents with the pooled mean subtracted in the operand loads.  The image already IS the centred data.
//
// One workgroup = one 128 x 128 tile (I, J >= I) of C over one chunk of the rows; four waves own its 64 x 64 quadrants
// (2 x 2 blocks of v_mfma_f32_32x32x16_f16: 64 accumulator registers).  C[i][j] = sum_r X[r][i] X[r][j] has BOTH operands
// transposed with respect to the row-major image, so a step of 32 rows is staged in LDS in the swizzled block format of
// mmd_q32.hip (32 rows x 32 features = 2 KiB, LDS-DMA fill, ds_read_b64_tr_b16 reads: conflict-free) and each operand is one
// transposed read.  A step's 16 blocks (I / J range x hi / lo x four 32-feature blocks) are fetched by the four waves
// together -- one barrier per step, two workgroups per CU cover each other's.  Partial tiles go to fp32 slabs; a second
// kernel adds them in a fixed order, undoes the image's scale and writes both triangles.
#include "mmd_common.h"

namespace tvae {
namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 g_f16x8;
typedef __attribute__((ext_vector_type(4))) short g_s16x4;

__device__ __forceinline__ int g16_swz(int q) { return (2 * (q & 1)) ^ ((q >> 1) & 1) ^ (3 * ((q >> 2) & 1)); }

struct Gram16Args {
    const _Float16* img;      // hi image of the set, row stride DPAD; the lo image lo_off elements behind it
    long long lo_off;
    int row0, n_rows, DPAD;   // rows [row0, row0 + n_rows) of the image
    int n_blk;                // 128-feature blocks per side (DPAD / 128)
    int steps, steps_per_chunk, n_chunks;     // 32-row steps
    float* slab;              // [n_chunks][n_tri][128][128]
};

constexpr int G16_BLOCK_B = 2048;                        // one 32 x 32 block
constexpr int G16_BUF_B = 2 * 2 * 4 * G16_BLOCK_B;       // [I / J][hi / lo][4 blocks]
constexpr int G16_LDS_B = 2 * G16_BUF_B;                 // two step buffers: 64 KiB

__global__ __launch_bounds__(256, 2) void gram16_kernel(Gram16Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int n_tri = a.n_blk * (a.n_blk + 1) / 2;
    const int tile = (int)blockIdx.x % n_tri, chunk = (int)blockIdx.x / n_tri;
    int I = 0, rem = tile;
    while (rem >= a.n_blk - I) { rem -= a.n_blk - I; ++I; }
    const int J = I + rem;
    const int s0 = chunk * a.steps_per_chunk;
    const int s1 = min(a.steps, s0 + a.steps_per_chunk);
    const int DPAD = a.DPAD;

    // LDS-DMA lane offsets of a block (mmd_q32.hip): 16 rows x 4 chunks per instruction, the second piece 1 KiB further
    unsigned dma_voff0, dma_voff1;
    {
        const int r = lane >> 2, pos = lane & 3;
        dma_voff0 = (unsigned)(r * DPAD + 8 * (pos ^ g16_swz(r >> 2))) * 2u;
        dma_voff1 = (unsigned)((16 + r) * DPAD + 8 * (pos ^ g16_swz(4 + (r >> 2)))) * 2u - 1024u;
    }
    // transposed operand reads: [k-half kk][j], + 2048 per block
    int tr_off[2][2];
    {
        const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pt = 16 * kk + 8 * j + 4 * (g >> 1) + tq;
                tr_off[kk][j] = 1024 * kk + 64 * (pt & 15) + 16 * ((2 * (g & 1) + (tp >> 1)) ^ g16_swz(pt >> 2)) + 8 * (tp & 1);
            }
    }
    auto tr_read = [&](unsigned addr) __attribute__((always_inline)) -> g_s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) g_s16x4*>(addr));
    };
    // wave w stages 32-feature block w of both ranges and both planes of a step: four 2 KiB pieces
    auto stage = [&](int step, int buf) __attribute__((always_inline)) {
        const long long rowbase = ((long long)a.row0 + 32ll * step) * DPAD;
#pragma unroll
        for (int ij = 0; ij < 2; ++ij)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const _Float16* src = a.img + (pl ? a.lo_off : 0) + rowbase + 128 * (ij ? J : I) + 32 * wave;
                dma16_pair(reinterpret_cast<const float*>(src), dma_voff0, dma_voff1,
                           smem_byte + (unsigned)(buf * G16_BUF_B + ((ij * 2 + pl) * 4 + wave) * G16_BLOCK_B));
            }
    };
    // rows past the end of the range (the last step of a ragged range; whatever lies behind it in the image): zeroed in LDS
    // by the wave that fetched them
    auto zero_tail = [&](int step, int buf) __attribute__((always_inline)) {
        const int valid = a.n_rows - 32 * step;
        if (valid >= 32) return;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int piece = 0; piece < 2; ++piece) {
                const int point = 16 * piece + (lane >> 2);
                if (point >= valid)
                    *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(
                        smem_byte + (unsigned)(buf * G16_BUF_B + (q * 4 + wave) * G16_BLOCK_B + 1024 * piece + lane * 16)) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

    if (s0 < s1) {
        stage(s0, 0);
        wait_vmcnt<0>();
        zero_tail(s0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int s = s0; s < s1; ++s) {
            const int buf = (s - s0) & 1;
            if (s + 1 < s1) stage(s + 1, buf ^ 1);
            const unsigned ib = smem_byte + (unsigned)(buf * G16_BUF_B), jb = ib + 2u * 4u * G16_BLOCK_B;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                g_f16x8 A[2][2], B[2][2];                     // [hi / lo][block]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const unsigned ao = ib + (unsigned)((pl * 4 + 2 * wi + b) * G16_BLOCK_B);
                        const unsigned bo = jb + (unsigned)((pl * 4 + 2 * wj + b) * G16_BLOCK_B);
                        A[pl][b] = __builtin_bit_cast(g_f16x8, __builtin_shufflevector(tr_read(ao + tr_off[kk][0]), tr_read(ao + tr_off[kk][1]), 0, 1, 2, 3, 4, 5, 6, 7));
                        B[pl][b] = __builtin_bit_cast(g_f16x8, __builtin_shufflevector(tr_read(bo + tr_off[kk][0]), tr_read(bo + tr_off[kk][1]), 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) {
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][x], B[0][y], acc[x][y], 0, 0, 0);
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][x], B[1][y], acc[x][y], 0, 0, 0);
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][x], B[0][y], acc[x][y], 0, 0, 0);
                    }
            }
            // the next step has landed (and its tail is zeroed); everybody is done with this buffer
            wait_vmcnt<0>();
            if (s + 1 < s1) zero_tail(s + 1, buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // D[m][n]: n = lane % 32, m = 8 (r / 4) + 4 (lane / 32) + r % 4
    float* out = a.slab + ((long long)chunk * n_tri + tile) * (128 * 128);
    const int n32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 8 * (r >> 2) + 4 * h + (r & 3);
                out[(64 * wi + 32 * x + m) * 128 + 64 * wj + 32 * y + n32] = acc[x][y][r];
            }
}

// C[i][j] = inv_s2 * sum over chunks of the tile that holds (i, j) or (j, i); fixed order
__global__ __launch_bounds__(256) void gram16_reduce_kernel(const float* __restrict__ slab, int n_chunks, int n_blk, int D, int ld,
                                                            const unsigned* __restrict__ amax_bits, float* __restrict__ C) {
    const float inv_s2 = split_scales(*amax_bits).inv_s2;
    const int n_tri = n_blk * (n_blk + 1) / 2;
    const long long total = (long long)D * D;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx / D), j = (int)(idx - (long long)i * D);
        const int lo = i < j ? i : j, hi = i < j ? j : i;      // the tile (I, J >= I) holds element (row in I, column in J)
        const int I = lo >> 7, J = hi >> 7;
        int r = lo & 127, c = hi & 127;
        if (I == J && (i > j)) { r = i & 127; c = j & 127; }    // diagonal tiles are computed in full
        const int tile = I * n_blk - I * (I - 1) / 2 + (J - I);
        const float* p = slab + (long long)tile * (128 * 128) + r * 128 + c;
        float v = 0.f;
#pragma unroll 4
        for (int k = 0; k < n_chunks; ++k) v += p[(long long)k * n_tri * (128 * 128)];
        C[(long long)i * ld + j] = v * inv_s2;
    }
}

}  // namespace

// slab floats this launch can use: one 128 x 128 partial per (chunk, upper-triangle tile)
int64_t gram16_slab_floats(int64_t DPAD, int64_t n_rows) {
    const int64_t nb = DPAD / 128, n_tri = nb * (nb + 1) / 2, steps = (n_rows + 31) / 32;
    int64_t chunks = (768 + n_tri - 1) / n_tri;          // ~3 workgroups per CU
    if (chunks > steps) chunks = steps;
    if (chunks < 1) chunks = 1;
    return chunks * n_tri * 128 * 128;
}

// C (D x D, row stride ld, fp32) = sum over rows [row0, row0 + n_rows) of x x^T, x = (hi + lo) / s of the image
int gram16_launch(const void* img_hi, long long lo_off, int64_t row0, int64_t n_rows, int64_t DPAD, int64_t D, int64_t ld,
                  const unsigned* amax_bits, float* C, float* slab, int64_t slab_floats, hipStream_t s) {
    TVAE_REQUIRE(DPAD % 128 == 0 && DPAD <= 1024 && D <= DPAD && n_rows > 0, "gram16: bad shape (DPAD=%lld D=%lld rows=%lld)",
                 (long long)DPAD, (long long)D, (long long)n_rows);
    Gram16Args a;
    a.img = static_cast<const _Float16*>(img_hi);
    a.lo_off = lo_off;
    a.row0 = (int)row0; a.n_rows = (int)n_rows; a.DPAD = (int)DPAD;
    a.n_blk = (int)(DPAD / 128);
    const int n_tri = a.n_blk * (a.n_blk + 1) / 2;
    a.steps = (int)((n_rows + 31) / 32);
    int64_t chunks = gram16_slab_floats(DPAD, n_rows) / ((int64_t)n_tri * 128 * 128);
    TVAE_REQUIRE(slab && slab_floats >= chunks * n_tri * 128 * 128, "gram16: slab too small");
    a.steps_per_chunk = (int)((a.steps + chunks - 1) / chunks);
    a.n_chunks = (a.steps + a.steps_per_chunk - 1) / a.steps_per_chunk;
    a.slab = slab;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G16_LDS_B);
    }));
    hipLaunchKernelGGL(gram16_kernel, dim3(n_tri * a.n_chunks), dim3(256), G16_LDS_B, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    long long blocks = ((long long)D * D + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gram16_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)slab, a.n_chunks, a.n_blk, (int)D, (int)ld,
                       amax_bits, C);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

}  // namespace tvae
