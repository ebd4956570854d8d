// fp32-accurate projection GEMM on the gfx950 matrix cores.
//
// Every contraction of the bottleneck (encode, decode, and the four backward products)
// runs through this one LDS-tiled kernel built on v_mfma_f32_32x32x2_f32: fp32 (or bf16
// widened to fp32) operands, fp32 accumulate — bit-for-bit an fmaf chain, so the fp32
// path meets the 1e-5 parity bar without split-precision tricks.
//
//   workgroup = 4 waves = one 64x64 tile of C, each wave one 32x32 accumulator
//   K is walked in 32-deep LDS tiles; operands are stored k-major in LDS ([k][m], ld 65)
//   so that lane (i = l&31, h = l>>5) of MFMA step s reads A[k=2s+h][i] / B[k=2s+h][j]
//   with conflict-free ds_read_b32 whatever the layout in HBM was.
//   Operands may be "row-gathered" (RowMap) — that is how the first-n-token slice of h is
//   read and how dh[:, :n] is scattered without a copy.
//   Thin outputs with long K (the weight gradients: K = B*n rows) are split along K over
//   blockIdx.z into fp32 slabs and summed in a fixed order (deterministic, no atomics).
#include <cstdlib>

#include "tvae_common.h"

namespace tvae {
namespace {

// (the TVAE_* macros below exist for A/B builds of single constants: tools/build_variant.sh)
#ifndef TVAE_BK
#define TVAE_BK 32
#endif
#ifndef TVAE_SPLIT_TARGET
#define TVAE_SPLIT_TARGET 2048
#endif
#ifndef TVAE_PAIR_MAX
#define TVAE_PAIR_MAX 4096
#endif
#ifndef TVAE_FUSED_LN_MIN_ROW_BLOCKS
#define TVAE_FUSED_LN_MIN_ROW_BLOCKS 1024      // fused decode + LayerNorm from this many 64-row blocks on (4 per CU)
#endif
#ifdef TVAE_GEMM_MINWAVES
#define TVAE_GEMM_BOUNDS __launch_bounds__(256, TVAE_GEMM_MINWAVES)
#else
#define TVAE_GEMM_BOUNDS __launch_bounds__(256)
#endif
constexpr int BM = 64, BN = 64, BK = TVAE_BK, LDS_LD = 65;
// Latency-bound launches (few workgroups, short K per workgroup -- every product of the reference's own batch
// sizes) walk K in 96-deep tiles instead: a cold global load costs ~2 us and the K loop pays it once per tile,
// so K = 77 (the default latent size) is ONE load phase instead of three, and a 768-deep product split eight
// ways is one phase per workgroup.  2 x 96 x 65 floats = 49.9 KB of LDS, three workgroups per CU.
constexpr int BK_DEEP = 96;
constexpr int DEEP_MAX_BLOCKS = 512;       // use the deep tiles when the whole launch is at most this many workgroups (measured: at 1 000-2 600 workgroups, BASELINE cfg3, they are 15-30 % slower than the 32-deep ones)

struct GemmArgs {
    const void* A; const void* B; void* C; const void* bias; float* slab;
    int M, N, K, k_per_split, splits;
    RowMap amap, bmap, cmap;
    int vecA, vecB;
    // weight-gradient products only (A and B k-major): B gets a virtual column of ones at index Nb (= N - 1), so
    // that output column Nb is the column sum of A over k = the bias gradient, written to csum[m] instead of C
    void* csum; int Nb;
    int deep;      // 1: BK_DEEP-deep K tiles (small launches)
    const float* a_sub; const float* b_sub;      // GemmDesc::a_sub / b_sub (nullable)
};

// A (64 rows x BKT k) operand tile travels global -> registers -> LDS (as tile[k][row]) in BKT/16 pieces per
// thread, so that the global loads of tile t+1 are in flight while tile t is multiplied.
// (Measured and not kept: a branch-free form -- out-of-range chunks read element 0 and are zeroed by selects, so
// that no load sits behind a bounds check -- was 0-20 % SLOWER at the BASELINE cfg2/cfg3 shapes.)
template <typename T, bool KMAJOR, int BKT>
__device__ __forceinline__ void tile_load(float v[BKT / 16][4], const T* __restrict__ P, const RowMap& map, int r0, int R,
                                          int k0, int kend, bool vec, int tid, const float* __restrict__ sub = nullptr,
                                          const float* ksub = nullptr) {
    // sub: GemmDesc::a_sub / b_sub of a row-major operand (indexed by k: re-read per tile).  ksub: the same for a k-major
    // operand, whose four contiguous indices are the same for every tile of a thread -- hoisted into registers by the caller
#pragma unroll
    for (int j = 0; j < BKT / 16; ++j) {
        const int q = tid + 256 * j;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = 0.f;
        if (!KMAJOR) {
            const int row = q / (BKT / 4), kq = (q % (BKT / 4)) * 4;
            const int r = r0 + row, k = k0 + kq;
            if (r < R && k < kend) {
                const T* p = P + map(r) + k;
                const int nv = min(4, kend - k);
                if (vec && nv == 4) load4_vec(p, v[j]); else load4_guard(p, nv, v[j]);
                if (sub) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nv) v[j][e] -= sub[k + e];
                }
            }
        } else {
            const int krow = q >> 4, rq = (q & 15) * 4;
            const int k = k0 + krow, r = r0 + rq;
            if (k < kend && r < R) {
                const T* p = P + map(k) + r;
                const int nv = min(4, R - r);
                if (vec && nv == 4) load4_vec(p, v[j]); else load4_guard(p, nv, v[j]);
                if (ksub) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < nv) v[j][e] -= ksub[e];
                }
            }
        }
    }
}

// the virtual ones column of a k-major B tile (see GemmArgs::csum): element (k, Nb) = 1 for k < kend
template <int BKT>
__device__ __forceinline__ void tile_ones(float v[BKT / 16][4], int r0, int Nb, int k0, int kend, int tid) {
#pragma unroll
    for (int j = 0; j < BKT / 16; ++j) {
        const int q = tid + 256 * j;
        const int krow = q >> 4, rq = (q & 15) * 4;
        const int e = Nb - (r0 + rq);
        if (k0 + krow < kend && e >= 0 && e < 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[j][t] = (t == e) ? 1.f : v[j][t];
        }
    }
}

template <bool KMAJOR, int BKT>
__device__ __forceinline__ void tile_store(float (*tile)[LDS_LD], const float v[BKT / 16][4], int tid) {
#pragma unroll
    for (int j = 0; j < BKT / 16; ++j) {
        const int q = tid + 256 * j;
        if (!KMAJOR) {
            const int row = q / (BKT / 4), kq = (q % (BKT / 4)) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[kq + e][row] = v[j][e];
        } else {
            const int krow = q >> 4, rq = (q & 15) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[krow][rq + e] = v[j][e];
        }
    }
}

template <typename T, bool A_KMAJOR, bool B_KMAJOR, int EPI, int BKT>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& a, int bx, int by, int bz, float (*As)[LDS_LD],
                                               float (*Bs)[LDS_LD]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i32 = lane & 31, h = lane >> 5;
    const int m0 = by * BM, n0 = bx * BN, z = bz;
    const int kbeg = z * a.k_per_split;
    const int kend = min(a.K, kbeg + a.k_per_split);
    const T* __restrict__ A = static_cast<const T*>(a.A);
    const T* __restrict__ B = static_cast<const T*>(a.B);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    constexpr bool CSUM = A_KMAJOR && B_KMAJOR && EPI == EPI_NONE;   // the weight-gradient instantiation
    const bool csum = CSUM && a.csum != nullptr;
    const int nb_real = csum ? a.Nb : a.N;                            // columns B really has
    float ra[BKT / 16][4], rb[BKT / 16][4];
    float asub[4] = {0.f, 0.f, 0.f, 0.f}, bsub[4] = {0.f, 0.f, 0.f, 0.f};
    const float* aks = nullptr;
    const float* bks = nullptr;
    if (A_KMAJOR && a.a_sub) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int r = m0 + (tid & 15) * 4 + e; asub[e] = r < a.M ? a.a_sub[r] : 0.f; }
        aks = asub;
    }
    if (B_KMAJOR && a.b_sub) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int r = n0 + (tid & 15) * 4 + e; bsub[e] = r < nb_real ? a.b_sub[r] : 0.f; }
        bks = bsub;
    }
    const float* asr = A_KMAJOR ? nullptr : a.a_sub;
    const float* bsr = B_KMAJOR ? nullptr : a.b_sub;
    tile_load<T, A_KMAJOR, BKT>(ra, A, a.amap, m0, a.M, kbeg, kend, a.vecA, tid, asr, aks);
    tile_load<T, B_KMAJOR, BKT>(rb, B, a.bmap, n0, nb_real, kbeg, kend, a.vecB, tid, bsr, bks);
    if (CSUM && csum) tile_ones<BKT>(rb, n0, a.Nb, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += BKT) {
        tile_store<A_KMAJOR, BKT>(As, ra, tid);
        tile_store<B_KMAJOR, BKT>(Bs, rb, tid);
        __syncthreads();
        if (k0 + BKT < kend) {      // next tile's loads fly during this tile's MFMAs
            tile_load<T, A_KMAJOR, BKT>(ra, A, a.amap, m0, a.M, k0 + BKT, kend, a.vecA, tid, asr, aks);
            tile_load<T, B_KMAJOR, BKT>(rb, B, a.bmap, n0, nb_real, k0 + BKT, kend, a.vecB, tid, bsr, bks);
            if (CSUM && csum) tile_ones<BKT>(rb, n0, a.Nb, k0 + BKT, kend, tid);
        }
#pragma unroll
        for (int s = 0; s < BKT / 2; ++s) {
            const float av = As[2 * s + h][wr * 32 + i32];
            const float bv = Bs[2 * s + h][wc * 32 + i32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int n = n0 + wc * 32 + i32;
    if (n < a.N) {          // (no early return: the row-block kernel below walks several tiles between barriers)
        float bias = 0.f;
        if (EPI != EPI_NONE && a.splits == 1) bias = to_f32(static_cast<const T*>(a.bias)[n]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= a.M) continue;
            float v = acc[r];
            if (a.splits > 1) {
                a.slab[((long long)z * a.M + m) * a.N + n] = v;
            } else {
                if (EPI != EPI_NONE) v += bias;
                if (EPI == EPI_BIAS_TANH) v = tanhf(v);
                if (CSUM && csum && n == a.Nb) static_cast<T*>(a.csum)[m] = from_f32<T>(v);
                else static_cast<T*>(a.C)[a.cmap(m) + n] = from_f32<T>(v);
            }
        }
    }
}

// Decoder forward in ONE launch: pre = z Wd^T + bd for a block of 64 rows over ALL d columns, then LayerNorm of those
// rows by the same workgroup while they are still in L2 (autoencoders.py:38-39 and :66-67 back to back).  Saves the
// ln_fwd launch and its read of `pre` from HBM; `pre` is still written (the backward needs it).  The LayerNorm code
// is ln_fwd_vec_kernel's (rowops.hip: one wave per row, same load pattern, same reductions), so the results are
// bit-identical to the two-launch form.  d % 4 == 0, d <= 2048, K in one split.
struct LnEpilogue { const void* gamma; const void* beta; void* out; float* mean; float* rstd; float eps; };

__device__ __forceinline__ void store4_out(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4_out(bf16* p, const float v[4]) {
    uint2 u;
    u.x = (uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[0])) | ((uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[1])) << 16);
    u.y = (uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[2])) | ((uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[3])) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}

// (kept out of line: inlined into the column-tile loop the tile body grew from 76 to 184 VGPRs and lost the occupancy
//  that hides its loads)
template <typename T, int BKT>
__device__ __noinline__ void gemm_tile_call(const GemmArgs& a, int bx, int by, float (*As)[LDS_LD], float (*Bs)[LDS_LD]) {
    gemm_tile_body<T, false, false, EPI_BIAS, BKT>(a, bx, by, 0, As, Bs);
}

template <typename T, int BKT>
__global__ TVAE_GEMM_BOUNDS void gemm_rows_ln_kernel(GemmArgs a, LnEpilogue e) {
    __shared__ float As[BKT][LDS_LD];
    __shared__ float Bs[BKT][LDS_LD];
    const int nx = (a.N + BN - 1) / BN;
#pragma unroll 1
    for (int bx = 0; bx < nx; ++bx) gemm_tile_call<T, BKT>(a, bx, blockIdx.x, As, Bs);
    __syncthreads();            // this workgroup's stores of `pre` are visible to all of its waves
    // The row is read three times from L2 (sum, centred squares, output) instead of once into 32 registers: the GEMM
    // part hides its loads through occupancy (76 VGPRs, six waves per SIMD); a register-resident row cost 184 and ran
    // the whole launch 1.8x slower than the two-launch form.  Per-lane summation order = ln_fwd_vec_kernel's.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = a.N, ng = d >> 2;
    const T* gamma = static_cast<const T*>(e.gamma);
    const T* beta = static_cast<const T*>(e.beta);
    for (int rr = wave; rr < BM; rr += 4) {
        const long long row = (long long)blockIdx.x * BM + rr;
        if (row >= a.M) break;
        const T* x = static_cast<const T*>(a.C) + a.cmap((int)row);
        float s = 0.f;
#pragma unroll 1
        for (int gidx = lane; gidx < ng; gidx += 64) {
            float v[4];
            load4_vec(x + 4 * gidx, v);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll 1
        for (int gidx = lane; gidx < ng; gidx += 64) {
            float v[4];
            load4_vec(x + 4 * gidx, v);
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = v[c] - mu; q = fmaf(t, t, q); }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)d + e.eps);
        T* o = static_cast<T*>(e.out) + row * d;
#pragma unroll 1
        for (int gidx = lane; gidx < ng; gidx += 64) {
            float v[4], gm[4], bt[4], w[4];
            load4_vec(x + 4 * gidx, v);
            load4_vec(gamma + 4 * gidx, gm);
            load4_vec(beta + 4 * gidx, bt);
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = (v[c] - mu) * rs * gm[c] + bt[c];
            store4_out(o + 4 * gidx, w);
        }
        if (lane == 0) { e.mean[row] = mu; e.rstd[row] = rs; }
    }
}

template <typename T, bool A_KMAJOR, bool B_KMAJOR, int EPI, int BKT>
__global__ TVAE_GEMM_BOUNDS void gemm_f32_mfma_kernel(GemmArgs a) {
    __shared__ float As[BKT][LDS_LD];
    __shared__ float Bs[BKT][LDS_LD];
    gemm_tile_body<T, A_KMAJOR, B_KMAJOR, EPI, BKT>(a, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Two independent products in ONE launch (the two backward products of a projection: the activation gradient
// dX = dY W, layouts <row-major, k-major>, and the weight gradient dW = dY^T X, layouts <k-major, k-major>).
// At the reference's batch sizes each fills a fraction of the chip for ~10 us; side by side they cost one.
struct GemmPairArgs {
    GemmArgs p[2];
    int gx[2], gy[2], nblk0;
    // encoder backward: the zero-fill of dh's untouched rows rides in the same launch -- its zt_blocks workgroups
    // come FIRST in the grid, so the HBM-bound stores stream while the latency-bound products wait on their loads
    ZeroTail zt; int zt_blocks;
};

template <typename T, int BKT>
__global__ TVAE_GEMM_BOUNDS void gemm_pair_kernel(GemmPairArgs q) {
    __shared__ float As[BKT][LDS_LD];
    __shared__ float Bs[BKT][LDS_LD];
    int b = blockIdx.x;
    if (b < q.zt_blocks) {
        zero_tail_body(q.zt, b % q.zt.gx, b / q.zt.gx, (int)threadIdx.x);
        return;
    }
    b -= q.zt_blocks;
    if (b < q.nblk0) {
        const int bx = b % q.gx[0], t = b / q.gx[0];
        gemm_tile_body<T, false, true, EPI_NONE, BKT>(q.p[0], bx, t % q.gy[0], t / q.gy[0], As, Bs);
    } else {
        b -= q.nblk0;
        const int bx = b % q.gx[1], t = b / q.gx[1];
        gemm_tile_body<T, true, true, EPI_NONE, BKT>(q.p[1], bx, t % q.gy[1], t / q.gy[1], As, Bs);
    }
}

template <typename T, int EPI>
__device__ __forceinline__ void reduce_slabs_body(const GemmArgs& a, long long first, long long stride) {
    const long long total = (long long)a.M * a.N;
    for (long long idx = first; idx < total; idx += stride) {
        const int m = (int)(idx / a.N), n = (int)(idx % a.N);
        float v = 0.f;
#pragma unroll 8
        for (int z = 0; z < a.splits; ++z) v += a.slab[(long long)z * total + idx];      // loads of 8 slabs in flight
        if (EPI != EPI_NONE) v += to_f32(static_cast<const T*>(a.bias)[n]);
        if (EPI == EPI_BIAS_TANH) v = tanhf(v);
        if (EPI == EPI_NONE && a.csum != nullptr && n == a.Nb) static_cast<T*>(a.csum)[m] = from_f32<T>(v);
        else static_cast<T*>(a.C)[a.cmap(m) + n] = from_f32<T>(v);
    }
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_reduce_slabs_kernel(GemmArgs a) {
    reduce_slabs_body<T, EPI>(a, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256);
}

// slab sums of both products of a pair: the first nblk0 workgroups take product 0
template <typename T>
__global__ __launch_bounds__(256) void gemm_reduce_pair_kernel(GemmPairArgs q) {
    const int b = blockIdx.x;
    if (b < q.nblk0) reduce_slabs_body<T, EPI_NONE>(q.p[0], (long long)b * 256 + threadIdx.x, (long long)q.nblk0 * 256);
    else reduce_slabs_body<T, EPI_NONE>(q.p[1], (long long)(b - q.nblk0) * 256 + threadIdx.x,
                                        (long long)(gridDim.x - q.nblk0) * 256);
}

// split count and K range per split for K tiles of depth `bk`
void choose_split_bk(int M, int N, int K, int bk, int* splits, int* k_per_split) {
    const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int s = 1;
    // ... and the same for launches of up to a few workgroups per CU whose K is in the thousands (the D x D products of the
    // MMD's closed form and the weight gradients at D, d_model = 1024: 256 tiles, K = the row count -- one workgroup per CU
    // walking K = 65 536 alone ran at a sixth of the split form's rate, BASELINE cfg5)
    if ((tiles < 256 && K > (bk == BK ? 256 : bk)) || (tiles < TVAE_SPLIT_TARGET && K >= 8192)) {
        // enough workgroups for ~8 per CU: with one 32x32 accumulator per wave the kernel hides its global
        // loads through occupancy (2-3 workgroups per CU left the weight-gradient products at 30 % MFMA-busy)
        s = (int)((TVAE_SPLIT_TARGET + tiles - 1) / tiles);
        const int max_s = bk == BK ? (K + 127) / 128 : (K + bk - 1) / bk;
        if (s > max_s) s = max_s;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
    }
    int kps = (K + s - 1) / s;
    kps = (kps + bk - 1) / bk * bk;
    if (kps < bk) kps = bk;
    *k_per_split = kps;
    *splits = (K + kps - 1) / kps;
    if (*splits < 1) *splits = 1;
}

int deep_max_blocks() {
#ifdef TVAE_TUNING
    static const int v = getenv("TVAE_GEMM_DEEP_MAX_BLOCKS") ? atoi(getenv("TVAE_GEMM_DEEP_MAX_BLOCKS")) : DEEP_MAX_BLOCKS;
    return v;
#else
    return DEEP_MAX_BLOCKS;
#endif
}

// Small launches take the deep K tiles (and split K down to one tile per workgroup); everything else the 32-deep ones.
void choose_split(int M, int N, int K, int* splits, int* k_per_split, int* deep) {
    const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    choose_split_bk(M, N, K, BK_DEEP, splits, k_per_split);
    *deep = 1;
    if (tiles * *splits > deep_max_blocks()) {
        choose_split_bk(M, N, K, BK, splits, k_per_split);
        *deep = 0;
    }
}

template <typename T, bool AK, bool BKM, int BKT>
int launch_epi_bk(const GemmArgs& a, int epi, dim3 grid, hipStream_t s) {
    switch (epi) {
        case EPI_NONE: hipLaunchKernelGGL((gemm_f32_mfma_kernel<T, AK, BKM, EPI_NONE, BKT>), grid, dim3(256), 0, s, a); break;
        case EPI_BIAS: hipLaunchKernelGGL((gemm_f32_mfma_kernel<T, AK, BKM, EPI_BIAS, BKT>), grid, dim3(256), 0, s, a); break;
        case EPI_BIAS_TANH: hipLaunchKernelGGL((gemm_f32_mfma_kernel<T, AK, BKM, EPI_BIAS_TANH, BKT>), grid, dim3(256), 0, s, a); break;
        default: return fail(TVAE_ERR_ARG, "gemm: bad epilogue %d", epi);
    }
    return TVAE_OK;
}

template <typename T, bool AK, bool BKM>
int launch_epi(const GemmArgs& a, int epi, dim3 grid, hipStream_t s) {
    return a.deep ? launch_epi_bk<T, AK, BKM, BK_DEEP>(a, epi, grid, s) : launch_epi_bk<T, AK, BKM, BK>(a, epi, grid, s);
}

template <typename T>
int launch_typed(const GemmDesc& g, GemmArgs& a, hipStream_t s) {
    dim3 grid((a.N + BN - 1) / BN, (g.M + BM - 1) / BM, a.splits);
    // the split-K main pass writes raw partials; the epilogue runs in the reduce pass
    const int epi_main = a.splits > 1 ? EPI_NONE : g.epilogue;
    int st;
    if (!g.a_kmajor && !g.b_kmajor) st = launch_epi<T, false, false>(a, epi_main, grid, s);
    else if (!g.a_kmajor && g.b_kmajor) st = launch_epi<T, false, true>(a, epi_main, grid, s);
    else if (g.a_kmajor && g.b_kmajor) st = launch_epi<T, true, true>(a, epi_main, grid, s);
    else return fail(TVAE_ERR_ARG, "gemm: layout (A k-major, B n-major) is not instantiated");
    TVAE_PROPAGATE(st);
    if (a.splits > 1) {
        const long long total = (long long)a.M * a.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        switch (g.epilogue) {
            case EPI_NONE: hipLaunchKernelGGL((gemm_reduce_slabs_kernel<T, EPI_NONE>), dim3(blocks), dim3(256), 0, s, a); break;
            case EPI_BIAS: hipLaunchKernelGGL((gemm_reduce_slabs_kernel<T, EPI_BIAS>), dim3(blocks), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((gemm_reduce_slabs_kernel<T, EPI_BIAS_TANH>), dim3(blocks), dim3(256), 0, s, a); break;
        }
    }
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

}  // namespace

int64_t gemm_slab_floats(int M, int N, int K) {
    int s, kps, deep;
    choose_split(M, N, K, &s, &kps, &deep);
    return s > 1 ? (int64_t)s * M * N : 0;
}

namespace {
int prepare_args(const GemmDesc& g, GemmArgs& a);
}

int gemm_launch(const GemmDesc& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return TVAE_OK;
    GemmArgs a;
    TVAE_PROPAGATE(prepare_args(g, a));
    if (g.dtype == TVAE_F32) return launch_typed<float>(g, a, stream);
    if (g.dtype == TVAE_BF16) return launch_typed<bf16>(g, a, stream);
    return fail(TVAE_ERR_ARG, "gemm: unknown dtype %d", g.dtype);
}

// pre = A B^T + bias (row-major A (M, K), row-major B (N, K): the decoder's linear), then out = LayerNorm(pre) gamma + beta
// with per-row mean / rstd, in one launch.  Returns TVAE_OK with *done = false when the shape is outside the fused
// form (few row blocks: the two-launch form spreads the tiles over the chip; unaligned or wide rows; a K that wants
// splitting) -- the caller then runs the two launches.
int gemm_bias_layernorm(const GemmDesc& g, const void* gamma, const void* beta, float eps, void* out, float* mean,
                        float* rstd, hipStream_t stream, bool* done) {
    *done = false;
    const int row_blocks = (g.M + BM - 1) / BM;
    const bool aligned = aligned16(g.C) && aligned16(out) && aligned16(gamma) && aligned16(beta);
    // Opt-in (environment TVAE_FUSED_DECODE_LN=1): measured at BASELINE cfg4 (3 840 row blocks, d = 1024, K = 768) the
    // one launch takes 7.0 ms against 4.1 + 0.5 ms for the two -- a workgroup walks its sixteen column tiles one after
    // the other and the out-of-line tile body runs at four waves per SIMD instead of six -- so the default stays two.
    const char* opt = getenv("TVAE_FUSED_DECODE_LN");
    if (!opt || opt[0] != '1') return TVAE_OK;
    if (g.a_kmajor || g.b_kmajor || g.epilogue != EPI_BIAS || g.N % 4 != 0 || g.N > 2048 || !aligned ||
        g.cmap.grp != 1 || g.cmap.so != g.N || row_blocks < TVAE_FUSED_LN_MIN_ROW_BLOCKS)
        return TVAE_OK;
    GemmArgs a;
    TVAE_PROPAGATE(prepare_args(g, a));
    // one split, 32-deep tiles: with >= 1024 row blocks occupancy hides the loads
    a.splits = 1; a.k_per_split = (g.K + BK - 1) / BK * BK; a.deep = 0; a.slab = nullptr;
    LnEpilogue e{gamma, beta, out, mean, rstd, eps};
    if (g.dtype == TVAE_F32) hipLaunchKernelGGL((gemm_rows_ln_kernel<float, BK>), dim3(row_blocks), dim3(256), 0, stream, a, e);
    else if (g.dtype == TVAE_BF16) hipLaunchKernelGGL((gemm_rows_ln_kernel<bf16, BK>), dim3(row_blocks), dim3(256), 0, stream, a, e);
    else return fail(TVAE_ERR_ARG, "gemm: unknown dtype %d", g.dtype);
    TVAE_CHECK_HIP(hipGetLastError());
    *done = true;
    return TVAE_OK;
}

namespace {
template <typename T>
int launch_pair_typed(const GemmPairArgs& q, int nblk, hipStream_t s) {
    if (q.p[0].deep) hipLaunchKernelGGL((gemm_pair_kernel<T, BK_DEEP>), dim3(nblk + q.zt_blocks), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((gemm_pair_kernel<T, BK>), dim3(nblk + q.zt_blocks), dim3(256), 0, s, q);
    const GemmArgs& a0 = q.p[0];
    const GemmArgs& a1 = q.p[1];
    auto rblocks = [](const GemmArgs& a) {
        const long long total = (long long)a.M * a.N;
        long long b = (total + 255) / 256;
        return (int)(b > 2048 ? 2048 : b);
    };
    if (a0.splits > 1 && a1.splits > 1) {
        GemmPairArgs r = q;
        r.zt_blocks = 0;
        r.nblk0 = rblocks(a0);
        hipLaunchKernelGGL((gemm_reduce_pair_kernel<T>), dim3(r.nblk0 + rblocks(a1)), dim3(256), 0, s, r);
    } else if (a0.splits > 1) {
        hipLaunchKernelGGL((gemm_reduce_slabs_kernel<T, EPI_NONE>), dim3(rblocks(a0)), dim3(256), 0, s, a0);
    } else if (a1.splits > 1) {
        hipLaunchKernelGGL((gemm_reduce_slabs_kernel<T, EPI_NONE>), dim3(rblocks(a1)), dim3(256), 0, s, a1);
    }
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}
}  // namespace

// g0: activation gradient (A row-major, B k-major), g1: weight gradient (A and B k-major), both without epilogue
// and of one dtype.  Small launches share one grid; large ones (or other layouts) go out one after the other.
int gemm_launch_pair(const GemmDesc& g0, const GemmDesc& g1, hipStream_t stream, const ZeroTail* zt, bool* zt_done) {
    if (zt_done) *zt_done = false;
    const bool shapes_ok = g0.M > 0 && g0.N > 0 && g1.M > 0 && g1.N > 0;
    const bool layouts_ok = !g0.a_kmajor && g0.b_kmajor && g1.a_kmajor && g1.b_kmajor &&
                            g0.epilogue == EPI_NONE && g1.epilogue == EPI_NONE && g0.dtype == g1.dtype;
    if (shapes_ok && layouts_ok) {
        GemmPairArgs q;
        TVAE_PROPAGATE(prepare_args(g0, q.p[0]));
        TVAE_PROPAGATE(prepare_args(g1, q.p[1]));
        long long nb[2];
        const GemmDesc* gs[2] = {&g0, &g1};
        for (int i = 0; i < 2; ++i) {
            q.gx[i] = (q.p[i].N + BN - 1) / BN;
            q.gy[i] = (gs[i]->M + BM - 1) / BM;
            nb[i] = (long long)q.gx[i] * q.gy[i] * q.p[i].splits;
        }
        if (nb[0] + nb[1] <= TVAE_PAIR_MAX && q.p[0].deep == q.p[1].deep) {      // (one K-tile depth per launch)
            q.nblk0 = (int)nb[0];
            q.zt = ZeroTail{};
            q.zt_blocks = 0;
            if (zt && zt->gx > 0 && zt_done) {
                q.zt = *zt;
                q.zt_blocks = zt->gx * zt->gy;
                *zt_done = true;
            }
            if (g0.dtype == TVAE_F32) return launch_pair_typed<float>(q, (int)(nb[0] + nb[1]), stream);
            if (g0.dtype == TVAE_BF16) return launch_pair_typed<bf16>(q, (int)(nb[0] + nb[1]), stream);
            return fail(TVAE_ERR_ARG, "gemm: unknown dtype %d", g0.dtype);
        }
    }
    TVAE_PROPAGATE(gemm_launch(g0, stream));
    return gemm_launch(g1, stream);
}

namespace {
int prepare_args(const GemmDesc& g, GemmArgs& a) {
    TVAE_REQUIRE(g.K > 0, "gemm: K must be positive (got %d)", g.K);
    TVAE_REQUIRE(g.A && g.B && g.C, "gemm: null operand");
    TVAE_REQUIRE(g.epilogue == EPI_NONE || g.bias, "gemm: epilogue needs a bias pointer");
    TVAE_REQUIRE(!g.colsum_out || (g.a_kmajor && g.b_kmajor && g.epilogue == EPI_NONE),
                 "gemm: a fused column sum needs the weight-gradient layout (A and B k-major, no epilogue)");
    a.A = g.A; a.B = g.B; a.C = g.C; a.bias = g.bias; a.slab = g.slab;
    a.M = g.M; a.N = g.N + (g.colsum_out ? 1 : 0); a.K = g.K;
    a.csum = g.colsum_out; a.Nb = g.N;
    a.a_sub = g.a_sub; a.b_sub = g.b_sub;
    a.amap = g.amap; a.bmap = g.bmap; a.cmap = g.cmap;
    choose_split(a.M, a.N, g.K, &a.splits, &a.k_per_split, &a.deep);
    if (a.splits > 1 && (!g.slab || g.slab_floats < (int64_t)a.splits * a.M * a.N)) {
        a.splits = 1;
        a.deep = 0;
        a.k_per_split = (g.K + BK - 1) / BK * BK;
    }
    const size_t align = g.dtype == TVAE_F32 ? 16 : 8;
    a.vecA = g.amap.vec_ok() && (reinterpret_cast<uintptr_t>(g.A) % align == 0);
    a.vecB = g.bmap.vec_ok() && (reinterpret_cast<uintptr_t>(g.B) % align == 0);
    return TVAE_OK;
}
}  // namespace

}  // namespace tvae
