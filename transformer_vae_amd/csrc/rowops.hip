// HBM-bound row/column passes around the projections: LayerNorm forward/backward,
// tanh backward, deterministic column reductions (bias / gamma / beta gradients) and the
// zero-fill of dh rows the latent slice never touched.  One wave (64 lanes) per row,
// coalesced along the feature dimension, fp32 math on fp32 or bf16 storage.
#include "tvae_common.h"

namespace tvae {
namespace {

constexpr int ROWS_PER_BLOCK = 4;     // one wave each
constexpr int COL_CHUNK_ROWS = 128;   // rows summed per partial in the column reductions

// ---- LayerNorm forward: autoencoders.py:64-67 (nn.LayerNorm(d, eps), biased variance) ----
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ r, const T* __restrict__ gamma,
                                                     const T* __restrict__ beta, float eps, T* __restrict__ out,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     long long rows, int d) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* x = r + row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += to_f32(x[c]);
    const float mu = wave_sum(s) / (float)d;
    float v = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = to_f32(x[c]) - mu; v += t * t; }
    const float rs = rsqrtf(wave_sum(v) / (float)d + eps);
    T* o = out + row * d;
    for (int c = lane; c < d; c += 64)
        o[c] = from_f32<T>((to_f32(x[c]) - mu) * rs * to_f32(gamma[c]) + to_f32(beta[c]));
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// ---- LayerNorm backward, input gradient ----
//   dxh = dout * gamma;  dr = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat))
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const T* __restrict__ dout, const T* __restrict__ r,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const T* __restrict__ gamma, T* __restrict__ dr,
                                                          long long rows, int d) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* g = dout + row * d;
    const T* x = r + row * d;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float dxh = to_f32(g[c]) * to_f32(gamma[c]);
        s1 += dxh;
        s2 += dxh * ((to_f32(x[c]) - mu) * rs);
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
    T* o = dr + row * d;
    for (int c = lane; c < d; c += 64) {
        const float xh = (to_f32(x[c]) - mu) * rs;
        o[c] = from_f32<T>(rs * (to_f32(g[c]) * to_f32(gamma[c]) - s1 - xh * s2));
    }
}

// ---- register-resident variants: one wave per row, the row is read ONCE with 16-byte (fp32) / 8-byte (bf16)
//      loads and stays in registers for the statistics and the output pass.  d % 4 == 0, d <= 256 * LN_MAXG,
//      16-byte aligned rows; anything else takes the scalar kernels above.  The summation order differs from
//      the scalar kernels only in how a lane strides the row (groups of four), not in the wave reduction.
constexpr int LN_MAXG = 8;      // 4-element groups per lane -> d <= 2048

__device__ __forceinline__ void store4_vec(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4_vec(bf16* p, const float v[4]) {
    uint2 u;
    u.x = (uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[0])) | ((uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[1])) << 16);
    u.y = (uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[2])) | ((uint32_t)__bfloat16_as_ushort(__float2bfloat16(v[3])) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const T* __restrict__ r, const T* __restrict__ gamma,
                                                         const T* __restrict__ beta, float eps, T* __restrict__ out,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         long long rows, int d) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* x = r + row * d;
    const int ng = d >> 2;
    float v[LN_MAXG][4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXG; ++k) {
        const int gidx = lane + 64 * k;
        if (gidx < ng) {
            load4_vec(x + 4 * gidx, v[k]);
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        }
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXG; ++k)
        if (lane + 64 * k < ng) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = v[k][e] - mu; q = fmaf(t, t, q); }
        }
    const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
    T* o = out + row * d;
#pragma unroll
    for (int k = 0; k < LN_MAXG; ++k) {
        const int gidx = lane + 64 * k;
        if (gidx < ng) {
            float gm[4], bt[4], w[4];
            load4_vec(gamma + 4 * gidx, gm);
            load4_vec(beta + 4 * gidx, bt);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (v[k][e] - mu) * rs * gm[e] + bt[e];
            store4_vec(o + 4 * gidx, w);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_rows_vec_kernel(const T* __restrict__ dout, const T* __restrict__ r,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const T* __restrict__ gamma, T* __restrict__ dr,
                                                              long long rows, int d) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* g = dout + row * d;
    const T* x = r + row * d;
    const float mu = mean[row], rs = rstd[row];
    const int ng = d >> 2;
    float dxh[LN_MAXG][4], xh[LN_MAXG][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXG; ++k) {
        const int gidx = lane + 64 * k;
        if (gidx < ng) {
            float gv[4], gm[4];
            load4_vec(g + 4 * gidx, gv);
            load4_vec(gamma + 4 * gidx, gm);
            load4_vec(x + 4 * gidx, xh[k]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dxh[k][e] = gv[e] * gm[e];
                xh[k][e] = (xh[k][e] - mu) * rs;
                s1 += dxh[k][e];
                s2 = fmaf(dxh[k][e], xh[k][e], s2);
            }
        }
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
    T* o = dr + row * d;
#pragma unroll
    for (int k = 0; k < LN_MAXG; ++k) {
        const int gidx = lane + 64 * k;
        if (gidx < ng) {
            float w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = rs * (dxh[k][e] - s1 - xh[k][e] * s2);
            store4_vec(o + 4 * gidx, w);
        }
    }
}

// LayerNorm backward with the parameter gradients folded in: the wave that turns row r of dout into dr also holds
// dout * xhat and dout of that row, so it accumulates them per column over its `rows_per_wave` consecutive rows, the
// four waves of a workgroup add their sums through LDS in a fixed order (deterministic), and ONE partial per
// workgroup goes out for col_final_pair_kernel.  Replaces a second pass over dout and the pre-norm activations
// (col_partial_kernel<T, 1>) and one of the two final launches.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_fused_vec_kernel(const T* __restrict__ dout, const T* __restrict__ r,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const T* __restrict__ gamma, T* __restrict__ dr,
                                                               long long rows, int d, int rows_per_wave,
                                                               float* __restrict__ part_g, float* __restrict__ part_b) {
    extern __shared__ float ln_sm[];      // [ROWS_PER_BLOCK][2][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = d >> 2;
    const long long row0 = ((long long)blockIdx.x * ROWS_PER_BLOCK + wave) * rows_per_wave;
    const long long row1 = min(rows, row0 + (long long)rows_per_wave);
    float ag[LN_MAXG][4], ab[LN_MAXG][4];
#pragma unroll
    for (int k = 0; k < LN_MAXG; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; }
    }
    for (long long row = row0; row < row1; ++row) {
        const T* g = dout + row * d;
        const T* x = r + row * d;
        const float mu = mean[row], rs = rstd[row];
        float dxh[LN_MAXG][4], xh[LN_MAXG][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXG; ++k) {
            const int gidx = lane + 64 * k;
            if (gidx < ng) {
                float gv[4], gm[4];
                load4_vec(g + 4 * gidx, gv);
                load4_vec(gamma + 4 * gidx, gm);      // (L1/L2-resident; in the loop so that it flies with the row)
                load4_vec(x + 4 * gidx, xh[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dxh[k][e] = gv[e] * gm[e];
                    xh[k][e] = (xh[k][e] - mu) * rs;
                    s1 += dxh[k][e];
                    s2 = fmaf(dxh[k][e], xh[k][e], s2);
                    ag[k][e] = fmaf(gv[e], xh[k][e], ag[k][e]);
                    ab[k][e] += gv[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
        T* o = dr + row * d;
#pragma unroll
        for (int k = 0; k < LN_MAXG; ++k) {
            const int gidx = lane + 64 * k;
            if (gidx < ng) {
                float w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = rs * (dxh[k][e] - s1 - xh[k][e] * s2);
                store4_vec(o + 4 * gidx, w);
            }
        }
    }
    // workgroup sum: every wave parks its column sums in LDS (a wave without rows parks zeros), one barrier,
    // then the four are added in wave order
    float* mine = ln_sm + (size_t)wave * 2 * d;
#pragma unroll
    for (int k = 0; k < LN_MAXG; ++k) {
        const int gidx = lane + 64 * k;
        if (gidx < ng) {
            *reinterpret_cast<float4*>(mine + 4 * gidx) = make_float4(ag[k][0], ag[k][1], ag[k][2], ag[k][3]);
            *reinterpret_cast<float4*>(mine + d + 4 * gidx) = make_float4(ab[k][0], ab[k][1], ab[k][2], ab[k][3]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += 256) {
        float v = ln_sm[c];
#pragma unroll
        for (int w = 1; w < ROWS_PER_BLOCK; ++w) v += ln_sm[(size_t)w * 2 * d + c];
        if (c < d) part_g[(long long)blockIdx.x * d + c] = v;
        else part_b[(long long)blockIdx.x * d + (c - d)] = v;
    }
}

inline bool ln_vec_ok(int64_t d, int dtype, const void* a, const void* b, const void* c, const void* e) {
    const uintptr_t align = dtype == TVAE_F32 ? 16 : 8;
    auto ok = [&](const void* p) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % align == 0; };
    return d % 4 == 0 && d <= 256 * LN_MAXG && ok(a) && ok(b) && ok(c) && ok(e);
}

// ---- column reductions: partial[chunk][c] over COL_CHUNK_ROWS rows, then a fixed-order sum ----
// MODE 0: sum_r x[r][c];  MODE 1: (sum_r dout*xhat, sum_r dout) for LayerNorm gamma/beta.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const T* __restrict__ x, const T* __restrict__ r,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          long long rows, int cols, float* __restrict__ part0,
                                                          float* __restrict__ part1) {
    __shared__ float sm[2][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long rbeg = (long long)blockIdx.y * COL_CHUNK_ROWS;
    const long long rend = min(rows, rbeg + (long long)COL_CHUNK_ROWS);
    float a0 = 0.f, a1 = 0.f;
    if (c < cols) {
#pragma unroll 8      // eight rows in flight per thread: the loop is a chain of dependent-latency loads otherwise
        for (long long row = rbeg + ry; row < rend; row += 4) {
            const float v = to_f32(x[row * cols + c]);
            if (MODE == 0) {
                a0 += v;
            } else {
                a0 += v * ((to_f32(r[row * cols + c]) - mean[row]) * rstd[row]);
                a1 += v;
            }
        }
    }
    sm[0][ry][cx] = a0;
    sm[1][ry][cx] = a1;
    __syncthreads();
    if (ry == 0 && c < cols) {
        part0[(long long)blockIdx.y * cols + c] = (sm[0][0][cx] + sm[0][1][cx]) + (sm[0][2][cx] + sm[0][3][cx]);
        if (MODE == 1)
            part1[(long long)blockIdx.y * cols + c] = (sm[1][0][cx] + sm[1][1][cx]) + (sm[1][2][cx] + sm[1][3][cx]);
    }
}

// out[c] = sum over chunks (fp64, fixed order): 64 columns per workgroup, 16 row-lanes stride the chunks
template <typename T>
__global__ __launch_bounds__(1024) void col_final_kernel(const float* __restrict__ part, int chunks, int cols,
                                                         T* __restrict__ out) {
    __shared__ double red[16][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s = 0.0;
    if (c < cols)
        for (int k = ry; k < chunks; k += 16) s += (double)part[(long long)k * cols + c];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < cols) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        out[c] = from_f32<T>((float)t);
    }
}

// the two column sums of a LayerNorm backward (gamma: blockIdx.y = 0, beta: 1) in one launch
template <typename T>
__global__ __launch_bounds__(1024) void col_final_pair_kernel(const float* __restrict__ part0, const float* __restrict__ part1,
                                                              int chunks, int cols, T* __restrict__ out0, T* __restrict__ out1) {
    __shared__ double red[16][64];
    const float* __restrict__ part = blockIdx.y ? part1 : part0;
    T* __restrict__ out = blockIdx.y ? out1 : out0;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s = 0.0;
    if (c < cols) {
#pragma unroll 8
        for (int k = ry; k < chunks; k += 16) s += (double)part[(long long)k * cols + c];
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < cols) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        out[c] = from_f32<T>((float)t);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                                       T* __restrict__ dpre, long long count) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float zv = to_f32(z[i]);
        dpre[i] = from_f32<T>(to_f32(dz[i]) * (1.f - zv * zv));
    }
}

// dpre = (dz + *scale * dx) * (1 - z^2): the decoder's and the regulariser's shares of the latent cotangent meet here
// (the regulariser's arrives unscaled, with its upstream gradient as a device scalar)
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_sum_kernel(const T* __restrict__ dz, const T* __restrict__ dx,
                                                           const float* __restrict__ scale, const T* __restrict__ z,
                                                           T* __restrict__ dpre, long long count) {
    const float sc = *scale;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float zv = to_f32(z[i]);
        dpre[i] = from_f32<T>(fmaf(sc, to_f32(dx[i]), to_f32(dz[i])) * (1.f - zv * zv));
    }
}

// dh[b, n_eff:S, :] = 0 (zero_tail_body, tvae_common.h)
__global__ __launch_bounds__(256) void zero_tail_kernel(ZeroTail z) {
    zero_tail_body(z, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x);
}

inline int row_blocks(int64_t rows) { return (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK); }
inline int col_chunks(int64_t rows) { return (int)((rows + COL_CHUNK_ROWS - 1) / COL_CHUNK_ROWS); }

// rows per wave of the fused LayerNorm backward: one at the reference's batch sizes (latency-bound: as many waves
// as rows), more once there are enough rows to fill the chip, so that the partials stay small next to the input
inline int ln_fused_rows_per_wave(int64_t rows) {
    const int64_t r = (rows + 4095) / 4096;
    return (int)(r < 1 ? 1 : r);
}
inline int ln_fused_blocks(int64_t rows) {
    const int64_t per_block = (int64_t)ROWS_PER_BLOCK * ln_fused_rows_per_wave(rows);
    return (int)((rows + per_block - 1) / per_block);
}

}  // namespace

int64_t layernorm_bwd_scratch_floats(int64_t rows, int64_t d) {
    const int64_t fused = 2 * (int64_t)ln_fused_blocks(rows) * d;
    const int64_t plain = 2 * (int64_t)col_chunks(rows) * d;
    return fused > plain ? fused : plain;
}

// dr and (dgamma, dbeta) in one pass over dout and the pre-norm activations + one final launch; returns
// TVAE_ERR_ARG (without a message change) when the shape needs the separate kernels -- callers test
// layernorm_bwd_can_fuse first.
bool layernorm_bwd_can_fuse(const void* dout, const void* r, const void* gamma, const void* dr, int64_t d, int dtype) {
    return ln_vec_ok(d, dtype, dout, r, gamma, dr) && ROWS_PER_BLOCK * 2 * d * 4 <= 64 * 1024;
}

int layernorm_bwd_fused(const void* dout, const void* r, const float* mean, const float* rstd, const void* gamma,
                        void* dr, void* dgamma, void* dbeta, int64_t rows, int64_t d, int dtype, float* scratch,
                        hipStream_t s) {
    TVAE_REQUIRE(rows > 0 && d > 0, "layernorm_bwd_fused: empty input");
    TVAE_REQUIRE(layernorm_bwd_can_fuse(dout, r, gamma, dr, d, dtype), "layernorm_bwd_fused: shape not supported");
    const int rpw = ln_fused_rows_per_wave(rows), blocks = ln_fused_blocks(rows);
    float* p0 = scratch;
    float* p1 = scratch + (int64_t)blocks * d;
    const size_t lds = (size_t)ROWS_PER_BLOCK * 2 * d * 4;
    const dim3 fgrid((unsigned)((d + 63) / 64), 2);
    if (dtype == TVAE_F32) {
        hipLaunchKernelGGL(ln_bwd_fused_vec_kernel<float>, dim3(blocks), dim3(256), lds, s, (const float*)dout,
                           (const float*)r, mean, rstd, (const float*)gamma, (float*)dr, (long long)rows, (int)d, rpw, p0, p1);
        hipLaunchKernelGGL(col_final_pair_kernel<float>, fgrid, dim3(1024), 0, s, p0, p1, blocks, (int)d, (float*)dgamma,
                           (float*)dbeta);
    } else {
        hipLaunchKernelGGL(ln_bwd_fused_vec_kernel<bf16>, dim3(blocks), dim3(256), lds, s, (const bf16*)dout,
                           (const bf16*)r, mean, rstd, (const bf16*)gamma, (bf16*)dr, (long long)rows, (int)d, rpw, p0, p1);
        hipLaunchKernelGGL(col_final_pair_kernel<bf16>, fgrid, dim3(1024), 0, s, p0, p1, blocks, (int)d, (bf16*)dgamma,
                           (bf16*)dbeta);
    }
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

namespace {
}  // namespace

int layernorm_fwd(const void* r, const void* gamma, const void* beta, float eps, void* out, float* mean, float* rstd,
                  int64_t rows, int64_t d, int dtype, hipStream_t s) {
    if (rows == 0) return TVAE_OK;
    if (ln_vec_ok(d, dtype, r, gamma, beta, out)) {
        if (dtype == TVAE_F32)
            hipLaunchKernelGGL(ln_fwd_vec_kernel<float>, dim3(row_blocks(rows)), dim3(256), 0, s, (const float*)r,
                               (const float*)gamma, (const float*)beta, eps, (float*)out, mean, rstd, (long long)rows, (int)d);
        else
            hipLaunchKernelGGL(ln_fwd_vec_kernel<bf16>, dim3(row_blocks(rows)), dim3(256), 0, s, (const bf16*)r,
                               (const bf16*)gamma, (const bf16*)beta, eps, (bf16*)out, mean, rstd, (long long)rows, (int)d);
        TVAE_CHECK_HIP(hipGetLastError());
        return TVAE_OK;
    }
    if (dtype == TVAE_F32)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(row_blocks(rows)), dim3(256), 0, s, (const float*)r,
                           (const float*)gamma, (const float*)beta, eps, (float*)out, mean, rstd, (long long)rows, (int)d);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<bf16>, dim3(row_blocks(rows)), dim3(256), 0, s, (const bf16*)r,
                           (const bf16*)gamma, (const bf16*)beta, eps, (bf16*)out, mean, rstd, (long long)rows, (int)d);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

int layernorm_bwd_rows(const void* dout, const void* r, const float* mean, const float* rstd, const void* gamma,
                       void* dr, int64_t rows, int64_t d, int dtype, hipStream_t s) {
    if (rows == 0) return TVAE_OK;
    if (ln_vec_ok(d, dtype, dout, r, gamma, dr)) {
        if (dtype == TVAE_F32)
            hipLaunchKernelGGL(ln_bwd_rows_vec_kernel<float>, dim3(row_blocks(rows)), dim3(256), 0, s, (const float*)dout,
                               (const float*)r, mean, rstd, (const float*)gamma, (float*)dr, (long long)rows, (int)d);
        else
            hipLaunchKernelGGL(ln_bwd_rows_vec_kernel<bf16>, dim3(row_blocks(rows)), dim3(256), 0, s, (const bf16*)dout,
                               (const bf16*)r, mean, rstd, (const bf16*)gamma, (bf16*)dr, (long long)rows, (int)d);
        TVAE_CHECK_HIP(hipGetLastError());
        return TVAE_OK;
    }
    if (dtype == TVAE_F32)
        hipLaunchKernelGGL(ln_bwd_rows_kernel<float>, dim3(row_blocks(rows)), dim3(256), 0, s, (const float*)dout,
                           (const float*)r, mean, rstd, (const float*)gamma, (float*)dr, (long long)rows, (int)d);
    else
        hipLaunchKernelGGL(ln_bwd_rows_kernel<bf16>, dim3(row_blocks(rows)), dim3(256), 0, s, (const bf16*)dout,
                           (const bf16*)r, mean, rstd, (const bf16*)gamma, (bf16*)dr, (long long)rows, (int)d);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

int64_t colreduce_scratch_floats(int64_t rows, int64_t cols) { return (int64_t)col_chunks(rows) * cols; }

int colsum(const void* x, int64_t rows, int64_t cols, void* out, int dtype, float* scratch, hipStream_t s) {
    TVAE_REQUIRE(rows > 0 && cols > 0, "colsum: empty input");
    const int chunks = col_chunks(rows);
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)chunks);
    if (dtype == TVAE_F32) {
        hipLaunchKernelGGL((col_partial_kernel<float, 0>), grid, dim3(256), 0, s, (const float*)x, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (long long)rows, (int)cols, scratch, (float*)nullptr);
        hipLaunchKernelGGL(col_final_kernel<float>, dim3((unsigned)((cols + 63) / 64)), dim3(1024), 0, s, scratch, chunks,
                           (int)cols, (float*)out);
    } else {
        hipLaunchKernelGGL((col_partial_kernel<bf16, 0>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (long long)rows, (int)cols, scratch, (float*)nullptr);
        hipLaunchKernelGGL(col_final_kernel<bf16>, dim3((unsigned)((cols + 63) / 64)), dim3(1024), 0, s, scratch, chunks,
                           (int)cols, (bf16*)out);
    }
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

int layernorm_bwd_params(const void* dout, const void* r, const float* mean, const float* rstd, void* dgamma,
                         void* dbeta, int64_t rows, int64_t d, int dtype, float* scratch, hipStream_t s) {
    TVAE_REQUIRE(rows > 0 && d > 0, "layernorm_bwd_params: empty input");
    const int chunks = col_chunks(rows);
    float* p0 = scratch;
    float* p1 = scratch + (int64_t)chunks * d;
    dim3 grid((unsigned)((d + 63) / 64), (unsigned)chunks);
    const dim3 fgrid((unsigned)((d + 63) / 64));
    if (dtype == TVAE_F32) {
        hipLaunchKernelGGL((col_partial_kernel<float, 1>), grid, dim3(256), 0, s, (const float*)dout, (const float*)r, mean,
                           rstd, (long long)rows, (int)d, p0, p1);
        hipLaunchKernelGGL(col_final_kernel<float>, fgrid, dim3(1024), 0, s, p0, chunks, (int)d, (float*)dgamma);
        hipLaunchKernelGGL(col_final_kernel<float>, fgrid, dim3(1024), 0, s, p1, chunks, (int)d, (float*)dbeta);
    } else {
        hipLaunchKernelGGL((col_partial_kernel<bf16, 1>), grid, dim3(256), 0, s, (const bf16*)dout, (const bf16*)r, mean,
                           rstd, (long long)rows, (int)d, p0, p1);
        hipLaunchKernelGGL(col_final_kernel<bf16>, fgrid, dim3(1024), 0, s, p0, chunks, (int)d, (bf16*)dgamma);
        hipLaunchKernelGGL(col_final_kernel<bf16>, fgrid, dim3(1024), 0, s, p1, chunks, (int)d, (bf16*)dbeta);
    }
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

int tanh_bwd(const void* dz, const void* z, void* dpre, int64_t count, int dtype, hipStream_t s) {
    if (count == 0) return TVAE_OK;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == TVAE_F32)
        hipLaunchKernelGGL(tanh_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dz, (const float*)z,
                           (float*)dpre, (long long)count);
    else
        hipLaunchKernelGGL(tanh_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)dz, (const bf16*)z,
                           (bf16*)dpre, (long long)count);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

int tanh_bwd_sum(const void* dz, const void* dx, const float* scale, const void* z, void* dpre, int64_t count, int dtype,
                 hipStream_t s) {
    if (count == 0) return TVAE_OK;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == TVAE_F32)
        hipLaunchKernelGGL(tanh_bwd_sum_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dz, (const float*)dx,
                           scale, (const float*)z, (float*)dpre, (long long)count);
    else
        hipLaunchKernelGGL(tanh_bwd_sum_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (const bf16*)dz, (const bf16*)dx,
                           scale, (const bf16*)z, (bf16*)dpre, (long long)count);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

ZeroTail zero_tail_plan(void* dh, int64_t B, int64_t S, int64_t n_eff, int64_t d, int dtype) {
    ZeroTail z{};
    if (S <= n_eff || B == 0) return z;       // gx == 0
    const int64_t esz = dtype == TVAE_F32 ? 4 : 2;
    z.dh = static_cast<char*>(dh);
    z.seq_bytes = S * d * esz;
    z.head_bytes = n_eff * d * esz;
    z.B = B;
    const long long tail_vec = (z.seq_bytes - z.head_bytes) / 16 + 1;
    // enough workgroups to fill the chip: x over one sequence's tail, y over sequences
    int gx = (int)((tail_vec + 255) / 256);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    int gy = (int)B;
    const int want = 4096 / gx;
    if (gy > want) gy = want;
    if (gy < 1) gy = 1;
    z.gx = gx; z.gy = gy;
    return z;
}

int zero_tail_rows(void* dh, int64_t B, int64_t S, int64_t n_eff, int64_t d, int dtype, hipStream_t s) {
    const ZeroTail z = zero_tail_plan(dh, B, S, n_eff, d, dtype);
    if (z.gx == 0) return TVAE_OK;
    hipLaunchKernelGGL(zero_tail_kernel, dim3(z.gx, z.gy), dim3(256), 0, s, z);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

}  // namespace tvae
