// MMD forward + gradient for latent sizes beyond the tile kernels (D > 1024): any D.
//
// The tile kernels keep a row block's features in registers, which caps D at 1024.  The reference has no
// such cap (autoencoders.py:117-136 works for any latent_size), so wider latents take this strip-wise
// route through the projection GEMM (exact fp32 MFMA, gemm.hip) — never a CPU path:
//   for each strip of R local rows (R sized so that the strip of the kernel matrix stays ~256 MiB):
//     S  = Zs Z^T | Zs Y^T            two GEMMs into one R x 2N strip
//     E  = expm1(-max((|z_i|^2 + |x_j|^2 - 2 S) / D^2, 0)), prior columns negated, in place;
//          per-row fp64 sums of Q = E + t (zz, zy) and the signed row sum R_i of E
//          (the part of the sums linear in t is added in closed form, see em1_split)   one workgroup per row
//     G  = E[:, :N] Z + E[:, N:] Y    two GEMMs (split-K slabs, deterministic)
//     dz = grad_scale (-z R + (colsum z - colsum y) + G)
//   the y-y sum runs the same strips with one GEMM and the sums only.
// Same math and the same E-form as mmd.hip (the "+1"s cancel exactly); only strips of the N x N matrices
// ever exist.  bf16 inputs are widened once to fp32 copies (exact).
#include "mmd_common.h"

namespace tvae {
namespace {

constexpr int64_t STRIP_BYTES = 256ll << 20;

struct WidePlan {
    int R;                      // rows per strip
    int64_t slab_floats;
    int64_t off_zf, off_yf, off_zn, off_yn, off_csz, off_csy, off_cszl, off_csyl, off_cscr, off_S, off_G1, off_G2, off_slab;
    int64_t off_rsum, off_rzz, off_rzy, off_ryy, bytes;
};

WidePlan wide_plan(int64_t n_total, int64_t n_rows, int64_t D, bool want_grad) {
    WidePlan p;
    int64_t R = STRIP_BYTES / (2 * n_total * 4);
    R = R / 64 * 64;
    if (R < 64) R = 64;
    if (R > n_rows) R = n_rows;
    p.R = (int)R;
    const int64_t s1 = gemm_slab_floats((int)R, (int)n_total, (int)D);
    const int64_t s2 = want_grad ? gemm_slab_floats((int)R, (int)D, (int)n_total) : 0;
    p.slab_floats = s1 > s2 ? s1 : s2;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off = align_up(off + bytes, 256); return o; };
    p.off_zf = take(n_total * D * 4);          // fp32 copies (used for bf16 inputs; sized always: the query has no dtype)
    p.off_yf = take(n_total * D * 4);
    p.off_zn = take(n_total * 4);
    p.off_yn = take(n_total * 4);
    p.off_csz = take(D * 4);
    p.off_csy = take(D * 4);
    p.off_cszl = take(D * 4);
    p.off_csyl = take(D * 4);
    p.off_cscr = take(colreduce_scratch_floats(n_total, D) * 4);
    p.off_S = take(R * 2 * n_total * 4);
    p.off_G1 = take(want_grad ? R * D * 4 : 0);
    p.off_G2 = take(want_grad ? R * D * 4 : 0);
    p.off_slab = take(p.slab_floats * 4);
    p.off_rsum = take(n_rows * 8);
    p.off_rzz = take(n_rows * 8);
    p.off_rzy = take(n_rows * 8);
    p.off_ryy = take(n_rows * 8);
    p.bytes = off;
    return p;
}

__global__ __launch_bounds__(256) void wide_cast_kernel(const bf16* __restrict__ x, float* __restrict__ out, long long count) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256)
        out[i] = to_f32(x[i]);
}

// one wave per row: out[row] = sum_c x[row][c]^2
__global__ __launch_bounds__(256) void wide_rownorm_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           long long rows, int D) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + row * D;
    float sq = 0.f;
    for (int c = lane; c < D; c += 64) sq = fmaf(p[c], p[c], sq);
    sq = wave_sum(sq);
    if (lane == 0) out[row] = sq;
}

// One workgroup per strip row i: S[i][j] (dot products) -> signed E in place; fp64 row totals.
// Columns [0, n0) belong to the first set (sign +, total -> out_a), [n0, n0 + n1) to the second (sign -, -> out_b).
__global__ __launch_bounds__(256) void wide_e_kernel(float* __restrict__ S, long long ld, const float* __restrict__ rnorm,
                                                     const float* __restrict__ cn0, int n0, const float* __restrict__ cn1,
                                                     int n1, float inv_d2, double* __restrict__ out_a,
                                                     double* __restrict__ out_b, double* __restrict__ out_signed) {
    __shared__ double red[2][256];
    const int i = blockIdx.x, tid = threadIdx.x;
    float* row = S + (long long)i * ld;
    const float rn = rnorm[i];
    double qa = 0.0, qb = 0.0, es = 0.0;      // sums of Q per set, signed sum of E
    auto eq = [](float t) -> EQ {
        if (t < EM1_POLY_MAX) return em1_split(t);
        const float e = expm1f(-t);
        return EQ{e, e + t};
    };
    for (int j = tid; j < n0; j += 256) {
        const EQ v = eq(fmaxf((rn + cn0[j] - 2.f * row[j]) * inv_d2, 0.f));
        row[j] = v.e;
        qa += (double)v.q;
        es += (double)v.e;
    }
    for (int j = tid; j < n1; j += 256) {
        const EQ v = eq(fmaxf((rn + cn1[j] - 2.f * row[n0 + j]) * inv_d2, 0.f));
        row[n0 + j] = -v.e;
        qb += (double)v.q;
        es -= (double)v.e;
    }
    __shared__ double red3[256];
    red[0][tid] = qa; red[1][tid] = qb; red3[tid] = es;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red3[tid] += red3[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        if (out_a) out_a[i] = red[0][0];
        if (out_b) out_b[i] = red[1][0];
        if (out_signed) out_signed[i] = red3[0];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wide_dz_kernel(const float* __restrict__ zrows, const double* __restrict__ rsum,
                                                      const float* __restrict__ csz, const float* __restrict__ csy,
                                                      const float* __restrict__ G1, const float* __restrict__ G2,
                                                      T* __restrict__ dz, long long rows, int D, double grad_scale) {
    const long long total = rows * D;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / D;
        const int c = (int)(idx - r * D);
        const double v = grad_scale * (-(double)zrows[idx] * rsum[r] + ((double)csz[c] - (double)csy[c]) +
                                       (double)G1[idx] + (double)G2[idx]);
        dz[idx] = from_f32<T>((float)v);
    }
}

__global__ __launch_bounds__(256) void wide_final_kernel(const double* __restrict__ rzz, const double* __restrict__ rzy,
                                                         const double* __restrict__ ryy, long long rows,
                                                         const float* __restrict__ zn, const float* __restrict__ yn,
                                                         long long n_total, long long row_offset,
                                                         const float* __restrict__ csz, const float* __restrict__ csy,
                                                         const float* __restrict__ cszl, const float* __restrict__ csyl,
                                                         int D, double* __restrict__ sums3, float* __restrict__ loss,
                                                         double loss_scale) {
    __shared__ double red[10][256];
    const int tid = threadIdx.x;
    double v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long long i = tid; i < rows; i += 256) { v[0] += rzz[i]; v[1] += rzy[i]; v[2] += ryy[i]; }
    for (long long i = tid; i < n_total; i += 256) {         // squared norms: all points / this call's rows
        const double a = (double)zn[i], b = (double)yn[i];
        v[3] += a; v[4] += b;
        if (i >= row_offset && i < row_offset + rows) { v[5] += a; v[6] += b; }
    }
    for (int c = tid; c < D; c += 256) {                     // <c_z^loc, c_z>, <c_z^loc, c_y>, <c_y^loc, c_y>
        v[7] += (double)cszl[c] * (double)csz[c];
        v[8] += (double)cszl[c] * (double)csy[c];
        v[9] += (double)csyl[c] * (double)csy[c];
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) red[k][tid] = v[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int k = 0; k < 10; ++k) red[k][tid] += red[k][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double N = (double)n_total, nl = (double)rows, inv = 1.0 / ((double)D * (double)D);
        const double szz = red[0][0] - (N * red[5][0] + nl * red[3][0] - 2.0 * red[7][0]) * inv;
        const double szy = red[1][0] - (N * red[5][0] + nl * red[4][0] - 2.0 * red[8][0]) * inv;
        const double syy = red[2][0] - (N * red[6][0] + nl * red[4][0] - 2.0 * red[9][0]) * inv;
        sums3[0] = szz; sums3[1] = szy; sums3[2] = syy;
        if (loss) *loss = (float)(loss_scale * (syy + szz - 2.0 * szy));
    }
}

int strip_gemm(const float* A, long long lda, const float* B, long long ldb, bool b_kmajor, float* C, long long ldc,
               int M, int N, int K, float* slab, int64_t slab_floats, hipStream_t s) {
    GemmDesc g{};
    g.A = A; g.B = B; g.C = C; g.bias = nullptr;
    g.M = M; g.N = N; g.K = K;
    g.a_kmajor = false; g.b_kmajor = b_kmajor;
    g.amap = dense_rows(lda); g.bmap = dense_rows(ldb); g.cmap = dense_rows(ldc);
    g.epilogue = EPI_NONE; g.dtype = TVAE_F32;
    g.slab = slab; g.slab_floats = slab_floats;
    return gemm_launch(g, s);
}

}  // namespace

int64_t mmd_wide_workspace_bytes(int64_t n_total, int64_t n_rows, int64_t D, bool want_grad) {
    return wide_plan(n_total, n_rows, D, want_grad).bytes;
}

int mmd_wide_launch(const void* z_all, const void* y_all, int64_t n_total, int64_t row_offset, int64_t n_rows,
                    int64_t D, int dtype, double* sums3, float* loss, double loss_scale, void* d_z,
                    double grad_scale, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    TVAE_REQUIRE(D < (1ll << 24), "mmd: latent dimension %lld too large", (long long)D);
    const bool want_grad = d_z != nullptr;
    const WidePlan p = wide_plan(n_total, n_rows, D, want_grad);
    if (workspace_bytes < p.bytes)
        return fail(TVAE_ERR_WORKSPACE, "mmd: workspace %lld B < required %lld B", (long long)workspace_bytes,
                    (long long)p.bytes);
    TVAE_REQUIRE(aligned16(workspace), "mmd: workspace must be 16-byte aligned");
    char* ws = static_cast<char*>(workspace);
    const int N = (int)n_total, Di = (int)D;
    const float *zf, *yf;
    if (dtype == TVAE_F32) {
        zf = static_cast<const float*>(z_all);
        yf = static_cast<const float*>(y_all);
    } else {
        float* zc = reinterpret_cast<float*>(ws + p.off_zf);
        float* yc = reinterpret_cast<float*>(ws + p.off_yf);
        const long long count = (long long)n_total * D;
        const unsigned blocks = (unsigned)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256);
        hipLaunchKernelGGL(wide_cast_kernel, dim3(blocks), dim3(256), 0, s, (const bf16*)z_all, zc, count);
        hipLaunchKernelGGL(wide_cast_kernel, dim3(blocks), dim3(256), 0, s, (const bf16*)y_all, yc, count);
        TVAE_CHECK_HIP(hipGetLastError());
        zf = zc; yf = yc;
    }
    float* zn = reinterpret_cast<float*>(ws + p.off_zn);
    float* yn = reinterpret_cast<float*>(ws + p.off_yn);
    float* csz = reinterpret_cast<float*>(ws + p.off_csz);
    float* csy = reinterpret_cast<float*>(ws + p.off_csy);
    float* cscr = reinterpret_cast<float*>(ws + p.off_cscr);
    float* S = reinterpret_cast<float*>(ws + p.off_S);
    float* G1 = reinterpret_cast<float*>(ws + p.off_G1);
    float* G2 = reinterpret_cast<float*>(ws + p.off_G2);
    float* slab = p.slab_floats ? reinterpret_cast<float*>(ws + p.off_slab) : nullptr;
    double* rsum = reinterpret_cast<double*>(ws + p.off_rsum);
    double* rzz = reinterpret_cast<double*>(ws + p.off_rzz);
    double* rzy = reinterpret_cast<double*>(ws + p.off_rzy);
    double* ryy = reinterpret_cast<double*>(ws + p.off_ryy);
    const float inv_d2 = (float)(1.0 / ((double)D * (double)D));

    const unsigned nb = (unsigned)((n_total + 3) / 4);
    hipLaunchKernelGGL(wide_rownorm_kernel, dim3(nb), dim3(256), 0, s, zf, zn, (long long)n_total, Di);
    hipLaunchKernelGGL(wide_rownorm_kernel, dim3(nb), dim3(256), 0, s, yf, yn, (long long)n_total, Di);
    TVAE_CHECK_HIP(hipGetLastError());
    float* cszl = reinterpret_cast<float*>(ws + p.off_cszl);
    float* csyl = reinterpret_cast<float*>(ws + p.off_csyl);
    TVAE_PROPAGATE(colsum(zf, n_total, D, csz, TVAE_F32, cscr, s));
    TVAE_PROPAGATE(colsum(yf, n_total, D, csy, TVAE_F32, cscr, s));
    TVAE_PROPAGATE(colsum(zf + row_offset * D, n_rows, D, cszl, TVAE_F32, cscr, s));
    TVAE_PROPAGATE(colsum(yf + row_offset * D, n_rows, D, csyl, TVAE_F32, cscr, s));

    for (int64_t r0 = 0; r0 < n_rows; r0 += p.R) {
        const int Rc = (int)(n_rows - r0 < p.R ? n_rows - r0 : p.R);
        const int64_t g0 = row_offset + r0;              // global index of the strip's first row
        const long long ld = 2ll * N;
        // z rows against z and y columns
        TVAE_PROPAGATE(strip_gemm(zf + g0 * D, D, zf, D, false, S, ld, Rc, N, Di, slab, p.slab_floats, s));
        TVAE_PROPAGATE(strip_gemm(zf + g0 * D, D, yf, D, false, S + N, ld, Rc, N, Di, slab, p.slab_floats, s));
        hipLaunchKernelGGL(wide_e_kernel, dim3(Rc), dim3(256), 0, s, S, ld, zn + g0, zn, N, yn, N, inv_d2, rzz + r0,
                           rzy + r0, rsum + r0);
        TVAE_CHECK_HIP(hipGetLastError());
        if (want_grad) {
            TVAE_PROPAGATE(strip_gemm(S, ld, zf, D, true, G1, D, Rc, Di, N, slab, p.slab_floats, s));
            TVAE_PROPAGATE(strip_gemm(S + N, ld, yf, D, true, G2, D, Rc, Di, N, slab, p.slab_floats, s));
            const long long total = (long long)Rc * D;
            const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
            if (dtype == TVAE_F32)
                hipLaunchKernelGGL(wide_dz_kernel<float>, dim3(blocks), dim3(256), 0, s, zf + g0 * D, rsum + r0, csz, csy,
                                   G1, G2, static_cast<float*>(d_z) + r0 * D, (long long)Rc, Di, grad_scale);
            else
                hipLaunchKernelGGL(wide_dz_kernel<bf16>, dim3(blocks), dim3(256), 0, s, zf + g0 * D, rsum + r0, csz, csy,
                                   G1, G2, static_cast<bf16*>(d_z) + r0 * D, (long long)Rc, Di, grad_scale);
            TVAE_CHECK_HIP(hipGetLastError());
        }
        // y rows against y columns: the sum only
        TVAE_PROPAGATE(strip_gemm(yf + g0 * D, D, yf, D, false, S, N, Rc, N, Di, slab, p.slab_floats, s));
        hipLaunchKernelGGL(wide_e_kernel, dim3(Rc), dim3(256), 0, s, S, (long long)N, yn + g0, yn, N,
                           (const float*)nullptr, 0, inv_d2, ryy + r0, (double*)nullptr, (double*)nullptr);
        TVAE_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(wide_final_kernel, dim3(1), dim3(256), 0, s, rzz, rzy, ryy, (long long)n_rows, zn, yn,
                       (long long)n_total, (long long)row_offset, csz, csy, cszl, csyl, Di, sums3, loss, loss_scale);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

}  // namespace tvae
