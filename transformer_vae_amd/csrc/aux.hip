// The two steps either side of the bottleneck that SURVEY.md 8f ranks "next", as HBM-bound row kernels:
//   * upsample  — transformers.models.funnel.modeling_funnel.upsample as called on
//     reconstructed_encoding at transformer_vae/model.py:265-271 (repeat_interleave along the sequence,
//     optional separate cls token / truncate_seq padding, cut to target_len), forward and backward;
//   * slerp     — transformer_vae/utils.py:17-34, the spherical interpolation of latent rows used by
//     trainer.py:283-301 (forward only: its callers work on detached latents).
// One wave per output row, 16 B per lane when the rows are 16-B aligned.
#include "tvae_common.h"

namespace tvae {
namespace {

struct UpsampleShape {
    long long B, n, d, stride, out_len;
    int separate_cls;
};

// 16 B of T as fp32 values (PER = 4 fp32 / 8 bf16) and back
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int PER = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float f[4]) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float f[4]) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Vec16<bf16> {
    static constexpr int PER = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float f[8]) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_bits_to_f32(w[i] & 0xffffu); f[2 * i + 1] = bf16_bits_to_f32(w[i] >> 16); }
    }
    static __device__ __forceinline__ uint4 pack(const float f[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16 lo = __float2bfloat16(f[2 * i]), hi = __float2bfloat16(f[2 * i + 1]);
            w[i] = (uint32_t) * reinterpret_cast<const uint16_t*>(&lo) | ((uint32_t) * reinterpret_cast<const uint16_t*>(&hi) << 16);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// output positions [j0, j1) that copy source row src (clipped to the output length)
__device__ __forceinline__ void up_span(const UpsampleShape& s, long long src, long long& j0, long long& j1) {
    if (s.separate_cls) {
        if (src == 0) { j0 = 0; j1 = 1; }
        else { j0 = 1 + (src - 1) * s.stride; j1 = j0 + s.stride; }
    } else {
        j0 = src * s.stride; j1 = j0 + s.stride;
    }
    if (j1 > s.out_len) j1 = s.out_len;
    if (j0 > s.out_len) j0 = s.out_len;
}
// One wave per OUTPUT row; 16 B per lane, the CH chunks of a pass are all loaded before the first store.  The up to
// `stride` waves that copy one source row sit next to each other, so all but the first read it from the caches.
// (Measured and not kept: one wave per SOURCE row storing its `stride` copies -- a quarter of the load
// instructions, but 5.0 TB/s against 5.9 here.)
template <typename T, int CH, bool STREAM>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, UpsampleShape s,
                                                           int vec) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= s.B * s.out_len) return;
    const long long b = row / s.out_len, j = row - b * s.out_len;
    // source row of output position j (-1: zero padding introduced by truncate_seq)
    long long src;
    if (s.separate_cls) {
        const long long q = (j - 1) / s.stride;
        src = j == 0 ? 0 : (q < s.n - 1 ? 1 + q : -1);
    } else {
        src = j / s.stride;
    }
    T* o = out + row * s.d;
    const T* in = x + (b * s.n + (src < 0 ? 0 : src)) * s.d;
    if (vec) {
        constexpr int PER = 16 / sizeof(T);
        const long long nch = s.d / PER;
        uint4* ov = reinterpret_cast<uint4*>(o);
        const uint4* iv = reinterpret_cast<const uint4*>(in);
        for (long long c0 = lane; c0 < nch; c0 += 64 * CH) {
            uint4 v[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) v[k] = (src >= 0 && c0 + 64 * k < nch) ? iv[c0 + 64 * k] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < CH; ++k) if (c0 + 64 * k < nch) store16<STREAM>(&ov[c0 + 64 * k], v[k]);
        }
    } else {
        for (long long c = lane; c < s.d; c += 64) o[c] = src >= 0 ? in[c] : from_f32<T>(0.f);
    }
}

// d_x[b, src] = sum of d_out over the (contiguous) output positions that copied row src, in ascending position
// order; 16 B per lane, four positions' loads in flight
template <typename T, bool STREAM>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, UpsampleShape s,
                                                           int vec) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= s.B * s.n) return;
    const long long b = row / s.n, src = row - b * s.n;
    long long j0, j1;
    up_span(s, src, j0, j1);
    const T* g = dout + b * s.out_len * s.d;
    T* o = dx + row * s.d;
    if (vec) {
        constexpr int PER = Vec16<T>::PER;
        const long long nch = s.d / PER;
        uint4* ov = reinterpret_cast<uint4*>(o);
        for (long long c = lane; c < nch; c += 64) {
            float acc[PER];
#pragma unroll
            for (int e = 0; e < PER; ++e) acc[e] = 0.f;
            long long j = j0;
            for (; j + 4 <= j1; j += 4) {
                uint4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const uint4*>(g + (j + k) * s.d)[c];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float f[PER];
                    Vec16<T>::unpack(v[k], f);
#pragma unroll
                    for (int e = 0; e < PER; ++e) acc[e] += f[e];
                }
            }
            for (; j < j1; ++j) {
                float f[PER];
                Vec16<T>::unpack(reinterpret_cast<const uint4*>(g + j * s.d)[c], f);
#pragma unroll
                for (int e = 0; e < PER; ++e) acc[e] += f[e];
            }
            store16<STREAM>(&ov[c], Vec16<T>::pack(acc));
        }
    } else {
        for (long long c = lane; c < s.d; c += 64) {
            float acc = 0.f;
            for (long long j = j0; j < j1; ++j) acc += to_f32(g[j * s.d + c]);
            o[c] = from_f32<T>(acc);
        }
    }
}

// res = sin((1-r) w)/sin(w) * t1 + sin(r w)/sin(w) * t2,  w = acos(<t1/|t1|, t2/|t2|>)   (utils.py:29-33)
// REG: both rows live in registers (16 B per lane, up to CH chunks each: dim <= 64 * CH * PER), read once;
// otherwise (unaligned or longer rows) the three passes re-read them through the caches.
template <typename T, int CH, bool REG, bool STREAM>
__global__ __launch_bounds__(256) void slerp_kernel(const T* __restrict__ t1, const T* __restrict__ t2,
                                                    const float* __restrict__ ratios, float ratio, T* __restrict__ out,
                                                    long long rows, int dim) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* a = t1 + row * dim;
    const T* b = t2 + row * dim;
    T* o = out + row * dim;
    const float r = ratios ? ratios[row] : ratio;
    if constexpr (REG) {
        constexpr int PER = Vec16<T>::PER;
        const int nch = dim / PER;
        uint4 va[CH], vb[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = lane + 64 * k;
            va[k] = c < nch ? reinterpret_cast<const uint4*>(a)[c] : make_uint4(0, 0, 0, 0);
            vb[k] = c < nch ? reinterpret_cast<const uint4*>(b)[c] : make_uint4(0, 0, 0, 0);
        }
        float fa[CH][PER], fb[CH][PER];
        float na = 0.f, nb = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            Vec16<T>::unpack(va[k], fa[k]);
            Vec16<T>::unpack(vb[k], fb[k]);
#pragma unroll
            for (int e = 0; e < PER; ++e) { na = fmaf(fa[k][e], fa[k][e], na); nb = fmaf(fb[k][e], fb[k][e], nb); }
        }
        const float ia = 1.f / sqrtf(wave_sum(na)), ib = 1.f / sqrtf(wave_sum(nb));
        float dot = 0.f;       // the reference normalises first, then multiplies and sums
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int e = 0; e < PER; ++e) dot = fmaf(fa[k][e] * ia, fb[k][e] * ib, dot);
        const float omega = acosf(wave_sum(dot));
        const float so = sinf(omega);
        const float ca = sinf((1.f - r) * omega) / so, cb = sinf(r * omega) / so;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = lane + 64 * k;
            float res[PER];
#pragma unroll
            for (int e = 0; e < PER; ++e) res[e] = ca * fa[k][e] + cb * fb[k][e];
            if (c < nch) store16<STREAM>(&reinterpret_cast<uint4*>(o)[c], Vec16<T>::pack(res));
        }
    } else {
        float na = 0.f, nb = 0.f;
        for (int c = lane; c < dim; c += 64) {
            const float av = to_f32(a[c]), bv = to_f32(b[c]);
            na = fmaf(av, av, na);
            nb = fmaf(bv, bv, nb);
        }
        const float ia = 1.f / sqrtf(wave_sum(na)), ib = 1.f / sqrtf(wave_sum(nb));
        float dot = 0.f;
        for (int c = lane; c < dim; c += 64) dot = fmaf(to_f32(a[c]) * ia, to_f32(b[c]) * ib, dot);
        const float omega = acosf(wave_sum(dot));
        const float so = sinf(omega);
        const float ca = sinf((1.f - r) * omega) / so, cb = sinf(r * omega) / so;
        for (int c = lane; c < dim; c += 64) o[c] = from_f32<T>(ca * to_f32(a[c]) + cb * to_f32(b[c]));
    }
}

int64_t up_out_len(int64_t n, int64_t stride, int64_t target_len, int separate_cls, int truncate_seq) {
    if (stride == 1) return n;
    if (separate_cls) {
        int64_t body = (n - 1) * stride + (truncate_seq ? stride - 1 : 0);
        int64_t keep = target_len - 1 < 0 ? 0 : target_len - 1;
        if (body > keep) body = keep;
        return 1 + body;
    }
    const int64_t body = n * stride;
    return body < target_len ? body : target_len;
}

}  // namespace
}  // namespace tvae

using namespace tvae;

extern "C" int64_t tvae_upsample_out_len(int64_t n, int64_t stride, int64_t target_len, int separate_cls,
                                         int truncate_seq) {
    if (n <= 0 || stride <= 0) return -1;
    return up_out_len(n, stride, target_len, separate_cls, truncate_seq);
}

extern "C" int tvae_upsample_fwd(const void* x, void* out, int64_t B, int64_t n, int64_t d, int64_t stride,
                                 int64_t target_len, int separate_cls, int truncate_seq, int dtype, void* stream) {
    TVAE_REQUIRE(dtype == TVAE_F32 || dtype == TVAE_BF16, "upsample_fwd: unknown dtype %d", dtype);
    TVAE_REQUIRE(x && out, "upsample_fwd: null pointer argument");
    TVAE_REQUIRE(B > 0 && n > 0 && d > 0 && stride > 0, "upsample_fwd: bad shape");
    UpsampleShape s{B, n, d, stride, up_out_len(n, stride, target_len, separate_cls, truncate_seq), separate_cls};
    if (stride == 1) s.separate_cls = 0;     // identity copy
    TVAE_REQUIRE(s.out_len > 0, "upsample_fwd: empty output (target_len %lld)", (long long)target_len);
    const int64_t esz = dtype == TVAE_F32 ? 4 : 2;
    const int vec = aligned16(x) && aligned16(out) && (d * esz) % 16 == 0;
    TVAE_REQUIRE((B * s.out_len + 3) / 4 < (1ll << 31), "upsample_fwd: too many rows");
    const unsigned blocks = (unsigned)((B * s.out_len + 3) / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool stream_out = B * s.out_len * d * esz >= STREAM_MIN_BYTES;
#define TVAE_UP_FWD(T, ST) \
    hipLaunchKernelGGL((upsample_fwd_kernel<T, 4, ST>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)out, s, vec)
    if (dtype == TVAE_F32) { if (stream_out) TVAE_UP_FWD(float, true); else TVAE_UP_FWD(float, false); }
    else { if (stream_out) TVAE_UP_FWD(bf16, true); else TVAE_UP_FWD(bf16, false); }
#undef TVAE_UP_FWD
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

extern "C" int tvae_upsample_bwd(const void* d_out, void* d_x, int64_t B, int64_t n, int64_t d, int64_t stride,
                                 int64_t target_len, int separate_cls, int truncate_seq, int dtype, void* stream) {
    TVAE_REQUIRE(dtype == TVAE_F32 || dtype == TVAE_BF16, "upsample_bwd: unknown dtype %d", dtype);
    TVAE_REQUIRE(d_out && d_x, "upsample_bwd: null pointer argument");
    TVAE_REQUIRE(B > 0 && n > 0 && d > 0 && stride > 0, "upsample_bwd: bad shape");
    UpsampleShape s{B, n, d, stride, up_out_len(n, stride, target_len, separate_cls, truncate_seq), separate_cls};
    if (stride == 1) s.separate_cls = 0;
    const int64_t esz = dtype == TVAE_F32 ? 4 : 2;
    const int vec = aligned16(d_out) && aligned16(d_x) && (d * esz) % 16 == 0;
    const unsigned blocks = (unsigned)((B * n + 3) / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool stream_out = B * n * d * esz >= STREAM_MIN_BYTES;
#define TVAE_UP_BWD(T, ST) \
    hipLaunchKernelGGL((upsample_bwd_kernel<T, ST>), dim3(blocks), dim3(256), 0, st, (const T*)d_out, (T*)d_x, s, vec)
    if (dtype == TVAE_F32) { if (stream_out) TVAE_UP_BWD(float, true); else TVAE_UP_BWD(float, false); }
    else { if (stream_out) TVAE_UP_BWD(bf16, true); else TVAE_UP_BWD(bf16, false); }
#undef TVAE_UP_BWD
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

// ---- _compute_kernel as a materialised (Nx, Ny) matrix (autoencoders.py:117-126), for callers that want the matrix itself
// (notebooks, subclasses).  The training path never forms it: tvae_mmd_fwd_bwd regenerates tiles.  Direct-difference form
// like the reference (sum of (x - y)^2, then / D / D, then exp), fp32; 16 x 16 outputs per workgroup, 32-feature LDS stages.
namespace tvae {
namespace {
__global__ __launch_bounds__(256) void rbf_kernel_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int nx, int ny,
                                                            int D, float* __restrict__ K) {
    __shared__ float xs[16][33], ys[16][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    float s = 0.f;
    for (int c0 = 0; c0 < D; c0 += 32) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = threadIdx.x + 256 * u, r = e >> 5, c = e & 31;
            xs[r][c] = (i0 + r < nx && c0 + c < D) ? x[(long long)(i0 + r) * D + c0 + c] : 0.f;
            ys[r][c] = (j0 + r < ny && c0 + c < D) ? y[(long long)(j0 + r) * D + c0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float dlt = xs[ty][c] - ys[tx][c];
            s = fmaf(dlt, dlt, s);
        }
        __syncthreads();
    }
    if (i0 + ty < nx && j0 + tx < ny) K[(long long)(i0 + ty) * ny + j0 + tx] = expf(-(s / (float)D) / (float)D);
}

// out[r][c] = -(2 / D^2) sum_q W(r, q) (a[r][c] - b[q][c]),  W(r, q) = gK[r sr + q sq] K[r sr + q sq]:
// d<K, C>/dx with (a, b, sr, sq) = (x, y, ny, 1) and d/dy with (y, x, 1, ny).  One workgroup per row r.
__global__ __launch_bounds__(256) void rbf_kernel_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ K,
                                                            const float* __restrict__ gK, int nb, int D, long long sr, long long sq,
                                                            float* __restrict__ out) {
    const int r = blockIdx.x;
    const float scale = -2.f / ((float)D * (float)D);
    for (int c = threadIdx.x; c < D; c += 256) {
        const float ac = a[(long long)r * D + c];
        float acc = 0.f;
        for (int q = 0; q < nb; ++q) {
            const long long at = (long long)r * sr + (long long)q * sq;
            acc = fmaf(gK[at] * K[at], ac - b[(long long)q * D + c], acc);
        }
        out[(long long)r * D + c] = scale * acc;
    }
}
}  // namespace
}  // namespace tvae

extern "C" int tvae_rbf_kernel_fwd(const float* x, const float* y, int64_t nx, int64_t ny, int64_t D, float* K, void* stream) {
    TVAE_REQUIRE(x && y && K, "rbf_kernel_fwd: null pointer argument");
    TVAE_REQUIRE(nx > 0 && ny > 0 && D > 0 && nx < (1ll << 20) && ny < (1ll << 20) && D < (1ll << 24), "rbf_kernel_fwd: bad shape");
    hipLaunchKernelGGL(tvae::rbf_kernel_fwd_kernel, dim3((unsigned)((ny + 15) / 16), (unsigned)((nx + 15) / 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, (int)nx, (int)ny, (int)D, K);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

extern "C" int tvae_rbf_kernel_bwd(const float* x, const float* y, const float* K, const float* gK, int64_t nx, int64_t ny, int64_t D,
                                   float* dx, float* dy, void* stream) {
    TVAE_REQUIRE(x && y && K && gK, "rbf_kernel_bwd: null pointer argument");
    TVAE_REQUIRE(nx > 0 && ny > 0 && D > 0 && nx < (1ll << 20) && ny < (1ll << 20) && D < (1ll << 24), "rbf_kernel_bwd: bad shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dx) hipLaunchKernelGGL(tvae::rbf_kernel_bwd_kernel, dim3((unsigned)nx), dim3(256), 0, st, x, y, K, gK, (int)ny, (int)D,
                               (long long)ny, 1ll, dx);
    if (dy) hipLaunchKernelGGL(tvae::rbf_kernel_bwd_kernel, dim3((unsigned)ny), dim3(256), 0, st, y, x, K, gK, (int)nx, (int)D,
                               1ll, (long long)ny, dy);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

extern "C" int tvae_slerp(const void* t1, const void* t2, const float* ratios, float ratio, void* out, int64_t rows,
                          int64_t dim, int dtype, void* stream) {
    TVAE_REQUIRE(dtype == TVAE_F32 || dtype == TVAE_BF16, "slerp: unknown dtype %d", dtype);
    TVAE_REQUIRE(t1 && t2 && out, "slerp: null pointer argument");
    TVAE_REQUIRE(rows > 0 && dim > 0 && dim < (1ll << 31), "slerp: bad shape");
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t esz = dtype == TVAE_F32 ? 4 : 2;
    const int64_t nch = (dim * esz) % 16 == 0 && aligned16(t1) && aligned16(t2) && aligned16(out) ? dim * esz / 16 : 0;
    const bool stream_out = rows * dim * esz >= STREAM_MIN_BYTES;
#define TVAE_SLERP1(T, CH, REG, ST)                                                                                      \
    hipLaunchKernelGGL((slerp_kernel<T, CH, REG, ST>), dim3(blocks), dim3(256), 0, st, (const T*)t1, (const T*)t2, ratios, \
                       ratio, (T*)out, (long long)rows, (int)dim)
#define TVAE_SLERP(T, CH, REG) do { if (stream_out) TVAE_SLERP1(T, CH, REG, true); else TVAE_SLERP1(T, CH, REG, false); } while (0)
    if (dtype == TVAE_F32) {
        if (nch >= 1 && nch <= 64) TVAE_SLERP(float, 1, true);
        else if (nch >= 1 && nch <= 128) TVAE_SLERP(float, 2, true);
        else if (nch >= 1 && nch <= 256) TVAE_SLERP(float, 4, true);
        else if (nch >= 1 && nch <= 512) TVAE_SLERP(float, 8, true);
        else TVAE_SLERP(float, 1, false);
    } else {
        if (nch >= 1 && nch <= 64) TVAE_SLERP(bf16, 1, true);
        else if (nch >= 1 && nch <= 128) TVAE_SLERP(bf16, 2, true);
        else if (nch >= 1 && nch <= 256) TVAE_SLERP(bf16, 4, true);
        else TVAE_SLERP(bf16, 1, false);
    }
#undef TVAE_SLERP
#undef TVAE_SLERP1
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}
