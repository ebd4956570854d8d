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

// source row of output position j (-1: zero padding introduced by truncate_seq)
__device__ __forceinline__ long long up_src(const UpsampleShape& s, long long j) {
    if (s.separate_cls) {
        if (j == 0) return 0;
        const long long q = (j - 1) / s.stride;
        return q < s.n - 1 ? 1 + q : -1;
    }
    return j / s.stride;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, UpsampleShape s,
                                                           int vec) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= s.B * s.out_len) return;
    const long long b = row / s.out_len, j = row - b * s.out_len;
    const long long src = up_src(s, j);
    T* o = out + row * s.d;
    const T* in = src >= 0 ? x + (b * s.n + src) * s.d : nullptr;
    constexpr int PER = 16 / sizeof(T);
    if (vec) {
        const uint4* iv = reinterpret_cast<const uint4*>(in);
        uint4* ov = reinterpret_cast<uint4*>(o);
        for (long long c = lane; c < s.d / PER; c += 64) ov[c] = in ? iv[c] : make_uint4(0, 0, 0, 0);
    } else {
        for (long long c = lane; c < s.d; c += 64) o[c] = in ? in[c] : from_f32<T>(0.f);
    }
}

// d_x[b, src] = sum of d_out over the (contiguous) output positions that copied row src
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, UpsampleShape s) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= s.B * s.n) return;
    const long long b = row / s.n, src = row - b * s.n;
    long long j0, j1;
    if (s.separate_cls) {
        if (src == 0) { j0 = 0; j1 = 1; }
        else { j0 = 1 + (src - 1) * s.stride; j1 = j0 + s.stride; }
    } else {
        j0 = src * s.stride; j1 = j0 + s.stride;
    }
    if (j1 > s.out_len) j1 = s.out_len;
    const T* g = dout + b * s.out_len * s.d;
    T* o = dx + row * s.d;
    for (long long c = lane; c < s.d; c += 64) {
        float acc = 0.f;
        for (long long j = j0; j < j1; ++j) acc += to_f32(g[j * s.d + c]);
        o[c] = from_f32<T>(acc);
    }
}

// res = sin((1-r) w)/sin(w) * t1 + sin(r w)/sin(w) * t2,  w = acos(<t1/|t1|, t2/|t2|>)   (utils.py:29-33)
template <typename T>
__global__ __launch_bounds__(256) void slerp_kernel(const T* __restrict__ t1, const T* __restrict__ t2,
                                                    const float* __restrict__ ratios, float ratio, T* __restrict__ out,
                                                    long long rows, int dim) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* a = t1 + row * dim;
    const T* b = t2 + row * dim;
    float na = 0.f, nb = 0.f;
    for (int c = lane; c < dim; c += 64) {
        const float av = to_f32(a[c]), bv = to_f32(b[c]);
        na = fmaf(av, av, na);
        nb = fmaf(bv, bv, nb);
    }
    const float ia = 1.f / sqrtf(wave_sum(na)), ib = 1.f / sqrtf(wave_sum(nb));
    float dot = 0.f;       // the reference normalises first, then multiplies and sums
    for (int c = lane; c < dim; c += 64) dot = fmaf(to_f32(a[c]) * ia, to_f32(b[c]) * ib, dot);
    const float omega = acosf(wave_sum(dot));
    const float so = sinf(omega);
    const float r = ratios ? ratios[row] : ratio;
    const float ca = sinf((1.f - r) * omega) / so, cb = sinf(r * omega) / so;
    T* o = out + row * dim;
    for (int c = lane; c < dim; c += 64) o[c] = from_f32<T>(ca * to_f32(a[c]) + cb * to_f32(b[c]));
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
    const unsigned blocks = (unsigned)((B * s.out_len + 3) / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == TVAE_F32)
        hipLaunchKernelGGL(upsample_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)out, s, vec);
    else
        hipLaunchKernelGGL(upsample_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)out, s, vec);
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
    const unsigned blocks = (unsigned)((B * n + 3) / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == TVAE_F32)
        hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)d_out, (float*)d_x, s);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)d_out, (bf16*)d_x, s);
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
    if (dtype == TVAE_F32)
        hipLaunchKernelGGL(slerp_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)t1, (const float*)t2, ratios,
                           ratio, (float*)out, (long long)rows, (int)dim);
    else
        hipLaunchKernelGGL(slerp_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)t1, (const bf16*)t2, ratios,
                           ratio, (bf16*)out, (long long)rows, (int)dim);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}
