// Shared helpers for libtvae_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <atomic>
#include <string>

#include "../../include/tvae_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace tvae {

typedef __hip_bfloat16 bf16;

// ---- error plumbing (thread-local message, int status) -------------------------------
std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define TVAE_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::tvae::fail(TVAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr,             \
                                hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define TVAE_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) return ::tvae::fail(TVAE_ERR_ARG, __VA_ARGS__);                      \
    } while (0)

#define TVAE_PROPAGATE(expr)                                                              \
    do {                                                                                  \
        int _s = (expr);                                                                  \
        if (_s != TVAE_OK) return _s;                                                     \
    } while (0)

// ---- element access: fp32 math on fp32 / bf16 storage ---------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return __float2bfloat16(v); }
__device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) { return (_Float16)v; }

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t u16) { return __uint_as_float(u16 << 16); }

// 4 consecutive elements -> fp32, vectorised (16 B fp32 / 8 B bf16); caller guarantees alignment.
__device__ __forceinline__ void load4_vec(const float* p, float out[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
__device__ __forceinline__ void load4_vec(const bf16* p, float out[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    out[0] = bf16_bits_to_f32(v.x & 0xffffu); out[1] = bf16_bits_to_f32(v.x >> 16);
    out[2] = bf16_bits_to_f32(v.y & 0xffffu); out[3] = bf16_bits_to_f32(v.y >> 16);
}
// Guarded form: element e is read only when e < nvalid (others are 0).
template <typename T>
__device__ __forceinline__ void load4_guard(const T* p, int nvalid, float out[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = (e < nvalid) ? to_f32(p[e]) : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Function attributes are per device: one bit per device ordinal remembers where `prepare` has succeeded
// (nn.DataParallel drives several devices from one process; running `prepare` twice is harmless).
struct PerDeviceOnce { std::atomic<uint64_t> done{0}; };
template <typename F>
inline hipError_t per_device_once(PerDeviceOnce& o, F&& prepare) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (o.done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = prepare();
    if (e == hipSuccess) o.done.fetch_or(bit, std::memory_order_release);
    return e;
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- row address maps -----------------------------------------------------------------
// offset(r) = (r / grp) * so + (r % grp) * si   in elements.  A dense (rows, ld) matrix is
// {grp = 1, so = ld, si = 0}; the first n tokens of every sequence of a (B,S,d) tensor are
// {grp = n, so = stride_b, si = stride_s}.
struct RowMap {
    long long so, si;
    int grp;
    __host__ __device__ __forceinline__ long long operator()(int r) const {
        return grp == 1 ? (long long)r * so : (long long)(r / grp) * so + (long long)(r % grp) * si;
    }
    bool vec_ok() const { return (so % 4 == 0) && (grp == 1 || si % 4 == 0); }
};
inline RowMap dense_rows(long long ld) { return RowMap{ld, 0, 1}; }

// 16-B store that bypasses cache allocation, for outputs that are written once and not read back soon
// (measured on upsample_fwd, 934 MB: 5.9 -> 6.2 TB/s against plain stores; the zero-fill of dh inside the training
// step got SLOWER with it, 98 -> 105 us per launch at BASELINE cfg3, and keeps plain stores)
template <bool STREAM>
__device__ __forceinline__ void store16(uint4* p, const uint4& v) {
    if constexpr (STREAM) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
    } else {
        *p = v;
    }
}
// outputs at least this large are stored past the caches (nothing downstream could find them there anyway)
constexpr int64_t STREAM_MIN_BYTES = 256ll << 20;

// ---- zero-fill of dh[:, n_eff:, :] (the rows the latent slice never touched), 16 B per lane --------------
// One virtual grid (gx, gy): x * 256 threads stride over the tail of one sequence, y walks the sequences.
// Shared by zero_tail_kernel (rowops.hip) and the backward GEMM pair launch, whose extra workgroups run it.
struct ZeroTail {
    char* dh; long long seq_bytes, head_bytes, B;
    int gx, gy;     // gx * gy workgroups of 256 threads; gx == 0: nothing to do
};
__device__ __forceinline__ void zero_tail_body(const ZeroTail& z, int bx, int by, int tid) {
    const long long tail = z.seq_bytes - z.head_bytes;
    for (long long b = by; b < z.B; b += z.gy) {
        char* base = z.dh + b * z.seq_bytes + z.head_bytes;
        // leading bytes up to 16-B alignment, then uint4 stores, then the remainder
        const long long mis = (16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15;
        const long long lead = mis < tail ? mis : tail;
        const long long nvec = (tail - lead) / 16;
        const long long t0 = (long long)bx * 256 + tid;
        if (t0 < lead) base[t0] = 0;
        uint4* v = reinterpret_cast<uint4*>(base + lead);
        for (long long i = t0; i < nvec; i += (long long)z.gx * 256) v[i] = make_uint4(0, 0, 0, 0);
        const long long done = lead + nvec * 16;
        if (t0 < tail - done) base[done + t0] = 0;
    }
}

// ---- internal launchers (one translation unit each) -----------------------------------
enum Epilogue { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_TANH = 2 };

struct GemmDesc {
    // C[m][n] = sum_k A(m,k) * B(k,n)  (+ bias[n], tanh)
    const void* A; const void* B; void* C; const void* bias;
    int M, N, K;
    bool a_kmajor;   // false: A(m,k) = A[amap(m) + k];  true: A(m,k) = A[amap(k) + m]
    bool b_kmajor;   // false: B(k,n) = B[bmap(n) + k];  true: B(k,n) = B[bmap(k) + n]
    RowMap amap, bmap, cmap;   // cmap: C(m,n) = C[cmap(m) + n]
    int epilogue;
    int dtype;
    float* slab; int64_t slab_floats;   // split-K scratch (fp32), may be null => no split
    // weight-gradient layout only (A and B k-major, no epilogue), nullable: colsum_out[m] = sum_k A(m,k), i.e. the
    // bias gradient, computed by the same launch through a virtual column of ones appended to B (size the slabs
    // with N + 1 columns)
    void* colsum_out;
    // nullable fp32 vectors subtracted from the operands as they are loaded, indexed by the operand's CONTIGUOUS index
    // (k for a row-major operand, m / n for a k-major one): products of centred data without a centred copy (mmd.hip).
    // Honoured by gemm_launch only (not by the pair / LayerNorm forms).
    const float* a_sub = nullptr; const float* b_sub = nullptr;
};
int64_t gemm_slab_floats(int M, int N, int K);
int gemm_launch(const GemmDesc& g, hipStream_t stream);
// linear + bias + LayerNorm in one launch when the shape allows (*done says whether it ran), gemm.hip
int gemm_bias_layernorm(const GemmDesc& g, const void* gamma, const void* beta, float eps, void* out, float* mean,
                        float* rstd, hipStream_t stream, bool* done);
// activation-gradient + weight-gradient products of one projection; one launch when both are small
// `zt` (nullable): a zero-fill that may ride in the same launch (its workgroups come first); *zt_done says whether it did
int gemm_launch_pair(const GemmDesc& g_act, const GemmDesc& g_weight, hipStream_t stream,
                     const ZeroTail* zt = nullptr, bool* zt_done = nullptr);

// row-wise / column-wise helpers (rowops.hip)
int layernorm_fwd(const void* r, const void* gamma, const void* beta, float eps, void* out,
                  float* mean, float* rstd, int64_t rows, int64_t d, int dtype, hipStream_t s);
int layernorm_bwd_rows(const void* dout, const void* r, const float* mean, const float* rstd,
                       const void* gamma, void* dr, int64_t rows, int64_t d, int dtype, hipStream_t s);
// column reductions; scratch = colreduce_scratch_floats(rows, cols) fp32
int64_t colreduce_scratch_floats(int64_t rows, int64_t cols);
int colsum(const void* x, int64_t rows, int64_t cols, void* out, int dtype, float* scratch, hipStream_t s);
int layernorm_bwd_params(const void* dout, const void* r, const float* mean, const float* rstd,
                         void* dgamma, void* dbeta, int64_t rows, int64_t d, int dtype,
                         float* scratch /* 2x colreduce_scratch_floats */, hipStream_t s);
// input gradient and (dgamma, dbeta) in one pass + one final launch (register-resident rows: d % 4 == 0,
// d <= 2048, aligned); scratch = layernorm_bwd_scratch_floats(rows, d) fp32 (covers layernorm_bwd_params too)
int64_t layernorm_bwd_scratch_floats(int64_t rows, int64_t d);
bool layernorm_bwd_can_fuse(const void* dout, const void* r, const void* gamma, const void* dr, int64_t d, int dtype);
int layernorm_bwd_fused(const void* dout, const void* r, const float* mean, const float* rstd, const void* gamma,
                        void* dr, void* dgamma, void* dbeta, int64_t rows, int64_t d, int dtype, float* scratch,
                        hipStream_t s);
int tanh_bwd(const void* dz, const void* z, void* dpre, int64_t count, int dtype, hipStream_t s);
int tanh_bwd_sum(const void* dz, const void* dx, const float* scale, const void* z, void* dpre, int64_t count,
                 int dtype, hipStream_t s);
int zero_tail_rows(void* dh, int64_t B, int64_t S, int64_t n_eff, int64_t d, int dtype, hipStream_t s);
ZeroTail zero_tail_plan(void* dh, int64_t B, int64_t S, int64_t n_eff, int64_t d, int dtype);   // gx == 0: nothing to fill

// strip-wise MMD for latent sizes beyond the tile kernels (mmd_wide.hip)
int64_t mmd_wide_workspace_bytes(int64_t n_total, int64_t n_rows, int64_t D, bool want_grad);
int mmd_wide_launch(const void* z_all, const void* y_all, int64_t n_total, int64_t row_offset, int64_t n_rows,
                    int64_t D, int dtype, double* sums3, float* loss, double loss_scale, void* d_z,
                    double grad_scale, void* workspace, int64_t workspace_bytes, hipStream_t s);

}  // namespace tvae
