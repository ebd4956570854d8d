// Whole bottleneck step in FOUR launches for the reference's own (small-batch) configurations: BASELINE cfg1 (N = 40
// latent rows) and cfg2 (N = 350), where the general path spends its time in ~16 launches of 3-12 us each.
//
//   forward   small_rows_fwd_kernel     per latent row: slice + encode + tanh -> decode + bias -> LayerNorm
//                                       (autoencoders.py:21-22, 38-39, 66-67); also row norms of z and of the prior
//             small_mmd_kernel          MMD sums, loss and d loss / d z of all N rows against all 2N columns
//                                       (autoencoders.py:117-136; closed-form gradient SURVEY.md 8a)
//   backward  small_rows_bwd_kernel     per latent row: LayerNorm backward -> dpre . Wd -> + w dz_mmd -> tanh backward
//                                       -> . We -> dh[:, :n]; zero-fill of dh[:, n:]
//             small_weights_bwd_kernel  dWd, dbd, dgamma, dbeta, dWe, dbe: column-sliced sums over ALL rows on the fp32
//                                       matrix cores (operands straight from global memory: one element per lane)
//
// fp32 only, stock decoder (Linear + LayerNorm), N <= 512, latent_size <= 128, d_model <= 2048 and a multiple of 16.
// Everything is deterministic: no atomics in any sum (the one atomic is a ticket that picks which workgroup adds the
// per-workgroup fp64 partials, in a fixed order).
#include <cstdlib>

#include "mmd_common.h"

namespace tvae {
namespace {

constexpr int SMALL_MAX_N = 512, SMALL_MAX_L = 128, SMALL_MAX_D = 2048;

// These kernels are LATENCY-bound (a few hundred KB per workgroup, all of it L2 / Infinity-Cache resident), and every
// instruction runs about once per workgroup: what counts is (i) the number of dependent global-memory round trips
// (~1 us each) on a workgroup's critical path and (ii) CODE SIZE -- a first version unrolled everything, reached
// 21 000 instructions per kernel and ran at the pace of its instruction-cache misses (27 us for ONE latent row).
// Hence: rolled loops, weight tiles copied into LDS by LDS-DMA (twenty fire-and-forget instructions, one wait),
// all other loads of a phase issued together.

// Sum over the 64 lanes without the LDS crossbar: four DPP steps inside each row of 16 lanes (every lane of a row ends
// up with the row's sum), then the four row sums through v_readlane.  ~10 instructions against six dependent
// ds_bpermute round trips (~100 cycles each) for the __shfl_xor ladder; the order of the additions is fixed.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats */, int tid) {
    v = wave_sum_dpp(v);
    __syncthreads();                      // red may still be read from a previous use
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// `count` contiguous floats (16-byte aligned source, count % 4 == 0) -> the same bytes at LDS byte address dst_byte.
// Each wave instruction moves 1 KiB; issue only -- the caller waits (wait_vmcnt<0>) and synchronises.
__device__ __forceinline__ void dma_tile(const float* __restrict__ src, int count, unsigned dst_byte, int wave, int lane) {
    const int bytes = count * 4;
    for (int off = wave * 1024; off < bytes; off += 4096)
        if (off + lane * 16 < bytes) dma16(src, (unsigned)(off + lane * 16), dst_byte + (unsigned)off);
}
// any count / alignment: plain copy through registers (slow path, eight loads in flight)
__device__ __forceinline__ void copy_tile(const float* __restrict__ src, int count, float* dst, int tid) {
    for (int e0 = tid; e0 < count; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = e0 + 256 * u < count ? src[e0 + 256 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (e0 + 256 * u < count) dst[e0 + 256 * u] = v[u];
    }
}

struct SmallFwdArgs {
    const float* h; long long hsb, hss;
    const float *We, *be, *Wd, *bd, *gamma, *beta, *prior;
    float eps;
    float *z, *recon, *pre, *mean, *rstd;
    float *zn, *yn;              // row norms of z and of the prior
    unsigned* ticket;            // zeroed here for small_mmd_kernel
    int N, n_eff, d, L, JT;
    int stop;                    // debug: leave after phase `stop` (0: run everything)
};

// One workgroup = RPB latent rows.  LDS: hs[RPB][d] (later the pre-norm rows) | zs[RPB][128] | bes[128] | red[8] | wt[JT][L]
// A tile row j of Wd is walked from column (j mod L) on, wrapping: lanes then hit different banks whatever the
// parity of L (row stride L floats).
template <int RPB>
__global__ __launch_bounds__(256) void small_rows_fwd_kernel(SmallFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hs = sm;
    float* zs = hs + RPB * a.d;
    float* bes = zs + RPB * 128;
    float* red = bes + 128;
    float* wt = red + 8;
    const unsigned wt_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wt;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * RPB;
    const int d = a.d, L = a.L, d4 = d >> 2;
    const int ntiles = (d + a.JT - 1) / a.JT;
    if (blockIdx.x == 0 && tid == 0) *a.ticket = 0u;

    // ---- round trip 1: the first tile of Wd (LDS-DMA), the hidden-state rows, the encoder bias ----
    dma_tile(a.Wd, (d < a.JT ? d : a.JT) * L, wt_byte, wave, lane);
    for (int r = 0; r < RPB; ++r) {
        const int row = r0 + r;
        const float4* src = reinterpret_cast<const float4*>(a.h + (long long)((row < a.N ? row : 0) / a.n_eff) * a.hsb +
                                                            (long long)((row < a.N ? row : 0) % a.n_eff) * a.hss);
        for (int k = tid; k < d4; k += 256)
            reinterpret_cast<float4*>(hs + r * d)[k] = row < a.N ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 128) bes[tid] = tid < L ? a.be[tid] : 0.f;
    __syncthreads();
    if (a.stop == 1) { wait_vmcnt<0>(); return; }

    // ---- encode: z[r][l] = tanh(h[r] . We[l] + be[l]); a wave takes eight outputs per round and has all their
    //      loads (up to 32 float4 per lane) in flight at once ----
    for (int lb = 8 * wave; lb < L; lb += 32) {
        float acc[8][RPB];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < RPB; ++r) acc[u][r] = 0.f;
        for (int kb = 0; kb < d4; kb += 256) {
            float4 w[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = kb + 64 * c + lane;
                    w[u][c] = (lb + u < L && k < d4) ? reinterpret_cast<const float4*>(a.We + (long long)(lb + u) * d)[k]
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = kb + 64 * c + lane;
                if (k < d4) {
#pragma unroll
                    for (int r = 0; r < RPB; ++r) {
                        const float4 hv = reinterpret_cast<const float4*>(hs + r * d)[k];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            acc[u][r] = fmaf(w[u][c].x, hv.x, fmaf(w[u][c].y, hv.y, fmaf(w[u][c].z, hv.z, fmaf(w[u][c].w, hv.w, acc[u][r]))));
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < RPB; ++r) {
                const float v = wave_sum_dpp(acc[u][r]);
                if (lane == 0 && lb + u < L) zs[r * 128 + lb + u] = tanhf(v + bes[lb + u]);
            }
    }
    wait_vmcnt<0>();                       // the first tile of Wd has landed (its DMA is invisible to the compiler)
    __syncthreads();
    if (a.stop == 2) return;
    // latent rows out; row norms of z and of the prior (wave 0: |z_row|^2, wave 1: |y_row|^2)
    if (tid < 128) {
        for (int r = 0; r < RPB; ++r) {
            const int row = r0 + r;
            if (row < a.N && tid < L) a.z[(long long)row * L + tid] = zs[r * 128 + tid];
        }
    }
    for (int r = 0; r < RPB; ++r) {
        const int row = r0 + r;
        if (row < a.N && wave < 2) {
            float s = 0.f;
            for (int l = lane; l < L; l += 64) {
                const float v = wave == 0 ? zs[r * 128 + l] : a.prior[(long long)row * L + l];
                s = fmaf(v, v, s);
            }
            s = wave_sum_dpp(s);
            if (lane == 0) (wave == 0 ? a.zn : a.yn)[row] = s;
        }
    }

    // ---- decode: pre[r][j] = z[r] . Wd[j] + bd[j], JT rows of Wd at a time through LDS; the rows go to LDS (over hs) ----
    float* pres = hs;
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * a.JT, rows = d - j0 < a.JT ? d - j0 : a.JT;
        if (t > 0) {
            __syncthreads();
            dma_tile(a.Wd + (long long)j0 * L, rows * L, wt_byte, wave, lane);
        }
        const float bdv = tid < rows ? a.bd[j0 + tid] : 0.f;
        if (t > 0) { wait_vmcnt<0>(); __syncthreads(); }
        if (tid < rows) {
            const float* w = wt + tid * L;
            float acc[RPB];
#pragma unroll
            for (int r = 0; r < RPB; ++r) acc[r] = bdv;
            int l = tid % L;
#pragma unroll 16
            for (int i = 0; i < L; ++i) {
                const float wv = w[l];
#pragma unroll
                for (int r = 0; r < RPB; ++r) acc[r] = fmaf(zs[r * 128 + l], wv, acc[r]);
                l = l + 1 == L ? 0 : l + 1;
            }
#pragma unroll
            for (int r = 0; r < RPB; ++r) pres[r * d + j0 + tid] = acc[r];
        }
    }
    __syncthreads();
    if (a.stop == 3) return;

    // ---- LayerNorm over d (two passes over the LDS-resident rows) ----
    for (int r = 0; r < RPB; ++r) {
        const int row = r0 + r;
        float s = 0.f;
        for (int k = tid; k < d; k += 256) s += pres[r * d + k];
        const float mean = block_sum_256(s, red, tid) / (float)d;
        float v = 0.f;
        for (int k = tid; k < d; k += 256) { const float c = pres[r * d + k] - mean; v = fmaf(c, c, v); }
        const float rstd = rsqrtf(block_sum_256(v, red, tid) / (float)d + a.eps);
        if (row < a.N) {
            for (int k = tid; k < d4; k += 256) {
                const float4 pv = reinterpret_cast<const float4*>(pres + r * d)[k];
                const float4 gm = reinterpret_cast<const float4*>(a.gamma)[k], bt = reinterpret_cast<const float4*>(a.beta)[k];
                reinterpret_cast<float4*>(a.pre + (long long)row * d)[k] = pv;
                reinterpret_cast<float4*>(a.recon + (long long)row * d)[k] =
                    make_float4(fmaf((pv.x - mean) * rstd, gm.x, bt.x), fmaf((pv.y - mean) * rstd, gm.y, bt.y),
                                fmaf((pv.z - mean) * rstd, gm.z, bt.z), fmaf((pv.w - mean) * rstd, gm.w, bt.w));
            }
            if (tid == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
        }
    }
}

struct SmallMmdArgs {
    const float *z, *y, *zn, *yn;
    float* dz;                   // nullable
    double* sums3; float* loss;  // loss nullable
    double* part;                // [blocks][3]
    unsigned* ticket;
    double loss_scale; float grad_scale, inv_d2;
    int N, D;
};

// One workgroup = two z rows (and the two prior rows of the same index for the y-y sum) against all 2N columns,
// 128 columns at a time through LDS, double buffered: the LDS-DMA of chunk c + 1 runs under the work on chunk c.
// LDS: xs[2][128][D] | ar[2][D] | yr[2][D] | es[2][128] | gred[4][2][128] | cred[4][128] | xn[2][128]
// Row p of a chunk is walked from feature (p mod D) on, wrapping (bank spread for any D).
__global__ __launch_bounds__(256) void small_mmd_kernel(SmallMmdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = a.D, N = a.N;
    float* xs = sm;
    float* ar = xs + 2 * 128 * D;
    float* yr = ar + 2 * D;
    float* es = yr + 2 * D;
    float* gred = es + 2 * 128;
    float* cred = gred + 8 * 128;
    float* xn = cred + 4 * 128;
    __shared__ double dred[4][3];
    __shared__ float rred[4];
    __shared__ unsigned last;
    const unsigned xs_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)xs;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 2;
    const int nchunks = (N + 127) >> 7;
    const bool can_dma = (reinterpret_cast<uintptr_t>(a.z) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
    auto fetch = [&](int c) {                            // chunk c -> buffer c & 1 (issue only)
        const int set = c >= nchunks ? 1 : 0;
        const int p0 = (c - set * nchunks) << 7;
        const int cols = N - p0 < 128 ? N - p0 : 128;
        const float* src = (set ? a.y : a.z) + (long long)p0 * D;
        const int count = cols * D;
        float* dst = xs + (c & 1) * 128 * D;
        if (can_dma) {
            dma_tile(src, count & ~3, xs_byte + (unsigned)((c & 1) * 128 * D * 4), wave, lane);
            if (tid < (count & 3)) dst[(count & ~3) + tid] = src[(count & ~3) + tid];
        } else {
            copy_tile(src, count, dst, tid);
        }
        if (tid < 128) xn[(c & 1) * 128 + tid] = tid < cols ? (set ? a.yn : a.zn)[p0 + tid] : 0.f;
    };
    fetch(0);
    for (int e = tid; e < 2 * D; e += 256) {             // rows of this block
        const int r = e / D, k = e - r * D, row = i0 + r;
        ar[e] = row < N ? a.z[(long long)row * D + k] : 0.f;
        yr[e] = row < N ? a.y[(long long)row * D + k] : 0.f;
    }
    const int pr = tid >> 7, pp = tid & 127;            // phase 1: (row, column of the chunk)
    const int prow = i0 + pr;
    const bool prow_ok = prow < N;
    const float na = prow_ok ? a.zn[prow] : 0.f, nyr = prow_ok ? a.yn[prow] : 0.f;
    const int gr = lane >> 5, gl = lane & 31;           // phase 2: (row, features gl + 32 m), the wave takes 32 columns
    float G[4] = {0.f, 0.f, 0.f, 0.f}, CS[4] = {0.f, 0.f, 0.f, 0.f};   // CS: this wave's share of sum_p (+z_p | -y_p)
    double s_zz = 0.0, s_zy = 0.0, s_yy = 0.0;
    float rsum = 0.f;                                   // sum of signed E over this thread's columns (its row pr)
    const int k_start = pp % D;
    for (int c = 0; c < 2 * nchunks; ++c) {
        const int set = c >= nchunks ? 1 : 0;
        const int p0 = (c - set * nchunks) << 7;
        const int cols = N - p0 < 128 ? N - p0 : 128;
        const float* xc = xs + (c & 1) * 128 * D;
        wait_vmcnt<0>();
        __syncthreads();                                 // chunk c is in LDS; everybody is done with chunk c - 1
        if (c + 1 < 2 * nchunks) fetch(c + 1);
        // ---- phase 1: one (row, column) pair per thread ----
        float ev = 0.f;
        if (pp < cols && prow_ok) {
            const float* x = xc + pp * D;
            const float* av = ar + pr * D;
            const float* yv = yr + pr * D;
            float dz_ = 0.f, dy_ = 0.f;
            int k = k_start;
            if (set) {
#pragma unroll 16
                for (int i = 0; i < D; ++i) {
                    const float xv = x[k];
                    dz_ = fmaf(xv, av[k], dz_);
                    dy_ = fmaf(xv, yv[k], dy_);
                    k = k + 1 == D ? 0 : k + 1;
                }
            } else {
#pragma unroll 16
                for (int i = 0; i < D; ++i) {
                    dz_ = fmaf(x[k], av[k], dz_);
                    k = k + 1 == D ? 0 : k + 1;
                }
            }
            const float nx = xn[(c & 1) * 128 + pp];
            const float t = fmaxf((na + nx - 2.f * dz_) * a.inv_d2, 0.f);
            const float e = t < EM1_POLY_MAX ? em1_split(t).e : expm1f(-t);
            if (set) {
                s_zy += (double)e;
                ev = -e;
                const float t2 = fmaxf((nyr + nx - 2.f * dy_) * a.inv_d2, 0.f);
                s_yy += (double)(t2 < EM1_POLY_MAX ? em1_split(t2).e : expm1f(-t2));
            } else {
                s_zz += (double)e;
                ev = e;
            }
            rsum += ev;
        }
        es[pr * 128 + pp] = ev;
        __syncthreads();
        // ---- phase 2: G[r][dd] += sum_p E[r][p] x[p][dd]; wave w takes columns [32 w, 32 w + 32) ----
        if (a.dz) {
            const int pe = cols - 32 * wave < 32 ? cols - 32 * wave : 32;
            const float sg = set ? -1.f : 1.f;
#pragma unroll 4
            for (int p = 0; p < pe; ++p) {
                const float e = es[gr * 128 + 32 * wave + p];
                const float* x = xc + (32 * wave + p) * D;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float xv = gl + 32 * m < D ? x[gl + 32 * m] : 0.f;
                    G[m] = fmaf(e, xv, G[m]);
                    CS[m] = fmaf(sg, xv, CS[m]);
                }
            }
        }
    }
    // ---- per-row results ----
    __syncthreads();
    if (a.dz) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            gred[(wave * 2 + gr) * 128 + gl + 32 * m] = G[m];
            if (gr == 0) cred[wave * 128 + gl + 32 * m] = CS[m];
        }
        // row sums of the signed E: threads of row pr are the two waves 2 pr, 2 pr + 1
        const float rs = wave_sum_dpp(rsum);
        if (lane == 0) rred[wave] = rs;
        __syncthreads();
        for (int e = tid; e < 2 * D; e += 256) {
            const int r = e / D, k = e - r * D, row = i0 + r;
            if (row < N) {
                const float g = (gred[(0 * 2 + r) * 128 + k] + gred[(1 * 2 + r) * 128 + k]) + (gred[(2 * 2 + r) * 128 + k] + gred[(3 * 2 + r) * 128 + k]);
                const float cs = (cred[0 * 128 + k] + cred[1 * 128 + k]) + (cred[2 * 128 + k] + cred[3 * 128 + k]);   // sum z - sum y
                const float R = rred[2 * r] + rred[2 * r + 1];
                a.dz[(long long)row * D + k] = a.grad_scale * (fmaf(-ar[r * D + k], R, cs) + g);
            }
        }
    }
    // ---- sums: per-workgroup fp64 partials; the workgroup that draws the last ticket adds them in a fixed order ----
    const double w0 = wave_sum(s_zz), w1 = wave_sum(s_zy), w2 = wave_sum(s_yy);
    if (lane == 0) { dred[wave][0] = w0; dred[wave][1] = w1; dred[wave][2] = w2; }
    __syncthreads();
    if (tid < 3) {
        a.part[(long long)blockIdx.x * 3 + tid] = (dred[0][tid] + dred[1][tid]) + (dred[2][tid] + dred[3][tid]);
        __threadfence();
    }
    __syncthreads();
    if (tid == 0) last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last) {
        __threadfence();
        if (wave < 3) {                                  // lanes take blocks round-robin, then a fixed tree
            double s = 0.0;
            for (unsigned b = lane; b < gridDim.x; b += 64) s += a.part[(long long)b * 3 + wave];
            s = wave_sum(s);
            if (lane == 0) { dred[0][wave] = s; a.sums3[wave] = s; }
        }
        __syncthreads();
        if (tid == 0) {
            if (a.loss) *a.loss = (float)(a.loss_scale * (dred[0][2] + dred[0][0] - 2.0 * dred[0][1]));
            *a.ticket = 0u;
        }
    }
}

struct SmallBwdArgs {
    const float *g, *pre, *mean, *rstd, *gamma, *Wd, *z, *dz_mmd, *g_reg, *We;
    float *dpre, *dpe, *dh;
    long long dh_sb;             // elements between sequences of dh (S * d)
    long long tail_vec;          // float4 per sequence tail, B sequences
    int B;
    int N, n_eff, d, L, JT;
};

// One workgroup = RPB latent rows.  LDS: dps[RPB][d] | dzs[2][RPB][128] | dpes[RPB][128] | red[8 RPB] | wt[JT][L]
template <int RPB>
__global__ __launch_bounds__(256) void small_rows_bwd_kernel(SmallBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = a.d, L = a.L, d4 = d >> 2;
    float* dps = sm;
    float* dzs = dps + RPB * d;
    float* dpes = dzs + 2 * RPB * 128;
    float* red = dpes + RPB * 128;
    float* wt = red + 8 * RPB;
    const unsigned wt_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wt;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * RPB;
    const int ntiles = (d + a.JT - 1) / a.JT;

    // ---- the first tile of Wd on its way; zero-fill of dh[:, n:]: plain stores that drain while the rest computes ----
    dma_tile(a.Wd, (d < a.JT ? d : a.JT) * L, wt_byte, wave, lane);
    {
        const long long total = (long long)a.B * a.tail_vec;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) {
            const long long b = i / a.tail_vec, o = i - b * a.tail_vec;
            reinterpret_cast<float4*>(a.dh + b * a.dh_sb + (long long)a.n_eff * d)[o] = zero;
        }
    }
    // ---- LayerNorm backward: dpre = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)), dxh = g gamma (rows one by one) ----
    const float wreg = a.g_reg ? *a.g_reg : 0.f;
    float zrow[RPB], dmm[RPB];
    for (int r = 0; r < RPB; ++r) {
        const int row = r0 + r;
        const bool ok = row < a.N;
        zrow[r] = (ok && tid < L) ? a.z[(long long)row * L + tid] : 0.f;
        dmm[r] = (ok && tid < L && a.dz_mmd) ? a.dz_mmd[(long long)row * L + tid] : 0.f;
        const float mean = ok ? a.mean[row] : 0.f, rstd = ok ? a.rstd[row] : 0.f;
        float4 dx[2], xh[2];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = tid + 256 * c;
            float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), pv = gv, gm = gv;
            if (ok && k < d4) {
                gv = reinterpret_cast<const float4*>(a.g + (long long)row * d)[k];
                pv = reinterpret_cast<const float4*>(a.pre + (long long)row * d)[k];
                gm = reinterpret_cast<const float4*>(a.gamma)[k];
            }
            dx[c] = make_float4(gv.x * gm.x, gv.y * gm.y, gv.z * gm.z, gv.w * gm.w);
            xh[c] = make_float4((pv.x - mean) * rstd, (pv.y - mean) * rstd, (pv.z - mean) * rstd, (pv.w - mean) * rstd);
            if (k < d4) {
                s1 += (dx[c].x + dx[c].y) + (dx[c].z + dx[c].w);
                s2 += fmaf(dx[c].x, xh[c].x, fmaf(dx[c].y, xh[c].y, fmaf(dx[c].z, xh[c].z, dx[c].w * xh[c].w)));
            }
        }
        const float m1 = block_sum_256(s1, red, tid) / (float)d;
        const float m2 = block_sum_256(s2, red + 4, tid) / (float)d;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = tid + 256 * c;
            if (k < d4) {
                float4 o;
                o.x = rstd * (dx[c].x - m1 - xh[c].x * m2); o.y = rstd * (dx[c].y - m1 - xh[c].y * m2);
                o.z = rstd * (dx[c].z - m1 - xh[c].z * m2); o.w = rstd * (dx[c].w - m1 - xh[c].w * m2);
                reinterpret_cast<float4*>(dps + r * d)[k] = o;
                if (ok) reinterpret_cast<float4*>(a.dpre + (long long)row * d)[k] = o;
            }
        }
    }
    // ---- dz_dec[r][l] = sum_j dpre[r][j] Wd[j][l]: two halves of every JT-row tile, then their sum ----
    const int l = tid & 127, part = tid >> 7;
    float acc[RPB];
#pragma unroll
    for (int r = 0; r < RPB; ++r) acc[r] = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * a.JT, rows = d - j0 < a.JT ? d - j0 : a.JT;
        if (t > 0) {
            __syncthreads();
            dma_tile(a.Wd + (long long)j0 * L, rows * L, wt_byte, wave, lane);
        }
        wait_vmcnt<0>();
        __syncthreads();
        if (l < L) {
            const int half = (rows + 1) >> 1;
            const int jb = part * half, je = jb + half < rows ? jb + half : rows;
#pragma unroll 16
            for (int j = jb; j < je; ++j) {
                const float wv = wt[j * L + l];
#pragma unroll
                for (int r = 0; r < RPB; ++r) acc[r] = fmaf(dps[r * d + j0 + j], wv, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPB; ++r) dzs[(part * RPB + r) * 128 + l] = acc[r];
    __syncthreads();
    // ---- + w * dz_mmd, tanh backward ----
    if (tid < 128) {
#pragma unroll
        for (int r = 0; r < RPB; ++r) {
            const int row = r0 + r;
            float v = 0.f;
            if (row < a.N && tid < L) {
                const float dz = fmaf(wreg, dmm[r], dzs[(0 * RPB + r) * 128 + tid] + dzs[(1 * RPB + r) * 128 + tid]);
                v = dz * (1.f - zrow[r] * zrow[r]);
                a.dpe[(long long)row * L + tid] = v;
            }
            dpes[r * 128 + tid] = v;
        }
    }
    __syncthreads();
    // ---- dh[r][k] = sum_l dpe[r][l] We[l][k]: 16 rows of We in flight per thread ----
    if (a.dh) {
        for (int k = tid; k < d4; k += 256) {
            float4 o[RPB];
#pragma unroll
            for (int r = 0; r < RPB; ++r) o[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l0 = 0; l0 < L; l0 += 16) {
                float4 wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    wv[u] = l0 + u < L ? reinterpret_cast<const float4*>(a.We + (long long)(l0 + u) * d)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int r = 0; r < RPB; ++r) {
                        const float e = l0 + u < L ? dpes[r * 128 + l0 + u] : 0.f;
                        o[r].x = fmaf(e, wv[u].x, o[r].x); o[r].y = fmaf(e, wv[u].y, o[r].y);
                        o[r].z = fmaf(e, wv[u].z, o[r].z); o[r].w = fmaf(e, wv[u].w, o[r].w);
                    }
            }
#pragma unroll
            for (int r = 0; r < RPB; ++r) {
                const int row = r0 + r;
                if (row < a.N)
                    reinterpret_cast<float4*>(a.dh + (long long)(row / a.n_eff) * a.dh_sb + (long long)(row % a.n_eff) * d)[k] = o[r];
            }
        }
    }
}

struct SmallWgtArgs {
    const float *dpre, *z, *g, *pre, *mean, *rstd, *dpe, *h;
    long long hsb, hss;
    float *dWd, *dbd, *dgamma, *dbeta, *dWe, *dbe;      // each nullable
    int N, n_eff, d, L;
    int nblk_d;                  // d / 16 column blocks per product
};

// Workgroups [0, nblk_d): 16 columns j of dWd (d x L) = dpre^T z, with dbd, dgamma, dbeta of the same columns.
// Workgroups [nblk_d, 2 nblk_d): 16 columns k of dWe (L x d) = dpe^T h; the first of them also sums dbe.
// fp32 matrix cores, v_mfma_f32_16x16x4_f32: the K dimension is the N latent rows, four per instruction, split over the
// four waves; operands come straight from global memory (one element per lane, KU k-steps of loads in flight), the
// four partial tiles meet in LDS.  Column sums (dbd, dgamma, dbeta, dbe) are products with a column of ones.
__global__ __launch_bounds__(256) void small_weights_bwd_kernel(SmallWgtArgs a) {
    constexpr int MAXT = 9;                              // 16-wide tiles of the L (+1) dimension: L <= 128
    constexpr int NACC = 2 * 8 + 1;                      // second group, first workgroup: L tiles twice; first group: MAXT + 2
    constexpr int KU = 4;                                // k-steps (of four rows) whose operand loads are in flight together
    extern __shared__ __attribute__((aligned(16))) float sm_w[];
    float (*redt)[NACC][4][64] = reinterpret_cast<float (*)[NACC][4][64]>(sm_w);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int N = a.N, d = a.d, L = a.L;
    const bool second = (int)blockIdx.x >= a.nblk_d;
    const int c0 = ((int)blockIdx.x - (second ? a.nblk_d : 0)) * 16;
    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksteps = (N + 3) >> 2;
    if (!second) {
        // A[i = column j][k = row] = dpre[row][c0 + i]; B[k = row][n = l] = z[row][16 t + n], n == L: the ones column (dbd)
        // tiles MAXT, MAXT + 1: A = g xhat resp. g against B = 1 -> dgamma, dbeta in every column n
        const int nt = (L + 1 + 15) >> 4;
        for (int ks0 = wave; ks0 < ksteps; ks0 += 4 * KU) {
            float av[KU], gq[KU], pq[KU], mq[KU], rq[KU], bv[KU][MAXT];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int row = 4 * (ks0 + 4 * u) + lk;
                const bool ok = ks0 + 4 * u < ksteps && row < N;
                av[u] = ok ? a.dpre[(long long)row * d + c0 + li] : 0.f;
                gq[u] = ok ? a.g[(long long)row * d + c0 + li] : 0.f;
                pq[u] = ok ? a.pre[(long long)row * d + c0 + li] : 0.f;
                mq[u] = ok ? a.mean[row] : 0.f;
                rq[u] = ok ? a.rstd[row] : 0.f;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    const int n = 16 * t + li;
                    bv[u][t] = (t < nt && ok) ? (n < L ? a.z[(long long)row * L + n] : (n == L ? 1.f : 0.f)) : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
                    if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][t], acc[t], 0, 0, 0);
                acc[MAXT] = __builtin_amdgcn_mfma_f32_16x16x4f32(gq[u] * (pq[u] - mq[u]) * rq[u], 1.f, acc[MAXT], 0, 0, 0);
                acc[MAXT + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(gq[u], 1.f, acc[MAXT + 1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < MAXT + 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) redt[wave][t][r][lane] = acc[t][r];
        __syncthreads();
        // D[i = 4 lk + r][n = li]: column j = c0 + i, l = 16 t + n
        for (int e = tid; e < (MAXT + 2) * 256; e += 256) {
            const int t = e >> 8, rr = (e >> 6) & 3, ln = e & 63;
            if (t >= nt && t < MAXT) continue;
            const float v = (redt[0][t][rr][ln] + redt[1][t][rr][ln]) + (redt[2][t][rr][ln] + redt[3][t][rr][ln]);
            const int j = c0 + 4 * (ln >> 4) + rr, n = 16 * t + (ln & 15);
            if (t < MAXT) {
                if (n < L) { if (a.dWd) a.dWd[(long long)j * L + n] = v; }
                else if (n == L && a.dbd) a.dbd[j] = v;
            } else if ((ln & 15) == 0) {
                float* out = t == MAXT ? a.dgamma : a.dbeta;
                if (out) out[j] = v;
            }
        }
    } else {
        // A[i = l][k = row] = dpe[row][16 t + i]; B[k = row][n = column k] = h[row][c0 + n]; D[i][n] -> dWe[l][c0 + n]
        // first workgroup of the group: tiles 8 + t with B = 1 -> dbe[l] in every column n
        const int nt = (L + 15) >> 4;
        const bool with_dbe = (int)blockIdx.x == a.nblk_d && a.dbe;
        for (int ks0 = wave; ks0 < ksteps; ks0 += 4 * KU) {
            float bv[KU], av[KU][8];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int row = 4 * (ks0 + 4 * u) + lk;
                const bool ok = ks0 + 4 * u < ksteps && row < N;
                bv[u] = ok ? a.h[(long long)(row / a.n_eff) * a.hsb + (long long)(row % a.n_eff) * a.hss + c0 + li] : 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int i = 16 * t + li;
                    av[u][t] = (t < nt && ok && i < L) ? a.dpe[(long long)row * L + i] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (t < nt) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t], bv[u], acc[t], 0, 0, 0);
                        if (with_dbe) acc[8 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t], 1.f, acc[8 + t], 0, 0, 0);
                    }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) redt[wave][t][r][lane] = acc[t][r];
        __syncthreads();
        for (int e = tid; e < 16 * 256; e += 256) {
            const int t = e >> 8, rr = (e >> 6) & 3, ln = e & 63;
            if ((t & 7) >= nt || (t >= 8 && !with_dbe)) continue;
            const float v = (redt[0][t][rr][ln] + redt[1][t][rr][ln]) + (redt[2][t][rr][ln] + redt[3][t][rr][ln]);
            const int i = 16 * (t & 7) + 4 * (ln >> 4) + rr, n = c0 + (ln & 15);
            if (i < L) {
                if (t < 8) { if (a.dWe) a.dWe[(long long)i * d + n] = v; }
                else if ((ln & 15) == 0) a.dbe[i] = v;
            }
        }
    }
}

struct SmallShape { int RPB, JT; int64_t lds_rows_fwd, lds_rows_bwd, lds_mmd; };

bool small_shape(int64_t N, int64_t d, int64_t L, SmallShape* s) {
    if (N < 1 || N > SMALL_MAX_N || L < 1 || L > SMALL_MAX_L || d < 16 || d > SMALL_MAX_D || (d & 15)) return false;
    s->RPB = N > 128 ? 2 : 1;
    s->JT = L <= 96 ? 256 : 128;                         // rows of Wd per LDS tile (<= 96 KiB)
    s->lds_rows_fwd = ((int64_t)s->RPB * d + s->RPB * 128 + 128 + 8 + (int64_t)s->JT * L) * 4;
    s->lds_rows_bwd = ((int64_t)s->RPB * d + 3 * s->RPB * 128 + 8 * s->RPB + (int64_t)s->JT * L) * 4;
    s->lds_mmd = ((int64_t)2 * 128 * L + 4 * L + 2 * 128 + 12 * 128 + 2 * 128) * 4;
    return s->lds_rows_fwd <= 150 * 1024 && s->lds_rows_bwd <= 150 * 1024 && s->lds_mmd <= 150 * 1024;
}

struct FwdWs { int64_t zn, yn, part, ticket, bytes; };
FwdWs fwd_ws(int64_t N, int64_t L, const SmallShape& s) {
    FwdWs w; int64_t off = 0;
    auto take = [&](int64_t b) { int64_t o = off; off = align_up(off + b, 256); return o; };
    const int64_t mblocks = (N + 1) / 2;
    (void)L;
    w.zn = take(N * 4); w.yn = take(N * 4); w.part = take(mblocks * 3 * 8); w.ticket = take(256);
    w.bytes = off;
    return w;
}

template <typename K>
int set_lds(K kernel, int64_t bytes, PerDeviceOnce& once) {
    TVAE_CHECK_HIP(per_device_once(once, [&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }));
    (void)bytes;
    return TVAE_OK;
}

}  // namespace
}  // namespace tvae

using namespace tvae;

extern "C" int tvae_small_step_supported(int64_t B, int64_t n_eff, int64_t d, int64_t L) {
    SmallShape s;
    return (B > 0 && n_eff > 0 && small_shape(B * n_eff, d, L, &s)) ? 1 : 0;
}

extern "C" int64_t tvae_small_step_fwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L) {
    SmallShape s;
    if (B <= 0 || n_eff <= 0 || !small_shape(B * n_eff, d, L, &s)) return -1;
    return fwd_ws(B * n_eff, L, s).bytes;
}

extern "C" int64_t tvae_small_step_bwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L) {
    SmallShape s;
    if (B <= 0 || n_eff <= 0 || !small_shape(B * n_eff, d, L, &s)) return -1;
    return align_up(B * n_eff * d * 4, 256) + align_up(B * n_eff * L * 4, 256);
}

extern "C" int tvae_small_step_fwd(const void* h, int64_t h_stride_b, int64_t h_stride_s, const void* We, const void* be,
                                   const void* Wd, const void* bd, const void* gamma, const void* beta, float eps,
                                   const void* prior, int64_t B, int64_t n_eff, int64_t d, int64_t L, int dtype,
                                   void* latent, void* recon, void* pre_norm, float* mean, float* rstd, void* d_latent_mmd,
                                   double* sums3, float* loss, double loss_scale, double grad_scale, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    TVAE_REQUIRE(dtype == TVAE_F32, "small_step_fwd: fp32 only (dtype %d)", dtype);
    TVAE_REQUIRE(h && We && be && Wd && bd && gamma && beta && prior && latent && recon && pre_norm && mean && rstd && sums3 && workspace,
                 "small_step_fwd: null pointer argument");
    SmallShape s;
    TVAE_REQUIRE(B > 0 && n_eff > 0 && small_shape(B * n_eff, d, L, &s),
                 "small_step_fwd: shape outside the small-batch path (N=%lld <= %d, L=%lld <= %d, d=%lld <= %d and a multiple of 16)",
                 (long long)(B * n_eff), SMALL_MAX_N, (long long)L, SMALL_MAX_L, (long long)d, SMALL_MAX_D);
    TVAE_REQUIRE(aligned16(h) && aligned16(We) && aligned16(Wd) && aligned16(gamma) && aligned16(beta) && aligned16(recon) &&
                     aligned16(pre_norm) && h_stride_b % 4 == 0 && h_stride_s % 4 == 0,
                 "small_step_fwd: h, We, Wd, gamma, beta, recon, pre_norm must be 16-byte aligned and the strides of h multiples of 4 elements");
    const int64_t N = B * n_eff;
    const FwdWs w = fwd_ws(N, L, s);
    if (workspace_bytes < w.bytes)
        return fail(TVAE_ERR_WORKSPACE, "small_step_fwd: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)w.bytes);
    char* ws = static_cast<char*>(workspace);
    SmallFwdArgs a;
    a.h = (const float*)h; a.hsb = h_stride_b; a.hss = h_stride_s;
    a.We = (const float*)We; a.be = (const float*)be; a.Wd = (const float*)Wd; a.bd = (const float*)bd;
    a.gamma = (const float*)gamma; a.beta = (const float*)beta; a.prior = (const float*)prior; a.eps = eps;
    a.z = (float*)latent; a.recon = (float*)recon; a.pre = (float*)pre_norm; a.mean = mean; a.rstd = rstd;
    a.zn = (float*)(ws + w.zn); a.yn = (float*)(ws + w.yn); a.ticket = (unsigned*)(ws + w.ticket);
    a.N = (int)N; a.n_eff = (int)n_eff; a.d = (int)d; a.L = (int)L; a.JT = s.JT;
    a.stop = 0;
    const int blocks = (int)((N + s.RPB - 1) / s.RPB);
    static PerDeviceOnce o1, o2, o3;
    if (s.RPB == 1) {
        TVAE_PROPAGATE(set_lds(&small_rows_fwd_kernel<1>, s.lds_rows_fwd, o1));
        hipLaunchKernelGGL(small_rows_fwd_kernel<1>, dim3(blocks), dim3(256), (size_t)s.lds_rows_fwd, st, a);
    } else {
        TVAE_PROPAGATE(set_lds(&small_rows_fwd_kernel<2>, s.lds_rows_fwd, o2));
        hipLaunchKernelGGL(small_rows_fwd_kernel<2>, dim3(blocks), dim3(256), (size_t)s.lds_rows_fwd, st, a);
    }
    TVAE_CHECK_HIP(hipGetLastError());
    SmallMmdArgs m;
    m.z = a.z; m.y = a.prior; m.zn = a.zn; m.yn = a.yn;
    m.dz = (float*)d_latent_mmd; m.sums3 = sums3; m.loss = loss; m.part = (double*)(ws + w.part); m.ticket = a.ticket;
    m.loss_scale = loss_scale; m.grad_scale = (float)grad_scale; m.inv_d2 = (float)(1.0 / ((double)L * (double)L));
    m.N = (int)N; m.D = (int)L;
    TVAE_PROPAGATE(set_lds(&small_mmd_kernel, s.lds_mmd, o3));
    hipLaunchKernelGGL(small_mmd_kernel, dim3((unsigned)((N + 1) / 2)), dim3(256), (size_t)s.lds_mmd, st, m);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

extern "C" int tvae_small_step_bwd(const void* d_recon, const float* d_reg, const void* latent, const void* pre_norm,
                                   const float* mean, const float* rstd, const void* gamma, const void* Wd, const void* We,
                                   const void* h, int64_t h_stride_b, int64_t h_stride_s, const void* d_latent_mmd,
                                   void* dh, void* dWe, void* dbe, void* dWd, void* dbd, void* dgamma, void* dbeta,
                                   int64_t B, int64_t S, int64_t n_eff, int64_t d, int64_t L, int dtype, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    TVAE_REQUIRE(dtype == TVAE_F32, "small_step_bwd: fp32 only (dtype %d)", dtype);
    TVAE_REQUIRE(d_recon && latent && pre_norm && mean && rstd && gamma && Wd && We && h && workspace, "small_step_bwd: null pointer argument");
    TVAE_REQUIRE(!d_latent_mmd || d_reg, "small_step_bwd: the regulariser's share needs its upstream gradient");
    SmallShape s;
    TVAE_REQUIRE(B > 0 && n_eff > 0 && S >= n_eff && small_shape(B * n_eff, d, L, &s), "small_step_bwd: shape outside the small-batch path");
    TVAE_REQUIRE(aligned16(d_recon) && aligned16(pre_norm) && aligned16(gamma) && aligned16(We) && aligned16(Wd) && (!dh || aligned16(dh)),
                 "small_step_bwd: d_recon, pre_norm, gamma, We, Wd, dh must be 16-byte aligned");
    const int64_t N = B * n_eff;
    const int64_t need = tvae_small_step_bwd_workspace_bytes(B, n_eff, d, L);
    if (workspace_bytes < need)
        return fail(TVAE_ERR_WORKSPACE, "small_step_bwd: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
    char* ws = static_cast<char*>(workspace);
    SmallBwdArgs a;
    a.g = (const float*)d_recon; a.pre = (const float*)pre_norm; a.mean = mean; a.rstd = rstd; a.gamma = (const float*)gamma;
    a.Wd = (const float*)Wd; a.z = (const float*)latent; a.dz_mmd = (const float*)d_latent_mmd; a.g_reg = d_reg; a.We = (const float*)We;
    a.dpre = (float*)ws; a.dpe = (float*)(ws + align_up(N * d * 4, 256)); a.dh = (float*)dh;
    a.dh_sb = S * d; a.tail_vec = dh ? (S - n_eff) * d / 4 : 0; a.B = (int)B;
    a.N = (int)N; a.n_eff = (int)n_eff; a.d = (int)d; a.L = (int)L; a.JT = s.JT;
    const int blocks = (int)((N + s.RPB - 1) / s.RPB);
    static PerDeviceOnce o1, o2;
    if (s.RPB == 1) {
        TVAE_PROPAGATE(set_lds(&small_rows_bwd_kernel<1>, s.lds_rows_bwd, o1));
        hipLaunchKernelGGL(small_rows_bwd_kernel<1>, dim3(blocks), dim3(256), (size_t)s.lds_rows_bwd, st, a);
    } else {
        TVAE_PROPAGATE(set_lds(&small_rows_bwd_kernel<2>, s.lds_rows_bwd, o2));
        hipLaunchKernelGGL(small_rows_bwd_kernel<2>, dim3(blocks), dim3(256), (size_t)s.lds_rows_bwd, st, a);
    }
    TVAE_CHECK_HIP(hipGetLastError());
    if (dWd || dbd || dgamma || dbeta || dWe || dbe) {
        SmallWgtArgs w;
        w.dpre = a.dpre; w.z = a.z; w.g = a.g; w.pre = a.pre; w.mean = mean; w.rstd = rstd; w.dpe = a.dpe; w.h = (const float*)h;
        w.hsb = h_stride_b; w.hss = h_stride_s;
        w.dWd = (float*)dWd; w.dbd = (float*)dbd; w.dgamma = (float*)dgamma; w.dbeta = (float*)dbeta; w.dWe = (float*)dWe; w.dbe = (float*)dbe;
        w.N = (int)N; w.n_eff = (int)n_eff; w.d = (int)d; w.L = (int)L; w.nblk_d = (int)(d / 16);
        static PerDeviceOnce o3;
        TVAE_PROPAGATE(set_lds(&small_weights_bwd_kernel, 0, o3));
        hipLaunchKernelGGL(small_weights_bwd_kernel, dim3((unsigned)(2 * w.nblk_d)), dim3(256), (size_t)(4 * 17 * 4 * 64 * 4), st, w);
        TVAE_CHECK_HIP(hipGetLastError());
    }
    return TVAE_OK;
}
