// MMD tile kernel for narrow latents (padded dimension <= 80): ONE wave per workgroup owns 64 rows.
//
// Same math, LDS image, in-place refill and counted waits as the single-wave instantiations of
// mmd_tile_kernel (mmd.hip), with four 16-row tiles per wave instead of two.  Why: below ~128 features a
// point tile carries few MFMAs (256 cycles per 16-column block and row-tile pair) against fixed per-tile
// costs that do not shrink with D -- the 1-KiB LDS-DMA pieces (~150 cycles of issue each) and the loop
// overhead -- so the 32-row kernel sat at 55-66 % MFMA-busy for D = 48..112.  Doubling the
// rows per wave halves the DMA pieces, the L2 traffic and the loop overhead per MFMA; the registers allow
// it while 2 NB row-feature blocks + 2 NB accumulator blocks fit (NB <= 5 at two waves per SIMD).
// Symmetric y-y enumeration on 64-point blocks (four tiles), same parity rule as mmd.hip.
#include <mutex>

#include "mmd_common.h"

namespace tvae {
namespace {

constexpr int NROWS = 64, NRT = 4;

template <int NB, bool GRAD>
__global__ __launch_bounds__(64, 2) void mmd_narrow_kernel(MmdArgs a) {
    constexpr int SW = 16 * NB, DPAD = SW, NORM_OFF = TILE_P * SW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int i16 = lane & 15, g = lane >> 4;
    const unsigned xs_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const float* xs = smem;

    const int per_problem = a.row_blocks * a.col_splits;
    const int problem = GRAD ? 0 : a.problem_base + (int)blockIdx.x / per_problem;
    const int rem = (int)blockIdx.x - (problem - a.problem_base) * per_problem;
    const int rb = rem / a.col_splits, sp = rem - rb * a.col_splits;
    const int i0 = rb * NROWS;
    const long long bid = (long long)a.sums_base + blockIdx.x;

    // ---- the 64 rows of this wave, all features, in registers ----
    const float* rows = a.cols[problem] + (long long)a.row_offset * DPAD;
    const float* rnorm = a.cnorm[problem] + a.row_offset;
    f32x4 zreg[NRT][NB];
    float rn[NRT];
    bool valid_i[NRT];
    bool all_valid = true;
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        const int row = i0 + 16 * t + i16;
        valid_i[t] = row < a.n_rows;
        all_valid = all_valid && valid_i[t];
        const float* zp = rows + (long long)row * DPAD + 4 * g;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            if (valid_i[t]) zreg[t][m] = *reinterpret_cast<const f32x4*>(zp + 16 * m);
            else zreg[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        rn[t] = valid_i[t] ? rnorm[row] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t) {       // retire the loads before the loop (see mmd_tile_kernel)
#pragma unroll
        for (int m = 0; m < NB; ++m) asm volatile("" : "+v"(zreg[t][m]));
        asm volatile("" : "+v"(rn[t]));
    }
    const bool any_invalid = __any(!all_valid);

    f32x4 acc[GRAD ? NB : 1][NRT];
    if (GRAD) {
#pragma unroll
        for (int db = 0; db < NB; ++db)
#pragma unroll
            for (int t = 0; t < NRT; ++t) acc[db][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // column walk; symmetric y-y problem: unordered pairs of 64-point blocks, parity rule of mmd.hip
    const long long tps = a.tiles_per_set;
    const bool sym = !GRAD && problem == 1 && a.sym;
    const int blkI = (a.row_offset >> 6) + rb;
    const int nJ = (int)(tps >> 2);
    const int symA = (blkI + 1) >> 1;
    const int symM = symA + 1 + ((nJ - 1 - blkI) >> 1);
    const long long total = problem == 0 ? 2 * tps : (sym ? 4 * (long long)symM : tps);
    const long long base = (problem == 0 || sym) ? 0 : tps;
    const int tb = (int)(base + sp * total / a.col_splits);
    const int te = (int)(base + (sp + 1) * total / a.col_splits);
    auto actual = [&](int vt) -> int {
        if (!sym) return vt;
        const int q = vt >> 2;
        const int J = q < symA ? ((blkI + 1) & 1) + 2 * q : (q == symA ? blkI : blkI + 2 * (q - symA));
        return (int)tps + 4 * J + (vt & 3);
    };

    // per-lane constants of the swizzled image (TileLds in mmd.hip)
    const unsigned dma_voff = (unsigned)((i16 ^ g) * DPAD + 4 * g) * 4u;
    const int gram_off = 64 * g + 4 * (i16 ^ g);
    const int c4 = i16 >> 2;
    int grad_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) grad_off[r] = 64 * c4 + 16 * g + 4 * (r ^ c4) + (i16 & 3);

    auto tile_src = [&](int vt) -> const float* {
        const int tile = actual(vt);
        const int set = tile >= tps ? 1 : 0;
        const int p0 = (tile - set * (int)tps) * TILE_P;
        return a.cols[set] + (long long)p0 * DPAD;
    };
    auto stage_norms = [&](int vt, int nbuf) {
        const int tile = actual(vt);
        const int set = tile >= tps ? 1 : 0;
        const int p0 = (tile - set * (int)tps) * TILE_P;
        const unsigned dst = __builtin_amdgcn_readfirstlane(xs_byte + (unsigned)(NORM_OFF + nbuf * TILE_P) * 4u);
        const float* src = uniform_ptr(a.cnorm[set] + p0);
        if (lane < TILE_P) dma4(src, (unsigned)lane * 4u, dst);
    };

    double sum0 = 0.0, sum1 = 0.0, rowsum[NRT];
    float f0 = 0.f, f1 = 0.f, fr[NRT];        // fp32 partials, flushed to fp64 every 8 tiles
#pragma unroll
    for (int t = 0; t < NRT; ++t) { rowsum[t] = 0.0; fr[t] = 0.f; }

    // DMA issue order per tile k: norms(k) ... X(k) blocks 0..NB-1; norms(k+1) at the top of iteration k.
    // The last tile prefetches itself again, so no DMA sits behind a branch and the counted waits always hold.
    if (tb < te) {
        stage_norms(tb, 0);
        const float* X0 = tile_src(tb);
#pragma unroll
        for (int i = 0; i < NB; ++i) dma16(X0 + 16 * i, dma_voff, xs_byte + 1024u * i);
    }
    for (int tile = tb; tile < te; ++tile) {
        const int atile = actual(tile);
        const int set = atile >= tps ? 1 : 0;
        const int p0 = (atile - set * (int)tps) * TILE_P;
        const float wsym = (sym && (p0 >> 6) != blkI) ? 2.f : 1.f;
        const int buf = (tile - tb) & 1;
        const int ntile = tile + 1 < te ? tile + 1 : tile;
        const float* Xn = tile_src(ntile);
        stage_norms(ntile, buf ^ 1);

        // 1. Gram: S[t][r] = <x_p, z_i>, p = p0 + 4g + r, i = i0 + 16t + i16
        f32x4 S[NRT];
#pragma unroll
        for (int t = 0; t < NRT; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        wait_vmcnt<NB>();
        f32x4 av = *reinterpret_cast<const f32x4*>(xs + gram_off);
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            f32x4 avn = av;
            if (m + 1 < NB) {
                wait_vmcnt_n(GRAD ? NB - (m + 1) : NB - 1);
                avn = *reinterpret_cast<const f32x4*>(xs + gram_off + 256 * (m + 1));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < NRT; ++t)
                    S[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], zreg[t][m][q], S[t], 0, 0, 0);
            if (!GRAD) {
#pragma unroll
                for (int t = 0; t < NRT; ++t) asm volatile("" ::"v"(S[t]));
                dma16(Xn + 16 * m, dma_voff, xs_byte + 1024u * m);
            }
            av = avn;
        }

        // 2. E = expm1(-t)
        const f32x4 cn = *reinterpret_cast<const f32x4*>(smem + NORM_OFF + buf * TILE_P + 4 * g);
        f32x4 e[NRT], q[NRT];
        float tmax = 0.f;
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float tt = fmaxf((cn[r] + rn[t] - 2.f * S[t][r]) * a.inv_d2, 0.f);
                S[t][r] = tt;
                tmax = fmaxf(tmax, tt);
            }
        if (!__any(tmax >= EM1_TINY)) {
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const EQ v_ = em1_split_tiny(S[t][r]); e[t][r] = v_.e; q[t][r] = v_.q; }
        } else {
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const EQ v_ = em1_split(S[t][r]); e[t][r] = v_.e; q[t][r] = v_.q; }
            if (__any(tmax >= EM1_POLY_MAX)) {
                // |E| >= 0.1175 here, so exp(-t) - 1 loses at most 3 bits (relative error < 1e-6); sixteen
                // inlined expm1f bodies would not fit the register budget of this kernel
#pragma unroll
                for (int t = 0; t < NRT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ex = expf(-S[t][r]) - 1.f;
                        const bool big = S[t][r] >= EM1_POLY_MAX;
                        e[t][r] = big ? ex : e[t][r];
                        q[t][r] = big ? ex + S[t][r] : q[t][r];
                    }
            }
        }
        if (any_invalid) {
#pragma unroll
            for (int t = 0; t < NRT; ++t)
                if (!valid_i[t]) { e[t] = f32x4{0.f, 0.f, 0.f, 0.f}; q[t] = e[t]; }
        }
        if (p0 + TILE_P > a.n_cols) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (p0 + 4 * g + r >= a.n_cols) {
#pragma unroll
                    for (int t = 0; t < NRT; ++t) { e[t][r] = 0.f; q[t][r] = 0.f; }
                }
        }
        // sums carry Q (linear part: closed form in the finalize kernel); row sums carry E
        float qs_all = 0.f;
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            qs_all += (q[t][0] + q[t][1]) + (q[t][2] + q[t][3]);
            if (GRAD) {
                const float es = (e[t][0] + e[t][1]) + (e[t][2] + e[t][3]);
                fr[t] += set ? -es : es;
            }
        }
        if (set) f1 = fmaf(wsym, qs_all, f1);
        else f0 += qs_all;
        if (((tile - tb) & 7) == 7) {
            sum0 += (double)f0; sum1 += (double)f1;
            f0 = f1 = 0.f;
#pragma unroll
            for (int t = 0; t < NRT; ++t) { rowsum[t] += (double)fr[t]; fr[t] = 0.f; }
        }

        // 3. gradient: acc^T[d][i] += X^T[d][p] (s_p E[p][i]); block db of the tile is dead after its MFMAs
        if (GRAD) {
            const unsigned flip = set ? 0x80000000u : 0u;      // prior columns enter with s_p = -1
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) e[t][r] = __uint_as_float(__float_as_uint(e[t][r]) ^ flip);
            float ga[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[r] = xs[grad_off[r]];
#pragma unroll
            for (int db = 0; db < NB; ++db) {
                float gn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) gn[r] = (db + 1 < NB) ? xs[grad_off[r] + 256 * (db + 1)] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < NRT; ++t)
                        acc[db][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[r], e[t][r], acc[db][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NRT; ++t) asm volatile("" ::"v"(acc[db][t]));
                dma16(Xn + 16 * db, dma_voff, xs_byte + 1024u * db);
#pragma unroll
                for (int r = 0; r < 4; ++r) ga[r] = gn[r];
            }
        }
    }
    wait_vmcnt<0>();
    sum0 += (double)f0; sum1 += (double)f1;

    const double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    if (lane == 0) {
        a.wg_sums[2 * bid + 0] = w0;
        a.wg_sums[2 * bid + 1] = w1;
    }
    if (GRAD) {
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            double rv = rowsum[t] + (double)fr[t];
            rv += __shfl_xor(rv, 16, 64);
            rv += __shfl_xor(rv, 32, 64);
            if (g == 0) a.wg_rowsum[(long long)sp * a.n_rows_pad + i0 + 16 * t + i16] = rv;
            float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 16 * t + i16) * DPAD + 4 * g;
#pragma unroll
            for (int db = 0; db < NB; ++db) *reinterpret_cast<f32x4*>(grow + 16 * db) = acc[GRAD ? db : 0][t];
        }
    }
}

template <int NB, bool GRAD>
int launch_narrow(const MmdArgs& a, int nblocks, hipStream_t s) {
    constexpr int lds_bytes = (TILE_P * 16 * NB + 2 * TILE_P) * 4;
    hipLaunchKernelGGL((mmd_narrow_kernel<NB, GRAD>), dim3(nblocks), dim3(64), lds_bytes, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

template <bool GRAD>
int launch_narrow_shape(int nb, const MmdArgs& a, int nblocks, hipStream_t s) {
    switch (nb) {
        case 1: return launch_narrow<1, GRAD>(a, nblocks, s);
        case 2: return launch_narrow<2, GRAD>(a, nblocks, s);
        case 3: return launch_narrow<3, GRAD>(a, nblocks, s);
        case 4: return launch_narrow<4, GRAD>(a, nblocks, s);
        case 5: return launch_narrow<5, GRAD>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no narrow kernel for %d feature blocks", nb);
}

}  // namespace

int launch_mmd_narrow(bool grad, int nb, const MmdArgs& a, int nblocks, hipStream_t s) {
    return grad ? launch_narrow_shape<true>(nb, a, nblocks, s) : launch_narrow_shape<false>(nb, a, nblocks, s);
}

}  // namespace tvae
