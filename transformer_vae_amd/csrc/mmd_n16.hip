// MMD tile kernel for NARROW latents (D <= 192) on the 16-bit matrix cores: fp32 inputs as scaled fp16 pairs hi + lo
// (mmd_common.h, split_scales) -- the split-fp16 mode of DESIGN 4.1d / 4.1f for the shapes the reference actually ships
// (latent_size 77).
//
// Each of the four waves of a workgroup owns 32 rows (RT = 2 row tiles) with ALL features in registers (no exchange of
// partial products); DPAD = 32 KBN features.  Rows and accumulators leave room for two to four waves per SIMD, which is
// what covers one wave's waits (LDS round trip of E, matrix results) with the others' instructions.  The four waves
// SHARE the column tiles: one LDS-DMA of a step serves all 128 rows, and one barrier per step (two 16-point tiles)
// hands the double buffer over.  (A first version with one wave per workgroup, every wave streaming its own tiles, ran
// at the pace of the L2 -> LDS path: 124 us at N = 5 120, D = 77, about 30 GB/s per CU, the guide's Infinity-Cache
// gather rate.)
//   Gram      S'[p][i] over 32-feature k-steps, v_mfma_f32_16x16x32_f16:   X_hi Z_hi + X_hi Z_lo  (+ X_lo Z_hi up to 64
//             features: TVAE_GRAM_PRODUCTS in mmd_common.h)
//   gradient  acc^T[d][i] += X^T[d][p] (s_p E[p][i]), K = a tile's 16 points, v_mfma_f32_32x32x16_f16 on tiles of
//             32 features x 32 rows:   X_hi E_hi + X_lo E_hi + X_hi E_lo
// E goes through a per-wave 2 KiB LDS buffer (the producer lanes hold points x rows the way the Gram left them, the
// gradient's B operand wants eight points of one row per lane).  Steps are double buffered: the LDS-DMA of step k + 1 is
// issued at the top of step k, one M0 set-up per hi + lo pair (mmd_pipe.hip); a workgroup without gradient fetches the
// hi copy alone when the Gram does not read lo.
// Workgroups [0, nwg0) take problem 0 (z rows against both column sets, with the gradient when GRAD), the rest
// problem 1 (y rows against the priors, sum only, symmetric enumeration on 128-point blocks when a.sym).
#include <type_traits>

#include "mmd_common.h"

namespace tvae {
namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int N16_WAVES = 4;
constexpr int N16_PT = 2;          // 16-point tiles per step (one barrier and one round of LDS-DMA per step)
#ifndef TVAE_N16_RT_SMALL
#define TVAE_N16_RT_SMALL 2        // row tiles per wave for KBN <= 2 (4: measured 6 % slower at N = 8 192, D = 64 and 30 % at N = 2 048, D = 32)
#endif
// waves per SIMD asked of hipcc (registers per wave: 512 / that)
constexpr int N16_WPE(int kbn, int rt) { return rt == 2 ? (kbn == 1 ? 4 : (kbn <= 3 ? 3 : 2)) : (kbn == 1 ? 3 : 2); }

template <int KBN, int RT>
struct N16Lds {
    static constexpr int TILE_B = KBN * 2048;                 // [s][hi | lo] pieces of 16 points x 32 features
    static constexpr int STEP_B = N16_PT * TILE_B;            // [tile of the step][s][hi | lo]
    static constexpr int NORM_OFF = 2 * STEP_B;               // [2][64] floats (16 N16_PT used; the DMA moves 64)
    static constexpr int E_OFF = NORM_OFF + 2 * 64 * 4;       // [wave][hi | lo][16 RT rows][16 points] fp16
    static constexpr int EPIECE_B = 16 * RT * 32;
    static constexpr int BYTES = E_OFF + N16_WAVES * 2 * EPIECE_B;
    static_assert(N16_WAVES * 2 * EPIECE_B >= (2 * N16_WAVES) * 8, "final reduction reuses the E buffers");
};

// EDGE = false: the launch has no partial row block and no partial point step (picked on the host), so the
// invalid-row and ragged-tile selects are not even compiled in
template <int KBN, int RT, bool GRAD, bool EDGE>
__global__ __launch_bounds__(64 * N16_WAVES, N16_WPE(KBN, RT)) void mmd_n16_kernel(MmdArgs a, int nwg0) {
    using L = N16Lds<KBN, RT>;
    constexpr int DPAD = 32 * KBN, ROWS = 16 * RT, RB = RT / 2;      // ROWS: per wave; RB: 32-row blocks of the gradient
    constexpr int ROWS_WG = N16_WAVES * ROWS, PT = N16_PT, STEP_P = PT * TILE_P;
    // Gram products (mmd_common.h, TVAE_GRAM_PRODUCTS): the error of the two-product form grows as the width shrinks (the
    // 1 / D^2 in t no longer hides it; measured gradient error against fp64: 2e-7 at D = 192, 6e-7 at 77, 2.4e-6 at 32,
    // 1e-5 at 8), so widths up to 64 keep all three
    // Round 3: ALL three products at every width of this kernel.  The two-product form (column side hi only) held 1e-5 on
    // i.i.d. data but not on structured inputs at D = 77 (tests/test_gpu_r3.py, N = 8 192, against fp64): rows duplicated
    // 64x, latents quantised to 3 or 8 levels and one outlier row gave 5 - 16x the strict fp32 kernel's error on the
    // kernel-dependent part of the gradient and 6.7e-6 on the loss -- with 77 features there are too few roundings per dot
    // product to average out.  Cost at BASELINE cfg3: 79.5 -> 84.5 us for the kernel.
    constexpr int GP = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const unsigned sm_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;

    const int problem = (int)blockIdx.x < nwg0 ? 0 : 1;
    const bool do_grad = GRAD && problem == 0;
    const int rem = (int)blockIdx.x - (problem ? nwg0 : 0);
    const int rb = rem / a.col_splits, sp = rem - rb * a.col_splits;
    const int i0 = rb * ROWS_WG + wave * ROWS;
    const long long bid = blockIdx.x;

    // ---- the rows of this wave, all features, hi and lo, as MFMA B operands of the Gram ----
    const _Float16* rows_hi = reinterpret_cast<const _Float16*>(a.cols[problem]) + (long long)a.row_offset * DPAD;
    const _Float16* rows_lo = reinterpret_cast<const _Float16*>(a.cols_lo[problem]) + (long long)a.row_offset * DPAD;
    const float* rnorm = a.cnorm[problem] + a.row_offset;
    f16x8_t zreg[RT][KBN], zlo[RT][KBN];
    float rn[RT];
    bool valid_i[RT];
    bool all_valid = true;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = i0 + 16 * t + i16;
        valid_i[t] = row < a.n_rows;
        all_valid = all_valid && valid_i[t];
#pragma unroll
        for (int s = 0; s < KBN; ++s) {
            if (valid_i[t]) {
                zreg[t][s] = *reinterpret_cast<const f16x8_t*>(rows_hi + (long long)row * DPAD + 32 * s + 8 * g);
                zlo[t][s] = *reinterpret_cast<const f16x8_t*>(rows_lo + (long long)row * DPAD + 32 * s + 8 * g);
            } else {
                zreg[t][s] = f16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
                zlo[t][s] = zreg[t][s];
            }
        }
        rn[t] = valid_i[t] ? rnorm[row] : 0.f;
    }
    const SplitScales sc = split_scales(*a.split_amax);
    const float gram_unscale2 = 2.f * sc.inv_s2;
#pragma unroll
    for (int t = 0; t < RT; ++t) {       // retire the loads before the loop (hipcc would wait vmcnt(0) at first use, draining the DMA)
#pragma unroll
        for (int s = 0; s < KBN; ++s) { asm volatile("" : "+v"(zreg[t][s])); asm volatile("" : "+v"(zlo[t][s])); }
        asm volatile("" : "+v"(rn[t]));
    }
    const bool any_invalid = EDGE && __any(!all_valid);

    f32x16 acc[GRAD ? KBN : 1][GRAD ? RB : 1];
    if (GRAD) {
#pragma unroll
        for (int s = 0; s < KBN; ++s)
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][b][r] = 0.f;
    }

    // column walk in 16-point tiles; symmetric y-y problem: unordered pairs of ROWS_WG-point blocks (parity rule of mmd.hip)
    // (walked in steps of PT tiles: ROWS_WG, the padding unit of the point sets, is a multiple of STEP_P)
    const long long tps = a.tiles_per_set / PT;                 // steps per set
    const bool sym = problem == 1 && a.sym;
    constexpr int BT = ROWS_WG / STEP_P;                        // steps per block of the symmetric walk
    const int blkI = a.row_offset / ROWS_WG + rb;
    const int nJ = (int)(tps / BT);
    const int symA = (blkI + 1) >> 1;
    const int symM = symA + 1 + ((nJ - 1 - blkI) >> 1);
    const long long total = problem == 0 ? 2 * tps : (sym ? (long long)BT * symM : tps);
    const long long base = (problem == 0 || sym) ? 0 : tps;
    const int tb = (int)(base + sp * total / a.col_splits);
    const int te = (int)(base + (sp + 1) * total / a.col_splits);
    auto actual = [&](int vt) -> int {
        if (!sym) return vt;
        const int q = vt / BT;
        const int J = q < symA ? ((blkI + 1) & 1) + 2 * q : (q == symA ? blkI : blkI + 2 * (q - symA));
        return (int)tps + BT * J + (vt - q * BT);
    };

    // per-lane constants of the LDS image (pieces of 16 points x 32 features, chunks swizzled: mmd_bf16.hip / mmd_pipe.hip)
    const int dma_r = lane >> 2, dma_c = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const unsigned dma_voff = (unsigned)(dma_r * DPAD + 8 * dma_c) * 2u;
    const unsigned dma_voff_lo = dma_voff + (unsigned)(reinterpret_cast<const char*>(a.cols_lo[0]) - reinterpret_cast<const char*>(a.cols[0])) - 1024u;
    const int gram_off = 16 * (4 * i16 + (g ^ (2 * ((i16 >> 2) & 1))));
    const int tq = i16 >> 2, tp = i16 & 3;
    int tr_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        tr_off[h] = 64 * (8 * (g >> 1) + 4 * h + tq) + 16 * ((2 * (g & 1) + (tp >> 1)) ^ (2 * h)) + 8 * (tp & 1);
    auto tr_read = [&](int byte_off) -> s16x4_t {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)(sm + byte_off));
    };
    // both sets' bases in scalar registers (indexing a.cols[set] would be a scalar memory load per tile)
    const _Float16* const cbase0 = reinterpret_cast<const _Float16*>(a.cols[0]);
    const _Float16* const cbase1 = reinterpret_cast<const _Float16*>(a.cols[1]);
    const float* const nbase0 = a.cnorm[0];
    const float* const nbase1 = a.cnorm[1];
    // position of a tile of the walk: located once, then stepped (the symmetric walk relocates at block boundaries)
    struct TileRef { int set, p0; };
    auto locate = [&](int vt) -> TileRef {
        const int tile = actual(vt);
        const int set = tile >= tps ? 1 : 0;
        return TileRef{set, (tile - (set ? (int)tps : 0)) * STEP_P};
    };
    const int set_pts = (int)tps * STEP_P;
    auto step = [&](const TileRef& r, int vt_next) -> TileRef {       // the tile after r, vt_next its index in the walk
        if (sym) return (vt_next % BT) ? TileRef{1, r.p0 + STEP_P} : locate(vt_next);
        const bool wrap = r.p0 + STEP_P >= set_pts;                   // problem 0 only: latents, then priors
        return TileRef{wrap ? 1 : r.set, wrap ? 0 : r.p0 + STEP_P};
    };
    // this wave's share of a step -> buffer buf: the hi + lo pairs j = wave, wave + 4, .. of the step's PT KBN (pair j:
    // tile j / KBN, feature block j % KBN); the last wave also the norms
    auto fetch = [&](const TileRef& r, int buf) {
        const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane(r.p0);
        // all 64 lanes move a float (no lane mask, no branch): the words past the step's norms are never read, and
        // the floats past the end of a norm array are workspace of this call (mmd.hip, make_plan)
        if (wave == N16_WAVES - 1)
            dma4((r.set ? nbase1 : nbase0) + p0, (unsigned)lane * 4u, sm_byte + (unsigned)(L::NORM_OFF + buf * 256));
        const float* X = reinterpret_cast<const float*>((r.set ? cbase1 : cbase0) + p0 * (unsigned)DPAD);   // the plan keeps n_pad * DPAD < 2^31
#pragma unroll
        for (int j0 = 0; j0 < PT * KBN; j0 += N16_WAVES) {
            const int j = j0 + wave;
            if (j < PT * KBN) {
                const int u = j / KBN, s = j - u * KBN;
                // (a workgroup without gradient never reads the lo copy under the two-product Gram)
                if (GP == 3 || do_grad) dma16_pair(X + (u * TILE_P * DPAD + 32 * s) / 2, dma_voff, dma_voff_lo, sm_byte + (unsigned)(buf * L::STEP_B + 2048 * j));
                else dma16(X + (u * TILE_P * DPAD + 32 * s) / 2, dma_voff, sm_byte + (unsigned)(buf * L::STEP_B + 2048 * j));
            }
        }
    };

    // The vector work runs in a domain scaled by 2^12 (SPLIT_E_SCALE, the scale the gradient's fp16 pairs want):
    // X = -2^12 t.  All scalings are powers of two, so every value is the unscaled one times 2^12 exactly.
    //   X = min(kS S' - 2^12 (|x|^2 + |z|^2) / D^2, 0);   p(x) = 1/2 + x/6 + ... + x^4/720 in Horner form on X with
    //   coefficients c_j 2^-12(j+1);   px = P X;   2^12 E = px X + X;   2^12 Q = px X      (em1_split, mmd_common.h)
    const float kS = gram_unscale2 * a.inv_d2 * SPLIT_E_SCALE;
    const float nscale = -a.inv_d2 * SPLIT_E_SCALE;
    f32x2 rn2[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) rn2[t] = f32x2{rn[t] * nscale, rn[t] * nscale};
    constexpr float C4 = (1.f / 720.f) * 0x1p-60f, C3 = (1.f / 120.f) * 0x1p-48f, C2 = (1.f / 24.f) * 0x1p-36f,
                    C1 = (1.f / 6.f) * 0x1p-24f, C0 = 0.5f * 0x1p-12f;
    constexpr float X_EXP = -EM1_POLY_MAX * SPLIT_E_SCALE;      // at or below: library exp
    constexpr float UNSCALE = 1.f / SPLIT_E_SCALE;

    // fp32 partials (scaled, two slots per lane), flushed to fp64 every 8 steps and when the column set changes
    double sum0 = 0.0, sum1 = 0.0, rowsum[RT];
    f32x2 qacc = {0.f, 0.f}, facc[RT];
    int qset = 0;
#pragma unroll
    for (int t = 0; t < RT; ++t) { rowsum[t] = 0.0; facc[t] = f32x2{0.f, 0.f}; }
    auto flush = [&]() {
        const double v = (double)((qacc.x + qacc.y) * UNSCALE);
        if (qset) sum1 += v; else sum0 += v;
        qacc = f32x2{0.f, 0.f};
        if (GRAD) {
#pragma unroll
            for (int t = 0; t < RT; ++t) { rowsum[t] += (double)((facc[t].x + facc[t].y) * UNSCALE); facc[t] = f32x2{0.f, 0.f}; }
        }
    };

#ifdef TVAE_PHASE_TIMING
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    TileRef cur = locate(tb < te ? tb : 0), nxt = cur;
    if (tb < te) fetch(cur, 0);
    for (int tile = tb; tile < te; ++tile) {                // (`tile`: a step of PT tiles)
        PHASE_MARK(5);
        cur = nxt;
        const int set = __builtin_amdgcn_readfirstlane(cur.set), p0s = __builtin_amdgcn_readfirstlane(cur.p0);
        const float wsym = (sym && p0s / ROWS_WG != blkI) ? 2.f : 1.f;
        const float sgn = set ? -1.f : 1.f;                 // prior columns enter the gradient and the row sums with s_p = -1
        const int buf = (tile - tb) & 1;
        if (set != qset) { flush(); qset = set; }
        wait_vmcnt<0>();                                    // this wave's share of the step has landed ...
        __syncthreads();                                    // ... everyone's has, and the other buffer is free again
        PHASE_MARK(0);
        if (tile + 1 < te) { nxt = step(cur, tile + 1); fetch(nxt, buf ^ 1); }
        PHASE_MARK(1);
#pragma unroll 1       // (unrolled, the two tiles' live ranges overlap: 28-68 B of scratch per lane)
      for (int u = 0; u < PT; ++u) {
        const int p0 = p0s + u * TILE_P;
        const char* xs = sm + buf * L::STEP_B + u * L::TILE_B;

        // 1. Gram: S[t][r] = <x_p, z_i> (scaled), p = p0 + 4 g + r, i = i0 + 16 t + i16
        f32x4 S[RT];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KBN; ++s) {
            const f16x8_t ab = *reinterpret_cast<const f16x8_t*>(xs + gram_off + 2048 * s);
            f16x8_t al;
            if constexpr (GP == 3) al = *reinterpret_cast<const f16x8_t*>(xs + gram_off + 2048 * s + 1024);
#pragma unroll
            for (int t = 0; t < RT; ++t) S[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ab, zreg[t][s], s == 0 ? zero : S[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) S[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ab, zlo[t][s], S[t], 0, 0, 0);
            if constexpr (GP == 3) {
#pragma unroll
                for (int t = 0; t < RT; ++t) S[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, zreg[t][s], S[t], 0, 0, 0);
            }
        }
        PHASE_MARK(2);

        // 2. E = expm1(-t), two values per instruction (v_pk_*_f32); h = 0: points 4 g, 4 g + 1; h = 1: + 2, + 3.
        // One row tile at a time, through to its E pieces in LDS (short live ranges: the register budget of two or
        // three waves per SIMD is what this kernel lives on).
        const f32x4 cn = *reinterpret_cast<const f32x4*>(sm + L::NORM_OFF + buf * 256 + u * 64 + 16 * g);
        const f32x2 cn2[2] = {f32x2{cn[0], cn[1]} * nscale, f32x2{cn[2], cn[3]} * nscale};
        const bool ragged = EDGE && p0 + TILE_P > a.n_cols;
        char* eb = sm + L::E_OFF + wave * (2 * L::EPIECE_B);
        f32x2 qt = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            f32x2 X[2], es[2], qs[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 Sv = {S[t][2 * h], S[t][2 * h + 1]};
                X[h] = __builtin_elementwise_fma(f32x2{kS, kS}, Sv, cn2[h] + rn2[t]);
                X[h] = f32x2{fminf(X[h].x, 0.f), fminf(X[h].y, 0.f)};
                f32x2 P = __builtin_elementwise_fma(f32x2{C4, C4}, X[h], f32x2{C3, C3});
                P = __builtin_elementwise_fma(P, X[h], f32x2{C2, C2});
                P = __builtin_elementwise_fma(P, X[h], f32x2{C1, C1});
                P = __builtin_elementwise_fma(P, X[h], f32x2{C0, C0});
                const f32x2 px = P * X[h];
                es[h] = __builtin_elementwise_fma(px, X[h], X[h]);
                qs[h] = px * X[h];
            }
            const float xmin = fminf(fminf(X[0].x, X[0].y), fminf(X[1].x, X[1].y));
            if (__any(xmin <= X_EXP)) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float Xv = X[h][c];
                        const float ex = (expf(Xv * UNSCALE) - 1.f) * SPLIT_E_SCALE;      // |E| >= 0.1175 here: relative error < 1e-6
                        const bool big = Xv <= X_EXP;
                        es[h][c] = big ? ex : es[h][c];
                        qs[h][c] = big ? ex - Xv : qs[h][c];
                    }
            }
            if (EDGE && any_invalid && !valid_i[t]) { es[0] = es[1] = f32x2{0.f, 0.f}; qs[0] = qs[1] = f32x2{0.f, 0.f}; }
            if (EDGE && ragged) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        if (p0 + 4 * g + 2 * h + c >= a.n_cols) { es[h][c] = 0.f; qs[h][c] = 0.f; }
            }
            // sums carry Q (linear part: closed form in the finalize kernel); row sums carry s_p E
            qt += qs[0] + qs[1];
            if (GRAD && do_grad) {
                es[0] *= sgn; es[1] *= sgn;
                facc[t] += es[0] + es[1];
                // 3a. the E pieces of this row tile: [piece][row][16 points], the two 16-byte halves of a row swapped
                // on rows 8 .. 15 of a tile
                f16x4_t p0v, p1v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = es[r >> 1][r & 1];
                    p0v[r] = (_Float16)v;
                    p1v[r] = (_Float16)(v - (float)p0v[r]);
                }
                const int at = (t * 16 + i16) * 32 + 16 * ((g >> 1) ^ ((i16 >> 3) & 1)) + 8 * (g & 1);
                *reinterpret_cast<uint2*>(eb + at) = __builtin_bit_cast(uint2, p0v);
                *reinterpret_cast<uint2*>(eb + L::EPIECE_B + at) = __builtin_bit_cast(uint2, p1v);
            }
        }
        qacc = __builtin_elementwise_fma(f32x2{wsym, wsym}, qt, qacc);
        PHASE_MARK(3);

        // 3. gradient: acc^T[d][i] += X^T[d][p] (s_p E[p][i]) on 32 x 32 tiles, K = the 16 points of the tile
        if (GRAD && do_grad) {
            f16x8_t Bh[RB], Bl[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int n = lane & 31;
                const int at = (32 * b + n) * 32 + 16 * ((lane >> 5) ^ ((n >> 3) & 1));
                Bh[b] = *reinterpret_cast<const f16x8_t*>(eb + at);
                Bl[b] = *reinterpret_cast<const f16x8_t*>(eb + L::EPIECE_B + at);
            }
            const int xb0 = buf * L::STEP_B + u * L::TILE_B;
#pragma unroll
            for (int s = 0; s < KBN; ++s) {
                const s16x4_t h0 = tr_read(xb0 + 2048 * s + tr_off[0]), h1 = tr_read(xb0 + 2048 * s + tr_off[1]);
                const s16x4_t l0 = tr_read(xb0 + 2048 * s + 1024 + tr_off[0]), l1 = tr_read(xb0 + 2048 * s + 1024 + tr_off[1]);
                const f16x8_t Ah = __builtin_bit_cast(f16x8_t, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                const f16x8_t Al = __builtin_bit_cast(f16x8_t, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int b = 0; b < RB; ++b) acc[s][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[b], acc[s][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < RB; ++b) acc[s][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[b], acc[s][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < RB; ++b) acc[s][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[b], acc[s][b], 0, 0, 0);
            }
        }
        PHASE_MARK(4);
      }
        if (((tile - tb) & 7) == 7) flush();
    }
    flush();
    wait_vmcnt<0>();
#ifdef TVAE_PHASE_TIMING
    if ((blockIdx.x == 0 || (int)blockIdx.x == nwg0) && lane == 0 && te > tb)
        printf("problem %d tiles %d: wait %llu  fetch %llu  gram %llu  expm1 %llu  grad %llu  top %llu (cycles per tile)\n", problem, te - tb,
               ph[0] / (te - tb), ph[1] / (te - tb), ph[2] / (te - tb), ph[3] / (te - tb), ph[4] / (te - tb), ph[5] / (te - tb));
#endif

    // the four waves' sums in a fixed order
    const double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    __syncthreads();                                        // the E buffers are dead
    double* red = reinterpret_cast<double*>(sm + L::E_OFF);
    if (lane == 0) { red[2 * wave] = w0; red[2 * wave + 1] = w1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.wg_sums[2 * bid + 0] = (red[0] + red[2]) + (red[4] + red[6]);
        a.wg_sums[2 * bid + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
    if (GRAD && do_grad) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            double rv = rowsum[t];
            rv += __shfl_xor(rv, 16, 64);
            rv += __shfl_xor(rv, 32, 64);
            if (g == 0) a.wg_rowsum[(long long)sp * a.n_rows_pad + i0 + 16 * t + i16] = rv;
        }
        // 32x32 tiles: register r of a lane is feature 8 (r / 4) + 4 (lane / 32) + r % 4 of row lane % 32
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 32 * b + (lane & 31)) * DPAD + 4 * (lane >> 5);
#pragma unroll
            for (int s = 0; s < KBN; ++s)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *reinterpret_cast<f32x4*>(grow + 32 * s + 8 * q4) =
                        f32x4{acc[s][b][4 * q4], acc[s][b][4 * q4 + 1], acc[s][b][4 * q4 + 2], acc[s][b][4 * q4 + 3]};
        }
    }
}

template <int KBN, int RT, bool GRAD, bool EDGE>
void launch_n16_inst(const MmdArgs& a, int nwg0, int nwg1, hipStream_t s) {
    constexpr int lds_bytes = N16Lds<KBN, RT>::BYTES;
    hipLaunchKernelGGL((mmd_n16_kernel<KBN, RT, GRAD, EDGE>), dim3(nwg0 + nwg1), dim3(64 * N16_WAVES), lds_bytes, s, a, nwg0);
}

template <int KBN, int RT>
int launch_n16(bool grad, const MmdArgs& a, int nwg0, int nwg1, hipStream_t s) {
    // no partial row block, and no padding points behind the last column (the walk covers the PADDED sets: a row block of
    // whole workgroups out of a point count that is not a multiple of the padding unit still needs the masks)
    const bool edge = a.n_rows % (N16_WAVES * 16 * RT) != 0 || a.n_cols != a.tiles_per_set * TILE_P;
    if (grad) { if (edge) launch_n16_inst<KBN, RT, true, true>(a, nwg0, nwg1, s); else launch_n16_inst<KBN, RT, true, false>(a, nwg0, nwg1, s); }
    else { if (edge) launch_n16_inst<KBN, RT, false, true>(a, nwg0, nwg1, s); else launch_n16_inst<KBN, RT, false, false>(a, nwg0, nwg1, s); }
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

}  // namespace

int mmd_n16_rows(int kbn) { return N16_WAVES * 16 * (kbn <= 2 ? TVAE_N16_RT_SMALL : 2); }
int mmd_n16_wg_per_cu(int kbn) { return N16_WPE(kbn, kbn <= 2 ? TVAE_N16_RT_SMALL : 2); }      // four-wave workgroups: one wave per SIMD each

int launch_mmd_n16(bool grad, int kbn, const MmdArgs& a, int nwg0, int nwg1, hipStream_t s) {
    switch (kbn) {
        case 1: return launch_n16<1, TVAE_N16_RT_SMALL>(grad, a, nwg0, nwg1, s);
        case 2: return launch_n16<2, TVAE_N16_RT_SMALL>(grad, a, nwg0, nwg1, s);
        case 3: return launch_n16<3, 2>(grad, a, nwg0, nwg1, s);
        case 4: return launch_n16<4, 2>(grad, a, nwg0, nwg1, s);
        case 5: return launch_n16<5, 2>(grad, a, nwg0, nwg1, s);
        case 6: return launch_n16<6, 2>(grad, a, nwg0, nwg1, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no narrow 16-bit kernel for %d feature blocks", kbn);
}

}  // namespace tvae
