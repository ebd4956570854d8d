// MMD tile kernel on the bf16 matrix cores (bf16 latents and priors, latent_size > 192).
// Math, workspace layout and the per-workgroup outputs are those of mmd.hip; see DESIGN.md 4.1b.
#include <mutex>

#include "mmd_common.h"

namespace tvae {
namespace {

// =======================================================================================
// bf16 latents, wide dimension (DPAD = 128 KB, KB = 2..8): the same algorithm on the 16x bf16 matrix cores.
//
// One workgroup = 4 waves owns 32 rows (two 16-row tiles t) and walks 32-point steps (two 16-point
// tiles pt); the feature dimension is sliced over the waves in KB blocks of 32 (one
// v_mfma_f32_16x16x32_bf16 k-step each).  Differences from the fp32 kernel:
//   * the four partial S' tiles (pt, t) are REDUCE-SCATTERED: wave w finishes tile (pt = w>>1, t = w&1)
//     only (three remote partials), so the expm1 work is split over the waves instead of replicated;
//   * E (signed) is rounded to bf16 and shared through a 2 KiB LDS buffer laid out so that every wave
//     reads back, with one ds_read_b128 per row tile, exactly its B operand of the gradient MFMA
//     (k = 8g + j  <->  point tile j>>2, point 4g + (j&3));
//   * the gradient's A operand X^T comes from the SAME LDS image through ds_read_b64_tr_b16.
// LDS image of a wave's slice (32 points x SW bf16): DMA instruction 2 s + a carries features
// [32 s, 32 s + 32) of points [16 a, 16 a + 16) = 64 chunks of 16 B; lane l fetches chunk
// c = (l & 3) ^ 2 ((l >> 4) & 1) of point rr = l >> 2, so every point contributes 64 contiguous bytes per
// instruction (16 half cache lines per wave-instruction; a 32-B-per-point variant touched 32 lines and ran
// at a fifth of the matrix-core rate), and chunk (rr, c) sits at position 4 rr + (c ^ 2 ((rr >> 2) & 1)):
// the Gram row read (ds_read_b128) and the transposed gradient read are both bank-conflict free, each
// instruction's 1 KiB is contiguous (in-place refill as in the fp32 kernel) and all instructions share one
// per-lane source offset.
// Sums and row sums use the fp32 E; only the gradient product sees the bf16 rounding of E (2^-9).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

// 16x16x32 matrix instruction on 16-bit operands held in a bf16x8_t container: bf16, or (F16) the same bits read as fp16
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x4& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

constexpr int STEP_P = 32;


// RT = 16-row tiles per workgroup; with 2 RT tiles per step each of the 4 waves finishes TPW = RT / 2 of
// them.  Production uses RT = 2 (32 rows, two workgroups per CU).  RT = 4 (64 rows: half the L2 -> LDS
// traffic per flop, but 434 registers per lane = one wave per SIMD) was measured SLOWER at every width
// (N = 32768: D = 768 11.3 ms vs 9.05 ms, D = 256 5.06 vs 4.49): with 16-cycle MFMAs the second resident
// workgroup, which hides the exchange / expm1 / barrier phases, is worth more than the halved traffic.
#ifndef TVAE_QF_GRAM_PRODUCTS
#define TVAE_QF_GRAM_PRODUCTS 1     // Gram products of the Q-form instances (1: hi x hi; 2: + hi x lo of the rows)
#endif
#ifndef TVAE_QF32_MIN_KB
#define TVAE_QF32_MIN_KB 7         // the 32-row kernel has Q-form instances from this many feature blocks per wave on (narrower: mmd_q32.hip)
#endif
#ifndef TVAE_QF_TWO_WG
#define TVAE_QF_TWO_WG 1           // two workgroups per CU for the Q-form instances of the 32-row kernel at KB >= 7
#endif
// QF: the Q-form instance of a split launch pair never holds the lo image of the columns nor an E_lo piece in LDS
template <int KB, int RT, bool GRAD, bool SPLIT, bool QF = false>
struct Bf16Lds {
    static constexpr int SW = 32 * KB;                       // bf16 features per wave
    static constexpr int TPW = RT / 2;                       // S' tiles finished per wave
    static constexpr int SLICE_B = STEP_P * SW * 2;          // bytes of one wave's slice of a 32-point step (one image)
    static constexpr bool LO = SPLIT && !(QF && TVAE_GRAM_PRODUCTS == 2);
    static constexpr int SLICE_W = (LO ? 2 : 1) * SLICE_B;      // LO: the hi image, then the lo image
    static constexpr int NORM_B = 4 * SLICE_W;               // [2 buf][4 waves][32] float point norms
    static constexpr int XCH_B = NORM_B + 2 * 4 * STEP_P * 4;
    static constexpr int XCH_ONE_B = 2 * RT * 3 * 64 * 16;   // [tile][3 remote sources][64 lanes] float4
    // the gradient instantiation has a second barrier per step, which also protects a single exchange buffer
    static constexpr int XBUFS = GRAD ? 1 : (KB <= 6 ? 2 : 1);
    static constexpr int EBUF_B = XCH_B + XBUFS * XCH_ONE_B; // [RT row tiles][64 lanes][2 point tiles] 4 x bf16
    static constexpr int EBUF_ONE_B = RT * 64 * 16;          // SPLIT: E_hi, then E_lo
    static constexpr int BYTES = EBUF_B + (GRAD ? (LO ? 2 : 1) * EBUF_ONE_B : 0);
    // two workgroups per CU when LDS and registers (<= 256 per lane) both allow it
    // (the Q-form gradient instance at 768 < D <= 1 024: 79 KB and, capped by the launch bounds, 256 registers)
    static constexpr int WG_PER_CU = ((!SPLIT && RT == 2 && KB <= 6) || (TVAE_QF_TWO_WG && QF && RT == 2 && KB >= TVAE_QF32_MIN_KB && 2 * BYTES <= 160 * 1024)) ? 2 : 1;
    static_assert(RT == 2 || RT == 4, "2 or 4 row tiles");
    static_assert(BYTES <= 160 * 1024, "bf16 tile kernel exceeds the 160 KiB LDS");
    static_assert(XCH_ONE_B >= (8 + 4 * TPW * 16) * 8, "final reduction reuses the exchange buffer");
};

// SPLIT (fp32 inputs, two (Gram, TVAE_GRAM_PRODUCTS) or three (gradient) fp16 products per fp32 product): every fp32 value v is held as hi = fp16(s v) and
// lo = fp16(s v - hi), s a power of two that puts the largest |v| of the call at 2^11 (22 significant bits, error
// <= 2^-22 of the largest value: the fp32 input MFMA's own class).  The Gram is X_hi Z_hi + X_hi Z_lo + X_lo Z_hi and
// the gradient product X_hi^T E_hi + X_lo^T E_hi + X_hi^T E_lo (E in (-1, 0] scaled by 2^12), all accumulated in fp32
// by the matrix cores; the dropped lo x lo terms are 2^-22 of a product.  A first version on bf16 halves (16 bits)
// failed the 1e-5 bar on the gradient when the kernel values are O(1) (far-apart points: 1.003e-5).  Everything else
// (norms, t, expm1, sums, row sums) is the fp32 / fp64 arithmetic of the exact kernel; the scale is undone exactly.
// QF (split gradient launches): the Q-form instance of mmd_pipe.hip / DESIGN 4.1g -- gradient and row sums carry
// Q = E + t in ONE fp16 product, the lo images of columns and E are neither fetched nor formed.  A launch pair; the
// device-side flag picks the instance that runs.
#ifndef TVAE_QF_GRAM_PRODUCTS
#define TVAE_QF_GRAM_PRODUCTS 1     // Gram products of the Q-form instance: its sums carry Q = E + t, which tolerates the hi x hi Gram
#endif
template <int KB, int RT, bool GRAD, bool SPLIT, bool QF = false>
__global__ __launch_bounds__(256, (Bf16Lds<KB, RT, GRAD, SPLIT, QF>::WG_PER_CU)) void mmd_bf16_kernel(MmdArgs a) {
    static_assert(!QF || (SPLIT && TVAE_GRAM_PRODUCTS == 2), "Q-form: split launches with the two-product Gram");
    // launch pairs under the device flag (mmd.hip, QFlag): the gradient launch always; the sum-only launch of the 32-row
    // kernel's widths (KB >= 7), whose QF member differs from the other only in forming ONE Gram product
    if constexpr (SPLIT && (GRAD || (RT == 2 && KB >= TVAE_QF32_MIN_KB))) {
        if (a.qflag != nullptr && (__builtin_amdgcn_readfirstlane((int)*a.qflag) != 0) != QF) return;
    }
    using L = Bf16Lds<KB, RT, GRAD, SPLIT, QF>;
    static_assert(GRAD || RT == 2, "the symmetric y-y enumeration assumes 32-row blocks");
    constexpr int SW = L::SW, DPAD = 4 * SW, NI = 2 * KB;     // NI: 16-feature blocks per slice
    // (with the two-product Gram of mmd_common.h a sum-only launch never reads the lo copy of the columns)
    constexpr bool NEED_LO = SPLIT && ((GRAD && !QF) || TVAE_GRAM_PRODUCTS == 3);
    constexpr int DPS = NEED_LO ? 4 : 2;                      // DMA instructions per 32-feature block pair
    constexpr int TPW = L::TPW, ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned xs_byte = smem_byte + (unsigned)(wave * L::SLICE_W);
    const char* xs = sm + wave * L::SLICE_W;
    // the S' tiles this wave finishes: point tile my_pt, row tiles my_t0 .. my_t0 + TPW - 1
    const int my_pt = wave >> 1, my_t0 = (wave & 1) * TPW;

    const int per_problem = a.row_blocks * a.col_splits;
    const int problem = GRAD ? 0 : a.problem_base + (int)blockIdx.x / per_problem;
    const int rem = (int)blockIdx.x - (problem - a.problem_base) * per_problem;
    const int rb = a.sp_major ? rem % a.row_blocks : rem / a.col_splits;
    const int sp = a.sp_major ? rem / a.row_blocks : rem - rb * a.col_splits;
    const int i0 = rb * ROWS;
    const long long bid = (long long)a.sums_base + blockIdx.x;

    // ---- the rows of this workgroup: this wave's feature slice, in registers as MFMA B operands ----
    const bf16* rows = reinterpret_cast<const bf16*>(a.cols[problem]) + (long long)a.row_offset * DPAD;
    const float* rnorm = a.cnorm[problem] + a.row_offset;
    const bf16* rows_lo = SPLIT ? reinterpret_cast<const bf16*>(a.cols_lo[problem]) + (long long)a.row_offset * DPAD : rows;
    constexpr bool ROW_LO = SPLIT && !(QF && TVAE_QF_GRAM_PRODUCTS == 1);      // the Gram's second product: row side lo
    bf16x8_t zreg[RT][KB], zlo[ROW_LO ? RT : 1][ROW_LO ? KB : 1];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = i0 + 16 * t + i16;
        const bf16* zp = rows + (long long)row * DPAD + wave * SW + 8 * g;
        const bf16* zl = rows_lo + (long long)row * DPAD + wave * SW + 8 * g;
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            if (row < a.n_rows) zreg[t][s] = *reinterpret_cast<const bf16x8_t*>(zp + 32 * s);
            else zreg[t][s] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (ROW_LO) {
                if (row < a.n_rows) zlo[t][s] = *reinterpret_cast<const bf16x8_t*>(zl + 32 * s);
                else zlo[t][s] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    // SPLIT: the Gram was formed on operands scaled by s (a power of two): 2 S = (2 / s^2) S'
    const float gram_unscale2 = SPLIT ? 2.f * split_scales(*a.split_amax).inv_s2 : 2.f;
    bool my_valid[TPW];
    float rn[TPW];
    bool all_valid = true;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int my_row = i0 + 16 * (my_t0 + j) + i16;
        my_valid[j] = my_row < a.n_rows;
        rn[j] = my_valid[j] ? rnorm[my_row] : 0.f;
        all_valid = all_valid && my_valid[j];
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            asm volatile("" : "+v"(zreg[t][s]));
            if constexpr (ROW_LO) asm volatile("" : "+v"(zlo[t][s]));
        }
#pragma unroll
    for (int j = 0; j < TPW; ++j) asm volatile("" : "+v"(rn[j]));
    const bool any_invalid = __any(!all_valid);

    f32x4 acc[GRAD ? NI : 1][RT];
    if (GRAD) {
#pragma unroll
        for (int db = 0; db < NI; ++db)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[db][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // 32-point steps; the symmetric y-y enumeration is the fp32 kernel's (one step = one block J)
    const int nJ = a.tiles_per_set >> 1;
    const bool sym = !GRAD && problem == 1 && a.sym;
    const int blkI = (a.row_offset >> 5) + rb;
    const int symA = (blkI + 1) >> 1;
    const int symM = symA + 1 + ((nJ - 1 - blkI) >> 1);
    const long long total = problem == 0 ? 2ll * nJ : (sym ? (long long)symM : (long long)nJ);
    const long long base = (problem == 0 || sym) ? 0 : nJ;
    const int sb = (int)(base + sp * total / a.col_splits);
    const int se = (int)(base + (sp + 1) * total / a.col_splits);
    auto actual = [&](int v) -> int {
        if (!sym) return v;
        const int J = v < symA ? ((blkI + 1) & 1) + 2 * v : (v == symA ? blkI : blkI + 2 * (v - symA));
        return nJ + J;
    };

    // per-lane constants of the LDS image
    const int dma_r = lane >> 2, dma_c = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const unsigned dma_voff = (unsigned)(dma_r * DPAD + 8 * dma_c) * 2u;
    const int gram_off = 16 * (4 * i16 + (g ^ (2 * ((i16 >> 2) & 1))));            // + 1024 (2 s + pt)
    const int tq = i16 >> 2, tp = i16 & 3;
    // transposed read of 16-feature block db of point tile pt: + 1024 (2 (db >> 1) + pt), base by parity of db
    const int tr_off0 = 16 * (4 * (4 * g + tq) + 2 * (g & 1) + (tp >> 1)) + 8 * (tp & 1);
    const int tr_off1 = 16 * (4 * (4 * g + tq) + 2 * ((g & 1) ^ 1) + (tp >> 1)) + 8 * (tp & 1);
    auto tr_read = [&](int byte_off) -> s16x4_t {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)(sm + byte_off));
    };

    // this wave's slice of step v in the hi (only) copy and, SPLIT, the lo copy
    struct Src { const bf16* hi; const bf16* lo; };
    auto step_src = [&](int v) -> Src {
        const int u = actual(v);
        const int set = u >= nJ ? 1 : 0;
        const int p0 = (u - set * nJ) * STEP_P;
        const long long off = (long long)p0 * DPAD + wave * SW;
        return Src{reinterpret_cast<const bf16*>(a.cols[set]) + off,
                   SPLIT ? reinterpret_cast<const bf16*>(a.cols_lo[set]) + off : nullptr};
    };
    auto stage_norms = [&](int v, int nbuf) {
        const int u = actual(v);
        const int set = u >= nJ ? 1 : 0;
        const int p0 = (u - set * nJ) * STEP_P;
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            smem_byte + (unsigned)(L::NORM_B + (nbuf * 4 + wave) * STEP_P * 4));
        const float* src = uniform_ptr(a.cnorm[set] + p0);
        if (lane < STEP_P) dma4(src, (unsigned)lane * 4u, dst);
    };
    const unsigned dma_voff_b = dma_voff + 16u * DPAD * 2u - 1024u;      // second point tile, see dma16_pair
    auto dma_pair = [&](const Src& X, int s) {       // features [32 s, 32 s + 32) of both point tiles (SPLIT: hi, then lo)
        dma16_pair(reinterpret_cast<const float*>(X.hi + 32 * s), dma_voff, dma_voff_b, xs_byte + 2048u * s);
        if constexpr (NEED_LO)
            dma16_pair(reinterpret_cast<const float*>(X.lo + 32 * s), dma_voff, dma_voff_b,
                       xs_byte + (unsigned)L::SLICE_B + 2048u * s);
    };

    double sum0 = 0.0, sum1 = 0.0, rowsum[TPW];
    float f0 = 0.f, f1 = 0.f, fr[TPW];      // fp32 partials of the sums, flushed to fp64 every 8 steps
#pragma unroll
    for (int j = 0; j < TPW; ++j) { rowsum[j] = 0.0; fr[j] = 0.f; }

    // DMA issue order: norms(k) ... X(k) blocks 0..NI-1 ...; norms(k+1) at the top of iteration k.
    // The last step prefetches ITSELF again (harmless, once per workgroup): no branch around any DMA, and
    // the counted waits hold on every iteration.
    if (sb < se) {
        stage_norms(sb, 0);
        const Src X0 = step_src(sb);
#pragma unroll
        for (int s = 0; s < KB; ++s) dma_pair(X0, s);
    }
#ifdef TVAE_PHASE_TIMING
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    for (int step = sb; step < se; ++step) {
        PHASE_MARK(5);
        const int u = actual(step);
        const int set = u >= nJ ? 1 : 0;
        const int p0 = (u - set * nJ) * STEP_P;
        const float wsym = (sym && (p0 >> 5) != blkI) ? 2.f : 1.f;
        const int buf = (step - sb) & 1;
        const int nstep = step + 1 < se ? step + 1 : step;
        const Src Xn = step_src(nstep);
        stage_norms(nstep, buf ^ 1);

        // 1. partial Gram over the slice: S[pt][t] (16 points x 16 rows each)
        //    The operand reads run GD k-steps ahead of their MFMAs.  Before the read of k-step s the DMAs
        //    still allowed in flight are the younger blocks of this step, norms(k+1) and -- sum-only
        //    instantiation, which refills block pair s right after its MFMAs -- the pairs of the next step
        //    already issued.
        f32x4 S[2][RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) S[0][t] = S[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int GD = KB < 2 ? KB : 2;
        bf16x8_t ab[KB][2], al[SPLIT ? KB : 1][2];
        // before the read of k-step sr: all but the younger blocks of this step, norms(k+1) and (sum-only) the
        // refills already issued may still be in flight
        auto gram_wait = [&](int sr, int issued_pairs) {
            wait_vmcnt_n(DPS * (KB - 1 - sr) + 1 + (GRAD ? 0 : DPS * issued_pairs));
        };
        auto gram_read = [&](int sr) {
            ab[sr][0] = *reinterpret_cast<const bf16x8_t*>(xs + gram_off + 2048 * sr);
            ab[sr][1] = *reinterpret_cast<const bf16x8_t*>(xs + gram_off + 2048 * sr + 1024);
            if constexpr (L::LO) {
                al[sr][0] = *reinterpret_cast<const bf16x8_t*>(xs + L::SLICE_B + gram_off + 2048 * sr);
                al[sr][1] = *reinterpret_cast<const bf16x8_t*>(xs + L::SLICE_B + gram_off + 2048 * sr + 1024);
            }
        };
#pragma unroll
        for (int s = 0; s < GD; ++s) {
            gram_wait(s, 0);
            gram_read(s);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            if (s + GD < KB) {
                gram_wait(s + GD, s);
                gram_read(s + GD);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads GD k-steps ahead (hipcc sinks them otherwise)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                S[0][t] = mfma16<SPLIT>(ab[s][0], zreg[t][s], S[0][t]);
                S[1][t] = mfma16<SPLIT>(ab[s][1], zreg[t][s], S[1][t]);
                if constexpr (ROW_LO) {
                    S[0][t] = mfma16<SPLIT>(ab[s][0], zlo[t][s], S[0][t]);
                    S[1][t] = mfma16<SPLIT>(ab[s][1], zlo[t][s], S[1][t]);
#if TVAE_GRAM_PRODUCTS == 3
                    S[0][t] = mfma16<SPLIT>(al[s][0], zreg[t][s], S[0][t]);
                    S[1][t] = mfma16<SPLIT>(al[s][1], zreg[t][s], S[1][t]);
#endif
                }
            }
            if (!GRAD) {
#pragma unroll
                for (int t = 0; t < RT; ++t) asm volatile("" ::"v"(S[0][t]), "v"(S[1][t]));
                dma_pair(Xn, s);
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) asm volatile("" ::"v"(S[0][t]), "v"(S[1][t]));
        PHASE_MARK(0);

        // 2. reduce-scatter of the partials: tile q = RT pt + t goes to wave q / TPW
        f32x4* xb = reinterpret_cast<f32x4*>(sm + L::XCH_B + (L::XBUFS == 2 ? buf * L::XCH_ONE_B : 0));
#pragma unroll
        for (int q = 0; q < 2 * RT; ++q) {
            const int owner = q / TPW;
            if (wave != owner) xb[(q * 3 + ((wave - owner - 1) & 3)) * 64 + lane] = S[q / RT][q % RT];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PHASE_MARK(1);
        f32x4 sv[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            // own partial: tile (pt = wave >> 1, t = (wave & 1) TPW + j)
            sv[j] = wave == 0 ? S[0][j] : (wave == 1 ? S[0][TPW + j] : (wave == 2 ? S[1][j] : S[1][TPW + j]));
            const int q = wave * TPW + j;
            sv[j] += xb[(q * 3 + 0) * 64 + lane];
            sv[j] += xb[(q * 3 + 1) * 64 + lane];
            sv[j] += xb[(q * 3 + 2) * 64 + lane];
        }
        if (!GRAD && L::XBUFS == 1) {       // single exchange buffer without the E barrier: fence the reads
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }

        // 3. E = expm1(-t) for this wave's tiles; register r <-> point p0 + 16 pt + 4 g + r, column = row i16
        const f32x4 cn = *reinterpret_cast<const f32x4*>(sm + L::NORM_B + ((buf * 4 + wave) * STEP_P + 16 * my_pt + 4 * g) * 4);
        f32x4 e[TPW], q[TPW];
        float tmax = 0.f;
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float tt = fmaxf((cn[r] + rn[j] - gram_unscale2 * sv[j][r]) * a.inv_d2, 0.f);
                sv[j][r] = tt;
                tmax = fmaxf(tmax, tt);
            }
        if (!__any(tmax >= EM1_TINY)) {
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const EQ v_ = em1_split_tiny(sv[j][r]); e[j][r] = v_.e; q[j][r] = v_.q; }
        } else {
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const EQ v_ = em1_split(sv[j][r]); e[j][r] = v_.e; q[j][r] = v_.q; }
            if (__any(tmax >= EM1_POLY_MAX)) {
#pragma unroll
                for (int j = 0; j < TPW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (sv[j][r] >= EM1_POLY_MAX) { e[j][r] = expm1f(-sv[j][r]); q[j][r] = e[j][r] + sv[j][r]; }
            }
        }
        if (any_invalid) {
#pragma unroll
            for (int j = 0; j < TPW; ++j)
                if (!my_valid[j]) { e[j] = f32x4{0.f, 0.f, 0.f, 0.f}; q[j] = e[j]; }
        }
        if (p0 + STEP_P > a.n_cols) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (p0 + 16 * my_pt + 4 * g + r >= a.n_cols) {
#pragma unroll
                    for (int j = 0; j < TPW; ++j) { e[j][r] = 0.f; q[j][r] = 0.f; }
                }
        }
        // sums carry Q (linear part: closed form in the finalize kernel); row sums carry E (Q-form: Q as well)
        if constexpr (QF) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) e[j] = q[j];
        }
        float qs_all = 0.f;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            qs_all += (q[j][0] + q[j][1]) + (q[j][2] + q[j][3]);
            if (GRAD) {
                const float es = (e[j][0] + e[j][1]) + (e[j][2] + e[j][3]);
                fr[j] += set ? -es : es;
            }
        }
        if (set) f1 = fmaf(wsym, qs_all, f1);
        else f0 += qs_all;
        if (((step - sb) & 7) == 7) {
            sum0 += (double)f0; sum1 += (double)f1;
            f0 = f1 = 0.f;
#pragma unroll
            for (int j = 0; j < TPW; ++j) { rowsum[j] += (double)fr[j]; fr[j] = 0.f; }
        }

        // 4. gradient: acc^T[d][i] += X^T[d][p] (s_p E[p][i]) over the 32 points of the step in ONE k-step
        if (GRAD) {
            uint2* ebuf = reinterpret_cast<uint2*>(sm + L::EBUF_B);
            // prior columns enter the gradient with s_p = -1: flip the four sign bits of the packed values
            const unsigned flip = set ? 0x80008000u : 0u;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if constexpr (SPLIT) {
                    f16x4_t eh, el;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float es = e[j][r] * SPLIT_E_SCALE;
                        eh[r] = (_Float16)es;
                        if constexpr (!QF) el[r] = (_Float16)(es - (float)eh[r]);
                    }
                    uint2 hbits = __builtin_bit_cast(uint2, eh);
                    hbits.x ^= flip; hbits.y ^= flip;
                    ebuf[((my_t0 + j) * 64 + lane) * 2 + my_pt] = hbits;
                    if constexpr (!QF) {
                        uint2 lbits = __builtin_bit_cast(uint2, el);
                        lbits.x ^= flip; lbits.y ^= flip;
                        ebuf[(L::EBUF_ONE_B / 8) + ((my_t0 + j) * 64 + lane) * 2 + my_pt] = lbits;
                    }
                } else {
                    bf16x4_t eb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) eb[r] = (__bf16)e[j][r];
                    uint2 ebits = __builtin_bit_cast(uint2, eb);
                    ebits.x ^= flip; ebits.y ^= flip;
                    ebuf[((my_t0 + j) * 64 + lane) * 2 + my_pt] = ebits;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PHASE_MARK(2);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            PHASE_MARK(3);
            bf16x8_t Eb[RT], El[SPLIT ? RT : 1];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                Eb[t] = *reinterpret_cast<const bf16x8_t*>(sm + L::EBUF_B + (t * 64 + lane) * 16);
                if constexpr (SPLIT && !QF) El[t] = *reinterpret_cast<const bf16x8_t*>(sm + L::EBUF_B + L::EBUF_ONE_B + (t * 64 + lane) * 16);
            }
            const int tb0 = wave * L::SLICE_W + tr_off0, tb1 = wave * L::SLICE_W + tr_off1;
            // transposed operand reads run TD blocks ahead of their MFMAs
            constexpr int TD = NI < 3 ? NI : (SPLIT ? 2 : 3);
            s16x4_t ga[NI][2], gl[SPLIT ? NI : 1][2];
            auto grad_read = [&](int db) {
                const int nb = ((db & 1) ? tb1 : tb0) + 2048 * (db >> 1);
                ga[db][0] = tr_read(nb);
                ga[db][1] = tr_read(nb + 1024);
                if constexpr (SPLIT && !QF) {
                    gl[db][0] = tr_read(nb + L::SLICE_B);
                    gl[db][1] = tr_read(nb + L::SLICE_B + 1024);
                }
            };
#pragma unroll
            for (int db = 0; db < TD; ++db) grad_read(db);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < NI; ++db) {
                if (db + TD < NI) grad_read(db + TD);
                __builtin_amdgcn_sched_barrier(0);  // keep the reads TD blocks ahead
                const s16x8_t av = __builtin_shufflevector(ga[db][0], ga[db][1], 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8_t ab2 = __builtin_bit_cast(bf16x8_t, av);
#pragma unroll
                for (int t = 0; t < RT; ++t)
                    acc[db][t] = mfma16<SPLIT>(ab2, Eb[t], acc[db][t]);
                if constexpr (SPLIT && !QF) {
                    const s16x8_t lv = __builtin_shufflevector(gl[db][0], gl[db][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8_t al2 = __builtin_bit_cast(bf16x8_t, lv);
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        acc[db][t] = mfma16<SPLIT>(ab2, El[t], acc[db][t]);
                        acc[db][t] = mfma16<SPLIT>(al2, Eb[t], acc[db][t]);
                    }
                }
                // the MFMAs of block db are issued (so their operand reads have completed) before its refill below;
                // a scheduling fence, not an operand constraint: the accumulators may live in either register half
                __builtin_amdgcn_sched_barrier(0);
                // features [32 s, 32 s + 32) of the step are dead after block 2 s + 1 -> refill both point tiles
                if (db & 1) dma_pair(Xn, db >> 1);
            }
            PHASE_MARK(4);
        }
    }
#ifdef TVAE_PHASE_TIMING
    if (blockIdx.x == 0 && lane == 0)
        printf("wave %d steps %d: gram %llu  xch+barrier %llu  E %llu  barrier2 %llu  grad %llu  top %llu (cycles per step)\n", wave,
               se - sb, ph[0] / (se - sb), ph[1] / (se - sb), ph[2] / (se - sb), ph[3] / (se - sb), ph[4] / (se - sb),
               ph[5] / (se - sb));
#endif
    wait_vmcnt<0>();
    sum0 += (double)f0; sum1 += (double)f1;
#pragma unroll
    for (int j = 0; j < TPW; ++j) rowsum[j] += (double)fr[j];

    // ---- per-workgroup results: each wave holds the sums of ITS tiles; add the four in a fixed order ----
    __syncthreads();
    double* red = reinterpret_cast<double*>(sm + L::XCH_B);          // [4][2] sums, then [4 TPW][16] row sums
    const double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    if (lane == 0) { red[wave * 2 + 0] = w0; red[wave * 2 + 1] = w1; }
    if (GRAD) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            double rv = rowsum[j];
            rv += __shfl_xor(rv, 16, 64);
            rv += __shfl_xor(rv, 32, 64);
            if (g == 0) red[8 + (wave * TPW + j) * 16 + i16] = rv;
        }
    }
    __syncthreads();
    if (tid == 0) {
        a.wg_sums[2 * bid + 0] = (red[0] + red[2]) + (red[4] + red[6]);
        a.wg_sums[2 * bid + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
    if (GRAD) {
        if (tid < ROWS) {    // row 16 t + i: slot (wave * TPW + j) with t = (wave & 1) TPW + j, point tiles wave >> 1 = 0, 1
            const int t = tid >> 4, i = tid & 15;
            const int slot = (t / TPW) * TPW + (t % TPW);          // wave (t / TPW), j = t % TPW
            a.wg_rowsum[(long long)sp * a.n_rows_pad + i0 + tid] = red[8 + slot * 16 + i] + red[8 + (slot + 2 * TPW) * 16 + i];
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 16 * t + i16) * DPAD + wave * SW + 4 * g;
#pragma unroll
            for (int db = 0; db < NI; ++db) *reinterpret_cast<f32x4*>(grow + 16 * db) = acc[GRAD ? db : 0][t];
        }
    }
}

// With -DTVAE_PHASE_TIMING the environment variable TVAE_DEBUG_LDS_PAD (bytes) inflates the dynamic LDS of
// the bf16 kernels to lower the number of resident workgroups for an experiment; otherwise both are zero.
#ifdef TVAE_PHASE_TIMING
constexpr int kDebugLdsPadMax = 64 * 1024;
int debug_lds_pad() {
    static const int pad = getenv("TVAE_DEBUG_LDS_PAD") ? atoi(getenv("TVAE_DEBUG_LDS_PAD")) : 0;
    return pad < 0 ? 0 : (pad > kDebugLdsPadMax ? kDebugLdsPadMax : pad);
}
#else
constexpr int kDebugLdsPadMax = 0;
constexpr int debug_lds_pad() { return 0; }
#endif

template <int KB, int RT, bool GRAD, bool SPLIT>
int launch_bf16(const MmdArgs& a, int nblocks, hipStream_t s) {
    constexpr int lds_bytes = Bf16Lds<KB, RT, GRAD, SPLIT>::BYTES;
    constexpr int lds_attr = lds_bytes + kDebugLdsPadMax > 160 * 1024 ? 160 * 1024 : lds_bytes + kDebugLdsPadMax;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_bf16_kernel<KB, RT, GRAD, SPLIT>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr);
    }));
    const int lds_launch = lds_bytes + debug_lds_pad() > lds_attr ? lds_attr : lds_bytes + debug_lds_pad();
    hipLaunchKernelGGL((mmd_bf16_kernel<KB, RT, GRAD, SPLIT>), dim3(nblocks), dim3(256), lds_launch, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    if constexpr (SPLIT && (GRAD || (RT == 2 && KB >= TVAE_QF32_MIN_KB))) {
        if (a.qflag != nullptr) {          // the pair: exactly one of the two does the work (mmd.hip, QFlag)
            if constexpr (RT == 2 && KB >= TVAE_QF32_MIN_KB && TVAE_GRAM_PRODUCTS == 2) {
                if constexpr (!GRAD) {
                    // sum-only launches: the 128-row kernel without a K-split (mmd_q32s.hip; two step buffers at this width)
                    // under the conditions of mmd_q32.hip's launch_q32_sum -- same symmetric choice for both block sizes,
                    // more than half a chip of workgroups -- and images padded to whole 128-point blocks (mmd.hip, make_plan)
                    const int problems = nblocks / (a.row_blocks * a.col_splits);
                    const int rb2 = (a.n_rows + 127) / 128;
                    if (a.sym2 == a.sym && a.tiles_per_set % 8 == 0 && (long long)rb2 * a.col_splits * problems > 128) {
                        MmdArgs b = a;
                        b.launch_blocks = a.row_blocks * a.col_splits;
                        b.row_blocks = rb2;
                        return launch_mmd_q32_sum3(KB, b, rb2 * a.col_splits * problems, s);
                    }
                }
                // (its own LDS image: no lo copy of the columns, no E_lo piece -- two workgroups fit a CU)
                constexpr int lds_q = Bf16Lds<KB, RT, GRAD, SPLIT, true>::BYTES;
                static PerDeviceOnce once_q;
                TVAE_CHECK_HIP(per_device_once(once_q, [] {
                    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_bf16_kernel<KB, RT, GRAD, SPLIT, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
                }));
                hipLaunchKernelGGL((mmd_bf16_kernel<KB, RT, GRAD, SPLIT, true>), dim3(nblocks), dim3(256), lds_q, s, a);
                TVAE_CHECK_HIP(hipGetLastError());
            } else {
                return fail(TVAE_ERR_ARG, "mmd: no Q-form instance of the split kernel for %d feature blocks, %d row tiles", KB, RT);
            }
        }
    }
    return TVAE_OK;
}

template <bool GRAD, bool SPLIT>
int launch_bf16_shape(int kb, int rt, const MmdArgs& a, int nblocks, hipStream_t s) {
    if constexpr (GRAD && SPLIT) {
        if (rt == 4) {
            switch (kb) {
                case 2: return launch_bf16<2, 4, GRAD, SPLIT>(a, nblocks, s);
                case 3: return launch_bf16<3, 4, GRAD, SPLIT>(a, nblocks, s);
                case 4: return launch_bf16<4, 4, GRAD, SPLIT>(a, nblocks, s);
                case 5: return launch_bf16<5, 4, GRAD, SPLIT>(a, nblocks, s);
                case 6: return launch_bf16<6, 4, GRAD, SPLIT>(a, nblocks, s);
            }
            return fail(TVAE_ERR_ARG, "mmd: no 64-row split kernel for %d feature blocks", kb);
        }
    }
    if (rt != 2) return fail(TVAE_ERR_ARG, "mmd: no bf16 kernel with %d row tiles", rt);
    switch (kb) {
        case 2: return launch_bf16<2, 2, GRAD, SPLIT>(a, nblocks, s);
        case 3: return launch_bf16<3, 2, GRAD, SPLIT>(a, nblocks, s);
        case 4: return launch_bf16<4, 2, GRAD, SPLIT>(a, nblocks, s);
        case 5: return launch_bf16<5, 2, GRAD, SPLIT>(a, nblocks, s);
        case 6: return launch_bf16<6, 2, GRAD, SPLIT>(a, nblocks, s);
        case 7: return launch_bf16<7, 2, GRAD, SPLIT>(a, nblocks, s);
        case 8: return launch_bf16<8, 2, GRAD, SPLIT>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no bf16 kernel for %d feature blocks", kb);
}

}  // namespace

int launch_mmd_bf16(bool grad, bool split, int kb, int rt, const MmdArgs& a, int nblocks, hipStream_t s) {
    if (split) return grad ? launch_bf16_shape<true, true>(kb, rt, a, nblocks, s) : launch_bf16_shape<false, true>(kb, rt, a, nblocks, s);
    return grad ? launch_bf16_shape<true, false>(kb, rt, a, nblocks, s) : launch_bf16_shape<false, false>(kb, rt, a, nblocks, s);
}

}  // namespace tvae
