// Fused MMD forward + gradient for gfx950 (autoencoders.py:117-136 and its autograd).
//
// Math.  With t(a,b) = ||a-b||^2 / D^2, K = exp(-t), E = K - 1 = expm1(-t):
//   MMD       = [ SS E(y,y) + SS E(z,z) - 2 SS E(y,z) ] / N^2          (the +1's cancel exactly)
//   dMMD/dz_k = 4/(N^2 D^2) [ -z_k * R_k + (Zsum - Ysum) + G_k ]
//   R_k = S_p s_p E(z_k,x_p),  G_k = S_p s_p E(z_k,x_p) x_p,  p over all z (s=+1) and all y (s=-1)
// Working with E keeps every accumulated quantity ~|t| << 1, so the catastrophic
// cancellation of the reference's "three means near 1" never happens in fp32.
//
// Kernel structure (mmd_tile_kernel).  One workgroup = 4 waves owns 32 rows i of the kernel
// matrix and walks 16-point column tiles p.  The feature dimension is SLICED over the four
// waves (wave w owns columns [w*SW, (w+1)*SW) of the padded dimension DPAD = 4*SW):
//   1. each wave stages ITS slice of the 16 x DPAD point tile into a wave-private LDS region
//      (no workgroup barrier needed for the tile);
//   2. Gram: S'[p][i] partial over the slice with v_mfma_f32_16x16x4_f32 — A = point tile
//      from LDS (ds_read_b128, 4 k-steps per read), B = the 32 rows held in registers for the
//      whole kernel.  k order inside the dot product is permuted identically for A and B;
//   3. the partial S' tiles are REDUCE-SCATTERED through LDS: every wave sums (fixed order) and finishes
//      only its share of the 16 x 32 tile, so the expm1 work is split over the waves, not replicated;
//   4. E' = expm1(-t) (masked at ragged edges), sums in fp64; the gradient instantiation publishes E'
//      through LDS (second barrier) and every wave reads it back in the MFMA C layout
//      (row p = 4*(lane>>4)+reg, column i = lane&15);
//   5. gradient: acc^T[d][i] += X^T[d][p] * E'[p][i] — the accumulator tile E' is ALREADY in
//      the B-operand layout of the next MFMA (it sums over E's row index p), so it never
//      leaves registers; A = the same LDS tile read transposed (conflict-free ds_read_b32).
// The N x N matrices are never materialised; per-workgroup partial sums / row sums /
// gradient slabs are combined by mmd_finalize_kernel in a fixed order (no atomics).
#include <mutex>

#include "mmd_common.h"

namespace tvae {
namespace {

// LDS plan of mmd_tile_kernel.  A wave's slice of a point tile (16 points x SW floats) is stored
// CHUNK-MAJOR and XOR-swizzled: 16-byte chunk j (4 floats) of point p lives at chunk position
//     j * 16 + (p ^ (j & 3))
// which (a) is one contiguous range, so NB full-width LDS-DMA instructions fill it (instruction i,
// lane l brings chunk j = 4i + (l >> 4) of point p = (l & 15) ^ (l >> 4): the per-lane source offset
// is the same for every instruction; the scalar base advances by 64 B), (b) makes the Gram operand
// read (ds_read_b128, lane (p, g) reads chunk 4m+g of point p) and (c) the transposed gradient
// operand read (ds_read_b32, lane (dc, g) reads feature 16 db + dc of point 4g + r) both bank-conflict
// free.  DMA instruction i carries exactly the four chunk columns that Gram step m = i and gradient
// block db = i consume, which is what makes the single tile buffer below refillable in place.
template <int NB, int WAVES>
struct TileLds {
    static constexpr int SW = 16 * NB;                 // feature columns per wave
    static constexpr int SLICE = TILE_P * SW;          // floats per wave (ONE tile buffer)
    static constexpr int NORM_OFF = WAVES * SLICE;     // [2 buf][WAVES][16] point norms
    static constexpr int XCH_OFF = NORM_OFF + 2 * WAVES * TILE_P;
    static constexpr int XCH_ONE = WAVES * 2 * 64 * 4; // floats of one exchange buffer: [WAVES][2 tiles][64 lanes] float4
    static constexpr int FLOATS = XCH_OFF + 2 * XCH_ONE;
    static constexpr int BYTES = FLOATS * 4;
    static_assert(BYTES <= 160 * 1024, "tile kernel exceeds the 160 KiB LDS");
    // workgroups per CU the LDS admits; the kernel is written to live with two waves per SIMD from
    // DIFFERENT workgroups (independent phases), so the exchange / expm1 / DMA issue of one hide
    // behind the MFMAs of the other
    static constexpr int WG_PER_CU = (160 * 1024) / BYTES;
};

#ifndef TVAE_SYNC_TILES
#define TVAE_SYNC_TILES 1024
#endif
#ifndef TVAE_SYNC_POLLS
#define TVAE_SYNC_POLLS 32
#endif
constexpr int SYNC_TILES = TVAE_SYNC_TILES;     // soft rendezvous period in column tiles (~6 ms at D = 768), a power of two
constexpr int SYNC_POLLS = TVAE_SYNC_POLLS;     // polls (~1-2 us each) before a workgroup gives up waiting

// One workgroup = WAVES waves owns 32 rows i of the kernel matrix and walks 16-point column tiles;
// the feature dimension is sliced over the waves (DPAD = WAVES * SW).
//
// Single tile buffer, refilled in place: as soon as the LAST consumer of chunk-column block i of
// tile k (gradient block db = i; Gram step m = i in the sum-only instantiation) has issued its
// MFMAs, DMA instruction i of tile k+1 is issued into the same LDS bytes.  The Gram of tile k+1
// waits with a COUNTED vmcnt in front of each operand read (DMAs retire in order), so every DMA has
// between one and two phases to land and nothing is ever drained.  Slices and norms are wave-private:
// the only workgroup barrier per tile is the one of the partial-S' exchange.
template <int NB, int WAVES, bool GRAD>
__device__ __forceinline__ void mmd_tile_body(const MmdArgs& a, const int block) {
    using L = TileLds<NB, WAVES>;
    constexpr int SW = L::SW, DPAD = WAVES * SW;
    // reduce-scatter exchange: the S' tile is split over the lanes of the workgroup (EPL elements per lane)
    constexpr bool RS = WAVES > 1;
    constexpr int EPL = RS ? 8 / WAVES : 1;
    static_assert(WAVES == 1 || WAVES == 4 || WAVES == 8, "1, 4 or 8 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // Single-wave workgroups: two waves per SIMD already saturate the matrix cores, and the dispatcher does not
    // balance a third one (HW_ID dump at N = 5 120, D = 77: three waves on 102 SIMDs, one on 230; a wave alone
    // on its SIMD finishes in 100 us, one of two in ~150, one of three in ~200).  Touching v175 makes the
    // register footprint 176 = at most two waves per SIMD (kernel 187 -> 176 us there, neutral elsewhere).
    if constexpr (WAVES == 1) asm volatile("" ::: "v175");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_t = wave & 1, my_r = ((wave >> 1) * EPL) & 3;     // RS: the row group / first accumulator register this wave finishes
    const int i16 = lane & 15, g = lane >> 4;
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const float* xs = smem + wave * L::SLICE;
    const unsigned xs_byte = smem_byte + (unsigned)(wave * L::SLICE) * 4u;

    const int per_problem = a.row_blocks * a.col_splits;
    // problem 0: rows = z against both column sets (+ gradient when GRAD); problem 1: rows = y against y.
    // The GRAD instantiation is only ever launched for problem 0, so the gradient path is branch-free.
    const int bid = a.problem_base * per_problem + block;
    const int problem = GRAD ? 0 : bid / per_problem;
    const int rem = bid - problem * per_problem;
    const int rb = a.sp_major ? rem % a.row_blocks : rem / a.col_splits;
    const int sp = a.sp_major ? rem / a.row_blocks : rem - rb * a.col_splits;
    const int i0 = rb * ROWS_PER_WG;

    // ---- the 32 rows of this workgroup: this wave's feature slice, kept in registers ----
    const float* rows = a.cols[problem] + (long long)a.row_offset * DPAD;
    const float* rnorm = a.cnorm[problem] + a.row_offset;
    f32x4 zreg[2][NB];
    float rn[2];
    bool valid_i[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = i0 + 16 * t + i16;
        valid_i[t] = row < a.n_rows;
        const float* zp = rows + (long long)row * DPAD + wave * SW + 4 * g;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            if (valid_i[t]) zreg[t][m] = *reinterpret_cast<const f32x4*>(zp + 16 * m);
            else zreg[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        rn[t] = valid_i[t] ? rnorm[row] : 0.f;
    }
    const bool any_invalid = __any(!(valid_i[0] && valid_i[1]));
    // Retire these loads BEFORE the loop: otherwise hipcc must assume them pending at the loop header
    // and waits vmcnt(0) at their first use in every iteration, draining the DMA it cannot see.
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int m = 0; m < NB; ++m) asm volatile("" : "+v"(zreg[t][m]));
        asm volatile("" : "+v"(rn[t]));
    }
    const float rn_my = my_t ? rn[1] : rn[0];
    const bool valid_my = my_t ? valid_i[1] : valid_i[0];

    f32x4 acc[GRAD ? NB : 1][2];
    if (GRAD) {
#pragma unroll
        for (int db = 0; db < NB; ++db) {
            acc[db][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[db][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    const long long tps = a.tiles_per_set;
    // Kyy is symmetric: with 32-point column blocks J (two tiles each) and this workgroup's global row
    // block I, the pair {I, J} is evaluated from I's side iff J == I (weight 1) or
    // (I < J) == ((I + J) even) (weight 2).  Every unordered pair is then visited exactly once, and the
    // work per row block is balanced (which matters when ranks own contiguous row ranges).
    const bool sym = !GRAD && problem == 1 && a.sym;
    const int blkI = (a.row_offset >> 5) + rb;
    const int nJ = (int)(tps >> 1);
    const int symA = (blkI + 1) >> 1;                                   // # of J < I with parity != I's
    const int symM = symA + 1 + ((nJ - 1 - blkI) >> 1);                 // + J == I + # of J > I, same parity
    const long long total = problem == 0 ? 2 * tps : (sym ? 2 * (long long)symM : tps);
    const long long base = (problem == 0 || sym) ? 0 : tps;
    const int tb = (int)(base + sp * total / a.col_splits);
    const int te = (int)(base + (sp + 1) * total / a.col_splits);
    // loop index -> tile index in [0, 2 tps)
    auto actual = [&](int vt) -> int {
        if (!sym) return vt;
        const int q = vt >> 1;
        const int J = q < symA ? ((blkI + 1) & 1) + 2 * q : (q == symA ? blkI : blkI + 2 * (q - symA));
        return (int)tps + 2 * J + (vt & 1);
    };

    // per-lane constants of the swizzled image (see TileLds)
    const unsigned dma_voff = (unsigned)((i16 ^ g) * DPAD + 4 * g) * 4u;     // source byte offset of DMA 0
    const int gram_off = 64 * g + 4 * (i16 ^ g);                            // + 256 m   (floats)
    const int c4 = i16 >> 2;                                                 // chunk column of this lane's feature
    int grad_off[4];                                                         // + 256 db  (floats)
#pragma unroll
    for (int r = 0; r < 4; ++r) grad_off[r] = 64 * c4 + 16 * g + 4 * (r ^ c4) + (i16 & 3);

    // wave-uniform source of this wave's slice of point tile `tile` (rows past n_cols are zero in the
    // padded copies, so no bounds checks are needed), and its norms
    auto tile_src = [&](int vt) -> const float* {
        const int tile = actual(vt);
        const int set = tile >= tps ? 1 : 0;
        const int p0 = (tile - set * (int)tps) * TILE_P;
        return a.cols[set] + (long long)p0 * DPAD + wave * SW;
    };
    auto stage_norms = [&](int vt, int nbuf) {
        const int tile = actual(vt);
        const int set = tile >= tps ? 1 : 0;
        const int p0 = (tile - set * (int)tps) * TILE_P;
        // scalar operands are pinned to SGPRs OUTSIDE the lane-masked region (inside it hipcc may
        // rematerialise them in VGPRs, which the asm's "s" constraints cannot take)
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            smem_byte + (unsigned)(L::NORM_OFF + (nbuf * WAVES + wave) * TILE_P) * 4u);
        const float* src = uniform_ptr(a.cnorm[set] + p0);
        if (lane < TILE_P) dma4(src, (unsigned)lane * 4u, dst);
    };

    double sum0 = 0.0, sum1 = 0.0;     // per-lane partial of sum E over set 0 / set 1
    double rowsum0 = 0.0, rowsum1 = 0.0;  // signed row sums R for the two 16-row groups
    float f0 = 0.f, f1 = 0.f, fr0 = 0.f, fr1 = 0.f;   // their fp32 partials, flushed to fp64 every 8 tiles

    // DMA issue order per tile k: norms(k) ... X(k) blocks 0..NB-1 ... ; norms(k+1) goes out at the top of
    // iteration k.  The counted waits below rely on exactly this order.
    if (tb < te) {
        stage_norms(tb, 0);
        const float* X0 = tile_src(tb);
#pragma unroll
        for (int i = 0; i < NB; ++i) dma16(X0 + 16 * i, dma_voff, xs_byte + 1024u * i);
    }
#ifdef TVAE_PHASE_TIMING
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    for (int tile = tb; tile < te; ++tile) {
        PHASE_MARK(5);
        if constexpr (GRAD && WAVES > 1) {
            // SOFT RENDEZVOUS.  Every workgroup streams the same column tiles; while the ~64 workgroups of an XCD
            // stay within ~80 tiles of each other (4 MiB of L2) each tile leaves HBM once per XCD.  Left alone they
            // drift apart (measured L2 -> fabric reads of this kernel: 2x the once-per-XCD minimum at N = 32 768,
            // 21x at N = 131 072, 30x at BASELINE cfg4).  Every SYNC_TILES tiles the workgroups that were dispatched
            // together (`sync_group` consecutive block indices = one resident set) count in and wait for each other
            // -- for at most SYNC_POLLS polls, so a set that is not co-resident (shared GPU, last partial set) costs
            // microseconds, never a hang.  Counters are per (set, epoch): nothing is ever reset or reused.
            const int rel = tile - tb;
            if (a.sync_ctr != nullptr && rel > 0 && (rel & (SYNC_TILES - 1)) == 0 && rel / SYNC_TILES <= a.sync_epochs) {
                if (tid == 0) {
                    // one rendezvous per XCD share of the set (workgroups go to the 8 XCDs round-robin by index;
                    // were that ever not so, the sets would merely be different ones)
                    const int grp = block / a.sync_group, x = block & 7;
                    int left = a.launch_blocks - grp * a.sync_group;
                    left = left < a.sync_group ? left : a.sync_group;
                    const unsigned want = (unsigned)((left - x + 7) >> 3);
                    unsigned* c = a.sync_ctr + ((long long)grp * 8 + x) * a.sync_epochs + (rel / SYNC_TILES - 1);
                    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int poll = 0; poll < SYNC_POLLS; ++poll) {
                        if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                        __builtin_amdgcn_s_sleep(16);
                    }
                }
                __builtin_amdgcn_s_barrier();
            }
        }
        const int atile = actual(tile);
        const int set = atile >= tps ? 1 : 0;
        const int p0 = (atile - set * (int)tps) * TILE_P;
        const float wsym = (sym && (p0 >> 5) != blkI) ? 2.f : 1.f;
        const int buf = (tile - tb) & 1;
        // the last tile prefetches ITSELF again (harmless, once per workgroup): no DMA sits behind a branch
        // and the counted waits below hold on every iteration
        const int ntile = tile + 1 < te ? tile + 1 : tile;
        const float* Xn = tile_src(ntile);
        stage_norms(ntile, buf ^ 1);

        // 1+2. partial Gram over the slice: S'[p][i], p = point (A rows), i = our rows (B columns).
        //      Before the read of block m: all but the (younger) DMAs still allowed in flight have landed.
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        wait_vmcnt<NB>();                       // block 0 of this tile (and its norms) are in LDS
        f32x4 av = *reinterpret_cast<const f32x4*>(xs + gram_off);
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            f32x4 avn = av;
            if (m + 1 < NB) {
                wait_vmcnt_n(GRAD ? NB - (m + 1) : NB - 1);
                avn = *reinterpret_cast<const f32x4*>(xs + gram_off + 256 * (m + 1));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], zreg[0][m][q], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], zreg[1][m][q], s1, 0, 0, 0);
            }
            if (!GRAD) {
                // sum-only: block m is dead once its MFMAs are issued -> refill it with the next tile
                asm volatile("" ::"v"(s0), "v"(s1));
                dma16(Xn + 16 * m, dma_voff, xs_byte + 1024u * m);
            }
            av = avn;
        }

        asm volatile("" ::"v"(s0), "v"(s1));
        PHASE_MARK(0);
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
        if constexpr (RS) {
            // 3'. REDUCE-SCATTER exchange: the 16 x 32 tile of S' is split over the lanes of the workgroup (one
            //     element per lane with 8 waves, two with 4).  Wave w finishes registers r = EPL (w >> 1) ... of
            //     row group t = w & 1 only -- conflict-free scalar reads instead of 16-byte ones from every wave,
            //     and EPL expm1 per lane instead of eight replicated in every wave (VALU work issues through the
            //     same port as the MFMAs of the SIMD's other wave: N = 32768, D = 1024 85.4 -> 77.7 ms,
            //     D = 768 56.7 -> 54.4 ms, D = 256 22.0 -> 19.4 ms).
            //     Partials and E are stored component-major ([.][t][r][lane]).  Gradient instantiation: a second
            //     barrier publishes E, which also protects the single partial buffer; sum-only: buffers alternate.
            float* xp = smem + L::XCH_OFF + (GRAD ? 0 : buf * L::XCH_ONE);
            float* eb = smem + L::XCH_OFF + L::XCH_ONE;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xp[((wave * 2 + 0) * 4 + r) * 64 + lane] = s0[r];
                xp[((wave * 2 + 1) * 4 + r) * 64 + lane] = s1[r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            float sv[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const float* mine = xp + (my_t * 4 + my_r + j) * 64 + lane;
                sv[j] = mine[0];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) sv[j] += mine[w * 512];      // fixed order: deterministic
            }
            PHASE_MARK(1);
            float tv[EPL], tmax = 0.f;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const float cnv = smem[L::NORM_OFF + (buf * WAVES + wave) * TILE_P + 4 * g + my_r + j];
                tv[j] = fmaxf((cnv + rn_my - 2.f * sv[j]) * a.inv_d2, 0.f);
                tmax = fmaxf(tmax, tv[j]);
            }
            EQ v_[EPL];
            if (!__any(tmax >= EM1_TINY)) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) v_[j] = em1_split_tiny(tv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < EPL; ++j) v_[j] = em1_split(tv[j]);
                if (__any(tmax >= EM1_POLY_MAX)) {
#pragma unroll
                    for (int j = 0; j < EPL; ++j)
                        if (tv[j] >= EM1_POLY_MAX) { v_[j].e = expm1f(-tv[j]); v_[j].q = v_[j].e + tv[j]; }
                }
            }
            if (any_invalid && !valid_my) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) { v_[j].e = 0.f; v_[j].q = 0.f; }
            }
            if (p0 + TILE_P > a.n_cols) {
#pragma unroll
                for (int j = 0; j < EPL; ++j)
                    if (p0 + 4 * g + my_r + j >= a.n_cols) { v_[j].e = 0.f; v_[j].q = 0.f; }
            }
            float qs = v_[0].q, es[EPL];
#pragma unroll
            for (int j = 1; j < EPL; ++j) qs += v_[j].q;
            if (set) f1 = fmaf(wsym, qs, f1);
            else f0 += qs;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                es[j] = __uint_as_float(__float_as_uint(v_[j].e) ^ (set ? 0x80000000u : 0u));   // s_p E
                if (GRAD) fr0 += es[j];
            }
            if (((tile - tb) & 7) == 7) {
                sum0 += (double)f0; sum1 += (double)f1; rowsum0 += (double)fr0;
                f0 = f1 = fr0 = 0.f;
            }
            if (GRAD) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) eb[(my_t * 4 + my_r + j) * 64 + lane] = es[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e0[r] = eb[(0 * 4 + r) * 64 + lane];
                    e1[r] = eb[(1 * 4 + r) * 64 + lane];
                }
            }
        } else {
            // WAVES == 1 (narrow latents): one wave holds the whole feature dimension, nothing to exchange
            PHASE_MARK(1);
            // 4. E' = expm1(-t), t = (|x_p|^2 + |z_i|^2 - 2 S') / D^2 clamped at 0; register r <-> p = p0 + 4g + r
            const f32x4 cn = *reinterpret_cast<const f32x4*>(smem + L::NORM_OFF + (buf * WAVES + wave) * TILE_P + 4 * g);
            f32x4 t0, t1, q0, q1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t0[r] = fmaxf((cn[r] + rn[0] - 2.f * s0[r]) * a.inv_d2, 0.f);
                t1[r] = fmaxf((cn[r] + rn[1] - 2.f * s1[r]) * a.inv_d2, 0.f);
            }
            const float tmax = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])),
                                     fmaxf(fmaxf(t1[0], t1[1]), fmaxf(t1[2], t1[3])));
            if (!__any(tmax >= EM1_TINY)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    { const EQ v_ = em1_split_tiny(t0[r]); e0[r] = v_.e; q0[r] = v_.q; }
                    { const EQ v_ = em1_split_tiny(t1[r]); e1[r] = v_.e; q1[r] = v_.q; }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    { const EQ v_ = em1_split(t0[r]); e0[r] = v_.e; q0[r] = v_.q; }
                    { const EQ v_ = em1_split(t1[r]); e1[r] = v_.e; q1[r] = v_.q; }
                }
                if (__any(tmax >= EM1_POLY_MAX)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (t0[r] >= EM1_POLY_MAX) { e0[r] = expm1f(-t0[r]); q0[r] = e0[r] + t0[r]; }
                        if (t1[r] >= EM1_POLY_MAX) { e1[r] = expm1f(-t1[r]); q1[r] = e1[r] + t1[r]; }
                    }
                }
            }
            if (any_invalid) {
                if (!valid_i[0]) { e0 = f32x4{0.f, 0.f, 0.f, 0.f}; q0 = e0; }
                if (!valid_i[1]) { e1 = f32x4{0.f, 0.f, 0.f, 0.f}; q1 = e1; }
            }
            if (p0 + TILE_P > a.n_cols) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p0 + 4 * g + r >= a.n_cols) { e0[r] = 0.f; e1[r] = 0.f; q0[r] = 0.f; q1[r] = 0.f; }
            }
            // sums carry Q (the linear part is added in closed form by the finalize kernel); row sums carry E
            const float qs = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q1[0] + q1[1]) + (q1[2] + q1[3]));
            if (set) f1 = fmaf(wsym, qs, f1);
            else f0 += qs;
            if (GRAD) {
                const float es0 = (e0[0] + e0[1]) + (e0[2] + e0[3]);
                const float es1 = (e1[0] + e1[1]) + (e1[2] + e1[3]);
                fr0 += set ? -es0 : es0;
                fr1 += set ? -es1 : es1;
            }
            if (((tile - tb) & 7) == 7) {
                sum0 += (double)f0; sum1 += (double)f1; rowsum0 += (double)fr0; rowsum1 += (double)fr1;
                f0 = f1 = fr0 = fr1 = 0.f;
            }

        }

        // 5. gradient accumulation: acc^T[d][i] += sum_p X[p][d] * (s_p E'[p][i]); block db of the tile is
        //    dead after its MFMAs -> DMA block db of the NEXT tile over it (operand reads run one block ahead)
        asm volatile("" ::"v"(e0), "v"(e1));
        PHASE_MARK(2);
        if (GRAD) {
            if constexpr (!RS) {      // (the reduce-scatter path published s_p E already)
                const unsigned flip = set ? 0x80000000u : 0u;      // prior columns enter with s_p = -1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e0[r] = __uint_as_float(__float_as_uint(e0[r]) ^ flip);
                    e1[r] = __uint_as_float(__float_as_uint(e1[r]) ^ flip);
                }
            }
            float ga[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[r] = xs[grad_off[r]];
#pragma unroll
            for (int db = 0; db < NB; ++db) {
                float gn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) gn[r] = (db + 1 < NB) ? xs[grad_off[r] + 256 * (db + 1)] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[db][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[r], e0[r], acc[db][0], 0, 0, 0);
                    acc[db][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[r], e1[r], acc[db][1], 0, 0, 0);
                }
                asm volatile("" ::"v"(acc[db][0]), "v"(acc[db][1]));   // MFMAs of block db are issued before its refill
                dma16(Xn + 16 * db, dma_voff, xs_byte + 1024u * db);
#pragma unroll
                for (int r = 0; r < 4; ++r) ga[r] = gn[r];
            }
            PHASE_MARK(4);
        }
    }
#ifdef TVAE_PHASE_TIMING
    if (block == 0 && lane == 0)
        printf("wave %d tiles %d: gram %llu  xch+barrier %llu  E %llu  grad %llu  top %llu (timer ticks per tile)\n", wave,
               te - tb, ph[0] / (te - tb), ph[1] / (te - tb), ph[2] / (te - tb), ph[4] / (te - tb), ph[5] / (te - tb));
#endif
    wait_vmcnt<0>();
    sum0 += (double)f0; sum1 += (double)f1; rowsum0 += (double)fr0; rowsum1 += (double)fr1;

    // ---- per-workgroup results (every wave holds identical sums; wave 0 writes) ----
    double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    double r0v = rowsum0, r1v = rowsum1;
    if (GRAD) {
        r0v += __shfl_xor(r0v, 16, 64); r0v += __shfl_xor(r0v, 32, 64);
        r1v += __shfl_xor(r1v, 16, 64); r1v += __shfl_xor(r1v, 32, 64);
    }
    if constexpr (RS) {
        // reduce-scatter: every wave holds the sums of ITS elements -> combine the waves in a fixed order
        // (the exchange buffers are free once every wave has left the tile loop)
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem + L::XCH_OFF);     // [WAVES][2] sums, then [WAVES][16] row sums
        if (lane == 0) { red[wave * 2 + 0] = w0; red[wave * 2 + 1] = w1; }
        if (GRAD && g == 0) red[2 * WAVES + wave * 16 + i16] = r0v;      // row group my_t = wave & 1, row i16
        __syncthreads();
        w0 = 0.0; w1 = 0.0;
        for (int w = 0; w < WAVES; ++w) { w0 += red[w * 2 + 0]; w1 += red[w * 2 + 1]; }
        if (GRAD) {
            r0v = 0.0; r1v = 0.0;
            for (int w = 0; w < WAVES; w += 2) {
                r0v += red[2 * WAVES + w * 16 + i16];
                r1v += red[2 * WAVES + (w + 1) * 16 + i16];
            }
        }
    }
    if (tid == 0) {
        a.wg_sums[2 * (long long)bid + 0] = w0;
        a.wg_sums[2 * (long long)bid + 1] = w1;
    }
    if (GRAD) {
        if (wave == 0 && g == 0) {
            double* rs = a.wg_rowsum + (long long)sp * a.n_rows_pad + i0;
            rs[i16] = r0v;
            rs[16 + i16] = r1v;
        }
        // acc[db][t][reg] = G[row i0+16t+i16][d = wave*SW + 16 db + 4 g + reg]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 16 * t + i16) * DPAD + wave * SW + 4 * g;
#pragma unroll
            for (int db = 0; db < NB; ++db) *reinterpret_cast<f32x4*>(grow + 16 * db) = acc[GRAD ? db : 0][t];
        }
    }
}

template <int NB, int WAVES, bool GRAD>
__global__ __launch_bounds__(64 * WAVES, (NB * WAVES <= 48 ? 2 : 1)) void mmd_tile_kernel(MmdArgs a) {
    mmd_tile_body<NB, WAVES, GRAD>(a, (int)blockIdx.x);
}

// Single-wave shapes: the gradient problem (z rows against z and y) and the sum-only problem (y rows against y)
// in ONE launch -- the first nwg0 workgroups take problem 0.  The y-y workgroups are short (symmetric
// enumeration, no gradient) and fill the tail of the gradient launch instead of a second, poorly filled one
// (BASELINE cfg2: 12 + 6 us -> one launch; cfg3: the 30 us second launch disappears into the first one's tail).
template <int NB>
__global__ __launch_bounds__(64, 2) void mmd_tile_both_kernel(MmdArgs a, int nwg0) {
    if ((int)blockIdx.x < nwg0) {
        mmd_tile_body<NB, 1, true>(a, (int)blockIdx.x);
    } else {
        MmdArgs b = a;
        b.problem_base = 1;
        mmd_tile_body<NB, 1, false>(b, (int)blockIdx.x - nwg0);
    }
}

// ---- prep: zero-padded copies (row stride DPAD; fp32, or bf16 for the bf16 kernels), squared norms,
//      per-wave column sums ----
// SPLIT (TC = fp16, T = float): dst holds hi = fp16(s v), and dst + lo_off holds lo = fp16(s v - hi) (split_scales)
template <typename TC, bool SPLIT>
__device__ __forceinline__ void prep_store(TC* dst, long long idx, long long lo_off, float v, float scale) {
    if constexpr (SPLIT) {
        const float sv = v * scale;
        const TC hi = from_f32<TC>(sv);
        dst[idx] = hi;
        dst[idx + lo_off] = from_f32<TC>(sv - to_f32(hi));
    } else {
        dst[idx] = from_f32<TC>(v);
    }
}

// ---- The call in three stages (split modes; the other modes have no stage A) --------------------------------------
//   A  stats : column sums and max |value| of the SOURCE rows of this call                      -> rec_a
//   B  prep  : images of the source rows CENTRED on the pooled mean mu = (sum z + sum y) / 2N, squared norms,
//              column sums / third moments / norm sums of the centred rows (-> rec_b), Z'^T Z' and Y'^T Y' of the
//              source rows (Q-form gradient)
//   C  tiles : closed-form pieces from the COMPLETE records, tile kernels, finalize
// One process: the source rows are all rows and the stages run back to back (tvae_mmd_fwd_bwd).  Row-sharded over R
// ranks (tvae_mmd_stats_local / _prep_local / _tiles): the source rows are the rank's own n / R rows, rec_a is summed
// over ranks between A and B, and between B and C the image rows and norms are all-gathered (fp16: half the bytes of
// the fp32 latents) and rec_b and the two D x D matrices summed -- no rank touches another rank's rows before C, so
// the O(N D) prep and the O(N D^2) closed-form products shrink with R like the tile work does.
// Centring: MMD and its gradient are invariant under a common translation of both sets, and the split arithmetic is
// not -- a common offset eats the fp16 pairs' 22 bits (both+5: gradient 1e-4 wrong, tests/test_gpu_r3.py).  The
// images, norms and every closed-form moment are therefore those of x - mu; only the explicit mean-shift term of the
// gradient, sum z - sum y, is taken from the uncentred column sums (it is translation invariant and exact there).
//
// rec_a (doubles): [0, DPAD) sum z, [DPAD, 2 DPAD) sum y, [2 DPAD] max |value|, [2 DPAD + 1] 0
// rec_b (doubles): [0, DPAD) sum z', [DPAD ..) sum y', [2 DPAD ..) sum |z'|^2 z', [3 DPAD ..) sum |y'|^2 y',
//                  [4 DPAD] sum |z'|^2, [+1] sum |y'|^2, [+2] max squared norm, [+3] 0
__host__ __device__ constexpr int rec_a_doubles(int DPAD) { return 2 * DPAD + 2; }
__host__ __device__ constexpr int rec_b_doubles(int DPAD) { return 4 * DPAD + 4; }

struct QFlag { unsigned on; unsigned pad; double az, ay; };      // az, ay: sums of squared norms over all points

__device__ __forceinline__ float wave_maxf(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
}

// stage A, first half: per-WORKGROUP column sums and maxima of 4 x rows_per_wave source rows (plain stores, no atomics;
// the four waves' sums meet in LDS, so the second half reads a quarter of the partials).
// fp64 from the first add: the mean-shift term of the gradient, sum z - sum y, comes from these sums, and under a common
// offset (both+50, tests/test_gpu_r3.py) fp32 partial sums of ~64 rows already cost 1.5e-5 of the gradient
__global__ __launch_bounds__(256) void mmd_stats_kernel(const float* __restrict__ z, const float* __restrict__ y, int n_src, int D,
                                                        int DPAD, int rows_per_wave, int nwaves, double* __restrict__ colpart,
                                                        float* __restrict__ amaxpart) {
    __shared__ double red[3][1024];
    __shared__ float redm[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wg = blockIdx.x * 4 + wave;
    const int set = blockIdx.y;
    const float* __restrict__ src = set ? y : z;
    const int NJ = (DPAD + 63) / 64;
    double csum[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) csum[j] = 0.0;
    float m = 0.f;
    const int r0 = wg * rows_per_wave, r1 = wg < nwaves ? min(n_src, r0 + rows_per_wave) : r0;
    if (NJ <= 4) {
        // narrow rows: four rows at a time, every load issued before the first use (see mmd_prep_kernel)
        for (int row = r0; row < r1; row += 4) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = lane + 64 * j;
                    const bool ok = j < NJ && c < D && row + u < r1;
                    const float t = src[ok ? (long long)(row + u) * D + c : 0ll];
                    v[u][j] = ok ? t : 0.f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) { csum[j] += (double)v[u][j]; m = fmaxf(m, fabsf(v[u][j])); }
        }
    } else {
        for (int row = r0; row < r1; ++row) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = lane + 64 * j;
                if (j < NJ && c < D) {
                    const float v = src[(long long)row * D + c];
                    csum[j] += (double)v;
                    m = fmaxf(m, fabsf(v));
                }
            }
        }
    }
    m = wave_maxf(m);
    if (lane == 0) redm[wave] = m;
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < NJ) red[wave - 1][lane + 64 * j] = csum[j];
    }
    __syncthreads();
    if (wave == 0) {
        double* cp = colpart + ((long long)set * gridDim.x + blockIdx.x) * DPAD;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < NJ && lane + 64 * j < DPAD)
                cp[lane + 64 * j] = ((csum[j] + red[0][lane + 64 * j]) + red[1][lane + 64 * j]) + red[2][lane + 64 * j];
        if (lane == 0) amaxpart[set * gridDim.x + blockIdx.x] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    }
}

// stage A, second half: fp64 sums of the partials in a fixed order, one workgroup per 64 columns; workgroup 0 also
// reduces the maxima
__global__ __launch_bounds__(1024) void mmd_stats_final_kernel(const double* __restrict__ colpart, const float* __restrict__ amaxpart,
                                                               int nwaves, int DPAD, double* __restrict__ rec_a) {
    __shared__ double red[2][16][64];
    __shared__ float redm[16];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double cz = 0.0, cy = 0.0;
    if (c < DPAD) {
#pragma unroll 8
        for (int w = ry; w < nwaves; w += 16) {
            cz += colpart[(long long)w * DPAD + c];
            cy += colpart[((long long)nwaves + w) * DPAD + c];
        }
    }
    red[0][ry][cx] = cz; red[1][ry][cx] = cy;
    if (blockIdx.x == 0) {
        float m = 0.f;
        for (int i = threadIdx.x; i < 2 * nwaves; i += 1024) m = fmaxf(m, amaxpart[i]);
        m = wave_maxf(m);
        if (cx == 0) redm[ry] = m;
    }
    __syncthreads();
    if (ry == 0 && c < DPAD) {
        double tz = 0.0, ty = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { tz += red[0][k][cx]; ty += red[1][k][cx]; }
        rec_a[c] = tz; rec_a[DPAD + c] = ty;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float m = 0.f;
        for (int k = 0; k < 16; ++k) m = fmaxf(m, redm[k]);
        rec_a[2 * DPAD] = (double)m; rec_a[2 * DPAD + 1] = 0.0;
    }
}

// stage B: zero-padded images (row stride DPAD; fp32, bf16, or the fp16 pairs of split mode), squared norms, per-wave column
// sums (and third moments).  The source arrays hold rows [src_row0, ...) of the problem; rows are cut into
// [0, lo) | [lo, hi) = the rows this call owns | [hi2, n_pad) and each segment into chunks of rows_per_wave, so that
// waves nw0 .. nw0 + nw1 - 1 hold exactly the owned rows.  One process: src_row0 = 0, hi2 = hi.  A rank of a sharded
// call: src_row0 = lo, nw0 = 0, hi2 = n_total -- its own rows and the zero rows behind the last point.
// rec_a non-null (split mode): centre on the pooled mean and take the fp16 scale from max |value| + max |mu|, which
// bounds every centred magnitude; wave 0 publishes mu and the scale word for the later kernels.
template <typename T, typename TC, bool SPLIT>
__global__ __launch_bounds__(256) void mmd_prep_kernel(const T* __restrict__ z, const T* __restrict__ y, int n_total,
                                                       int n_pad, int D, int DPAD, TC* __restrict__ zp,
                                                       TC* __restrict__ yp, float* __restrict__ zn,
                                                       float* __restrict__ yn, float* __restrict__ colpart,
                                                       int rows_per_wave, int nwaves, int lo, int hi, int hi2, int nw0, int nw1,
                                                       int src_row0, long long lo_off, const double* __restrict__ rec_a,
                                                       double inv_2n, float* __restrict__ mu_out, unsigned* __restrict__ amax_out,
                                                       float* __restrict__ colpart3) {
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wg >= nwaves) return;
    const int set = blockIdx.y;
    const T* __restrict__ src = set ? y : z;
    TC* __restrict__ dst = set ? yp : zp;
    float* __restrict__ nrm = set ? yn : zn;
    const int NJ = (DPAD + 63) / 64;      // 64-column groups per row (<= 16)
    float csum[16], csum3[16], mu[16];      // csum3 (Q-form gradient only): partial sums of |row|^2 row
#pragma unroll
    for (int j = 0; j < 16; ++j) { csum[j] = 0.f; csum3[j] = 0.f; mu[j] = 0.f; }
    float scale = 1.f;
    if (rec_a) {
        float mm = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = lane + 64 * j;
            if (j < NJ && c < D) {
                mu[j] = (float)((rec_a[c] + rec_a[DPAD + c]) * inv_2n);
                mm = fmaxf(mm, fabsf(mu[j]));
            }
        }
        mm = wave_maxf(mm);
        const unsigned bits = __float_as_uint((float)rec_a[2 * DPAD] + mm);
        if (SPLIT) scale = split_scales(bits).s;
        if (wg == 0 && set == 0) {
            if (lane == 0) *amax_out = bits;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < NJ && lane + 64 * j < DPAD) mu_out[lane + 64 * j] = mu[j];
        }
    }
    int r0, r1;
    if (wg < nw0) { r0 = wg * rows_per_wave; r1 = min(lo, r0 + rows_per_wave); }
    else if (wg < nw0 + nw1) { r0 = lo + (wg - nw0) * rows_per_wave; r1 = min(hi, r0 + rows_per_wave); }
    else { r0 = hi2 + (wg - nw0 - nw1) * rows_per_wave; r1 = min(n_pad, r0 + rows_per_wave); }
    src -= (long long)src_row0 * D;       // indexed by the row of the problem from here on (never dereferenced below src_row0)
    if (NJ <= 4) {
        // narrow rows (DPAD <= 256) carry one to four loads per lane: four rows at a time, every load issued before
        // the first use (out-of-range elements read the first source element and are zeroed by a select), so a wave pays
        // one memory latency per four rows instead of one per row (BASELINE cfg3, ten rows per wave: 15 -> 7 us)
        for (int row = r0; row < r1; row += 4) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = row + u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = lane + 64 * j;
                    const bool ok = j < NJ && c < D && rr < r1 && rr < n_total;
                    const float t = to_f32(src[ok ? (long long)rr * D + c : (long long)src_row0 * D]);
                    v[u][j] = ok ? t - mu[j] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = row + u;
                if (rr < r1) {
                    float sq = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = lane + 64 * j;
                        if (j < NJ && c < DPAD) {
                            prep_store<TC, SPLIT>(dst, (long long)rr * DPAD + c, lo_off, v[u][j], scale);
                            csum[j] += v[u][j];
                            sq = fmaf(v[u][j], v[u][j], sq);
                        }
                    }
                    sq = wave_sum(sq);
                    if (lane == 0) nrm[rr] = sq;
                    if (colpart3) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) csum3[j] = fmaf(sq, v[u][j], csum3[j]);
                    }
                }
            }
        }
    } else {
        for (int row = r0; row < r1; ++row) {
            float sq = 0.f, vr[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = lane + 64 * j;
                vr[j] = 0.f;
                if (j < NJ && c < DPAD) {
                    const float v = (row < n_total && c < D) ? to_f32(src[(long long)row * D + c]) - mu[j] : 0.f;
                    prep_store<TC, SPLIT>(dst, (long long)row * DPAD + c, lo_off, v, scale);    // TC == bf16 without SPLIT only with T == bf16: exact
                    csum[j] += v;
                    sq = fmaf(v, v, sq);
                    vr[j] = v;
                }
            }
            sq = wave_sum(sq);
            if (lane == 0) nrm[row] = sq;
            if (colpart3) {
#pragma unroll
                for (int j = 0; j < 16; ++j) csum3[j] = fmaf(sq, vr[j], csum3[j]);
            }
        }
    }
    float* cp = colpart + ((long long)set * nwaves + wg) * DPAD;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < NJ && lane + 64 * j < DPAD) cp[lane + 64 * j] = csum[j];
    if (colpart3) {
        float* cp3 = colpart3 + ((long long)set * nwaves + wg) * DPAD;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < NJ && lane + 64 * j < DPAD) cp3[lane + 64 * j] = csum3[j];
    }
}

// stage B, the record: fp64 sums of the per-wave partials in a fixed order -- over every wave (rec_all) and over the
// waves of the owned rows (rec_loc).  Workgroups 0 .. ncb - 1 take 64 columns each; workgroup ncb sums the squared
// norms of the source rows [g0, g1) and of the owned rows [lo, hi) and finds the largest.
// One process (`shift` non-null): every record is complete here, so the kernel also emits what mmd_closed_kernel would --
// shift, m3, the dot products (one partial triple per column workgroup: dots[3 ncb]) and the Q-form flag -- and that
// launch is skipped.
__global__ __launch_bounds__(1024) void mmd_recb_kernel(const float* __restrict__ colpart, const float* __restrict__ colpart3,
                                                        int nwaves, int nw0, int nw1, int DPAD, const float* __restrict__ zn,
                                                        const float* __restrict__ yn, int g0, int g1, int lo, int hi, int ncb,
                                                        double* __restrict__ rec_all, double* __restrict__ rec_loc,
                                                        const double* __restrict__ rec_a, int D, float* __restrict__ shift,
                                                        float* __restrict__ m3, double* __restrict__ dots, QFlag* __restrict__ qflag) {
    __shared__ double red[4][16][64];
    __shared__ double prod[3][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    if ((int)blockIdx.x == ncb) {
        double az = 0.0, ay = 0.0, azl = 0.0, ayl = 0.0;
        float mx = 0.f, mxl = 0.f;
        for (int i = g0 + (int)threadIdx.x; i < g1; i += 1024) {
            const float a = zn[i], b = yn[i];
            const bool own = i >= lo && i < hi;
            az += (double)a; ay += (double)b;
            mx = fmaxf(mx, fmaxf(a, b));
            if (own) { azl += (double)a; ayl += (double)b; mxl = fmaxf(mxl, fmaxf(a, b)); }
        }
        az = wave_sum(az); ay = wave_sum(ay); azl = wave_sum(azl); ayl = wave_sum(ayl);
        mx = wave_maxf(mx); mxl = wave_maxf(mxl);
        if (cx == 0) {
            red[0][ry][0] = az; red[0][ry][1] = ay; red[0][ry][2] = azl; red[0][ry][3] = ayl;
            red[0][ry][4] = (double)mx; red[0][ry][5] = (double)mxl;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t[4] = {0.0, 0.0, 0.0, 0.0}, m = 0.0, ml = 0.0;
            for (int k = 0; k < 16; ++k) {
                for (int q = 0; q < 4; ++q) t[q] += red[0][k][q];
                m = fmax(m, red[0][k][4]); ml = fmax(ml, red[0][k][5]);
            }
            double* ra = rec_all + 4 * DPAD;
            double* rl = rec_loc + 4 * DPAD;
            ra[0] = t[0]; ra[1] = t[1]; ra[2] = m; ra[3] = 0.0;
            rl[0] = t[2]; rl[1] = t[3]; rl[2] = ml; rl[3] = 0.0;
            if (shift && qflag) {      // as mmd_closed_kernel
                const float bound = 4.f * (float)m / ((float)D * (float)D);
                qflag->on = (bound < 0.5f * EM1_POLY_MAX) ? 1u : 0u;
                qflag->pad = 0u;
                qflag->az = t[0]; qflag->ay = t[1];
            }
        }
        return;
    }
    const int c = blockIdx.x * 64 + cx;
    for (int pass = 0; pass < 2; ++pass) {
        const float* __restrict__ part = pass ? colpart3 : colpart;
        double sz = 0.0, sy = 0.0, szl = 0.0, syl = 0.0;
        if (part && c < DPAD) {
#pragma unroll 8
            for (int w = ry; w < nwaves; w += 16) {      // eight independent loads in flight per lane
                const double vz = (double)part[(long long)w * DPAD + c];
                const double vy = (double)part[((long long)nwaves + w) * DPAD + c];
                const bool own = w >= nw0 && w < nw0 + nw1;
                sz += vz; sy += vy;
                szl += own ? vz : 0.0; syl += own ? vy : 0.0;
            }
        }
        __syncthreads();
        red[0][ry][cx] = sz; red[1][ry][cx] = sy; red[2][ry][cx] = szl; red[3][ry][cx] = syl;
        __syncthreads();
        if (ry == 0 && c < DPAD) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 16; ++k) { t[0] += red[0][k][cx]; t[1] += red[1][k][cx]; t[2] += red[2][k][cx]; t[3] += red[3][k][cx]; }
            rec_all[(2 * pass) * DPAD + c] = t[0]; rec_all[(2 * pass + 1) * DPAD + c] = t[1];
            rec_loc[(2 * pass) * DPAD + c] = t[2]; rec_loc[(2 * pass + 1) * DPAD + c] = t[3];
            if (shift && pass == 0) shift[c] = rec_a ? (float)(rec_a[c] - rec_a[DPAD + c]) : (float)(t[0] - t[1]);
            if (shift && pass == 1 && m3 && part) m3[c] = (float)(t[0] - t[1]);
            if (pass == 0) { prod[0][cx] = t[2] * t[0]; prod[1][cx] = t[2] * t[1]; prod[2][cx] = t[3] * t[1]; }
        } else if (ry == 0 && pass == 0) {
            prod[0][cx] = 0.0; prod[1][cx] = 0.0; prod[2][cx] = 0.0;
        }
        if (shift && pass == 0) {
            __syncthreads();
            if (threadIdx.x < 3) {
                double d = 0.0;
                for (int k = 0; k < 64; ++k) d += prod[threadIdx.x][k];
                dots[blockIdx.x * 3 + threadIdx.x] = d;
            }
        }
    }
}

// ---- Q-form gradient (split pipelined path): E = -t + Q(t), the part linear in t in closed form ----------------------
//   sum_p s_p E_kp x_p = -(1/D^2) [ m3 + |z_k|^2 shift - 2 M2 z_k ] + sum_p s_p Q_kp x_p
//   sum_p s_p E_kp     = -(1/D^2) [ A_z - A_y - 2 shift . z_k ]     + sum_p s_p Q_kp          (sum_p s_p = 0)
// with m3 = sum_z |z|^2 z - sum_y |y|^2 y, shift = sum z - sum y, M2 = Z^T Z - Y^T Y, A = sums of squared norms.
// For the shapes that matter t ~ 2 / D, so Q ~ t^2 / 2 is ~1e-3 of E and ONE fp16 product carries the tile kernel's share of
// the gradient at the accuracy of the fp32 kernel.  Valid while every t of the call is below the
// polynomial range: the closed-form kernel bounds t by 4 max |.|^2 / D^2 on the device, and tile kernel and finalize read it.

// M2 = Z^T Z - Y^T Y (in place over the first) with shift appended as column D: row r of the D x (D + 1) matrix the
// row product below multiplies by, so that column D of its result is shift . z_k
__global__ __launch_bounds__(256) void mmd_m2_fix_kernel(float* __restrict__ m2z, const float* __restrict__ m2y, const float* __restrict__ shift,
                                                          int D, int ld) {
    const long long total = (long long)D * (D + 1);
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / (D + 1)), c = (int)(idx - (long long)r * (D + 1));
        const long long at = (long long)r * ld + c;
        m2z[at] = c < D ? m2z[at] - m2y[at] : shift[r];
    }
}

// stage C, first kernel (one workgroup, a thread per column): everything the tile kernels and finalize need from the
// COMPLETE records --
//   shift[c] = sum z - sum y      (mean-shift term of the gradient; from rec_a, the uncentred sums, when there is one)
//   m3[c]                         (Q-form gradient)
//   dots[3] = <c_z^loc, c_z>, <c_z^loc, c_y>, <c_y^loc, c_y>,   c = column sums of the centred points
//             (the closed-form linear part of the three sums, see em1_split in mmd_common.h)
//   the Q-form flag and the norm sums
__global__ __launch_bounds__(1024) void mmd_closed_kernel(const double* __restrict__ rec_all, const double* __restrict__ rec_loc,
                                                          const double* __restrict__ rec_a, int D, int DPAD, float* __restrict__ shift,
                                                          float* __restrict__ m3, double* __restrict__ dots, QFlag* __restrict__ qflag) {
    __shared__ double red[3][16];
    const int c = threadIdx.x;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;
    if (c < DPAD) {
        const double cz = rec_all[c], cy = rec_all[DPAD + c], czl = rec_loc[c], cyl = rec_loc[DPAD + c];
        p0 = czl * cz; p1 = czl * cy; p2 = cyl * cy;
        shift[c] = rec_a ? (float)(rec_a[c] - rec_a[DPAD + c]) : (float)(cz - cy);
        if (m3) m3[c] = (float)(rec_all[2 * DPAD + c] - rec_all[3 * DPAD + c]);
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = p0; red[1][threadIdx.x >> 6] = p1; red[2][threadIdx.x >> 6] = p2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double d = 0.0;
        for (int k = 0; k < 16; ++k) d += red[threadIdx.x][k];
        dots[threadIdx.x] = d;
    }
    if (threadIdx.x == 0 && qflag) {
        // t <= (|x| + |z|)^2 / D^2 <= 4 max|.|^2 / D^2; half the polynomial range leaves room for rounding (NaN: off)
        const float bound = 4.f * (float)rec_all[4 * DPAD + 2] / ((float)D * (float)D);
        qflag->on = (bound < 0.5f * EM1_POLY_MAX) ? 1u : 0u;
        qflag->pad = 0u;
        qflag->az = rec_all[4 * DPAD]; qflag->ay = rec_all[4 * DPAD + 1];
    }
}

struct FinalArgs {
    const double* wg_sums; int nwg0, nwg1;   // nwg0 problem-0 workgroups come first, then nwg1 problem-1 ones
    const double* wg_rowsum; const float* wg_grad; const float* shift; const void* z_src;   // z_src: the caller's owned z rows (n_rows, D)
    const float* mu;                         // pooled mean the images were centred on (null: not centred)
    int col_splits, n_rows, n_rows_pad, row_offset, D, DPAD;
    double* sums3; float* loss; double loss_scale; double grad_scale;
    void* d_z;
    const unsigned* split_amax;   // split kernels: the gradient slabs carry 2^12 s G (split_scales); else null
    // closed-form linear part: squared norms of all points, the records, the column-moment dot products
    const float* zn; int n_total; const double* dots; int ndots; const double* rec_all; const double* rec_loc;
    // Q-form gradient (null: off): the flag, m3, and P = Z_rows [M2 | shift] (row stride ldP, column D = shift . z_k)
    const QFlag* qflag; const float* m3; const float* P; int ldP;
};

template <typename T>
__global__ __launch_bounds__(256) void mmd_finalize_kernel(FinalArgs f) {
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        double zz = 0.0, zy = 0.0, yy = 0.0;
#pragma unroll 4
        for (int w = tid; w < f.nwg0; w += 256) {
            zz += f.wg_sums[2 * (long long)w + 0];
            zy += f.wg_sums[2 * (long long)w + 1];
        }
#pragma unroll 4
        for (int w = tid; w < f.nwg1; w += 256) yy += f.wg_sums[2 * ((long long)f.nwg0 + w) + 1];
        red[0][tid] = zz; red[1][tid] = zy; red[2][tid] = yy;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                red[0][tid] += red[0][tid + o];
                red[1][tid] += red[1][tid + o];
                red[2][tid] += red[2][tid + o];
            }
            __syncthreads();
        }
        if (tid == 0) {
            const double qzz = red[0][0], qzy = red[1][0], qyy = red[2][0];     // sums of Q = E + t (second order)
            // linear part in closed form:  sum_{i owned, j} |a_i - b_j|^2 = N A_loc + n_loc A_all - 2 <c_loc, c_all>
            const double az = f.rec_all[4 * f.DPAD], ay = f.rec_all[4 * f.DPAD + 1];
            const double azl = f.rec_loc[4 * f.DPAD], ayl = f.rec_loc[4 * f.DPAD + 1];
            const double N = (double)f.n_total, nl = (double)f.n_rows, inv = 1.0 / ((double)f.D * (double)f.D);
            double dzz = 0.0, dzy = 0.0, dyy = 0.0;
            for (int k = 0; k < f.ndots; ++k) { dzz += f.dots[3 * k]; dzy += f.dots[3 * k + 1]; dyy += f.dots[3 * k + 2]; }
            const double szz = qzz - (N * azl + nl * az - 2.0 * dzz) * inv;
            const double szy = qzy - (N * azl + nl * ay - 2.0 * dzy) * inv;
            const double syy = qyy - (N * ayl + nl * ay - 2.0 * dyy) * inv;
            f.sums3[0] = szz; f.sums3[1] = szy; f.sums3[2] = syy;
            if (f.loss) *f.loss = (float)(f.loss_scale * (syy + szz - 2.0 * szy));
        }
    }
    if (!f.d_z) return;
    T* out = static_cast<T*>(f.d_z);
    const long long total = (long long)f.n_rows * f.D;
    const bool qf = f.qflag != nullptr && f.qflag->on != 0;      // the slabs carry the Q sums: add the closed-form linear part
    const double inv_d2 = 1.0 / ((double)f.D * (double)f.D);
    const double a_s = qf ? f.qflag->az - f.qflag->ay : 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + tid; idx < total; idx += (long long)gridDim.x * 256) {
        const int row = (int)(idx / f.D), c = (int)(idx - (long long)row * f.D);
        double rsum = 0.0;
        float gsum = 0.f;
#pragma unroll 8
        for (int s = 0; s < f.col_splits; ++s) {
            rsum += f.wg_rowsum[(long long)s * f.n_rows_pad + row];
            gsum += f.wg_grad[((long long)s * f.n_rows_pad + row) * f.DPAD + c];
        }
        if (f.split_amax) gsum *= split_scales(*f.split_amax).grad_unscale;       // a power of two: exact
        float zc = to_f32(static_cast<const T*>(f.z_src)[(long long)row * f.D + c]);
        if (f.mu) zc -= f.mu[c];      // the same fp32 difference the images were made from
        const double zv = (double)zc;
        double glin = 0.0;
        if (qf) {
            const float* pr = f.P + (long long)row * f.ldP;
            rsum -= inv_d2 * (a_s - 2.0 * (double)pr[f.D]);
            glin = -inv_d2 * ((double)f.m3[c] + (double)f.zn[f.row_offset + row] * (double)f.shift[c] - 2.0 * (double)pr[c]);
        }
        const double v = f.grad_scale * (-zv * rsum + (double)f.shift[c] + (double)gsum + glin);
        out[idx] = from_f32<T>((float)v);
    }
}

// ---- host side ------------------------------------------------------------------------
struct Plan {
    int bf16_tiles;   // 1: bf16 copies + mmd_bf16_kernel<KB = DPAD / 128> (bf16 inputs, or split fp32 inputs, D > 192)
    int split;        // 1 (with bf16_tiles): fp32 inputs as hi + lo fp16 copies, two (Gram) or three (gradient) products per fp32 product
    int pipe;         // 1 (with bf16_tiles): the gradient launch runs mmd_pipe_kernel (64 rows, software-pipelined)
    int pipe_sum;     // 1 (with bf16_tiles): the sum-only launches run mmd_pipe_sum_kernel (64-row blocks too)
    int RT, row_blocks_g;   // gradient launch: 16-row tiles per workgroup and its row blocks (32-row blocks otherwise)
    int narrow64;           // 1: mmd_narrow_kernel (padded dimension <= 80): 64 rows per single-wave workgroup
    int n16;                // 1 (with split, without bf16_tiles): mmd_n16_kernel, narrow latents on the fp16 matrix cores
    int qform;              // 1 (with pipe and split): Q-form gradient, linear part in closed form (see QFlag)
    int qsum;               // 1 (split pipelined sum-only launches): the device flag also picks the one-product sum kernel (mmd_q32.hip)
    int ldq;                // row stride of M2 and P (D + 1 columns, padded)
    int NB, WAVES, DPAD, n_pad, row_blocks, n_rows_pad, col_splits, tiles_per_set, nwaves, rows_per_wave;
    int sp_major;     // 1: split-major workgroup order (see make_plan)
    int sync_epochs, sync_group, sync_groups;   // soft rendezvous of the gradient launch (sync_epochs == 0: off)
    int64_t off_zp, off_yp, off_zn, off_yn, off_colpart, off_shift, off_dots, off_sums, off_rowsum, off_grad, off_sync, off_amax, off_col3, off_m3, off_m2, off_p, off_gslab, off_qflag, off_reca, off_recb, off_recb_loc, off_mu, bytes;
};

// (padded dimension, waves per workgroup, 16-column blocks per wave); DPAD = WAVES * 16 * NB
struct Shape { int dpad, waves, nb; };
// Narrow latents (<= 192) on the fp32 kernels: one wave per workgroup owns all features of its 32 rows (no exchange, no
// barrier); wider ones slice the features over 4 (8) waves.
const Shape kShapes[] = {{16, 1, 1}, {32, 1, 2}, {48, 1, 3}, {64, 1, 4}, {80, 1, 5}, {96, 1, 6}, {112, 1, 7},
                         {128, 1, 8}, {160, 1, 10}, {192, 1, 12}, {256, 4, 4}, {384, 4, 6},
                         {512, 4, 8}, {768, 4, 12}, {1024, 8, 8}};
#ifndef TVAE_PIPE_MAX_KB
#define TVAE_PIPE_MAX_KB 6
#endif
#ifndef TVAE_PIPE_SUM_BF16
#define TVAE_PIPE_SUM_BF16 1
#endif
constexpr bool kPipeSumBf16 = TVAE_PIPE_SUM_BF16 != 0;   // bf16 mode: sum-only launches on the pipelined kernel too
#ifndef TVAE_QFORM
#define TVAE_QFORM 1
#endif
constexpr bool kQForm = TVAE_QFORM != 0 && TVAE_GRAM_PRODUCTS == 2;        // Q-form gradient on the split pipelined path (0: E-form, three gradient products)
#ifndef TVAE_QF32_MIN_KB
#define TVAE_QF32_MIN_KB 7     // mmd_bf16.hip
#endif
constexpr int kPipeMaxKb = TVAE_PIPE_MAX_KB;    // pipelined gradient kernel up to this many feature blocks per wave (0: off)
const Shape* pick_shape(int64_t D) {
    for (const Shape& sh : kShapes)
        if (sh.dpad >= D) return &sh;
    return nullptr;
}

// Kernel family of a call.  Wide latents (192 < D <= 1024): bf16 inputs run on the bf16 matrix cores; fp32 inputs run
// there too, as scaled hi + lo fp16 pairs with two or three products per fp32 product (MODE_SPLIT, same 1e-5 accuracy class as the
// fp32-input MFMA at 16x its rate), unless the caller asks for TVAE_F32_STRICT.  Narrow latents (D <= 192): fp32
// inputs run the same split arithmetic on the single-wave kernel of mmd_n16.hip; bf16 inputs and TVAE_F32_STRICT stay
// on the single-wave fp32 kernels (bf16 inputs are widened exactly).
enum { MODE_F32 = 0, MODE_BF16 = 1, MODE_SPLIT = 2 };
#ifndef TVAE_N16
#define TVAE_N16 1
#endif
int mmd_mode(int dtype, int64_t D) {
    if (D > 1024) return MODE_F32;
    if (D <= 192) return (TVAE_N16 && dtype == TVAE_F32) ? MODE_SPLIT : MODE_F32;
    return dtype == TVAE_BF16 ? MODE_BF16 : (dtype == TVAE_F32 ? MODE_SPLIT : MODE_F32);
}

bool make_plan(int64_t n_total, int64_t n_rows, int64_t D, bool want_grad, int mode, Plan* p) {  // mode by value: the narrow split mode may fall back
    const Shape* sh = pick_shape(D);
    if (!sh) return false;
    // narrow split mode: padded to 32-feature blocks; the lo copy is reached through a 32-bit byte offset from the hi copy
    p->n16 = 0;
    if (mode == MODE_SPLIT && D <= 192) {
        const int64_t dpad = align_up(D, 32);
        // from 1 024 points on: below, the fp32 kernels finish in a few microseconds and the split mode's two extra
        // launches (max |value|) cost more than its kernel saves (BASELINE cfg2, N = 350: 114 -> 119 us per step)
        if (n_total >= 1024 && align_up(n_total, 256) * dpad < (1ll << 31) - (1ll << 20)) p->n16 = 1;
        else mode = MODE_F32;
    }
    const bool bf16_tiles = mode != MODE_F32 && !p->n16;
    p->bf16_tiles = bf16_tiles ? 1 : 0;
    p->split = mode == MODE_SPLIT ? 1 : 0;
    p->NB = sh->nb;
    p->WAVES = sh->waves;
    p->DPAD = sh->dpad;
    if (bf16_tiles) {
        p->DPAD = (int)align_up(D, 128);
        p->NB = p->DPAD / 128;        // KB
        p->WAVES = 4;
    }
    if (p->n16) {
        p->DPAD = (int)align_up(D, 32);
        p->NB = p->DPAD / 32;         // KBN
        p->WAVES = 1;
    }
    // Narrow latents: 64 rows per wave once the problem offers enough workgroups to fill the chip with them
    // (measured crossover at D = 77: N = 5 120 is 8 % faster on 32-row blocks, N = 8 192 is 35 % faster on 64).
    p->narrow64 = 0;
    if (!bf16_tiles && !p->n16 && p->WAVES == 1 && p->NB <= 5) {
        const int64_t tps64 = align_up(n_total, 64) / TILE_P;
        const int64_t cs64 = tps64 >= 32 ? (tps64 / 16 < 64 ? tps64 / 16 : 64) : 1;
        p->narrow64 = ((n_rows + 63) / 64) * cs64 >= 2048 ? 1 : 0;
    }
    // Launches of the 16-bit matrix-core modes: mmd_pipe_kernel / mmd_pipe_sum_kernel, 64 rows per workgroup (half the
    // L2 -> LDS stream of 32-row blocks; one workgroup per CU, its vector work software-pipelined between the matrix
    // instructions), for up to 6 feature blocks per wave (D <= 768: registers).  Wider: mmd_bf16_kernel, 32 rows.
    // The pipelined kernels address points with 32-bit element offsets (and the lo copy of split mode through a
    // 32-bit byte offset from the hi copy).
    const bool pipe_ok = bf16_tiles && p->NB <= kPipeMaxKb && align_up(n_total, 64) * p->DPAD < (1ll << 31) - (1ll << 20);
    p->pipe = (pipe_ok && want_grad) ? 1 : 0;
    p->pipe_sum = (pipe_ok && (p->split || kPipeSumBf16)) ? 1 : 0;
    const int rows_wg = p->n16 ? mmd_n16_rows(p->NB) : ((p->narrow64 || p->pipe_sum) ? 64 : ROWS_PER_WG);
    // whole 32- (64-) point blocks (symmetric y-y enumeration); 128 for the split plans on the 16-bit tile kernels, whose
    // sum-only launch may run 128-row workgroups over 128-point blocks (mmd_q32_sum3_kernel)
    p->n_pad = (int)align_up(n_total, (p->split && (p->pipe_sum || (bf16_tiles && p->NB > kPipeMaxKb))) ? 128 : rows_wg);
    p->row_blocks = (int)((n_rows + rows_wg - 1) / rows_wg);
    p->RT = p->n16 ? rows_wg / 16 : ((p->pipe || p->pipe_sum) ? 4 : 2);
    p->row_blocks_g = p->narrow64 ? p->row_blocks : (int)((n_rows + 16 * p->RT - 1) / (16 * p->RT));
    p->n_rows_pad = p->narrow64 ? p->row_blocks * 64 : p->row_blocks_g * 16 * p->RT;
    p->tiles_per_set = p->n_pad / TILE_P;
    // Column splits: enough workgroups to fill the chip, and a count that fills the LAST round of
    // resident workgroups too (a 1.9-round launch wastes 7 % of the machine on its half-empty tail).
    // (n16: four-wave workgroups, as many per CU as its waves' registers allow per SIMD)
    const int wg_per_cu = p->n16 ? mmd_n16_wg_per_cu(p->NB)
                        : bf16_tiles ? ((!p->split && !p->pipe && !p->pipe_sum && p->NB <= 6) ? 2 : 1)
                                     : (p->WAVES == 1 ? 8 : (p->WAVES == 4 && p->NB <= 12 ? 2 : 1));
    const int slots = 256 * wg_per_cu;
    // at least 32 of the 2 * tiles_per_set column tiles per split (4 when the whole problem is tiny)
    const int max_cs = p->tiles_per_set >= 32 ? (p->tiles_per_set / 16 < 64 ? p->tiles_per_set / 16 : 64)
                                              : (p->tiles_per_set >= 4 ? p->tiles_per_set / 2 : 1);
    int cs = 1;
    double best = 1e30;
    for (int c = 1; c <= max_cs; ++c) {
        const double rounds = (double)p->row_blocks_g * c / slots;
        const double waste = (rounds < 1.0 ? 1.0 / rounds : ((double)(long long)(rounds + 0.999999)) / rounds);
        // prefer fewer splits unless a larger count is at least 2 % better
        if (waste < best * 0.98) { best = waste; cs = c; }
        if (waste <= 1.0005) break;
    }
    p->col_splits = cs < 1 ? 1 : cs;
    // bf16 tiles: a step lasts ~1 us, less than an HBM round trip, so the kernel runs at the pace of its L2 misses.
    // While the column data (both sets) fits the 256 MB Infinity Cache those are cheap; past it the kernel lost a
    // quarter of its rate (N = 65 536 -> 98 304 at D = 768: 930 -> 750 TFLOP/s).  So split the columns into chunks of
    // ~24 MB and run split-major: all row blocks of chunk 0, then chunk 1, ... -- every resident set re-reads a
    // chunk the previous sets left in the cache (cfg4 size: 641 -> 492 ms; sweep in DESIGN 4.1b).  Costs one
    // gradient slab per split (capped at 32 GiB).
    p->sp_major = 0;
    // Q-form gradient: the split pipelined kernels, and the split 32-row kernels of 768 < D <= 1024
    p->qform = (kQForm && p->split && want_grad && !p->n16 && (p->pipe || (bf16_tiles && p->NB >= TVAE_QF32_MIN_KB && p->RT == 2))) ? 1 : 0;
    if (bf16_tiles) {
        const int64_t col_bytes = 2ll * p->n_pad * p->DPAD * (p->split ? 4 : 2);
        // the pipelined Q-form instance fetches tiles three intervals ahead into its ring and no longer cares (cfg4 size,
        // whole call: 1 split 618 ms, 4 -> 622, 16 -> 630, 40 -> 640; N = 65 536: 46.9 / 47.0 / 47.8 at 1 / 3 / 8), so it
        // keeps the load-balance count above and one gradient slab (cfg4: 3.0 GB of workspace instead of 13.8); the
        // rare E-form fallback of such a call runs unchunked (~15 % slower at that size)
        if (col_bytes > (64ll << 20) && !(p->qform && p->pipe)) {
            // chunk size: ~24 MB for the one-product kernels (a step lasts ~1 us); the split kernels do two to three times the
            // matrix work per byte (~5 us per step) and are at their best with ~96 MB (cfg4 size: 1 split 1 435 ms,
            // 4 -> 1 353, 8 -> 1 252, 16 -> 1 217, 32 -> 1 225, 63 -> 1 232)
            const int64_t chunk = p->split ? (96ll << 20) : (24ll << 20);
            int64_t want = (col_bytes + chunk - 1) / chunk;
            const int64_t hard_max = p->tiles_per_set / 16 < 64 ? p->tiles_per_set / 16 : 64;   // >= 16 tiles of a set per split
            if (want > hard_max) want = hard_max;
            const int64_t slab = (int64_t)p->n_rows_pad * p->DPAD * 4;
            while (want > 1 && want_grad && want * slab > (32ll << 30)) --want;
            if (want > p->col_splits) p->col_splits = (int)want;
            p->sp_major = 1;
        }
    }
#ifdef TVAE_PHASE_TIMING
    if (getenv("TVAE_DEBUG_COL_SPLITS")) p->col_splits = atoi(getenv("TVAE_DEBUG_COL_SPLITS"));   // experiments only
#endif
    int rpw = (p->n_pad + 511) / 512;
    if (rpw < 4) rpw = 4;
    if (rpw > 64) rpw = 64;
    p->rows_per_wave = rpw;
    p->nwaves = (p->n_pad + rpw - 1) / rpw + 2;       // upper bound: the row range of a call adds two chunk ends
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off = align_up(off + bytes, 256); return o; };
    const int copy_elt = (bf16_tiles && !p->split) ? 2 : 4;     // split: the hi copy, then the lo copy
    p->off_zp = take((int64_t)p->n_pad * p->DPAD * copy_elt);
    p->off_yp = take((int64_t)p->n_pad * p->DPAD * copy_elt);
    // the 32-point-step kernels touch (L2 prefetch) up to four steps past the last step of a column range: slack behind the
    // prior's image for the range that ends with it
    if (p->split) (void)take((int64_t)8 * 32 * p->DPAD * 2);
    p->off_zn = take((int64_t)p->n_pad * 4);
    p->off_yn = take((int64_t)p->n_pad * 4);
    p->off_colpart = take((int64_t)2 * p->nwaves * p->DPAD * (p->split ? 8 : 4));      // split: stage A's fp64 partials first
    p->off_shift = take((int64_t)p->DPAD * 4);
    p->off_dots = take((int64_t)((p->DPAD + 63) / 64) * 3 * 8);
    p->off_sums = take((int64_t)(2 * p->row_blocks + p->row_blocks_g) * p->col_splits * 2 * 8);
    p->off_rowsum = take(want_grad ? (int64_t)p->col_splits * p->n_rows_pad * 8 : 0);
    p->off_grad = take(want_grad ? (int64_t)p->col_splits * p->n_rows_pad * p->DPAD * 4 : 0);
    // soft rendezvous: multi-wave fp32 gradient launches whose workgroups stream at least two periods
    p->sync_epochs = 0; p->sync_group = slots; p->sync_groups = 0;
    if (want_grad && !bf16_tiles && p->WAVES > 1) {
        const int64_t stream = 2ll * p->tiles_per_set / p->col_splits;          // shortest stream of a workgroup
        const int64_t epochs = (stream - 1) / SYNC_TILES;
        const int64_t groups = ((int64_t)p->row_blocks_g * p->col_splits + slots - 1) / slots;
        if (epochs >= 1 && slots % 8 == 0 && groups * epochs <= (1 << 17)) { p->sync_epochs = (int)epochs; p->sync_groups = (int)groups; }
    }
    p->off_sync = take((int64_t)p->sync_groups * 8 * p->sync_epochs * 4);
    p->off_amax = take(p->split ? 256 + (int64_t)2 * p->nwaves * 4 : 0);      // the scale word, then the per-wave maxima of stage A
    p->off_reca = take(p->split ? (int64_t)rec_a_doubles(p->DPAD) * 8 : 0);
    p->off_recb = take((int64_t)rec_b_doubles(p->DPAD) * 8);
    p->off_recb_loc = take((int64_t)rec_b_doubles(p->DPAD) * 8);
    p->off_mu = take(p->split ? (int64_t)p->DPAD * 4 : 0);
    p->ldq = (int)align_up(D + 1, 4);
    p->off_col3 = take(p->qform ? (int64_t)2 * p->nwaves * p->DPAD * 4 : 0);
    p->off_m3 = take(p->qform ? (int64_t)p->DPAD * 4 : 0);
    p->off_m2 = take(p->qform ? (int64_t)2 * D * p->ldq * 4 : 0);                       // Z^T Z (then M2 | shift), Y^T Y
    p->off_p = take(p->qform ? (int64_t)n_rows * p->ldq * 4 : 0);
    int64_t gs_floats = 0;
    if (p->qform) {
        gs_floats = gram16_slab_floats(p->DPAD, n_total);
        const int64_t g2 = gemm_slab_floats((int)n_rows, (int)D + 1, (int)D);
        if (g2 > gs_floats) gs_floats = g2;
    }
    p->off_gslab = take(gs_floats * 4);
    p->qsum = (kQForm && p->split && p->pipe_sum) ? 1 : 0;
    p->off_qflag = take((p->qform || p->qsum) ? 256 : 0);
    p->bytes = off;
    return true;
}

template <int NB, int WAVES, bool GRAD>
int launch_tile(const MmdArgs& a, int nblocks, hipStream_t s) {
    constexpr int lds_bytes = TileLds<NB, WAVES>::BYTES;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_tile_kernel<NB, WAVES, GRAD>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }));
    hipLaunchKernelGGL((mmd_tile_kernel<NB, WAVES, GRAD>), dim3(nblocks), dim3(64 * WAVES), lds_bytes, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

template <int NB>
int launch_tile_both(const MmdArgs& a, int nwg0, int nwg1, hipStream_t s) {
    constexpr int lds_bytes = TileLds<NB, 1>::BYTES;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_tile_both_kernel<NB>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }));
    hipLaunchKernelGGL((mmd_tile_both_kernel<NB>), dim3(nwg0 + nwg1), dim3(64), lds_bytes, s, a, nwg0);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

// single-wave shapes only (padded dimension <= 192); returns -1 when there is no such kernel
int launch_tile_both_shape(int dpad, const MmdArgs& a, int nwg0, int nwg1, hipStream_t s) {
    switch (dpad) {
        case 16: return launch_tile_both<1>(a, nwg0, nwg1, s);
        case 32: return launch_tile_both<2>(a, nwg0, nwg1, s);
        case 48: return launch_tile_both<3>(a, nwg0, nwg1, s);
        case 64: return launch_tile_both<4>(a, nwg0, nwg1, s);
        case 80: return launch_tile_both<5>(a, nwg0, nwg1, s);
        case 96: return launch_tile_both<6>(a, nwg0, nwg1, s);
        case 112: return launch_tile_both<7>(a, nwg0, nwg1, s);
        case 128: return launch_tile_both<8>(a, nwg0, nwg1, s);
        case 160: return launch_tile_both<10>(a, nwg0, nwg1, s);
        case 192: return launch_tile_both<12>(a, nwg0, nwg1, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no single-wave kernel for padded dimension %d", dpad);
}

template <bool GRAD>
int launch_tile_shape(int dpad, const MmdArgs& a, int nblocks, hipStream_t s) {
    switch (dpad) {
        case 16: return launch_tile<1, 1, GRAD>(a, nblocks, s);
        case 32: return launch_tile<2, 1, GRAD>(a, nblocks, s);
        case 48: return launch_tile<3, 1, GRAD>(a, nblocks, s);
        case 64: return launch_tile<4, 1, GRAD>(a, nblocks, s);
        case 80: return launch_tile<5, 1, GRAD>(a, nblocks, s);
        case 96: return launch_tile<6, 1, GRAD>(a, nblocks, s);
        case 112: return launch_tile<7, 1, GRAD>(a, nblocks, s);
        case 128: return launch_tile<8, 1, GRAD>(a, nblocks, s);
        case 160: return launch_tile<10, 1, GRAD>(a, nblocks, s);
        case 192: return launch_tile<12, 1, GRAD>(a, nblocks, s);
        case 256: return launch_tile<4, 4, GRAD>(a, nblocks, s);
        case 384: return launch_tile<6, 4, GRAD>(a, nblocks, s);
        case 512: return launch_tile<8, 4, GRAD>(a, nblocks, s);
        case 768: return launch_tile<12, 4, GRAD>(a, nblocks, s);
        case 1024: return launch_tile<8, 8, GRAD>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no kernel for padded dimension %d", dpad);
}

template <bool GRAD>
int launch_tiles(const Plan& p, const MmdArgs& a, int nblocks, hipStream_t s) {
    if (p.bf16_tiles && GRAD && p.pipe) return launch_mmd_pipe(true, p.split != 0, p.NB, a, nblocks, s);
    if (p.bf16_tiles && !GRAD && p.pipe_sum) return launch_mmd_pipe(false, p.split != 0, p.NB, a, nblocks, s);
    if (p.bf16_tiles) return launch_mmd_bf16(GRAD, p.split != 0, p.NB, 2, a, nblocks, s);
    if (p.narrow64) return launch_mmd_narrow(GRAD, p.NB, a, nblocks, s);
    return launch_tile_shape<GRAD>(p.DPAD, a, nblocks, s);
}

}  // namespace
}  // namespace tvae

using namespace tvae;

namespace {
// one-shot, thread-local measurement hook (tvae_profile_next_mmd): events recorded around the tile kernel
thread_local hipEvent_t g_prof_start = nullptr;
thread_local hipEvent_t g_prof_stop = nullptr;
}  // namespace

extern "C" int tvae_profile_next_mmd(void* ev_start, void* ev_stop) {
    g_prof_start = static_cast<hipEvent_t>(ev_start);
    g_prof_stop = static_cast<hipEvent_t>(ev_stop);
    return TVAE_OK;
}

extern "C" int64_t tvae_mmd_workspace_bytes(int64_t n_total, int64_t n_rows, int64_t D, int want_grad) {
    // the dtype is not an argument here: report the larger of the two plans a call may pick
    int64_t best = -1;
    for (int dtype : {TVAE_F32, TVAE_BF16, TVAE_F32_STRICT}) {
        const int64_t b = tvae_mmd_workspace_bytes_for(n_total, n_rows, D, want_grad, dtype);
        if (b < 0) return -1;
        if (b > best) best = b;
    }
    return best;
}

extern "C" int64_t tvae_mmd_workspace_bytes_for(int64_t n_total, int64_t n_rows, int64_t D, int want_grad, int dtype) {
    if (dtype != TVAE_F32 && dtype != TVAE_BF16 && dtype != TVAE_F32_STRICT) return -1;
    if (n_total > 0 && n_rows > 0 && D > 1024 && n_total < (1ll << 30))
        return mmd_wide_workspace_bytes(n_total, n_rows, D, want_grad != 0);
    Plan p;
    if (n_total <= 0 || n_rows <= 0 || D <= 0 ||
        !make_plan(n_total, n_rows, D, want_grad != 0, mmd_mode(dtype, D), &p)) return -1;
    return p.bytes;
}

namespace {
// One evaluation, resolved: the plan, the element type of the inputs, and the row ranges.  The SOURCE rows
// [src_row0, src_row0 + n_src) are the rows whose values this call holds: all of them (one process), or the rows it
// owns (a rank of a sharded call).
struct Call {
    Plan p;
    int dtype;              // TVAE_F32 / TVAE_BF16: the element type (TVAE_F32_STRICT resolved into the plan)
    int64_t n_total, row_offset, n_rows, D, src_row0, n_src;
    char* ws;
    int rpw, lo, hi, hi2, nw0, nw1, nw2, nwaves;      // row chunks of the prep pass (mmd_prep_kernel)
    bool centred() const { return p.split != 0; }
    template <typename U> U* at(int64_t off) const { return reinterpret_cast<U*>(ws + off); }
};

int resolve_call(Call* c, int64_t n_total, int64_t row_offset, int64_t n_rows, int64_t D, int dtype, bool want_grad, bool sharded,
                 void* workspace, int64_t workspace_bytes) {
    TVAE_REQUIRE(workspace, "mmd: null workspace");
    TVAE_REQUIRE(dtype == TVAE_F32 || dtype == TVAE_BF16 || dtype == TVAE_F32_STRICT, "mmd: unknown dtype %d", dtype);
    TVAE_REQUIRE(n_total > 0 && n_rows > 0 && D > 0, "mmd: empty problem (n_total=%lld n_rows=%lld D=%lld)",
                 (long long)n_total, (long long)n_rows, (long long)D);
    TVAE_REQUIRE(row_offset >= 0 && row_offset + n_rows <= n_total, "mmd: row block [%lld,%lld) outside [0,%lld)",
                 (long long)row_offset, (long long)(row_offset + n_rows), (long long)n_total);
    TVAE_REQUIRE(n_total < (1ll << 30), "mmd: n_total too large");
    TVAE_REQUIRE(make_plan(n_total, n_rows, D, want_grad, mmd_mode(dtype, D), &c->p),
                 "mmd: latent dimension %lld not supported by the tile kernels (max 1024)", (long long)D);
    if (workspace_bytes < c->p.bytes)
        return fail(TVAE_ERR_WORKSPACE, "mmd: workspace %lld B < required %lld B", (long long)workspace_bytes,
                    (long long)c->p.bytes);
    TVAE_REQUIRE(aligned16(workspace), "mmd: workspace must be 16-byte aligned");
    c->dtype = dtype == TVAE_F32_STRICT ? TVAE_F32 : dtype;
    c->n_total = n_total; c->row_offset = row_offset; c->n_rows = n_rows; c->D = D;
    c->src_row0 = sharded ? row_offset : 0;
    c->n_src = sharded ? n_rows : n_total;
    c->ws = static_cast<char*>(workspace);
    const int rpw = c->p.rows_per_wave;
    c->rpw = rpw;
    c->lo = (int)row_offset; c->hi = (int)(row_offset + n_rows);
    c->hi2 = sharded ? (int)n_total : c->hi;
    c->nw0 = sharded ? 0 : (c->lo + rpw - 1) / rpw;
    c->nw1 = (c->hi - c->lo + rpw - 1) / rpw;
    c->nw2 = (c->p.n_pad - c->hi2 + rpw - 1) / rpw;
    c->nwaves = c->nw0 + c->nw1 + c->nw2;          // <= p.nwaves (the plan's bound)
    return TVAE_OK;
}

// stage A (split modes only): rec_a <- column sums and max |value| of the source rows
int stage_stats(const Call& c, const void* z_src, const void* y_src, hipStream_t s) {
    if (!c.centred()) return TVAE_OK;
    const Plan& p = c.p;
    const int nw = (int)((c.n_src + c.rpw - 1) / c.rpw);
    double* colpart = c.at<double>(p.off_colpart);
    float* amaxpart = c.at<float>(p.off_amax + 256);
    hipLaunchKernelGGL(mmd_stats_kernel, dim3((nw + 3) / 4, 2), dim3(256), 0, s, (const float*)z_src, (const float*)y_src,
                       (int)c.n_src, (int)c.D, p.DPAD, c.rpw, nw, colpart, amaxpart);
    TVAE_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(mmd_stats_final_kernel, dim3((p.DPAD + 63) / 64), dim3(1024), 0, s, (const double*)colpart,
                       (const float*)amaxpart, (nw + 3) / 4, p.DPAD, c.at<double>(p.off_reca));
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

// stage B: images, norms and records of the source rows (rec_a complete in the workspace), and for the Q-form gradient
// Z'^T Z' and Y'^T Y' over the source rows
int stage_prep(const Call& c, const void* z_src, const void* y_src, hipStream_t s) {
    const Plan& p = c.p;
    void* zp = c.ws + p.off_zp;
    void* yp = c.ws + p.off_yp;
    float* zn = c.at<float>(p.off_zn);
    float* yn = c.at<float>(p.off_yn);
    float* colpart = c.at<float>(p.off_colpart);
    float* colpart3 = p.qform ? c.at<float>(p.off_col3) : nullptr;
    const double* rec_a = c.centred() ? c.at<double>(p.off_reca) : nullptr;
    float* mu = c.centred() ? c.at<float>(p.off_mu) : nullptr;
    unsigned* amax = c.centred() ? c.at<unsigned>(p.off_amax) : nullptr;
    const double inv_2n = 0.5 / (double)c.n_total;
    const long long lo_off = (long long)p.n_pad * p.DPAD;      // split: elements from the hi copy to the lo copy
    dim3 pgrid((c.nwaves + 3) / 4, 2);
#define TVAE_PREP(T, TC, SPLIT)                                                                                          \
    hipLaunchKernelGGL((mmd_prep_kernel<T, TC, SPLIT>), pgrid, dim3(256), 0, s, (const T*)z_src, (const T*)y_src,       \
                       (int)c.n_total, p.n_pad, (int)c.D, p.DPAD, (TC*)zp, (TC*)yp, zn, yn, colpart, c.rpw, c.nwaves,  \
                       c.lo, c.hi, c.hi2, c.nw0, c.nw1, (int)c.src_row0, (SPLIT) ? lo_off : 0ll, rec_a, inv_2n, mu,    \
                       amax, colpart3)
    if (p.split) TVAE_PREP(float, _Float16, true);
    else if (c.dtype == TVAE_F32) TVAE_PREP(float, float, false);
    else if (!p.bf16_tiles) TVAE_PREP(bf16, float, false);
    else TVAE_PREP(bf16, bf16, false);
#undef TVAE_PREP
    TVAE_CHECK_HIP(hipGetLastError());
    const int ncb = (p.DPAD + 63) / 64;
    // one process: the records are complete, the closed-form pieces come out of the same launch
    const bool whole = c.n_src == c.n_total;
    hipLaunchKernelGGL(mmd_recb_kernel, dim3(ncb + 1), dim3(1024), 0, s, (const float*)colpart, (const float*)colpart3, c.nwaves,
                       c.nw0, c.nw1, p.DPAD, (const float*)zn, (const float*)yn, (int)c.src_row0, (int)(c.src_row0 + c.n_src),
                       c.lo, c.hi, ncb, c.at<double>(p.off_recb), c.at<double>(p.off_recb_loc), rec_a, (int)c.D,
                       whole ? c.at<float>(p.off_shift) : nullptr, p.qform ? c.at<float>(p.off_m3) : nullptr,
                       c.at<double>(p.off_dots), (p.qform || p.qsum) ? c.at<QFlag>(p.off_qflag) : nullptr);
    TVAE_CHECK_HIP(hipGetLastError());
    if (p.qform) {
        // Z'^T Z' and Y'^T Y' of the source rows, straight from the images just written (gram16.hip)
        float* m2z = c.at<float>(p.off_m2);
        float* m2y = m2z + (long long)c.D * p.ldq;
        float* gslab = c.at<float>(p.off_gslab);
        const int64_t gfloats = gram16_slab_floats(p.DPAD, c.n_total);
        TVAE_PROPAGATE(gram16_launch(zp, lo_off, c.src_row0, c.n_src, p.DPAD, c.D, p.ldq, amax, m2z, gslab, gfloats, s));
        TVAE_PROPAGATE(gram16_launch(yp, lo_off, c.src_row0, c.n_src, p.DPAD, c.D, p.ldq, amax, m2y, gslab, gfloats, s));
    }
    return TVAE_OK;
}

// stage C: the closed-form pieces from the complete records, the tile kernels, finalize.
// z_rows: the owned rows' values.  phase bit 0: everything that needs only the prior's image (the closed-form
// kernel and, with a gradient, the y-y sum-only launch); bit 1: the rest.  A sharded caller runs phase 1 while the
// latent images are still on the wire.
int stage_tiles(const Call& c, const void* z_rows, double* sums3, float* loss, double loss_scale, void* d_z, double grad_scale,
                int phase, hipStream_t s) {
    const Plan& p = c.p;
    const int64_t D = c.D, n_rows = c.n_rows, n_total = c.n_total, row_offset = c.row_offset;
    void* zp = c.ws + p.off_zp;
    void* yp = c.ws + p.off_yp;
    float* zn = c.at<float>(p.off_zn);
    float* yn = c.at<float>(p.off_yn);
    float* shift = c.at<float>(p.off_shift);
    double* dots = c.at<double>(p.off_dots);
    unsigned* split_amax = p.split ? c.at<unsigned>(p.off_amax) : nullptr;
    QFlag* qflag = (p.qform || p.qsum) ? c.at<QFlag>(p.off_qflag) : nullptr;
    float* m3 = p.qform ? c.at<float>(p.off_m3) : nullptr;
    float* pmat = p.qform ? c.at<float>(p.off_p) : nullptr;
    const long long lo_off = (long long)p.n_pad * p.DPAD;

    MmdArgs a;
    a.cols[0] = (const float*)zp; a.cols[1] = (const float*)yp; a.cnorm[0] = zn; a.cnorm[1] = yn;
    a.cols_lo[0] = a.cols_lo[1] = nullptr;
    a.split_amax = split_amax;
    a.qflag = qflag ? &qflag->on : nullptr;
    if (p.split) {
        a.cols_lo[0] = reinterpret_cast<const float*>(static_cast<const bf16*>(zp) + lo_off);
        a.cols_lo[1] = reinterpret_cast<const float*>(static_cast<const bf16*>(yp) + lo_off);
    }
    a.n_rows = (int)n_rows; a.n_cols = (int)n_total; a.row_offset = (int)row_offset;
    a.row_blocks = p.row_blocks; a.col_splits = p.col_splits; a.tiles_per_set = p.tiles_per_set;
    a.n_rows_pad = p.n_rows_pad;
    a.inv_d2 = (float)(1.0 / ((double)D * (double)D));
    a.sp_major = (p.sync_epochs > 0 || p.sp_major) ? 1 : 0;     // a resident set then shares ONE column range
    a.sync_ctr = nullptr; a.sync_epochs = p.sync_epochs; a.sync_group = p.sync_group; a.launch_blocks = 0;
    const int sym_rows = p.n16 ? 16 * p.RT : ((p.narrow64 || p.pipe_sum) ? 64 : ROWS_PER_WG);      // row block of the symmetric y-y enumeration
    // The symmetric enumeration shares the pairs {I, J} of row blocks between the calls that own I and J (mmd_tile_body),
    // so EVERY call of a partition has to make the same choice.  The rule is therefore one that all blocks of an EQUAL
    // partition answer alike: the whole problem in one call, or a block of whole sym_rows-row blocks that tiles
    // [0, n_total).  (Until round 3 a ragged LAST block qualified on its own: with three ranks of 752 rows the last one took
    // its symmetric share while the other two summed all of theirs -- y-y sum off by 6e-7, found by tools/fuzz_gpu.py.)
    const bool one_call = row_offset == 0 && n_rows == n_total;
    const bool equal_block = row_offset % sym_rows == 0 && n_rows % sym_rows == 0 && n_total % n_rows == 0 && row_offset % n_rows == 0;
    a.sym = (one_call || equal_block) ? 1 : 0;
    a.sym2 = (one_call || (equal_block && row_offset % 128 == 0 && n_rows % 128 == 0)) ? 1 : 0;
    a.wg_sums = c.at<double>(p.off_sums);
    a.wg_rowsum = c.at<double>(p.off_rowsum);
    a.wg_grad = c.at<float>(p.off_grad);
    // workgroups of the two problems (the bf16 gradient launch has its own, larger, row blocks)
    const int nwg_sum = p.row_blocks * p.col_splits;
    const int nwg0 = d_z ? p.row_blocks_g * p.col_splits : nwg_sum, nwg1 = nwg_sum;
    // the y-y problem as a launch of its own (the others put both problems into one launch)
    const bool yy_alone = d_z && !p.n16 && !(!p.bf16_tiles && !p.narrow64 && p.WAVES == 1);

    const bool whole = c.n_src == c.n_total;          // one process: mmd_recb_kernel has emitted the closed-form pieces already
    if (phase & 1) {
        if (!whole)
        hipLaunchKernelGGL(mmd_closed_kernel, dim3(1), dim3(1024), 0, s, (const double*)c.at<double>(p.off_recb),
                           (const double*)c.at<double>(p.off_recb_loc), (const double*)(c.centred() ? c.at<double>(p.off_reca) : nullptr),
                           (int)D, p.DPAD, shift, m3, dots, qflag);
        TVAE_CHECK_HIP(hipGetLastError());
        if (yy_alone && phase == 1) {      // (phase 3: behind the gradient launch, inside the profiling bracket)
            // a sharded caller's bracket opens here and closes behind the gradient launch of its phase-2 call
            if (g_prof_start) TVAE_CHECK_HIP(hipEventRecord(g_prof_start, s));
            g_prof_start = nullptr;
            a.problem_base = 1; a.sums_base = nwg0; a.row_blocks = p.row_blocks;
            TVAE_PROPAGATE(launch_tiles<false>(p, a, nwg1, s));    // y rows: sum only
        }
    }
    if (!(phase & 2)) return TVAE_OK;

    // Q-form gradient: M2 = Z^T Z - Y^T Y with shift as an extra column, P = Z_rows [M2 | shift]
    if (p.qform) {
        float* m2z = c.at<float>(p.off_m2);
        float* m2y = m2z + (long long)D * p.ldq;
        hipLaunchKernelGGL(mmd_m2_fix_kernel, dim3(256), dim3(256), 0, s, m2z, (const float*)m2y, (const float*)shift, (int)D, p.ldq);
        TVAE_CHECK_HIP(hipGetLastError());
        GemmDesc h{};
        h.bias = nullptr; h.epilogue = EPI_NONE; h.dtype = TVAE_F32; h.colsum_out = nullptr;
        h.M = (int)n_rows; h.N = (int)D + 1; h.K = (int)D;
        h.a_kmajor = false; h.b_kmajor = true;
        h.amap = dense_rows(D); h.bmap = dense_rows(p.ldq); h.cmap = dense_rows(p.ldq);
        h.slab = c.at<float>(p.off_gslab); h.slab_floats = gemm_slab_floats((int)n_rows, (int)D + 1, (int)D);
        h.a_sub = c.at<float>(p.off_mu);
        h.A = z_rows; h.B = m2z; h.C = pmat;
        TVAE_PROPAGATE(gemm_launch(h, s));
    }
    if (d_z && p.sync_epochs > 0)        // arrival counters of the soft rendezvous (a memset node under graph capture)
        TVAE_CHECK_HIP(hipMemsetAsync(c.ws + p.off_sync, 0, (size_t)p.sync_groups * 8 * p.sync_epochs * 4, s));
    const hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
    if (ev0) TVAE_CHECK_HIP(hipEventRecord(ev0, s));
    if (p.n16) {
        // narrow split mode: both problems in one launch, with or without the gradient
        a.problem_base = 0; a.sums_base = 0; a.row_blocks = p.row_blocks;
        TVAE_PROPAGATE(launch_mmd_n16(d_z != nullptr, p.NB, a, nwg0, nwg1, s));
    } else if (d_z && !yy_alone) {
        // single-wave shapes: both problems in one launch (row blocks are the same 32 rows in both)
        a.problem_base = 0; a.sums_base = 0; a.row_blocks = p.row_blocks;
        TVAE_PROPAGATE(launch_tile_both_shape(p.DPAD, a, nwg0, nwg1, s));
    } else if (d_z) {
        a.problem_base = 0; a.sums_base = 0; a.row_blocks = p.row_blocks_g;
        if (p.sync_epochs > 0) {
            a.sync_ctr = c.at<unsigned>(p.off_sync);
            a.launch_blocks = nwg0;
        }
        TVAE_PROPAGATE(launch_tiles<true>(p, a, nwg0, s));              // z rows: sums + gradient
        if (phase == 3) {
            a.sync_ctr = nullptr;
            a.problem_base = 1; a.sums_base = nwg0; a.row_blocks = p.row_blocks;
            TVAE_PROPAGATE(launch_tiles<false>(p, a, nwg1, s));    // y rows: sum only
        }
    } else {
        a.problem_base = 0; a.sums_base = 0;
        TVAE_PROPAGATE(launch_tiles<false>(p, a, 2 * nwg_sum, s));
    }
    if (ev1) TVAE_CHECK_HIP(hipEventRecord(ev1, s));

    FinalArgs f;
    f.wg_sums = a.wg_sums; f.nwg0 = nwg0; f.nwg1 = nwg1; f.wg_rowsum = a.wg_rowsum; f.wg_grad = a.wg_grad; f.shift = shift; f.z_src = z_rows;
    f.mu = c.centred() ? c.at<float>(p.off_mu) : nullptr;
    f.col_splits = p.col_splits; f.n_rows = (int)n_rows; f.n_rows_pad = p.n_rows_pad; f.row_offset = (int)row_offset;
    f.D = (int)D; f.DPAD = p.DPAD; f.sums3 = sums3; f.loss = loss; f.loss_scale = loss_scale; f.grad_scale = grad_scale;
    f.d_z = d_z;
    f.split_amax = split_amax;
    f.zn = zn; f.n_total = (int)n_total; f.dots = dots; f.ndots = whole ? (p.DPAD + 63) / 64 : 1;
    f.rec_all = c.at<double>(p.off_recb); f.rec_loc = c.at<double>(p.off_recb_loc);
    f.qflag = p.qform ? qflag : nullptr; f.m3 = m3; f.P = pmat; f.ldP = p.ldq;
    long long fb = d_z ? ((long long)n_rows * D + 255) / 256 : 1;
    if (fb > 4096) fb = 4096;
    if (fb < 1) fb = 1;
    if (c.dtype == TVAE_F32) hipLaunchKernelGGL((mmd_finalize_kernel<float>), dim3((unsigned)fb), dim3(256), 0, s, f);
    else hipLaunchKernelGGL((mmd_finalize_kernel<bf16>), dim3((unsigned)fb), dim3(256), 0, s, f);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

int mmd_fwd_bwd_impl(const void* z_all, const void* y_all, int64_t n_total, int64_t row_offset,
                     int64_t n_rows, int64_t D, int dtype, double* sums3, float* loss,
                     double loss_scale, void* d_z, double grad_scale, void* workspace,
                     int64_t workspace_bytes, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    TVAE_REQUIRE(z_all && y_all && sums3 && workspace, "mmd: null pointer argument");
    TVAE_REQUIRE(dtype == TVAE_F32 || dtype == TVAE_BF16 || dtype == TVAE_F32_STRICT, "mmd: unknown dtype %d", dtype);
    if (D > 1024) {      // beyond the tile kernels' register budget: strip-wise route through the GEMM kernel
        TVAE_REQUIRE(n_total > 0 && n_rows > 0, "mmd: empty problem (n_total=%lld n_rows=%lld D=%lld)",
                     (long long)n_total, (long long)n_rows, (long long)D);
        TVAE_REQUIRE(row_offset >= 0 && row_offset + n_rows <= n_total, "mmd: row block [%lld,%lld) outside [0,%lld)",
                     (long long)row_offset, (long long)(row_offset + n_rows), (long long)n_total);
        TVAE_REQUIRE(n_total < (1ll << 30), "mmd: n_total too large");
        const hipEvent_t w0 = g_prof_start, w1 = g_prof_stop;
        g_prof_start = g_prof_stop = nullptr;
        if (w0) TVAE_CHECK_HIP(hipEventRecord(w0, s));
        TVAE_PROPAGATE(mmd_wide_launch(z_all, y_all, n_total, row_offset, n_rows, D, dtype == TVAE_F32_STRICT ? TVAE_F32 : dtype, sums3,
                                       loss, loss_scale, d_z, grad_scale, workspace, workspace_bytes, s));
        if (w1) TVAE_CHECK_HIP(hipEventRecord(w1, s));
        return TVAE_OK;
    }
    Call c;
    TVAE_PROPAGATE(resolve_call(&c, n_total, row_offset, n_rows, D, dtype, d_z != nullptr, false, workspace, workspace_bytes));
    TVAE_PROPAGATE(stage_stats(c, z_all, y_all, s));
    TVAE_PROPAGATE(stage_prep(c, z_all, y_all, s));
    const size_t elt = c.dtype == TVAE_F32 ? 4 : 2;
    return stage_tiles(c, static_cast<const char*>(z_all) + (size_t)row_offset * D * elt, sums3, loss, loss_scale, d_z, grad_scale, 3, s);
}
}  // namespace

extern "C" int tvae_mmd_fwd_bwd(const void* z_all, const void* y_all, int64_t n_total, int64_t row_offset,
                                int64_t n_rows, int64_t D, int dtype, double* sums3, float* loss,
                                double loss_scale, void* d_z, double grad_scale, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    return mmd_fwd_bwd_impl(z_all, y_all, n_total, row_offset, n_rows, D, dtype, sums3, loss, loss_scale, d_z, grad_scale,
                            workspace, workspace_bytes, stream);
}

// ---- row-sharded evaluation: the three stages as entry points (see "The call in three stages" above) ---------------
// Every rank passes the SAME (n_total, n_rows, D, want_grad, dtype) plan arguments and its own row_offset; equal row
// counts per rank, row_offset = rank * n_rows.
extern "C" int tvae_mmd_shard_layout(int64_t n_total, int64_t n_rows, int64_t D, int want_grad, int dtype, int64_t* out) {
    TVAE_REQUIRE(out, "mmd_shard_layout: null output");
    if (dtype != TVAE_F32 && dtype != TVAE_BF16 && dtype != TVAE_F32_STRICT) return fail(TVAE_ERR_ARG, "mmd_shard_layout: unknown dtype %d", dtype);
    Plan p;
    if (n_total <= 0 || n_rows <= 0 || D <= 0 || D > 1024 || n_total >= (1ll << 30) ||
        !make_plan(n_total, n_rows, D, want_grad != 0, mmd_mode(dtype, D), &p))
        return fail(TVAE_ERR_ARG, "mmd_shard_layout: no sharded form for n_total=%lld n_rows=%lld D=%lld (tile kernels: D <= 1024)",
                    (long long)n_total, (long long)n_rows, (long long)D);
    const int64_t elt = p.split ? 2 : ((p.bf16_tiles) ? 2 : 4);
    const int64_t img = (int64_t)p.n_pad * p.DPAD * elt;          // one image (split: hi; the lo image follows it)
    out[0] = p.bytes;
    out[1] = p.off_zp; out[2] = p.off_yp;
    out[3] = (int64_t)p.DPAD * elt;                               // bytes per image row
    out[4] = p.split ? img : 0;                                   // bytes from an image to its lo image (0: there is none)
    out[5] = p.off_zn; out[6] = p.off_yn;                         // squared norms, fp32 per row
    out[7] = p.split ? p.off_reca : -1;                           // stage-A record (fp64), -1: the mode has no stage A
    out[8] = p.split ? rec_a_doubles(p.DPAD) : 0;
    out[9] = p.off_recb; out[10] = p.off_recb_loc; out[11] = rec_b_doubles(p.DPAD);
    out[12] = p.qform ? p.off_m2 : -1;                            // Z'^T Z' then Y'^T Y' (fp32), summed over ranks; -1: none
    out[13] = p.qform ? (int64_t)2 * D * p.ldq : 0;               // ... floats
    out[14] = (p.split && p.qsum && (!want_grad || (p.qform && p.pipe))) ? 1 : 0;   // 1: the lo images of OTHER ranks' rows are read only when the Q-form flag is off
    out[15] = p.DPAD;
    return TVAE_OK;
}

extern "C" int tvae_mmd_stats_local(const void* z_loc, const void* y_loc, int64_t n_total, int64_t row_offset, int64_t n_rows,
                                    int64_t D, int dtype, int want_grad, void* workspace, int64_t workspace_bytes, void* stream) {
    TVAE_REQUIRE(z_loc && y_loc, "mmd_stats_local: null pointer argument");
    TVAE_REQUIRE(D <= 1024, "mmd_stats_local: D = %lld has no sharded form (tile kernels: D <= 1024)", (long long)D);
    Call c;
    TVAE_PROPAGATE(resolve_call(&c, n_total, row_offset, n_rows, D, dtype, want_grad != 0, true, workspace, workspace_bytes));
    return stage_stats(c, z_loc, y_loc, static_cast<hipStream_t>(stream));
}

extern "C" int tvae_mmd_prep_local(const void* z_loc, const void* y_loc, int64_t n_total, int64_t row_offset, int64_t n_rows,
                                   int64_t D, int dtype, int want_grad, void* workspace, int64_t workspace_bytes, void* stream) {
    TVAE_REQUIRE(z_loc && y_loc, "mmd_prep_local: null pointer argument");
    TVAE_REQUIRE(D <= 1024, "mmd_prep_local: D = %lld has no sharded form (tile kernels: D <= 1024)", (long long)D);
    Call c;
    TVAE_PROPAGATE(resolve_call(&c, n_total, row_offset, n_rows, D, dtype, want_grad != 0, true, workspace, workspace_bytes));
    return stage_prep(c, z_loc, y_loc, static_cast<hipStream_t>(stream));
}

extern "C" int tvae_mmd_tiles(const void* z_loc, int64_t n_total, int64_t row_offset, int64_t n_rows, int64_t D, int dtype,
                              double* sums3, float* loss, double loss_scale, void* d_z, double grad_scale, int phase,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    TVAE_REQUIRE(z_loc && sums3, "mmd_tiles: null pointer argument");
    TVAE_REQUIRE(phase >= 1 && phase <= 3, "mmd_tiles: phase %d (1: prior-only part, 2: the rest, 3: both)", phase);
    TVAE_REQUIRE(D <= 1024, "mmd_tiles: D = %lld has no sharded form (tile kernels: D <= 1024)", (long long)D);
    Call c;
    TVAE_PROPAGATE(resolve_call(&c, n_total, row_offset, n_rows, D, dtype, d_z != nullptr, true, workspace, workspace_bytes));
    return stage_tiles(c, z_loc, sums3, loss, loss_scale, d_z, grad_scale, phase, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------
// Allocation-free variant (no counterpart in the reference; for callers that cannot allocate: code under HIP graph
// capture, tight serving loops).  Same kernels as tvae_mmd_fwd_bwd; the workspace is a file-scope device array that
// every GPU the library runs on holds until the process ends.
// ------------------------------------------------------------------------------------------
namespace {
constexpr int64_t SCRATCH_BYTES = TVAE_MMD_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
// `scratch` is a host-side stand-in in host code, never a device address: each device's copy is looked up once
// (with that device current) and only the looked-up pointer is ever handed to kernels / memsets.
constexpr int MAX_DEVICES = 64;
std::atomic<void*> g_scratch_ptr[MAX_DEVICES];

int scratch_of_current_device(void** out) {
    int dev = 0;
    TVAE_CHECK_HIP(hipGetDevice(&dev));
    TVAE_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "mmd_static: device ordinal %d beyond %d", dev, MAX_DEVICES);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        TVAE_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        TVAE_REQUIRE(p != nullptr, "mmd_static: hipGetSymbolAddress returned a null scratch address");
        g_scratch_ptr[dev].store(p, std::memory_order_release);     // racing first calls store the same address
    }
    *out = p;
    return TVAE_OK;
}
}  // namespace

extern "C" int tvae_mmd_fwd_bwd_static(const void* z_all, const void* y_all, int64_t n_total, int64_t row_offset,
                                       int64_t n_rows, int64_t D, int dtype, double* sums3, float* loss,
                                       double loss_scale, void* d_z, double grad_scale, void* stream) {
    TVAE_REQUIRE(dtype == TVAE_F32 || dtype == TVAE_BF16 || dtype == TVAE_F32_STRICT, "mmd_static: unknown dtype %d", dtype);
    const int64_t need = tvae_mmd_workspace_bytes_for(n_total, n_rows, D, d_z != nullptr, dtype);
    TVAE_REQUIRE(need >= 0, "mmd_static: unsupported problem (n_total=%lld n_rows=%lld D=%lld)", (long long)n_total,
                 (long long)n_rows, (long long)D);
    if (need > SCRATCH_BYTES)
        return fail(TVAE_ERR_WORKSPACE,
                    "mmd_static: this problem needs %lld B of scratch, the built-in buffer holds %lld B "
                    "(n_total=%lld n_rows=%lld D=%lld): use tvae_mmd_fwd_bwd with a caller workspace",
                    (long long)need, (long long)SCRATCH_BYTES, (long long)n_total, (long long)n_rows, (long long)D);
    void* ws = nullptr;
    TVAE_PROPAGATE(scratch_of_current_device(&ws));
    return mmd_fwd_bwd_impl(z_all, y_all, n_total, row_offset, n_rows, D, dtype, sums3, loss, loss_scale, d_z, grad_scale,
                            ws, SCRATCH_BYTES, stream);
}
