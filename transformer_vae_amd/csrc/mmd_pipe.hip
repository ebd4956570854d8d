// Software-pipelined MMD gradient kernel on the 16-bit matrix cores: 64 rows per workgroup, one barrier per step.
// Math, workspace layout and per-workgroup outputs are those of mmd.hip / mmd_bf16.hip.
//
// mmd_bf16_kernel runs Gram -> exchange -> barrier -> expm1 -> barrier -> gradient strictly one after the other; with one
// wave per SIMD (64-row blocks need the whole register file) the exchange / expm1 / publish phases, ~540 vector,
// scalar and LDS instructions per step, run with the matrix cores idle (PMC: 46 % busy).  Here the phases of
// NEIGHBOURING steps share a barrier interval, so their vector work sits between matrix instructions:
//
//   interval k:   Gram(k+1)  ||  expm1 / sums / publish E(k)        -> write partial S'(k+1), E(k) -> ONE barrier
//                 gradient(k) (+ in-place refill with tile k+2)  ||  reduce the partials of S'(k+1)
//
// which needs two point tiles resident (the Gram runs one tile ahead of the gradient) and the exchange and E buffers
// double-buffered: a fast wave may write the buffers of interval k+1 while a slow one still reads those of interval k;
// the buffers of interval k+2 are only written behind barrier k+1, which the slow wave reaches after its reads.
// Point tiles, norms and their LDS-DMA refill stay wave-private as in the other kernels (features are sliced over
// the four waves both for the Gram and for the gradient), so the barrier orders the exchange alone.
//
// SPLIT = false: bf16 inputs, 32-point steps (layouts of mmd_bf16_kernel, RT = 4).
// SPLIT = true : fp32 inputs as scaled fp16 pairs hi + lo (mmd_common.h, split_scales), 16-point steps so that two
//   hi + lo tiles fit the LDS.  The Gram stays on 16x16x32 tiles (K = 32 features); the gradient, whose K is the
//   step's 16 points, runs on v_mfma_f32_32x32x16_f16: 32 features x 32 rows per instruction, three instructions
//   (X_hi E_hi + X_lo E_hi + X_hi E_lo) per output block, a quarter of the instruction count of 16x16 tiles.
#include <type_traits>

#include "mmd_common.h"

namespace tvae {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 h16x8_t;      // 8 x 16-bit container (bf16 or fp16 bits)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(4))) __bf16 b16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

template <bool F16>
__device__ __forceinline__ f32x4 mma(const h16x8_t& a, const h16x8_t& b, const f32x4& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int KB, bool SPLIT>
struct PipeLds {
    static constexpr int P = SPLIT ? 16 : 32;                 // points per step
    static constexpr int PT = P / 16;                         // 16-point tiles per step
    static constexpr int RT = 4;                              // 16-row tiles per workgroup
    static constexpr int NT = PT * RT;                        // S' tiles per step
    static constexpr int TPW = NT / 4;                        // ... finished per wave
    static constexpr int SW = 32 * KB;                        // features per wave
    // a wave's slice of a step: KB blocks of 32 features, each 2 KiB = two 1 KiB pieces of 16 points x 32 features
    // (bf16: the step's two point tiles; split: the hi image and the lo image of its one point tile)
    static constexpr int SLICE_B = KB * 2048;
    static constexpr int XBUF_B = 4 * SLICE_B;                // one point-tile buffer (four wave slices)
    static constexpr int NORM_OFF = 2 * XBUF_B;               // [4 buf][4 waves][64] float point norms (P used; the DMA moves 64)
    static constexpr int XCH_OFF = NORM_OFF + 4 * 4 * 64 * 4; // (two buffers in use, four in the Q-form instance's ring)
    static constexpr int XCH_ONE_B = NT * 3 * 64 * 16;        // [tile][3 remote sources][64 lanes] float4
    static constexpr int EBUF_OFF = XCH_OFF + 2 * XCH_ONE_B;
    static constexpr int EPIECE_B = SPLIT ? 64 * 32 : RT * 64 * 16;        // SPLIT: [row][16 points]; else [t][lane][2 pt][4]
    static constexpr int EBUF_ONE_B = (SPLIT ? 2 : 1) * EPIECE_B;          // SPLIT: hi piece, lo piece
    static constexpr int BYTES = EBUF_OFF + 2 * EBUF_ONE_B;
    static_assert(BYTES <= 160 * 1024, "pipelined tile kernel exceeds the 160 KiB LDS");
    static_assert(XCH_ONE_B >= (8 + 4 * TPW * 16) * 8, "final reduction reuses an exchange buffer");
};

#ifndef TVAE_PIPE_TD
#define TVAE_PIPE_TD 2
#endif
#ifndef TVAE_PIPE_TD_QF
#define TVAE_PIPE_TD_QF 2
#endif
#ifndef TVAE_PIPE_GD
#define TVAE_PIPE_GD 1
#endif

#ifndef TVAE_Q32
#define TVAE_Q32 1
#endif

template <int V> using ic = std::integral_constant<int, V>;

// QF (split only): the Q-form instance. The device-side switch QFlag (mmd.hip) selects ONE of the two instances of a
// launch pair; the other returns at once. A runtime branch around the extra gradient products instead cost 50 registers
// and spilled at KB >= 5.
template <int KB, bool SPLIT, bool QF = false>
__global__ __launch_bounds__(256, 1) void mmd_pipe_kernel(MmdArgs a) {
    static_assert(SPLIT || !QF, "Q-form is a split-mode variant");
    if constexpr (SPLIT) {
        if (a.qflag != nullptr && (__builtin_amdgcn_readfirstlane((int)*a.qflag) != 0) != QF) return;
    }
    constexpr bool qform = QF;
    using L = PipeLds<KB, SPLIT>;
    constexpr int P = L::P, PT = L::PT, RT = L::RT, TPW = L::TPW, SW = L::SW, DPAD = 4 * SW, NI = 2 * KB;
    constexpr int NE = 4 * TPW;                               // E values per lane and step
    // DMA instructions per 32-feature block. Q-form: nothing reads the lo image of the column tiles (the Gram takes the
    // column side hi-only, the one gradient product is X_hi Q_hi), so it is not fetched
    constexpr int DPS = QF ? 1 : 2;
    static_assert(!QF || TVAE_GRAM_PRODUCTS == 2, "the Q-form instance does not fetch the lo image the third Gram product reads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // the S' tiles this wave finishes: q = wave * TPW + j  ->  point tile q / RT, row tile q % RT
    const int my_pt = (wave * TPW) / RT, my_t0 = (wave * TPW) % RT;

    const int rem = (int)blockIdx.x;                          // gradient launch only: problem 0
    const int rb = a.sp_major ? rem % a.row_blocks : rem / a.col_splits;
    const int sp = a.sp_major ? rem / a.row_blocks : rem - rb * a.col_splits;
    const int i0 = rb * 16 * RT;
    const long long bid = (long long)a.sums_base + blockIdx.x;

    // ---- the 64 rows of this workgroup: this wave's feature slice, in registers as MFMA B operands ----
    const bf16* rows = reinterpret_cast<const bf16*>(a.cols[0]) + (long long)a.row_offset * DPAD;
    const bf16* rows_lo = SPLIT ? reinterpret_cast<const bf16*>(a.cols_lo[0]) + (long long)a.row_offset * DPAD : rows;
    const float* rnorm = a.cnorm[0] + a.row_offset;
    // The Gram walks the row tiles and point tiles in an order ROTATED by the wave's ownership, local tile (pt', t') =
    // global ((pt' + my_pt) % PT, (t' + my_t0) % RT): the S' tiles a wave finishes itself are then always its
    // accumulators S[0][0 .. TPW-1], and the exchange needs neither branches nor register selects by wave number.
    h16x8_t zreg[RT][KB], zlo[SPLIT ? RT : 1][SPLIT ? KB : 1];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = i0 + 16 * ((t + my_t0) % RT) + i16;
        const bf16* zp = rows + (long long)row * DPAD + wave * SW + 8 * g;
        const bf16* zl = rows_lo + (long long)row * DPAD + wave * SW + 8 * g;
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            if (row < a.n_rows) zreg[t][s] = *reinterpret_cast<const h16x8_t*>(zp + 32 * s);
            else zreg[t][s] = h16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (SPLIT) {
                if (row < a.n_rows) zlo[t][s] = *reinterpret_cast<const h16x8_t*>(zl + 32 * s);
                else zlo[t][s] = h16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
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
    // retire these loads before the loop (hipcc would otherwise wait vmcnt(0) at their first use, draining the DMA)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            asm volatile("" : "+v"(zreg[t][s]));
            if constexpr (SPLIT) asm volatile("" : "+v"(zlo[t][s]));
        }
#pragma unroll
    for (int j = 0; j < TPW; ++j) asm volatile("" : "+v"(rn[j]));
    const bool any_invalid = __any(!all_valid);

    // gradient accumulators.  bf16: acc[16-feature block][row tile], 16x16 tiles.  split: acc[32-feature block][32-row
    // block], 32x32 tiles (v_mfma_f32_32x32x16_f16: K = this step's 16 points)
    using AccT = std::conditional_t<SPLIT, f32x16, f32x4>;
    constexpr int AN0 = SPLIT ? KB : NI, AN1 = SPLIT ? 2 : RT;
    AccT acc[AN0][AN1];
#pragma unroll
    for (int x = 0; x < AN0; ++x)
#pragma unroll
        for (int y = 0; y < AN1; ++y)
#pragma unroll
            for (int r = 0; r < (SPLIT ? 16 : 4); ++r) acc[x][y][r] = 0.f;

    // steps of P points over both column sets: [0, nS) latents, [nS, 2 nS) priors
    const int nS = a.tiles_per_set / PT;
    const long long total = 2ll * nS;
    const int sb = (int)(sp * total / a.col_splits);
    const int se = (int)((sp + 1) * total / a.col_splits);
    const int nsteps = se - sb;

    // ---- per-lane constants of the LDS images (one 1 KiB piece = 16 points x 32 features, mmd_bf16.hip) ----
    const int dma_r = lane >> 2, dma_c = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const unsigned dma_voff = (unsigned)(dma_r * DPAD + 8 * dma_c) * 2u;
    // second piece of a block, 1 KiB further in LDS through the instruction offset (dma16_pair), which the hardware
    // adds to the global address too: the second point tile (bf16), or the lo copy of the same tile (split; the plan
    // keeps the distance between the copies below 2^32 bytes)
    const unsigned dma_voff_b = SPLIT ? dma_voff + (unsigned)(reinterpret_cast<const char*>(a.cols_lo[0]) - reinterpret_cast<const char*>(a.cols[0])) - 1024u
                                      : dma_voff + 16u * DPAD * 2u - 1024u;
    const int gram_off = 16 * (4 * i16 + (g ^ (2 * ((i16 >> 2) & 1))));  // + 1024 * piece
    const int tq = i16 >> 2, tp = i16 & 3;
    // transposed operand reads (ds_read_b64_tr_b16: the 16 lanes of a group fetch 4 points x 16 features and lane i
    // receives feature i of the four points).
    //   bf16, 16-feature block db (16x16x32: k = 8 g + .. over 32 points): piece pair db >> 1, base by the parity of db
    //   split, 32-feature block (32x32x16: feature = lane % 32, k = 8 (lane / 32) + 4 h + ..): [h]
    int tr_off[2];
    if constexpr (SPLIT) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            tr_off[h] = 64 * (8 * (g >> 1) + 4 * h + tq) + 16 * ((2 * (g & 1) + (tp >> 1)) ^ (2 * h)) + 8 * (tp & 1);
    } else {
        tr_off[0] = 16 * (4 * (4 * g + tq) + 2 * (g & 1) + (tp >> 1)) + 8 * (tp & 1);
        tr_off[1] = 16 * (4 * (4 * g + tq) + 2 * ((g & 1) ^ 1) + (tp >> 1)) + 8 * (tp & 1);
    }
    auto tr_read = [&](int byte_off) -> s16x4_t {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)(sm + byte_off));
    };

    struct Src { const bf16* x; const float* nrm; int set; int p0; };
    auto step_meta = [&](int v, int& set, int& p0) {
        const int u = v < se ? v : se - 1;                    // past the end: the last step again (results unused)
        set = u >= nS ? 1 : 0;
        p0 = (u - (set ? nS : 0)) * P;
    };
    // both sets' bases in scalar registers, picked by a scalar select (indexing a.cols[set] is a scalar memory load
    // per step, and hipcc then did the pointer arithmetic in vector registers: ~25 instructions per step)
    const bf16* const cbase0 = reinterpret_cast<const bf16*>(a.cols[0]) + wave * SW;
    const bf16* const cbase1 = reinterpret_cast<const bf16*>(a.cols[1]) + wave * SW;
    const float* const nbase0 = a.cnorm[0];
    const float* const nbase1 = a.cnorm[1];
    auto step_src = [&](int v) -> Src {
        int set, p0;
        step_meta(v, set, p0);
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(p0) * (unsigned)DPAD;   // the plan keeps n_pad * DPAD < 2^31
        return Src{(set ? cbase1 : cbase0) + off, (set ? nbase1 : nbase0) + p0, set, p0};
    };
    // Q-form: the lo pieces are free (DPS above), so the same memory holds FOUR hi-only tiles, tile v in slot v & 3 = piece
    // (v & 1) of buffer (v >> 1) & 1.  Tile k + 3 is fetched during interval k and first read in interval k + 2: a whole
    // interval more for the L2 / Infinity Cache round trip than the in-place refill of the two-buffer scheme leaves
    // (measured with every step fetching the same tile: the refill latency was 8 % of the kernel).
    constexpr bool RING = QF;
    auto xbuf_byte = [&](int b) -> unsigned {
        if constexpr (RING) return (unsigned)(((b >> 1) & 1) * L::XBUF_B + wave * L::SLICE_B + (b & 1) * 1024);
        else return (unsigned)(b * L::XBUF_B + wave * L::SLICE_B);
    };
    auto stage_norms = [&](const Src& X, int nbuf) {
        // all 64 lanes move a float (no lane mask, no branch): the words past the step's P norms are never read
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(L::NORM_OFF + (nbuf * 4 + wave) * 64 * 4));
        dma4(uniform_ptr(X.nrm), (unsigned)lane * 4u, dst);
    };
    // features [32 s, 32 s + 32) of the step into buffer b: two DMA instructions under one M0
    auto dma_block = [&](const Src& X, int b, int s) {
        if constexpr (QF) dma16(reinterpret_cast<const float*>(X.x + 32 * s), dma_voff, smem_byte + xbuf_byte(b) + 2048u * s);
        else dma16_pair(reinterpret_cast<const float*>(X.x + 32 * s), dma_voff, dma_voff_b, smem_byte + xbuf_byte(b) + 2048u * s);
    };

#ifdef TVAE_PHASE_TIMING
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    f32x4 S[PT][RT];      // partial Gram accumulators of the tile in flight, local (rotated) tile order
    // reduce-scatter of the partials through exchange buffer xbuf: global tile q = RT pt + t goes to wave q / TPW;
    // in the rotated local order this wave's own tiles are S[0][0 .. TPW-1] and stay in registers
    auto xch_write = [&](int xbuf) {
        char* xb = sm + L::XCH_OFF + xbuf * L::XCH_ONE_B + lane * 16;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if (pt == 0 && t < TPW) continue;
                const int q = RT * ((pt + my_pt) % PT) + (t + my_t0) % RT;          // wave-uniform scalar arithmetic
                const int owner = q / TPW;
                *reinterpret_cast<f32x4*>(xb + (q * 3 + ((wave - owner - 1) & 3)) * 1024) = S[pt][t];
            }
    };
    // finish the S' tiles of this wave, then t = |x - z|^2 / D^2 per element (register r <-> point p0 + 16 my_pt + 4 g + r,
    // row i16 of row tile my_t0 + j) and whether any lane leaves the polynomial's range
    float tt[NE];
    bool t_big = false;
    f32x4 part[TPW][3], cn;
    auto xch_fetch = [&](int xbuf, int nbuf) {
        const f32x4* xb = reinterpret_cast<const f32x4*>(sm + L::XCH_OFF + xbuf * L::XCH_ONE_B);
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                part[j][w] = xb[((wave * TPW + j) * 3 + w) * 64 + lane];     // all reads in flight
            }
        cn = *reinterpret_cast<const f32x4*>(sm + L::NORM_OFF + ((nbuf * 4 + wave) * 64 + 16 * my_pt + 4 * g) * 4);
    };
    auto xch_finish = [&]() {
        float tm = 0.f;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {        // own partial (rotated order: still in registers) + the three others, fixed order
            const f32x4 sv = ((S[0][j] + part[j][0]) + part[j][1]) + part[j][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tt[4 * j + r] = fmaxf((cn[r] + rn[j] - gram_unscale2 * sv[r]) * a.inv_d2, 0.f);
                tm = fmaxf(tm, tt[4 * j + r]);
            }
        }
        t_big = __any(tm >= EM1_POLY_MAX);
    };
    auto xch_reduce = [&](int xbuf, int nbuf) { xch_fetch(xbuf, nbuf); xch_finish(); };

    double sum0 = 0.0, sum1 = 0.0, rowsum[TPW];     // per-lane fp64 partials of the Q sums (by column set) and the row sums
#pragma unroll
    for (int j = 0; j < TPW; ++j) rowsum[j] = 0.0;

    // k-step 0 operands of the NEXT interval's Gram, read at the end of this one (their LDS round trip would otherwise
    // open every interval): block 0 of the tile in buffer b has landed once all but the DPS (KB - 1) youngest DMA
    // instructions have
    h16x8_t nab[PT], nal[SPLIT ? PT : 1];
    auto gram_prefetch = [&](int b) {
        const char* xs = sm + xbuf_byte(b);
        wait_vmcnt_n(DPS * (KB - 1) + (RING ? 1 + DPS * KB : 0));      // ring: the norms and blocks of the tile after it
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            if constexpr (SPLIT) {
                nab[pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off);
                nal[pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 1024);
            } else {
                nab[pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 1024 * ((pt + my_pt) & (PT - 1)));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // First half of an interval: partial Gram of the tile in buffer b over this wave's feature slice, S[pt'][t'], with
    // the E work of the PREVIOUS tile (set x_set, first point x_p0: from tt, pieces into E buffer ebuf) cut into KB
    // slices that follow the matrix instructions of the KB k-steps.  The counted DMA waits in front of the operand
    // reads are inline asm, which hipcc does not schedule across: each k-step is a scheduling region of its own, and
    // vector work only lands between matrix instructions it shares a region with.  E = expm1(-t) as straight-line
    // code: the polynomial alone in the common instantiation; SLOW adds the exp form for t >= 1/8 and the masks of
    // ragged edges, and a wave-uniform branch picks it per step.  `slack`: DMA instructions issued after this tile's
    // that may still be in flight; WITH_E = false: the prologue's Gram.
    auto front = [&](auto slow_tag, auto with_e_tag, int b, int slack, int x_set, int x_p0, int ebuf) {
        constexpr bool SLOW = decltype(slow_tag)::value, WITH_E = decltype(with_e_tag)::value;
        constexpr bool PRE = WITH_E;                          // the loop's intervals: k-step 0 was read ahead (gram_prefetch)
        const char* xs = sm + xbuf_byte(b);
        constexpr int GD = KB < TVAE_PIPE_GD ? KB : TVAE_PIPE_GD;      // operand reads run GD k-steps ahead of their MFMAs
        h16x8_t ab[KB][PT], al[SPLIT ? KB : 1][PT];
        auto rd = [&](int sr) {
            if (PRE && sr == 0) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) { ab[0][pt] = nab[pt]; if constexpr (SPLIT) al[0][pt] = nal[pt]; }
                return;
            }
            wait_vmcnt_n(DPS * (KB - 1 - sr) + slack);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                if constexpr (SPLIT) {
                    ab[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 2048 * sr);
                    al[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 2048 * sr + 1024);
                } else {
                    const int piece = PT * sr + ((pt + my_pt) & (PT - 1));                   // local -> global point tile
                    ab[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 1024 * piece);
                }
            }
            // keep the reads HERE, ahead of the matrix instructions of the k-steps before sr: left alone, hipcc sinks
            // them to their first use and every k-step then waits out an LDS round trip (measured: 1 950 cycles for the
            // 1 152 cycles of matrix work of a split Gram)
            __builtin_amdgcn_sched_barrier(0);
        };
        f32x4 e[TPW], q[TPW];
        auto e_slice = [&](int s) {
            // elements el = 4 j + r spread over slices 0 .. FIN; sums, conversion and publication at the end of slice FIN,
            // with the matrix instructions of the last k-step still to come behind them
            constexpr int FIN = KB > 1 ? KB - 2 : 0;
#pragma unroll
            for (int el = 0; el < NE; ++el) {
                if (el * (FIN + 1) / NE != s) continue;
                const int j = el >> 2, r = el & 3;
                const EQ pl = em1_split(tt[el]);                                  // t < 1/8
                float ev = pl.e, qv = pl.q;
                if constexpr (SLOW) {
                    const float ex = __expf(-tt[el]) - 1.f;                       // beyond: |E| >= 0.1175, 1e-6 relative
                    const bool big = tt[el] >= EM1_POLY_MAX;
                    const bool ok = my_valid[j] && (x_p0 + 16 * my_pt + 4 * g + r < a.n_cols);
                    ev = ok ? (big ? ex : ev) : 0.f;
                    qv = ok ? (big ? ex + tt[el] : qv) : 0.f;
                }
                e[j][r] = qform ? qv : ev;      // Q-form (mmd.hip, QFlag): gradient and row sums carry Q = E + t
                q[j][r] = qv;
            }
            if (s != FIN) return;
            // sums carry Q (linear part: closed form in the finalize kernel); row sums carry E
            float qs_all = 0.f;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                qs_all += (q[j][0] + q[j][1]) + (q[j][2] + q[j][3]);
                const float es = (e[j][0] + e[j][1]) + (e[j][2] + e[j][3]);
                rowsum[j] += (double)(x_set ? -es : es);
            }
            sum0 += (double)(x_set ? 0.f : qs_all);
            sum1 += (double)(x_set ? qs_all : 0.f);
            // prior columns enter the gradient with s_p = -1: flip the sign bits of the packed values
            const unsigned flip = x_set ? 0x80008000u : 0u;
            char* eb = sm + L::EBUF_OFF + ebuf * L::EBUF_ONE_B;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if constexpr (SPLIT) {
                    f16x4_t p0v, p1v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float es = e[j][r] * SPLIT_E_SCALE;
                        p0v[r] = (_Float16)es;
                        if constexpr (!qform) p1v[r] = (_Float16)(es - (float)p0v[r]);
                    }
                    uint2 b0 = __builtin_bit_cast(uint2, p0v), b1 = {0u, 0u};
                    if constexpr (!qform) b1 = __builtin_bit_cast(uint2, p1v);
                    b0.x ^= flip; b0.y ^= flip; b1.x ^= flip; b1.y ^= flip;
                    // [piece][row][16 points], the two 16-byte halves of a row swapped on rows 8 .. 15 of a tile (bank
                    // spread of the consumer's reads): this lane holds points 4 g .. 4 g + 3 of row 16 (my_t0 + j) + i16
                    const int at = ((my_t0 + j) * 16 + i16) * 32 + 16 * ((g >> 1) ^ ((i16 >> 3) & 1)) + 8 * (g & 1);
                    *reinterpret_cast<uint2*>(eb + at) = b0;
                    if constexpr (!qform) *reinterpret_cast<uint2*>(eb + L::EPIECE_B + at) = b1;
                } else {
                    b16x4_t ev;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ev[r] = (__bf16)e[j][r];
                    uint2 bits = __builtin_bit_cast(uint2, ev);
                    bits.x ^= flip; bits.y ^= flip;
                    reinterpret_cast<uint2*>(eb)[((my_t0 + j) * 64 + lane) * 2 + my_pt] = bits;
                }
            }
        };
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < GD; ++s) rd(s);
#ifdef TVAE_PHASE_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PHASE_MARK(5);
#endif
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            if (s + GD < KB) rd(s + GD);
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(ab[s][pt], zreg[t][s], s == 0 ? zero : S[pt][t]);
            if constexpr (SPLIT) {
#if TVAE_GRAM_PRODUCTS >= 2
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(ab[s][pt], zlo[t][s], S[pt][t]);
#endif
#if TVAE_GRAM_PRODUCTS == 3
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(al[s][pt], zreg[t][s], S[pt][t]);
#endif
            }
            if constexpr (WITH_E) e_slice(s);
        }
    };

    // gradient of the tile in buffer b with the E pieces of ebuf: acc^T[d][i] += X^T[d][p] (s_p E[p][i]); a 32-feature
    // block of the buffer is dead after its MFMAs -> refilled in place with the same block of step Xn.
    // grad_fetch issues the first operand reads (in program order ahead of the exchange reads, so that the matrix
    // instructions of block 0 can start while the next tile's partials are still being summed); grad_run does the rest.
    constexpr int GB = SPLIT ? KB : NI;                       // feature blocks of the gradient (32 / 16 features)
    // transposed operand reads run TD blocks ahead (Q-form: two matrix instructions per block instead of six, so the same
    // LDS latency needs more blocks of cover)
    constexpr int TDW = QF ? TVAE_PIPE_TD_QF : TVAE_PIPE_TD;
    constexpr int TD = GB < TDW ? GB : TDW;
    h16x8_t Bh[SPLIT ? 2 : RT], Bl[SPLIT ? 2 : 1];
    s16x4_t gh[GB][2], gl[SPLIT ? GB : 1][2];
    auto grad_rd = [&](int xb0, int blk) {
        if constexpr (SPLIT) {
            const int at = xb0 + 2048 * blk;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gh[blk][h] = tr_read(at + tr_off[h]);
                if constexpr (!qform) gl[blk][h] = tr_read(at + 1024 + tr_off[h]);
            }
        } else {
            const int at = xb0 + 2048 * (blk >> 1);
            gh[blk][0] = tr_read(at + tr_off[blk & 1]);
            gh[blk][1] = tr_read(at + 1024 + tr_off[blk & 1]);
        }
    };
    auto grad_fetch_x = [&](int b) {          // wave-private tile data: legal ahead of the barrier
#pragma unroll
        for (int blk = 0; blk < TD && blk < GB; ++blk) grad_rd((int)xbuf_byte(b), blk);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto grad_fetch_e = [&](int ebuf) {
        const char* eb = sm + L::EBUF_OFF + ebuf * L::EBUF_ONE_B;
        if constexpr (SPLIT) {
            // B[k = point][n = row]: lane holds points 8 (lane / 32) .. + 7 of row 32 rbk + lane % 32
#pragma unroll
            for (int rbk = 0; rbk < 2; ++rbk) {
                const int n = lane & 31;
                const int at = (32 * rbk + n) * 32 + 16 * ((lane >> 5) ^ ((n >> 3) & 1));
                Bh[rbk] = *reinterpret_cast<const h16x8_t*>(eb + at);
                if constexpr (!qform) Bl[rbk] = *reinterpret_cast<const h16x8_t*>(eb + L::EPIECE_B + at);
            }
        } else {
#pragma unroll
            for (int t = 0; t < RT; ++t) Bh[t] = *reinterpret_cast<const h16x8_t*>(eb + (t * 64 + lane) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto grad_run = [&](int b, int fill, const Src& Xn) {
        const int xb0 = (int)xbuf_byte(b);
#pragma unroll
        for (int blk = 0; blk < GB; ++blk) {
            if (blk + TD < GB) grad_rd(xb0, blk + TD);
            if constexpr (SPLIT) {
                const f16x8_t Ah = __builtin_bit_cast(f16x8_t, __builtin_shufflevector(gh[blk][0], gh[blk][1], 0, 1, 2, 3, 4, 5, 6, 7));
                // X_hi E_hi + X_lo E_hi + X_hi E_lo, the two row blocks alternating
#pragma unroll
                for (int rbk = 0; rbk < 2; ++rbk)
                    acc[blk][rbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8_t, Bh[rbk]), acc[blk][rbk], 0, 0, 0);
                if constexpr (!qform) {          // Q is ~1e-3 of E: its sums need one product
                    const f16x8_t Al = __builtin_bit_cast(f16x8_t, __builtin_shufflevector(gl[blk][0], gl[blk][1], 0, 1, 2, 3, 4, 5, 6, 7));
#if TVAE_GRAD_PRODUCTS == 3
#pragma unroll
                    for (int rbk = 0; rbk < 2; ++rbk)
                        acc[blk][rbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(f16x8_t, Bh[rbk]), acc[blk][rbk], 0, 0, 0);
#endif
#pragma unroll
                    for (int rbk = 0; rbk < 2; ++rbk)
                        acc[blk][rbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8_t, Bl[rbk]), acc[blk][rbk], 0, 0, 0);
                }
            } else {
                const h16x8_t A1 = __builtin_bit_cast(h16x8_t, __builtin_shufflevector(gh[blk][0], gh[blk][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[blk][t] = mma<false>(A1, Bh[t], acc[blk][t]);
            }
            if (SPLIT || (blk & 1)) {
                // the MFMAs of the 32-feature block are issued (their operand reads have completed) before the refill
                __builtin_amdgcn_sched_barrier(0);
                dma_block(Xn, fill, SPLIT ? blk : blk >> 1);
            }
        }
    };

    Src Xn2_c{};
    int c_set_c = 0, c_p0_c = 0;
    bool slow_c = true;
    // one interval; the buffer indices are compile-time constants (the loop below is unrolled by two)
    auto interval = [&](int k, auto cur_tag) {
        const int cur = k & 1, nxt = cur ^ 1;
        // point-tile slots: this tile's, the next one's (Gram), the one to fill; norm buffers go with the slots
        const int xs_cur = RING ? (k & 3) : cur, xs_nxt = RING ? ((k + 1) & 3) : nxt, xs_fill = RING ? ((k + 3) & 3) : cur;
        constexpr int FSLACK = RING ? 1 + DPS * KB : 0;
        // this interval's scalars (c_set, c_p0, slow, Xn2) were worked out at the end of the previous one, behind its
        // last matrix instructions, instead of at the top of this one with the matrix pipe idle
        const Src Xn2 = Xn2_c;
        const int c_set = c_set_c, c_p0 = c_p0_c;
        // Gram of the next tile  ||  E of this one (from tt, left by the previous interval)
        PHASE_MARK(5);
        if (slow_c) front(std::true_type{}, std::true_type{}, xs_nxt, FSLACK, c_set, c_p0, cur);   // wave-uniform
        else front(std::false_type{}, std::true_type{}, xs_nxt, FSLACK, c_set, c_p0, cur);
        PHASE_MARK(0);
        // the barrier needs the partials' WRITES retired, not the gradient's operand reads: LDS operations retire in
        // order, so with the reads issued behind the writes a counted wait leaves them in flight across the barrier
        xch_write(nxt);
        asm volatile("" ::: "memory");          // the reads below stay behind the writes (the count depends on it)
        __builtin_amdgcn_sched_barrier(0);
        grad_fetch_x(xs_cur);
        {
            constexpr int NRD = (GB < TDW ? GB : TDW) * (SPLIT ? (QF ? 2 : 4) : 2);      // grad_fetch_x's reads
            static_assert(NRD <= 15, "lgkmcnt is a 4-bit counter");
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
        }
        PHASE_MARK(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PHASE_MARK(2);
        // gradient of this tile (+ refill with tile k + 2)  ||  reduce of the next tile's partials
        grad_fetch_e(cur);
        stage_norms(Xn2, xs_fill);
        // the partials' LDS round trip and the sums run behind the gradient's matrix instructions
        xch_fetch(nxt, xs_nxt);
        __builtin_amdgcn_sched_barrier(0);
        PHASE_MARK(3);
        grad_run(xs_cur, xs_fill, Xn2);
        xch_finish();
        step_meta(sb + k + 1, c_set_c, c_p0_c);
        Xn2_c = step_src(sb + k + (RING ? 4 : 3));
        slow_c = any_invalid || t_big || c_p0_c + P > a.n_cols;
        // the next interval's Gram: the tile just refilled into this buffer, or (ring) the slot after the next
        gram_prefetch(RING ? ((k + 2) & 3) : cur);
        PHASE_MARK(4);
    };

    if (nsteps > 0) {
        // ---- prologue: tiles sb, sb + 1 in flight; Gram and exchange of the first one ----
        {
            const Src X0 = step_src(sb), X1 = step_src(sb + 1);
            stage_norms(X0, 0);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X0, 0, s);
            stage_norms(X1, 1);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X1, 1, s);
            if constexpr (RING) {
                const Src X2 = step_src(sb + 2);
                stage_norms(X2, 2);
#pragma unroll
                for (int s = 0; s < KB; ++s) dma_block(X2, 2, s);
            }
            front(std::false_type{}, std::false_type{}, 0, (RING ? 2 : 1) * (1 + DPS * KB), 0, 0, 0);
            xch_write(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            xch_reduce(0, 0);
            gram_prefetch(1);
            step_meta(sb, c_set_c, c_p0_c);
            Xn2_c = step_src(sb + (RING ? 3 : 2));
            slow_c = any_invalid || t_big || c_p0_c + P > a.n_cols;
        }
        for (int k = 0; k < nsteps; ++k) interval(k, ic<0>{});
    }
    wait_vmcnt<0>();
#ifdef TVAE_PHASE_TIMING
    if (blockIdx.x == 0 && lane == 0 && nsteps > 0)
        printf("wave %d steps %d: front %llu  xch write %llu  barrier %llu  fetch+reduce %llu  grad %llu  top %llu (cycles per step)\n", wave,
               nsteps, ph[0] / nsteps, ph[1] / nsteps, ph[2] / nsteps, ph[3] / nsteps, ph[4] / nsteps, ph[5] / nsteps);
#endif

    // ---- per-workgroup results: each wave holds the sums of ITS tiles; add the four in a fixed order ----
    __syncthreads();
    double* red = reinterpret_cast<double*>(sm + L::XCH_OFF);          // [4][2] sums, then [4 TPW][16] row sums
    const double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    if (lane == 0) { red[wave * 2 + 0] = w0; red[wave * 2 + 1] = w1; }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        double rv = rowsum[j];
        rv += __shfl_xor(rv, 16, 64);
        rv += __shfl_xor(rv, 32, 64);
        if (g == 0) red[8 + (wave * TPW + j) * 16 + i16] = rv;
    }
    __syncthreads();
    if (tid == 0) {
        a.wg_sums[2 * bid + 0] = (red[0] + red[2]) + (red[4] + red[6]);
        a.wg_sums[2 * bid + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
    if (tid < 16 * RT) {      // row 16 t + i: the S' tiles q = RT pt + t of every point tile, in a fixed order
        const int t = tid >> 4, i = tid & 15;
        double r = 0.0;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) r += red[8 + (RT * pt + t) * 16 + i];
        a.wg_rowsum[(long long)sp * a.n_rows_pad + i0 + tid] = r;
    }
    if constexpr (SPLIT) {
        // 32x32 tiles: register r of a lane is feature 8 (r / 4) + 4 (lane / 32) + r % 4 of row lane % 32
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk) {
            float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 32 * rbk + (lane & 31)) * DPAD + wave * SW + 4 * (lane >> 5);
#pragma unroll
            for (int s = 0; s < KB; ++s)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *reinterpret_cast<f32x4*>(grow + 32 * s + 8 * q4) =
                        f32x4{acc[s][rbk][4 * q4], acc[s][rbk][4 * q4 + 1], acc[s][rbk][4 * q4 + 2], acc[s][rbk][4 * q4 + 3]};
        }
    } else {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 16 * t + i16) * DPAD + wave * SW + 4 * g;
#pragma unroll
            for (int db = 0; db < NI; ++db) *reinterpret_cast<f32x4*>(grow + 16 * db) = acc[db][t];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Sum-only companion: the y-rows launch of a gradient call (y-y, each unordered pair of 64-row blocks once when the
// row range allows) and both problems of a call without gradient.  Same 64-row blocks, K-sliced Gram and exchange;
// with no gradient phase an interval is
//
//     Gram(k+1)  ||  sum the partials of tile k, expm1, sums          -> write partial S'(k+1) -> ONE barrier
//
// and a feature block of the tile under the Gram is refilled right behind its matrix instructions with tile k + 3
// (two tile buffers: k + 1 under the Gram, k + 2 landed or landing).  Point norms sit in a ring of four.
// The exp form and the ragged-edge masks sit in a fix-up behind a wave-uniform test at the end of a step's E work.
// ------------------------------------------------------------------------------------------------------------------
template <int KB, bool SPLIT>
__global__ __launch_bounds__(256, 1) void mmd_pipe_sum_kernel(MmdArgs a) {
    // split mode with the device flag on: the one-product kernel of the launch pair does the work (mmd_q32.hip)
    if constexpr (SPLIT && TVAE_Q32) {
        if (a.qflag != nullptr && __builtin_amdgcn_readfirstlane((int)*a.qflag) != 0) return;
    }
    // With the two-product Gram (mmd_common.h) a sum-only launch never reads the lo copy of the columns, so split mode
    // takes the bf16 instantiation's LDS layout and 32-point steps (two point tiles of the hi copy per 2 KiB block)
    // with fp16 arithmetic and the row side's second product: half the barriers, exchanges and scalar bookkeeping per
    // point.  LS: the hi | lo piece layout of the 16-point split steps (three-product builds only).
    constexpr bool LS = SPLIT && TVAE_GRAM_PRODUCTS == 3;
    using L = PipeLds<KB, LS>;
    constexpr int P = L::P, PT = L::PT, RT = L::RT, TPW = L::TPW, SW = L::SW, DPAD = 4 * SW;
    constexpr int NE = 4 * TPW, DPS = 2, SPB = 64 / P;       // SPB: steps per 64-point block
    constexpr int NORM4_OFF = L::EBUF_OFF;                    // [4 ring][4 waves][64] floats, in the (unused) E buffers
    static_assert(2 * L::EBUF_ONE_B >= 4 * 4 * 64 * 4, "norm ring does not fit the E buffers");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int my_pt = (wave * TPW) / RT, my_t0 = (wave * TPW) % RT;

    const int per_problem = a.row_blocks * a.col_splits;
    const int problem = a.problem_base + (int)blockIdx.x / per_problem;
    const int rem = (int)blockIdx.x - (problem - a.problem_base) * per_problem;
    const int rb = a.sp_major ? rem % a.row_blocks : rem / a.col_splits;
    const int sp = a.sp_major ? rem / a.row_blocks : rem - rb * a.col_splits;
    const int i0 = rb * 16 * RT;
    const long long bid = (long long)a.sums_base + blockIdx.x;

    const bf16* rows = reinterpret_cast<const bf16*>(a.cols[problem]) + (long long)a.row_offset * DPAD;
    const bf16* rows_lo = SPLIT ? reinterpret_cast<const bf16*>(a.cols_lo[problem]) + (long long)a.row_offset * DPAD : rows;
    const float* rnorm = a.cnorm[problem] + a.row_offset;
    h16x8_t zreg[RT][KB], zlo[SPLIT ? RT : 1][SPLIT ? KB : 1];            // rotated tile order, see mmd_pipe_kernel
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = i0 + 16 * ((t + my_t0) % RT) + i16;
        const bf16* zp = rows + (long long)row * DPAD + wave * SW + 8 * g;
        const bf16* zl = rows_lo + (long long)row * DPAD + wave * SW + 8 * g;
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            if (row < a.n_rows) zreg[t][s] = *reinterpret_cast<const h16x8_t*>(zp + 32 * s);
            else zreg[t][s] = h16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (SPLIT) {
                if (row < a.n_rows) zlo[t][s] = *reinterpret_cast<const h16x8_t*>(zl + 32 * s);
                else zlo[t][s] = h16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    const float gram_unscale2 = SPLIT ? 2.f * split_scales(*a.split_amax).inv_s2 : 2.f;
    bool my_valid[TPW];
    float rn[TPW];
    bool all_valid_s = true;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int my_row = i0 + 16 * (my_t0 + j) + i16;
        my_valid[j] = my_row < a.n_rows;
        all_valid_s = all_valid_s && my_valid[j];
        rn[j] = my_valid[j] ? rnorm[my_row] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            asm volatile("" : "+v"(zreg[t][s]));
            if constexpr (SPLIT) asm volatile("" : "+v"(zlo[t][s]));
        }
#pragma unroll
    for (int j = 0; j < TPW; ++j) asm volatile("" : "+v"(rn[j]));
    const bool any_invalid = __any(!all_valid_s);

    // steps of P points.  Problem 0 (z rows): both column sets.  Problem 1 (y rows): the priors -- all of them, or (sym)
    // each unordered pair of 64-point blocks once, pairs other than the diagonal weighted 2: row block I takes the
    // column blocks J < I of the parity of I + 1, J = I, and J > I of the parity of I (mmd.hip).  Split sp owns the
    // column blocks [sp nJ / cs, (sp + 1) nJ / cs) and a workgroup walks those of ITS blocks J that fall in the range:
    // the workgroups of one split (neighbours in launch order under sp_major) then stream the same ~1 / cs of the
    // priors, which L2 and the Infinity Cache hold, instead of 3 840 different halves of all of them (cfg4 size:
    // 1.1e12 B per launch at the fabric, the kernel ran at its pace).
    const int nS = a.tiles_per_set / PT;
    const int nJ = nS / SPB;
    const bool sym = problem == 1 && a.sym;
    const int blkI = (a.row_offset >> 6) + rb;
    const int jlo = (int)((long long)sp * nJ / a.col_splits), jhi = (int)((long long)(sp + 1) * nJ / a.col_splits);
    const int aend = blkI < jhi ? blkI : jhi;                                  // part A: J in [jlo, aend), parity of I + 1
    const int ja0 = jlo + ((jlo ^ (blkI + 1)) & 1);
    const int nA = aend > ja0 ? (aend - ja0 + 1) >> 1 : 0;
    const int nB = (blkI >= jlo && blkI < jhi) ? 1 : 0;                        // part B: the diagonal block
    const int cbeg = blkI + 1 > jlo ? blkI + 1 : jlo;                          // part C: J in [cbeg, jhi), parity of I
    const int jc0 = cbeg + ((cbeg ^ blkI) & 1);
    const int nC = jhi > jc0 ? (jhi - jc0 + 1) >> 1 : 0;
    const long long total = problem == 0 ? 2ll * nS : (long long)nS;
    const long long base = problem == 0 ? 0 : nS;
    const int sb = sym ? 0 : (int)(base + sp * total / a.col_splits);
    const int se = sym ? (nA + nB + nC) * SPB : (int)(base + (sp + 1) * total / a.col_splits);
    const int nsteps = se - sb;

    const int dma_r = lane >> 2, dma_c = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const unsigned dma_voff = (unsigned)(dma_r * DPAD + 8 * dma_c) * 2u;
    const unsigned dma_voff_b = LS ? dma_voff + (unsigned)(reinterpret_cast<const char*>(a.cols_lo[0]) - reinterpret_cast<const char*>(a.cols[0])) - 1024u
                                      : dma_voff + 16u * DPAD * 2u - 1024u;
    const int gram_off = 16 * (4 * i16 + (g ^ (2 * ((i16 >> 2) & 1))));

    struct Meta { int set, p0; float w; };
    auto step_meta = [&](int v) -> Meta {
        const int u = v < se ? v : se - 1;
        if (sym) {
            const int blk = u / SPB, sub = u - blk * SPB;
            const int J = blk < nA ? ja0 + 2 * blk : (blk < nA + nB ? blkI : jc0 + 2 * (blk - nA - nB));
            return Meta{1, J * 64 + sub * P, J != blkI ? 2.f : 1.f};
        }
        const int set = u >= nS ? 1 : 0;
        return Meta{set, (u - (set ? nS : 0)) * P, 1.f};
    };
    auto xbuf_byte = [&](int b) -> unsigned { return (unsigned)(b * L::XBUF_B + wave * L::SLICE_B); };
    // all DMA of one step: its norms into ring slot `ring`; its feature blocks one by one (dma_block)
    struct Src { const bf16* x; const float* nrm; };
    const bf16* const cbase0 = reinterpret_cast<const bf16*>(a.cols[0]) + wave * SW;      // scalar bases: see mmd_pipe_kernel
    const bf16* const cbase1 = reinterpret_cast<const bf16*>(a.cols[1]) + wave * SW;
    const float* const nbase0 = a.cnorm[0];
    const float* const nbase1 = a.cnorm[1];
    auto step_src = [&](int v) -> Src {
        const Meta m = step_meta(v);
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(m.p0) * (unsigned)DPAD;
        return Src{(m.set ? cbase1 : cbase0) + off, (m.set ? nbase1 : nbase0) + m.p0};
    };
    auto stage_norms = [&](const Src& X, int ring) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(NORM4_OFF + (ring * 4 + wave) * 64 * 4));
        dma4(uniform_ptr(X.nrm), (unsigned)lane * 4u, dst);
    };
    auto dma_block = [&](const Src& X, int b, int s) {
        dma16_pair(reinterpret_cast<const float*>(X.x + 32 * s), dma_voff, dma_voff_b, smem_byte + xbuf_byte(b) + 2048u * s);
    };

    f32x4 S[PT][RT], own[TPW];
    auto xch_write = [&](int xbuf) {
        char* xb = sm + L::XCH_OFF + xbuf * L::XCH_ONE_B + lane * 16;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if (pt == 0 && t < TPW) { own[t] = S[0][t]; continue; }
                const int q = RT * ((pt + my_pt) % PT) + (t + my_t0) % RT;
                const int owner = q / TPW;
                *reinterpret_cast<f32x4*>(xb + (q * 3 + ((wave - owner - 1) & 3)) * 1024) = S[pt][t];
            }
    };
    double sum0 = 0.0, sum1 = 0.0;

    // One interval.  Tile in buffer b: partial Gram into S, each feature block refilled with step Xr behind its matrix
    // instructions (norms of Xr into ring slot ring_r).  WITH_E: the tile whose partials sit in exchange buffer xbuf
    // (own partial in `own`, norms in ring slot ring_e, meta m) is finished between them.
    // DMA in flight on entry: the 1 + DPS KB instructions of the tile in the OTHER buffer; everything older has to land.
    // (gram_prefetch: the k-step 0 operands of the next interval's Gram, read ahead of the barrier -- the tile is
    // wave-private and has landed; measured neutral, 2.9 -> 3.1 ms at N = 32 768, kept for symmetry with the gradient kernel)
    h16x8_t nab[PT], nal[LS ? PT : 1];
    auto gram_prefetch = [&](int b) {
        const char* xs = sm + xbuf_byte(b);
        wait_vmcnt_n(1 + DPS * KB);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            if constexpr (LS) {
                nab[pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off);
                nal[pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 1024);
            } else {
                nab[pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 1024 * ((pt + my_pt) & (PT - 1)));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto front = [&](auto with_e_tag, int b, const Src& Xr, int ring_r, int xbuf, int ring_e, const Meta& m) {
        constexpr bool WITH_E = decltype(with_e_tag)::value;
        constexpr bool PRE = WITH_E;                          // the loop's intervals: k-step 0 was read ahead
        const char* xs = sm + xbuf_byte(b);
        f32x4 part[TPW][3], cn;
        if constexpr (WITH_E) {
            const f32x4* xb = reinterpret_cast<const f32x4*>(sm + L::XCH_OFF + xbuf * L::XCH_ONE_B);
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int w = 0; w < 3; ++w) part[j][w] = xb[((wave * TPW + j) * 3 + w) * 64 + lane];
            cn = *reinterpret_cast<const f32x4*>(sm + NORM4_OFF + ((ring_e * 4 + wave) * 64 + 16 * my_pt + 4 * g) * 4);
        }
        constexpr int GD = KB < TVAE_PIPE_GD ? KB : TVAE_PIPE_GD;
        h16x8_t ab[KB][PT], al[LS ? KB : 1][PT];
        auto rd = [&](int sr) {
            if (PRE && sr == 0) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) { ab[0][pt] = nab[pt]; if constexpr (LS) al[0][pt] = nal[pt]; }
                return;
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                if constexpr (LS) {
                    ab[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 2048 * sr);
                    al[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 2048 * sr + 1024);
                } else {
                    const int piece = PT * sr + ((pt + my_pt) & (PT - 1));
                    ab[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 1024 * piece);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // hold the reads ahead of the previous k-steps' matrix instructions
        };
        float tt[NE], q[NE];
        auto e_slice = [&](int s) {
            constexpr int FIN = KB > 1 ? KB - 2 : 0;
            if (s == 0) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const f32x4 sv = ((own[j] + part[j][0]) + part[j][1]) + part[j][2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tt[4 * j + r] = fmaxf((cn[r] + rn[j] - gram_unscale2 * sv[r]) * a.inv_d2, 0.f);
                }
            }
            // the polynomial alone between the matrix instructions (every instruction issued here adds to the step:
            // DESIGN 4.1f); the exp form for t >= 1/8 and the masks of ragged edges in a fix-up behind a wave-uniform
            // test at the end, which almost no step takes
#pragma unroll
            for (int el = 0; el < NE; ++el) {
                if (el * (FIN + 1) / NE != s) continue;
                q[el] = em1_split(tt[el]).q;
            }
            if (s != FIN) return;
            float tm = tt[0];
#pragma unroll
            for (int el = 1; el < NE; ++el) tm = fmaxf(tm, tt[el]);
            if (any_invalid || m.p0 + P > a.n_cols || __any(tm >= EM1_POLY_MAX)) {
#pragma unroll
                for (int el = 0; el < NE; ++el) {
                    const int j = el >> 2, r = el & 3;
                    const float ex = __expf(-tt[el]) - 1.f;
                    const bool big = tt[el] >= EM1_POLY_MAX;
                    const bool ok = my_valid[j] && (m.p0 + 16 * my_pt + 4 * g + r < a.n_cols);
                    q[el] = ok ? (big ? ex + tt[el] : q[el]) : 0.f;
                }
            }
            float qs = 0.f;
#pragma unroll
            for (int j = 0; j < TPW; ++j) qs += (q[4 * j] + q[4 * j + 1]) + (q[4 * j + 2] + q[4 * j + 3]);
            sum0 += (double)(m.set ? 0.f : qs);
            sum1 += (double)(m.set ? m.w * qs : 0.f);
        };
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (!PRE) wait_vmcnt_n(1 + DPS * KB);
#pragma unroll
        for (int s = 0; s < GD; ++s) rd(s);
        stage_norms(Xr, ring_r);
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            if (s + GD < KB) rd(s + GD);
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(ab[s][pt], zreg[t][s], s == 0 ? zero : S[pt][t]);
            if constexpr (SPLIT) {
#if TVAE_GRAM_PRODUCTS >= 2
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(ab[s][pt], zlo[t][s], S[pt][t]);
#endif
#if TVAE_GRAM_PRODUCTS == 3
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(al[s][pt], zreg[t][s], S[pt][t]);
#endif
            }
            if constexpr (WITH_E) e_slice(s);
            // the MFMAs of block s are issued (their operand reads have completed) before its refill
            __builtin_amdgcn_sched_barrier(0);
            dma_block(Xr, b, s);
        }
    };

    if (nsteps > 0) {
        {
            const Src X0 = step_src(sb), X1 = step_src(sb + 1), X2 = step_src(sb + 2);
            stage_norms(X0, 0);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X0, 0, s);
            stage_norms(X1, 1);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X1, 1, s);
            front(std::false_type{}, 0, X2, 2, 0, 0, Meta{0, 0, 1.f});
            gram_prefetch(1);
            xch_write(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        Meta m = step_meta(sb);
        Src Xr = step_src(sb + 3);
        for (int k = 0; k < nsteps; ++k) {
            const int cur = k & 1, nxt = cur ^ 1;
            front(std::true_type{}, nxt, Xr, (k + 3) & 3, cur, k & 3, m);
            m = step_meta(sb + k + 1);                    // the next interval's scalars, ahead of the barrier wait
            Xr = step_src(sb + k + 4);
            // the barrier needs the partials' writes retired, not the operand reads behind them (see mmd_pipe_kernel)
            xch_write(nxt);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            gram_prefetch(cur);                           // interval k + 1 runs its Gram on buffer cur (tile k + 2)
            {
                constexpr int NRD = PT * (LS ? 2 : 1);                // gram_prefetch's reads
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    wait_vmcnt<0>();

    __syncthreads();
    double* red = reinterpret_cast<double*>(sm + L::XCH_OFF);
    const double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    if (lane == 0) { red[wave * 2 + 0] = w0; red[wave * 2 + 1] = w1; }
    __syncthreads();
    if (tid == 0) {
        a.wg_sums[2 * bid + 0] = (red[0] + red[2]) + (red[4] + red[6]);
        a.wg_sums[2 * bid + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

template <int KB, bool SPLIT>
int launch_pipe_sum(const MmdArgs& a, int nblocks, hipStream_t s) {
    constexpr int lds_bytes = PipeLds<KB, SPLIT && TVAE_GRAM_PRODUCTS == 3>::BYTES;      // the kernel's LS
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_pipe_sum_kernel<KB, SPLIT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }));
    hipLaunchKernelGGL((mmd_pipe_sum_kernel<KB, SPLIT>), dim3(nblocks), dim3(256), lds_bytes, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    if constexpr (SPLIT && TVAE_Q32) {
        if (a.qflag != nullptr) return launch_mmd_q32_sum(KB, a, nblocks, s);      // the pair: exactly one of the two does the work
    }
    return TVAE_OK;
}

template <int KB, bool SPLIT>
int launch_pipe(const MmdArgs& a, int nblocks, hipStream_t s) {
    constexpr int lds_bytes = PipeLds<KB, SPLIT>::BYTES;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_pipe_kernel<KB, SPLIT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }));
    hipLaunchKernelGGL((mmd_pipe_kernel<KB, SPLIT>), dim3(nblocks), dim3(256), lds_bytes, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    if constexpr (SPLIT) {
        if (a.qflag != nullptr) {          // the pair: exactly one of the two does the work (QFlag)
#if TVAE_Q32
            return launch_mmd_q32(KB, a, nblocks, s);      // round 3: one-product, 32-point, all-32x32x16 Q-form kernel (mmd_q32.hip)
#else       // -DTVAE_Q32=0: round 2's Q-form instance of this file (two-product Gram, 16-point steps, tile ring)
            static PerDeviceOnce once_q;
            TVAE_CHECK_HIP(per_device_once(once_q, [] {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_pipe_kernel<KB, SPLIT, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            }));
            hipLaunchKernelGGL((mmd_pipe_kernel<KB, SPLIT, true>), dim3(nblocks), dim3(256), lds_bytes, s, a);
            TVAE_CHECK_HIP(hipGetLastError());
#endif
        }
    }
    return TVAE_OK;
}

template <bool SPLIT>
int launch_pipe_shape(bool grad, int kb, const MmdArgs& a, int nblocks, hipStream_t s) {
    switch (kb) {
        case 2: return grad ? launch_pipe<2, SPLIT>(a, nblocks, s) : launch_pipe_sum<2, SPLIT>(a, nblocks, s);
        case 3: return grad ? launch_pipe<3, SPLIT>(a, nblocks, s) : launch_pipe_sum<3, SPLIT>(a, nblocks, s);
        case 4: return grad ? launch_pipe<4, SPLIT>(a, nblocks, s) : launch_pipe_sum<4, SPLIT>(a, nblocks, s);
        case 5: return grad ? launch_pipe<5, SPLIT>(a, nblocks, s) : launch_pipe_sum<5, SPLIT>(a, nblocks, s);
        case 6: return grad ? launch_pipe<6, SPLIT>(a, nblocks, s) : launch_pipe_sum<6, SPLIT>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no pipelined kernel for %d feature blocks", kb);
}

}  // namespace

int launch_mmd_pipe(bool grad, bool split, int kb, const MmdArgs& a, int nblocks, hipStream_t s) {
    return split ? launch_pipe_shape<true>(grad, kb, a, nblocks, s) : launch_pipe_shape<false>(grad, kb, a, nblocks, s);
}

}  // namespace tvae
