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
//   hi + lo tiles fit the LDS.  The gradient's K = 32 then holds 16 points TWICE:
//       acc += [X_hi | X_lo]^T [E_hi ; E_hi]  +  [X_hi | X_hi]^T [E_lo ; E_lo2]          (E = hi + lo + lo2)
//   i.e. X_hi E_hi + X_lo E_hi + X_hi (E_lo + E_lo2) in two matrix instructions per 16 x 16 output block.
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
    static constexpr int IMG_B = P * SW * 2;                  // one image (hi or lo) of a wave's slice of a step
    static constexpr int SLICE_B = (SPLIT ? 2 : 1) * IMG_B;   // SPLIT: the hi image, then the lo image
    static constexpr int XBUF_B = 4 * SLICE_B;                // one point-tile buffer (four wave slices)
    static constexpr int NORM_OFF = 2 * XBUF_B;               // [2 buf][4 waves][64] float point norms (P used; the DMA moves 64)
    static constexpr int XCH_OFF = NORM_OFF + 2 * 4 * 64 * 4;
    static constexpr int XCH_ONE_B = NT * 3 * 64 * 16;        // [tile][3 remote sources][64 lanes] float4
    static constexpr int EBUF_OFF = XCH_OFF + 2 * XCH_ONE_B;
    static constexpr int EPIECE_B = SPLIT ? RT * 16 * 32 : RT * 64 * 16;   // SPLIT: [t][row][16 points]; else [t][lane][2 pt][4]
    static constexpr int EBUF_ONE_B = (SPLIT ? 3 : 1) * EPIECE_B;
    static constexpr int BYTES = EBUF_OFF + 2 * EBUF_ONE_B;
    static_assert(BYTES <= 160 * 1024, "pipelined tile kernel exceeds the 160 KiB LDS");
    static_assert(XCH_ONE_B >= (8 + 4 * TPW * 16) * 8, "final reduction reuses an exchange buffer");
};

template <int KB, bool SPLIT>
__global__ __launch_bounds__(256, 1) void mmd_pipe_kernel(MmdArgs a) {
    using L = PipeLds<KB, SPLIT>;
    constexpr int P = L::P, PT = L::PT, RT = L::RT, TPW = L::TPW, SW = L::SW, DPAD = 4 * SW, NI = 2 * KB;
    constexpr int DPS = 2;                                    // DMA instructions per 32-feature block (pair of point tiles / hi + lo)
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

    f32x4 acc[NI][RT];
#pragma unroll
    for (int db = 0; db < NI; ++db)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[db][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // steps of P points over both column sets: [0, nS) latents, [nS, 2 nS) priors
    const int nS = a.tiles_per_set / PT;
    const long long total = 2ll * nS;
    const int sb = (int)(sp * total / a.col_splits);
    const int se = (int)((sp + 1) * total / a.col_splits);
    const int nsteps = se - sb;

    // ---- per-lane constants of the LDS images (one 1 KiB piece = 16 points x 32 features, mmd_bf16.hip) ----
    const int dma_r = lane >> 2, dma_c = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const unsigned dma_voff = (unsigned)(dma_r * DPAD + 8 * dma_c) * 2u;
    const unsigned dma_voff_b = dma_voff + 16u * DPAD * 2u - 1024u;      // second point tile of a pair (dma16_pair)
    const int gram_off = 16 * (4 * i16 + (g ^ (2 * ((i16 >> 2) & 1))));  // + 1024 * piece
    const int tq = i16 >> 2, tp = i16 & 3;
    // transposed reads of 16-feature block db: piece (db >> 1), base by the parity of db
    int tr_off[2][2];                 // bf16: [parity][0]; split: [parity][h = which four of a lane group's eight points]
    if constexpr (SPLIT) {
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                tr_off[par][h] = 16 * (4 * (8 * (g & 1) + 4 * h + tq) + ((2 * par + (tp >> 1)) ^ (2 * h))) + 8 * (tp & 1);
    } else {
        tr_off[0][0] = 16 * (4 * (4 * g + tq) + 2 * (g & 1) + (tp >> 1)) + 8 * (tp & 1);
        tr_off[1][0] = 16 * (4 * (4 * g + tq) + 2 * ((g & 1) ^ 1) + (tp >> 1)) + 8 * (tp & 1);
        tr_off[0][1] = tr_off[1][1] = 0;
    }
    auto tr_read = [&](int byte_off) -> s16x4_t {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(__attribute__((address_space(3))) char*)(sm + byte_off));
    };

    struct Src { const bf16* hi; const bf16* lo; const float* nrm; int set; int p0; };
    auto step_src = [&](int v) -> Src {
        const int u = v < se ? v : se - 1;                    // past the end: the last step again (results unused)
        const int set = u >= nS ? 1 : 0;
#ifdef TVAE_DEBUG_SAME_TILE
        const int p0 = 0 * u;
#else
        const int p0 = (u - set * nS) * P;
#endif
        const long long off = (long long)p0 * DPAD + wave * SW;
        return Src{reinterpret_cast<const bf16*>(a.cols[set]) + off,
                   SPLIT ? reinterpret_cast<const bf16*>(a.cols_lo[set]) + off : nullptr, a.cnorm[set] + p0, set, p0};
    };
    auto xbuf_byte = [&](int b) -> unsigned { return (unsigned)(b * L::XBUF_B + wave * L::SLICE_B); };
    auto stage_norms = [&](const Src& X, int nbuf) {
        // all 64 lanes move a float (no lane mask, no branch): the words past the step's P norms are never read
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(L::NORM_OFF + (nbuf * 4 + wave) * 64 * 4));
        dma4(uniform_ptr(X.nrm), (unsigned)lane * 4u, dst);
    };
    // features [32 s, 32 s + 32) of the step into buffer b: two DMA instructions
    auto dma_block = [&](const Src& X, int b, int s) {
        const unsigned base = smem_byte + xbuf_byte(b);
        if constexpr (SPLIT) {
            dma16(reinterpret_cast<const float*>(X.hi + 32 * s), dma_voff, base + 1024u * s);
            dma16(reinterpret_cast<const float*>(X.lo + 32 * s), dma_voff, base + (unsigned)L::IMG_B + 1024u * s);
        } else {
            dma16_pair(reinterpret_cast<const float*>(X.hi + 32 * s), dma_voff, dma_voff_b, base + 2048u * s);
        }
    };

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
    f32x4 sv[TPW];
    auto xch_reduce = [&](int xbuf) {
        const f32x4* xb = reinterpret_cast<const f32x4*>(sm + L::XCH_OFF + xbuf * L::XCH_ONE_B);
        f32x4 part[TPW][3];
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int w = 0; w < 3; ++w) part[j][w] = xb[((wave * TPW + j) * 3 + w) * 64 + lane];     // all reads in flight
#pragma unroll
        for (int j = 0; j < TPW; ++j)          // own partial (rotated order: still in registers) + the three others, fixed order
            sv[j] = ((S[0][j] + part[j][0]) + part[j][1]) + part[j][2];
    };

    double sum0 = 0.0, sum1 = 0.0, rowsum[TPW];     // per-lane fp64 partials of the Q sums (by column set) and the row sums
#pragma unroll
    for (int j = 0; j < TPW; ++j) rowsum[j] = 0.0;

    // First half of an interval: partial Gram of the tile in buffer b over this wave's feature slice, S[pt'][t'], with
    // the E work of the PREVIOUS tile (step X: from sv, its norms in nbuf, pieces into E buffer ebuf) cut into KB
    // slices that follow the matrix instructions of the KB k-steps.  The counted DMA waits in front of the operand
    // reads are inline asm, which hipcc does not schedule across: each k-step is a scheduling region of its own, and
    // vector work only lands between matrix instructions it shares a region with.  E = expm1(-t): register r <-> point
    // p0 + 16 my_pt + 4 g + r, column = row i16 of row tile my_t0 + j; straight-line code (polynomial and exp form both
    // evaluated and selected); the masks of ragged edges exist only in the EDGE instantiation, which a wave-uniform
    // branch picks per step.  `slack`: DMA instructions issued after this tile's that may still be in flight;
    // WITH_E = false: the prologue's Gram.
    auto front = [&](auto edge_tag, auto with_e_tag, int b, int slack, const Src& X, int nbuf, int ebuf) {
        constexpr bool EDGE = decltype(edge_tag)::value, WITH_E = decltype(with_e_tag)::value;
        constexpr int NE = 4 * TPW;                           // E values per lane
        const char* xs = sm + xbuf_byte(b);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int t = 0; t < RT; ++t) S[pt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int GD = KB < 2 ? KB : 2;                   // operand reads run GD k-steps ahead of their MFMAs
        h16x8_t ab[KB][PT], al[SPLIT ? KB : 1][PT];
        auto rd = [&](int sr) {
            wait_vmcnt_n(DPS * (KB - 1 - sr) + slack);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int piece = PT * sr + (PT == 1 ? 0 : ((pt + my_pt) & (PT - 1)));      // local -> global point tile
                ab[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + gram_off + 1024 * piece);
                if constexpr (SPLIT) al[sr][pt] = *reinterpret_cast<const h16x8_t*>(xs + L::IMG_B + gram_off + 1024 * piece);
            }
        };
        f32x4 cn, e[TPW], q[TPW];
        if constexpr (WITH_E)
            cn = *reinterpret_cast<const f32x4*>(sm + L::NORM_OFF + ((nbuf * 4 + wave) * 64 + 16 * my_pt + 4 * g) * 4);
        auto e_slice = [&](int s) {
            // elements el = 4 j + r spread over slices 0 .. KB-2; sums, conversion and publication in slice KB-1
#pragma unroll
            for (int el = 0; el < NE; ++el) {
                if ((KB > 1 ? el * (KB - 1) / NE : 0) != s) continue;
                const int j = el >> 2, r = el & 3;
                const float tt = fmaxf((cn[r] + rn[j] - gram_unscale2 * sv[j][r]) * a.inv_d2, 0.f);
                const EQ pl = em1_split(tt);                                  // t < 1/8
                const float ex = __expf(-tt) - 1.f;                           // beyond: |E| >= 0.1175, 1e-6 relative
                const bool big = tt >= EM1_POLY_MAX;
                float ev = big ? ex : pl.e, qv = big ? ex + tt : pl.q;
                if constexpr (EDGE) {
                    const bool ok = my_valid[j] && (X.p0 + 16 * my_pt + 4 * g + r < a.n_cols);
                    ev = ok ? ev : 0.f;
                    qv = ok ? qv : 0.f;
                }
                e[j][r] = ev;
                q[j][r] = qv;
            }
            if (s != KB - 1) return;
            // sums carry Q (linear part: closed form in the finalize kernel); row sums carry E
            float qs_all = 0.f;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                qs_all += (q[j][0] + q[j][1]) + (q[j][2] + q[j][3]);
                const float es = (e[j][0] + e[j][1]) + (e[j][2] + e[j][3]);
                rowsum[j] += (double)(X.set ? -es : es);
            }
            sum0 += (double)(X.set ? 0.f : qs_all);
            sum1 += (double)(X.set ? qs_all : 0.f);
            // prior columns enter the gradient with s_p = -1: flip the sign bits of the packed values
            const unsigned flip = X.set ? 0x80008000u : 0u;
            char* eb = sm + L::EBUF_OFF + ebuf * L::EBUF_ONE_B;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if constexpr (SPLIT) {
                    f16x4_t p0v, p1v, p2v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float es = e[j][r] * SPLIT_E_SCALE;
                        p0v[r] = (_Float16)es;
                        const float r1 = es - (float)p0v[r];
                        p1v[r] = (_Float16)r1;
                        p2v[r] = (_Float16)(r1 - (float)p1v[r]);
                    }
                    uint2 b0 = __builtin_bit_cast(uint2, p0v), b1 = __builtin_bit_cast(uint2, p1v), b2 = __builtin_bit_cast(uint2, p2v);
                    b0.x ^= flip; b0.y ^= flip; b1.x ^= flip; b1.y ^= flip; b2.x ^= flip; b2.y ^= flip;
                    // [piece][row tile][row][16 points]: this lane holds points 4 g .. 4 g + 3 of row i16
                    const int at = ((my_t0 + j) * 16 + i16) * 32 + 8 * g;
                    *reinterpret_cast<uint2*>(eb + at) = b0;
                    *reinterpret_cast<uint2*>(eb + L::EPIECE_B + at) = b1;
                    *reinterpret_cast<uint2*>(eb + 2 * L::EPIECE_B + at) = b2;
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
#pragma unroll
        for (int s = 0; s < GD; ++s) rd(s);
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            if (s + GD < KB) rd(s + GD);
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(ab[s][pt], zreg[t][s], S[pt][t]);
            if constexpr (SPLIT) {
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(ab[s][pt], zlo[t][s], S[pt][t]);
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) S[pt][t] = mma<SPLIT>(al[s][pt], zreg[t][s], S[pt][t]);
            }
            if constexpr (WITH_E) e_slice(s);
        }
    };

    // gradient of the tile in buffer b with the E pieces of ebuf: acc^T[d][i] += X^T[d][p] (s_p E[p][i]); block pair s of
    // the buffer is dead after its MFMAs -> refilled in place with the same block of step Xn
    auto grad = [&](int b, int ebuf, const Src& Xn) {
        const char* eb = sm + L::EBUF_OFF + ebuf * L::EBUF_ONE_B;
        h16x8_t B1[RT], B2[SPLIT ? RT : 1];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if constexpr (SPLIT) {
                const int at = (t * 16 + i16) * 32 + 16 * (g & 1);
                B1[t] = *reinterpret_cast<const h16x8_t*>(eb + at);
                B2[t] = *reinterpret_cast<const h16x8_t*>(eb + (g < 2 ? 1 : 2) * L::EPIECE_B + at);
            } else {
                B1[t] = *reinterpret_cast<const h16x8_t*>(eb + (t * 64 + lane) * 16);
            }
        }
        const int xb0 = (int)xbuf_byte(b);
        constexpr int TD = NI < 2 ? NI : 2;                   // transposed operand reads run TD blocks ahead
        s16x4_t ga[NI][2], gb[SPLIT ? NI : 1][2];
        auto rd = [&](int db) {
            const int at = xb0 + 1024 * PT * (db >> 1);
            if constexpr (SPLIT) {
                // A1 = [X_hi | X_lo]: lane groups 0, 1 read the hi image, 2, 3 the lo image; A2 = [X_hi | X_hi]
                ga[db][0] = tr_read(at + (g >> 1) * L::IMG_B + tr_off[db & 1][0]);
                ga[db][1] = tr_read(at + (g >> 1) * L::IMG_B + tr_off[db & 1][1]);
                gb[db][0] = tr_read(at + tr_off[db & 1][0]);
                gb[db][1] = tr_read(at + tr_off[db & 1][1]);
            } else {
                ga[db][0] = tr_read(at + tr_off[db & 1][0]);
                ga[db][1] = tr_read(at + 1024 + tr_off[db & 1][0]);
            }
        };
#pragma unroll
        for (int db = 0; db < TD; ++db) rd(db);
#pragma unroll
        for (int db = 0; db < NI; ++db) {
            if (db + TD < NI) rd(db + TD);
            const h16x8_t A1 = __builtin_bit_cast(h16x8_t, __builtin_shufflevector(ga[db][0], ga[db][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[db][t] = mma<SPLIT>(A1, B1[t], acc[db][t]);
            if constexpr (SPLIT) {
                const h16x8_t A2 = __builtin_bit_cast(h16x8_t, __builtin_shufflevector(gb[db][0], gb[db][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[db][t] = mma<SPLIT>(A2, B2[t], acc[db][t]);
            }
            if (db & 1) {
                // the MFMAs of blocks 2 s, 2 s + 1 are issued (their operand reads have completed) before the refill
                __builtin_amdgcn_sched_barrier(0);
                dma_block(Xn, b, db >> 1);
            }
        }
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
            front(std::false_type{}, std::false_type{}, 0, 1 + DPS * KB, X0, 0, 0);
            xch_write(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            xch_reduce(0);
        }
        for (int k = 0; k < nsteps; ++k) {
            const int cur = k & 1, nxt = cur ^ 1;
            const Src Xc = step_src(sb + k), Xn2 = step_src(sb + k + 2);
            // Gram of the next tile  ||  E of this one (from sv, reduced in the previous interval)
            if (any_invalid || Xc.p0 + P > a.n_cols) front(std::true_type{}, std::true_type{}, nxt, 0, Xc, cur, cur);   // wave-uniform
            else front(std::false_type{}, std::true_type{}, nxt, 0, Xc, cur, cur);
            xch_write(nxt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // gradient of this tile (+ refill with tile k + 2)  ||  reduce of the next tile's partials
            stage_norms(Xn2, cur);
            xch_reduce(nxt);
            grad(cur, cur, Xn2);
        }
    }
    wait_vmcnt<0>();

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
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 16 * t + i16) * DPAD + wave * SW + 4 * g;
#pragma unroll
        for (int db = 0; db < NI; ++db) *reinterpret_cast<f32x4*>(grow + 16 * db) = acc[db][t];
    }
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
    return TVAE_OK;
}

template <bool SPLIT>
int launch_pipe_shape(int kb, const MmdArgs& a, int nblocks, hipStream_t s) {
    switch (kb) {
        case 2: return launch_pipe<2, SPLIT>(a, nblocks, s);
        case 3: return launch_pipe<3, SPLIT>(a, nblocks, s);
        case 4: return launch_pipe<4, SPLIT>(a, nblocks, s);
        case 5: return launch_pipe<5, SPLIT>(a, nblocks, s);
        case 6: return launch_pipe<6, SPLIT>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no pipelined kernel for %d feature blocks", kb);
}

}  // namespace

int launch_mmd_pipe(bool split, int kb, const MmdArgs& a, int nblocks, hipStream_t s) {
    return split ? launch_pipe_shape<true>(kb, a, nblocks, s) : launch_pipe_shape<false>(kb, a, nblocks, s);
}

}  // namespace tvae
