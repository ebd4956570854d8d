// Q-form MMD gradient kernel, round 3: ONE fp16 product everywhere, 32-point steps, every matrix instruction 32x32x16.
//
// Same math, workspace and per-workgroup outputs as mmd_pipe_kernel<KB, true, true> (mmd_pipe.hip): fp32
// latents as scaled fp16 values, Q-form (the tile kernel sums Q = expm1(-t) + t, everything linear in t is closed form),
// 64 rows per workgroup of four waves, the features sliced over the waves, one barrier per step, Gram of step k + 1 beside
// the E work of step k, gradient of step k beside the reduction of step k + 1's partial sums.  What changed, and why:
//
//  * the Gram takes the hi halves alone on BOTH sides (one product instead of two).  With the Q-form the kernel's t only
//    feeds second-order terms; measured against fp64 the loss and the gradient do not move (tools/gram_numerics_cpu.py,
//    tests/test_gpu_r3.py).  The lo image of the rows leaves the registers (96 at D = 768), which is what a 32-point step
//    needs;
//  * the Gram runs on v_mfma_f32_32x32x16_f16 like the gradient: a 16x16x32 holds the SIMD's vector issue for 8 of its 16
//    cycles, a 32x32x16 for 8 of its 32 (MI355X_MICROARCH.md, cycle constants), and this kernel is a lone in-order wave
//    per SIMD whose ~400 vector / scalar / LDS instructions per step ran at their SUM with the matrix time on the 16x16
//    form.  Per 32-point step a wave issues 24 + 24 matrix instructions of 32 cycles;
//  * 32-point steps: one barrier, one exchange, one set of scalar bookkeeping per 32 points instead of 16.
//
// Layouts.  MFMA 32x32x16: A[m][k]: lane holds m = lane % 32, k = 8 (lane / 32) + 0..7; B[k][n]: n = lane % 32, same k;
// D[m][n]: n = lane % 32, m = 8 (r / 4) + 4 (lane / 32) + r % 4 for register r.
//   Gram:      D[point][row]  = sum_f X[point][f] Z[row][f]      A = point tile from LDS (ds_read_b128), B = rows (registers)
//   gradient:  D[feature][row] += sum_p X[p][feature] (s_p Q[p][row])   A = the same tile read transposed, B = Q through LDS
// The Gram's output registers r of a lane are points 8 (r / 4) + 4 (lane / 32) + r % 4 of ONE row: registers 0-7 / 8-15
// are the two 16-point halves, and eight consecutive registers are exactly the eight k-values that lane supplies as B
// operand of the gradient instruction of that half.  So the wave that finishes rows 32 wr .. + 31 x points 16 wp .. + 15
// (wave = 2 wr + wp) writes its Q as ONE 16-byte store per lane into [kk = wp][rbk = wr][lane], and every wave reads its
// four B operands back with four 16-byte loads -- no shuffles, no conflicts.
// LDS image of a step (per wave: its KB blocks of 32 features): block = two 1 KiB pieces (points 0-15, 16-31), a point's 32
// features are 64 contiguous bytes, its four 16-byte chunks stored at position c ^ f(point / 4) with
// f(q) = 2 (q & 1) ^ ((q >> 1) & 1) ^ 3 ((q >> 2) & 1): conflict-free for the Gram's ds_read_b128 (the sixteen lanes of a
// service group see four different f) and for ds_read_b64_tr_b16 (f of neighbouring point quads differs in bit 1).
#include <type_traits>

#include "mmd_common.h"

namespace tvae {
namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 q_f16x8;
typedef __attribute__((ext_vector_type(4))) short q_s16x4;
typedef __attribute__((ext_vector_type(8))) short q_s16x8;

template <int KB>
struct Q32Lds {
    static constexpr int SLICE_B = KB * 2048;                 // a wave's share of a 32-point step
    static constexpr int XBUF_B = 4 * SLICE_B;                // one step buffer
    static constexpr int NORM_OFF = 2 * XBUF_B;               // [2 buf][4 waves][64] floats (32 used; the DMA moves 64)
    static constexpr int XCH_OFF = NORM_OFF + 2 * 4 * 64 * 4;
    static constexpr int XCH_ONE_B = 4 * 3 * 2 * 1024;        // [owner][3 remote sources][2 x float4][64 lanes]
    static constexpr int EBUF_OFF = XCH_OFF + 2 * XCH_ONE_B;
    static constexpr int EBUF_ONE_B = 4 * 1024;               // [kk][rbk][64 lanes] 8 x fp16
    static constexpr int BYTES = EBUF_OFF + 2 * EBUF_ONE_B;
    static_assert(BYTES <= 160 * 1024, "q32 kernel exceeds the 160 KiB LDS");
    static_assert(XCH_ONE_B >= (8 + 4 * 64) * 8, "final reduction reuses an exchange buffer");
};

__device__ __forceinline__ int q32_swz(int q) { return (2 * (q & 1)) ^ ((q >> 1) & 1) ^ (3 * ((q >> 2) & 1)); }

#ifndef TVAE_Q32_TD
#define TVAE_Q32_TD 2       // gradient operand reads run this many feature blocks ahead
#endif
#ifndef TVAE_Q32_GD
#define TVAE_Q32_GD 2       // Gram operand reads run this many k-steps ahead
#endif

template <int KB>
__global__ __launch_bounds__(256, 1) void mmd_q32_kernel(MmdArgs a) {
    // one of the launch pair (mmd_pipe.hip, QFlag): the E-form instance does the work when the flag is off
    if (__builtin_amdgcn_readfirstlane((int)*a.qflag) == 0) return;
    using L = Q32Lds<KB>;
    constexpr int P = 32, SW = 32 * KB, DPAD = 4 * SW, NK = 2 * KB;      // NK: 16-feature k-steps of the Gram
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wp = wave & 1;                 // this wave finishes rows 32 wr .. x points 16 wp .. of every step
    const int n32 = lane & 31, h = lane >> 5;
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int rem = (int)blockIdx.x;
    const int rb = a.sp_major ? rem % a.row_blocks : rem / a.col_splits;
    const int sp = a.sp_major ? rem / a.row_blocks : rem - rb * a.col_splits;
    const int i0 = rb * 64;
    const long long bid = (long long)a.sums_base + blockIdx.x;

    // ---- the 64 rows of this workgroup, this wave's feature slice, as B operands.  Row blocks in an order ROTATED by the
    // wave's ownership (local block b' = actual block b' ^ wr; the A operand's points likewise, below): the part of S' a
    // wave finishes itself is then always registers 0-7 of S[0], whatever the wave
    const _Float16* rows = reinterpret_cast<const _Float16*>(a.cols[0]) + (long long)a.row_offset * DPAD;
    const float* rnorm = a.cnorm[0] + a.row_offset;
    q_f16x8 zreg[2][NK];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int row = i0 + 32 * (b ^ wr) + n32;
        const _Float16* zp = rows + (long long)row * DPAD + wave * SW + 8 * h;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (row < a.n_rows) zreg[b][ks] = *reinterpret_cast<const q_f16x8*>(zp + 16 * ks);
            else zreg[b][ks] = q_f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    const int my_row = i0 + 32 * wr + n32;
    const bool row_valid = my_row < a.n_rows;
    const SplitScales ss = split_scales(*a.split_amax);
    // x = -t = c1 S' - (|x_p|^2 + |z_i|^2) / D^2  with S' the Gram of the scaled values
    const float c1 = 2.f * ss.inv_s2 * a.inv_d2;
    float rnx = row_valid ? -rnorm[my_row] * a.inv_d2 : 0.f;
    // retire these loads before the loop (hipcc would otherwise wait vmcnt(0) at their first use, draining the DMA)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) asm volatile("" : "+v"(zreg[b][ks]));
    asm volatile("" : "+v"(rnx));
    const bool any_invalid = __any(!row_valid);

    f32x16 acc[KB][2];                                        // gradient: [32-feature block][ACTUAL 32-row block]
#pragma unroll
    for (int x = 0; x < KB; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    // zeros the first matrix instruction of every Gram accumulates onto: kept in registers (a literal zero vector costs
    // 32 register writes per step)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    asm volatile("" : "+v"(zero16));

    // steps of 32 points over both column sets: [0, nS) latents, [nS, 2 nS) priors
    const int nS = a.tiles_per_set / 2;
    const long long total = 2ll * nS;
    const int sb = (int)(sp * total / a.col_splits);
    const int se = (int)((sp + 1) * total / a.col_splits);
    const int nsteps = se - sb;

    // ---- per-lane constants of the LDS image ----
    unsigned dma_voff0, dma_voff1;                            // source byte offsets of the two pieces of a block
    {
        const int r = lane >> 2, pos = lane & 3;
        dma_voff0 = (unsigned)(r * DPAD + 8 * (pos ^ q32_swz(r >> 2))) * 2u;
        // the second piece rides on the instruction offset (1 KiB further in LDS), which the hardware adds to the global
        // address as well: taken off here
        dma_voff1 = (unsigned)((16 + r) * DPAD + 8 * (pos ^ q32_swz(4 + (r >> 2)))) * 2u - 1024u;
    }
    int gram_off[2];                                          // A operand of k-step ks: + 2048 (ks / 2), [ks & 1]
    {
        const int pa = n32 ^ (16 * wp);                       // this lane's row of A holds point pa (rotation, see zreg)
#pragma unroll
        for (int par = 0; par < 2; ++par)
            gram_off[par] = 1024 * (pa >> 4) + 64 * (pa & 15) + 16 * ((2 * par + h) ^ q32_swz(pa >> 2));
    }
    int tr_off[2][2];                                         // transposed reads of the gradient's A operand: [kk][j], + 2048 blk
    {
        const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pt = 16 * kk + 8 * j + 4 * (g >> 1) + tq;      // k = 8 (lane / 32) + 4 j + i  <->  point 16 kk + 8 j + 4 (lane / 32) + i
                tr_off[kk][j] = 1024 * kk + 64 * (pt & 15) + 16 * ((2 * (g & 1) + (tp >> 1)) ^ q32_swz(pt >> 2)) + 8 * (tp & 1);
            }
    }
    auto tr_read = [&](int byte_off) -> q_s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) q_s16x4*)(__attribute__((address_space(3))) char*)(sm + byte_off));
    };

    struct Src { const _Float16* x; const float* nrm; };
    auto step_meta = [&](int v, int& set, int& p0) {
        const int u = v < se ? v : se - 1;                    // past the end: the last step again (results unused)
        set = u >= nS ? 1 : 0;
        p0 = (u - (set ? nS : 0)) * P;
    };
    const _Float16* const cbase0 = reinterpret_cast<const _Float16*>(a.cols[0]) + wave * SW;
    const _Float16* const cbase1 = reinterpret_cast<const _Float16*>(a.cols[1]) + wave * SW;
    const float* const nbase0 = a.cnorm[0];
    const float* const nbase1 = a.cnorm[1];
    auto step_src = [&](int v) -> Src {
        int set, p0;
        step_meta(v, set, p0);
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(p0) * (unsigned)DPAD;   // the plan keeps n_pad * DPAD < 2^31
        return Src{(set ? cbase1 : cbase0) + off, (set ? nbase1 : nbase0) + p0};
    };
    auto xbuf_byte = [&](int b) -> unsigned { return (unsigned)(b * L::XBUF_B + wave * L::SLICE_B); };
    auto stage_norms = [&](const Src& X, int nbuf) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(L::NORM_OFF + (nbuf * 4 + wave) * 64 * 4));
        dma4(uniform_ptr(X.nrm), (unsigned)lane * 4u, dst);
    };
    auto dma_block = [&](const Src& X, int b, int s) {
        dma16_pair(reinterpret_cast<const float*>(X.x + 32 * s), dma_voff0, dma_voff1, smem_byte + xbuf_byte(b) + 2048u * s);
    };

    f32x16 S[2];                                              // partial Gram of the step in flight, local (rotated) order
    // reduce-scatter: local (block b', half hf') is actual (b' ^ wr, hf' ^ wp), finished by wave 2 (b' ^ wr) + (hf' ^ wp)
    auto xch_write = [&](int xbuf) {
        char* xb = sm + L::XCH_OFF + xbuf * L::XCH_ONE_B + lane * 16;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (b == 0 && hf == 0) continue;
                const int owner = 2 * (b ^ wr) + (hf ^ wp);                  // wave-uniform scalar arithmetic
                char* dst = xb + ((owner * 3 + ((wave - owner - 1) & 3)) * 2) * 1024;
                *reinterpret_cast<f32x4*>(dst) = f32x4{S[b][8 * hf], S[b][8 * hf + 1], S[b][8 * hf + 2], S[b][8 * hf + 3]};
                *reinterpret_cast<f32x4*>(dst + 1024) = f32x4{S[b][8 * hf + 4], S[b][8 * hf + 5], S[b][8 * hf + 6], S[b][8 * hf + 7]};
            }
    };
    float xv[8];                                              // x = -t of the eight S' entries this lane finishes
    f32x4 part[3][2], cn[2];
    auto xch_fetch = [&](int xbuf, int nbuf) {
        const char* xb = sm + L::XCH_OFF + xbuf * L::XCH_ONE_B + lane * 16;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int q = 0; q < 2; ++q) part[w][q] = *reinterpret_cast<const f32x4*>(xb + ((wave * 3 + w) * 2 + q) * 1024);
        // register r = 4 j + i of this lane is point 16 wp + 8 j + 4 h + i
#pragma unroll
        for (int j = 0; j < 2; ++j)
            cn[j] = *reinterpret_cast<const f32x4*>(sm + L::NORM_OFF + ((nbuf * 4 + wave) * 64 + 16 * wp + 8 * j + 4 * h) * 4);
    };
    auto xch_finish = [&]() {
#pragma unroll
        for (int r = 0; r < 8; ++r) {          // own partial (still in registers) + the three others, fixed order
            const float sv = ((S[0][r] + part[0][r >> 2][r & 3]) + part[1][r >> 2][r & 3]) + part[2][r >> 2][r & 3];
            xv[r] = fmaf(c1, sv, fmaf(cn[r >> 2][r & 3], -a.inv_d2, rnx));
        }
    };

    double sum0 = 0.0, sum1 = 0.0, rowsum = 0.0;              // per-lane fp64 partials: Q sums by column set, this lane's row sum

    q_f16x8 nab;                                              // k-step 0 operand of the NEXT interval's Gram (read at the end of this one)
    auto gram_prefetch = [&](int b) {
        wait_vmcnt_n(2 * (KB - 1));
        nab = *reinterpret_cast<const q_f16x8*>(sm + xbuf_byte(b) + gram_off[0]);
        __builtin_amdgcn_sched_barrier(0);
    };

    // First half of an interval: partial Gram of the step in buffer b, with the Q work of the PREVIOUS step (from xv; column
    // set x_set, first point x_p0; pieces into E buffer ebuf) spread over the k-steps.  SLOW: masks of ragged edges
    // (rows past n_rows, points past n_cols); a wave-uniform branch picks it per step.  `slack`: DMA instructions issued
    // after this step's that may still be in flight.  WITH_E false: the prologue's Gram.
    auto front = [&](auto slow_tag, auto with_e_tag, int b, int slack, int x_set, int x_p0, int ebuf) {
        constexpr bool SLOW = decltype(slow_tag)::value, WITH_E = decltype(with_e_tag)::value;
        const char* xs = sm + xbuf_byte(b);
        constexpr int GD = NK < TVAE_Q32_GD ? NK : TVAE_Q32_GD;
        q_f16x8 ab[NK];
        auto rd = [&](int ks) {
            if (WITH_E && ks == 0) { ab[0] = nab; return; }
            wait_vmcnt_n(2 * (KB - 1 - (ks >> 1)) + slack);
            ab[ks] = *reinterpret_cast<const q_f16x8*>(xs + gram_off[ks & 1] + 2048 * (ks >> 1));
            __builtin_amdgcn_sched_barrier(0);      // keep the read HERE, ahead of the earlier k-steps' matrix instructions
        };
        float q[8];
        auto e_slice = [&](int ks) {
            // element el finishes in k-step el * (FIN + 1) / 8; sums, conversion and publication behind k-step FIN, with the
            // matrix instructions of the remaining k-steps still to come
            constexpr int FIN = NK > 2 ? NK - 3 : 0;
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                if (el * (FIN + 1) / 8 != ks) continue;
                float qv = em1_split(-xv[el]).q;              // Q = x^2 r(x): second order, no cancellation
                if constexpr (SLOW) {
                    const bool ok = row_valid && (x_p0 + 16 * wp + 8 * (el >> 2) + 4 * h + (el & 3) < a.n_cols);
                    qv = ok ? qv : 0.f;
                }
                q[el] = qv;
            }
            if (ks != FIN) return;
            // this lane's eight entries belong to ONE row: their sum serves the Q sums and the row sum
            const float es = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
            rowsum += (double)(x_set ? -es : es);
            sum0 += (double)(x_set ? 0.f : es);
            sum1 += (double)(x_set ? es : 0.f);
            // prior columns enter the gradient with s_p = -1: folded into the fp16 scale
            const float esc = x_set ? -SPLIT_E_SCALE : SPLIT_E_SCALE;
            q_f16x8 ev;
#pragma unroll
            for (int r = 0; r < 8; ++r) ev[r] = (_Float16)(q[r] * esc);
            *reinterpret_cast<q_f16x8*>(sm + L::EBUF_OFF + ebuf * L::EBUF_ONE_B + (wp * 2 + wr) * 1024 + lane * 16) = ev;
        };
#pragma unroll
        for (int ks = 0; ks < GD; ++ks) rd(ks);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks + GD < NK) rd(ks + GD);
#pragma unroll
            for (int bq = 0; bq < 2; ++bq)
                S[bq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab[ks], zreg[bq][ks], ks == 0 ? zero16 : S[bq], 0, 0, 0);
            if constexpr (WITH_E) e_slice(ks);
        }
    };

    // gradient of the step in buffer b with the Q pieces of ebuf; a 32-feature block of the buffer is dead after its matrix
    // instructions -> refilled in place with the same block of step Xn
    constexpr int TD = KB < TVAE_Q32_TD ? KB : TVAE_Q32_TD;
    q_f16x8 Bq[2][2];                                         // [kk][actual rbk]
    q_s16x4 gh[KB][2][2];
    auto grad_rd = [&](int xb0, int blk) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) gh[blk][kk][j] = tr_read(xb0 + 2048 * blk + tr_off[kk][j]);
    };
    auto grad_fetch_x = [&](int b) {          // wave-private tile data: legal ahead of the barrier
#pragma unroll
        for (int blk = 0; blk < TD; ++blk) grad_rd((int)xbuf_byte(b), blk);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto grad_fetch_e = [&](int ebuf) {
        const char* eb = sm + L::EBUF_OFF + ebuf * L::EBUF_ONE_B + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rbk = 0; rbk < 2; ++rbk) Bq[kk][rbk] = *reinterpret_cast<const q_f16x8*>(eb + (kk * 2 + rbk) * 1024);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto grad_run = [&](int b, const Src& Xn) {
        const int xb0 = (int)xbuf_byte(b);
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
            if (blk + TD < KB) grad_rd(xb0, blk + TD);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const q_f16x8 A = __builtin_bit_cast(q_f16x8, __builtin_shufflevector(gh[blk][kk][0], gh[blk][kk][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int rbk = 0; rbk < 2; ++rbk)
                    acc[blk][rbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, Bq[kk][rbk], acc[blk][rbk], 0, 0, 0);
            }
            // the matrix instructions of the block are issued (their operand reads have completed) before its refill
            __builtin_amdgcn_sched_barrier(0);
            dma_block(Xn, b, blk);
        }
    };

#ifdef TVAE_PHASE_TIMING
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    Src Xn2_c{};
    int c_set_c = 0, c_p0_c = 0;
    bool slow_c = true;
    auto interval = [&](int k) {
        PHASE_MARK(5);
        const int cur = k & 1, nxt = cur ^ 1;
        // this interval's scalars were worked out at the end of the previous one, behind its last matrix instructions
        const Src Xn2 = Xn2_c;
        const int c_set = c_set_c, c_p0 = c_p0_c;
        // Gram of the next step  ||  Q of this one (from xv, left by the previous interval)
        if (slow_c) front(std::true_type{}, std::true_type{}, nxt, 0, c_set, c_p0, cur);   // wave-uniform
        else front(std::false_type{}, std::true_type{}, nxt, 0, c_set, c_p0, cur);
        PHASE_MARK(0);
        // the barrier needs the WRITES (Q piece, partials) retired, not the gradient's operand reads: LDS operations retire
        // in order, so with the reads issued behind the writes a counted wait leaves them in flight across the barrier
        xch_write(nxt);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        grad_fetch_x(cur);
        {
            constexpr int NRD = TD * 4;
            static_assert(NRD <= 15, "lgkmcnt is a 4-bit counter");
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
        }
        PHASE_MARK(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PHASE_MARK(2);
        // gradient of this step (+ refill with step k + 2)  ||  reduction of the next step's partials
        grad_fetch_e(cur);
        stage_norms(Xn2, cur);
        xch_fetch(nxt, nxt);
        __builtin_amdgcn_sched_barrier(0);
        PHASE_MARK(3);
        grad_run(cur, Xn2);
        xch_finish();
        step_meta(sb + k + 1, c_set_c, c_p0_c);
        Xn2_c = step_src(sb + k + 3);
        slow_c = any_invalid || c_p0_c + P > a.n_cols;
        gram_prefetch(cur);                   // the next interval's Gram: the step just refilled into this buffer
        PHASE_MARK(4);
    };

    if (nsteps > 0) {
        // ---- prologue: steps sb, sb + 1 in flight; Gram and exchange of the first one ----
        {
            const Src X0 = step_src(sb), X1 = step_src(sb + 1);
            stage_norms(X0, 0);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X0, 0, s);
            stage_norms(X1, 1);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X1, 1, s);
            front(std::false_type{}, std::false_type{}, 0, 1 + 2 * KB, 0, 0, 0);
            xch_write(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            xch_fetch(0, 0);
            xch_finish();
            gram_prefetch(1);
            step_meta(sb, c_set_c, c_p0_c);
            Xn2_c = step_src(sb + 2);
            slow_c = any_invalid || c_p0_c + P > a.n_cols;
        }
        for (int k = 0; k < nsteps; ++k) interval(k);
    }
    wait_vmcnt<0>();
#ifdef TVAE_PHASE_TIMING
    if (blockIdx.x == 0 && lane == 0 && nsteps > 0)
        printf("wave %d steps %d: front %llu  xch write %llu  barrier %llu  fetch %llu  grad %llu  top %llu (cycles per step)\n", wave,
               nsteps, ph[0] / nsteps, ph[1] / nsteps, ph[2] / nsteps, ph[3] / nsteps, ph[4] / nsteps, ph[5] / nsteps);
#endif

    // ---- per-workgroup results ----
    __syncthreads();
    double* red = reinterpret_cast<double*>(sm + L::XCH_OFF);          // [4][2] sums, then [4 waves][32] row sums
    const double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    if (lane == 0) { red[wave * 2 + 0] = w0; red[wave * 2 + 1] = w1; }
    {
        double rv = rowsum;
        rv += __shfl_xor(rv, 32, 64);                                  // the two point quads (lane / 32) of a row
        if (h == 0) red[8 + wave * 32 + n32] = rv;
    }
    __syncthreads();
    if (tid == 0) {
        a.wg_sums[2 * bid + 0] = (red[0] + red[2]) + (red[4] + red[6]);
        a.wg_sums[2 * bid + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
    if (tid < 64) {           // row 32 wr + n: the waves of both point halves, in a fixed order
        const int rwr = tid >> 5, rn = tid & 31;
        a.wg_rowsum[(long long)sp * a.n_rows_pad + i0 + tid] = red[8 + (2 * rwr) * 32 + rn] + red[8 + (2 * rwr + 1) * 32 + rn];
    }
    // 32x32 tiles: register r of a lane is feature 8 (r / 4) + 4 (lane / 32) + r % 4 of row lane % 32
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
        float* grow = a.wg_grad + ((long long)sp * a.n_rows_pad + i0 + 32 * rbk + n32) * DPAD + wave * SW + 4 * h;
#pragma unroll
        for (int s = 0; s < KB; ++s)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<f32x4*>(grow + 32 * s + 8 * q4) =
                    f32x4{acc[s][rbk][4 * q4], acc[s][rbk][4 * q4 + 1], acc[s][rbk][4 * q4 + 2], acc[s][rbk][4 * q4 + 3]};
    }
}

template <int KB>
int launch_q32(const MmdArgs& a, int nblocks, hipStream_t s) {
    constexpr int lds_bytes = Q32Lds<KB>::BYTES;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_q32_kernel<KB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   lds_bytes);
    }));
    hipLaunchKernelGGL((mmd_q32_kernel<KB>), dim3(nblocks), dim3(256), lds_bytes, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

}  // namespace

// the Q-form member of the gradient launch pair for kb = 2 .. 6 feature blocks per wave (192 < D <= 768)
int launch_mmd_q32(int kb, const MmdArgs& a, int nblocks, hipStream_t s) {
    switch (kb) {
        case 2: return launch_q32<2>(a, nblocks, s);
        case 3: return launch_q32<3>(a, nblocks, s);
        case 4: return launch_q32<4>(a, nblocks, s);
        case 5: return launch_q32<5>(a, nblocks, s);
        case 6: return launch_q32<6>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no q32 kernel for %d feature blocks", kb);
}

}  // namespace tvae
