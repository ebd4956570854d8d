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
#include "mmd_q32.h"

namespace tvae {
namespace {

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

#ifndef TVAE_Q32_LAG
#define TVAE_Q32_LAG 4        // k-steps by which local row block 1 leads block 0 in the Gram (>= NK: one block after the other)
#endif
#ifndef TVAE_Q32_SUM_LAG
#define TVAE_Q32_SUM_LAG 16
#endif
#ifndef TVAE_Q32_SUM_ROWS128
#define TVAE_Q32_SUM_ROWS128 1    // sum-only launches: 128 rows per workgroup without a K-split (mmd_q32s.hip) when the launch is large enough (0: always this file's 64-row kernel)
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
    // per-lane source byte offsets of the LDS-DMA of (32-feature block b, 16-point piece hh): lane = (point r, chunk position
    // pos).  The second piece rides on the instruction offset (1 KiB further in LDS), which the hardware adds to the global
    // address as well: taken off here, over a 1 KiB bias that the column pointers carry (cbase below)
    unsigned dvo[KB][2];
    {
        const int r = lane >> 2, pos = lane & 3;
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                dvo[b][hh] = (unsigned)((16 * hh + r) * DPAD + 32 * b + 8 * (pos ^ q32_swz(4 * hh + (r >> 2)))) * 2u + 1024u - 1024u * hh;
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
    // LDS is addressed by 32-bit byte addresses that already include the base of the dynamic segment (pointer arithmetic on
    // `smem` costs one vector addition of the -- relocated, zero -- segment base per access)
    const int lds0 = (int)smem_byte;
    auto tr_read = [&](int addr) __attribute__((always_inline)) -> q_s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) q_s16x4*>((unsigned)addr));
    };

    struct Src { const _Float16* x; const float* nrm; };
    auto step_meta = [&](int v, int& set, int& p0) __attribute__((always_inline)) {
        const int u = v < se ? v : se - 1;                    // past the end: the last step again (results unused)
        set = u >= nS ? 1 : 0;
        p0 = (u - (set ? nS : 0)) * P;
    };
    // column pointers carry the wave's feature slice and the -1 KiB bias of dvo
    const _Float16* const cbase0 = reinterpret_cast<const _Float16*>(a.cols[0]) + wave * SW - 512;
    const _Float16* const cbase1 = reinterpret_cast<const _Float16*>(a.cols[1]) + wave * SW - 512;
    const float* const nbase0 = a.cnorm[0];
    const float* const nbase1 = a.cnorm[1];
    auto step_src = [&](int v) __attribute__((always_inline)) -> Src {
        int set, p0;
        step_meta(v, set, p0);
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(p0) * (unsigned)DPAD;   // the plan keeps n_pad * DPAD < 2^31
        return Src{(set ? cbase1 : cbase0) + off, (set ? nbase1 : nbase0) + p0};
    };
    // The refill's column stream walks [sb, se) INCREMENTALLY, branch-free: one step further per interval, over the gap
    // between the two sets' images when it crosses nS, and no further than step se - 1 (a lone wave issues one instruction
    // per four cycles: working a step's pointers out from its index cost ~15 scalar instructions, three times per interval).
    const long long jump_x = (cbase1 - cbase0) - (long long)nS * (P * DPAD);      // in elements
    const long long jump_n = (nbase1 - nbase0) - (long long)nS * P;
    auto src_next = [&](Src& X, int& v) __attribute__((always_inline)) {
        const int v1 = v + 1;
        const bool in = v1 < se;
        const bool cross = in && v1 == nS;
        X.x += (in ? P * DPAD : 0) + (cross ? jump_x : 0ll);
        X.nrm += (in ? P : 0) + (cross ? jump_n : 0ll);
        v = in ? v1 : v;
    };
    auto xbuf_byte = [&](int b) __attribute__((always_inline)) -> unsigned { return (unsigned)(b * L::XBUF_B + wave * L::SLICE_B); };
    const unsigned nvo = (unsigned)lane * 4u;
    const unsigned nrm_lds0 = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(L::NORM_OFF + wave * 64 * 4));
    const unsigned dma_lds0 = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(wave * L::SLICE_B));
    auto stage_norms = [&](const Src& X, int nbuf) __attribute__((always_inline)) {
        q32_dma_dword(X.nrm, nvo, nrm_lds0 + (unsigned)(nbuf * 4 * 64 * 4));
    };
    auto dma_block = [&](const Src& X, int b, int s) __attribute__((always_inline)) {
        q32_dma_pair(X.x, dvo[s][0], dvo[s][1], dma_lds0 + (unsigned)(b * L::XBUF_B) + 2048u * s);
    };
    // L2 prefetch.  The refill of a block is issued ~700-1000 cycles before its first use (the buffer holds the step under
    // the gradient until then) -- enough for an L2 hit, not for the Infinity Cache or HBM behind it (timing diagnostic with
    // every step fetching the same tile: 3 950 -> 2 900 cycles per step).  So each wave touches the lines of ITS slice of the
    // step PFD steps ahead with two plain loads (one dword per 128-byte line: 32 points x 3 lines of 192 features) into a
    // register nothing reads.  They are issued behind the interval's last LDS-DMA, so the counted waits of the next Gram
    // allow them in flight (PF_SLACK) and they have a whole interval to land before anything waits on something younger.
    constexpr int PFD = 4, PF_OPS = (32 * ((SW * 2 + 127) / 128) + 63) / 64, PF_SLACK = PF_OPS;
    unsigned pf_voff[PF_OPS];
#pragma unroll
    for (int i = 0; i < PF_OPS; ++i) {
        constexpr int LPP = (SW * 2 + 127) / 128;             // 128-byte lines per point in a wave's slice
        int idx = 64 * i + lane;
        if (idx >= 32 * LPP) idx = 32 * LPP - 1;
        // relative to the REFILL's pointer (step k + 2): PFD steps further, its 1 KiB bias taken back.  Across the gap between
        // the sets and past the range's end these touch lines nobody needs for PFD steps -- harmless, and always inside the
        // workspace (the images are followed by their lo copies and a slack region: mmd.hip, make_plan)
        pf_voff[i] = (unsigned)((idx / LPP) * DPAD * 2 + (idx % LPP) * 128) + (unsigned)(PFD * P * DPAD * 2 + 1024);
    }
    float pf_sink = 0.f;
    auto l2_prefetch = [&](const Src& X) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PF_OPS; ++i)
            asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_sink) : "v"(pf_voff[i]), "s"(X.x) : "memory");
    };

    // ------------------------------------------------------------------------------------------------------------------
    // The loop body is written as SLOTS: one matrix instruction, then the few other instructions that issue in its shadow
    // (a 32x32x16 occupies the matrix pipe for 32 cycles and the vector issue port for 8 of them: ~5 single-issue
    // instructions per slot are free, MI355X_MICROARCH.md).  hipcc clusters matrix instructions when left alone, and every
    // LDS-DMA / counted wait here is an asm statement it does not schedule across anyway, so each slot is pinned with
    // sched_barrier(0) on both sides: the instruction stream is the source order.  (First version of this kernel, one
    // region per k-step / feature block: 7.24 ms at N = 32 768, D = 768; its phase timer showed 1 400 and 1 600 cycles for
    // the two 768-cycle matrix phases and 900 cycles of exchange / barrier / fetch with the matrix pipe idle.)
    // ------------------------------------------------------------------------------------------------------------------

    // Gram slots.  Each k-step's operand is read from LDS ONCE and feeds both row blocks (the kernel is bound by the LDS
    // pipe -- 260 KB per step per CU against 128 B per clock, DESIGN 4.1h -- and reading it once per block was 48 KB of
    // that).  Local row block 1 (all foreign: its partial sums leave through LDS) runs LAG k-steps ahead of block 0, so
    // that it completes -- and its stores go out -- 2 LAG - 1 slots before the interval's last matrix instruction:
    //     (1,0) .. (1,LAG-1) | (0,0) (1,LAG) (0,1) (1,LAG+1) ... | (0,NK-LAG) .. (0,NK-1)
    constexpr int NS = 2 * NK;
    constexpr int LAG = NK < TVAE_Q32_LAG ? NK : TVAE_Q32_LAG;
    constexpr int GD = 6, RING = LAG + 3 < NK ? LAG + 4 : NK; // operand reads run GD slots ahead of their first use, in a ring of RING registers
    struct Sched {
        static constexpr int lb(int s) { return s < LAG ? 1 : (s < 2 * NK - LAG ? ((s - LAG) & 1) : 0); }
        static constexpr int ks(int s) { return s < LAG ? s : (s < 2 * NK - LAG ? (((s - LAG) & 1) ? LAG + (s - LAG) / 2 : (s - LAG) / 2) : s - NK); }
        static constexpr int first(int k) { return k < LAG ? k : LAG + 2 * (k - LAG) + 1; }         // slot of block 1's use of k-step k
        static constexpr int last(int k) { return k < NK - LAG ? LAG + 2 * k : NK + k; }             // ... and of block 0's
        static constexpr int rd(int k) { return first(k) - GD; }                                     // slot behind which it is read (< 0: end of the previous interval)
        static constexpr bool ring_ok() {
            for (int k = 0; k + RING < NK; ++k)
                if (rd(k + RING) < last(k)) return false;
            return true;
        }
        static constexpr int pre() { int n = 0; for (int k = 0; k < NK; ++k) n += rd(k) < 0 ? 1 : 0; return n; }
    };
    static_assert(NS >= 8 && 4 * KB >= 8 && Sched::ring_ok(), "slot schedule");
    constexpr int PRE = Sched::pre();                         // operands read at the end of the previous interval
    // Per-lane LDS addresses of a buffer's operands, worked out ONCE per interval and pinned in registers (hipcc otherwise
    // re-adds buffer base + lane offset in front of every read: 46 vector additions per step, 8 cycles each on a lone wave)
    struct Bases { int g[2]; int t[2][2]; int xw[3]; int xr; int nr; int ew; int er; };
    // chunk c = 0, 1, 2 of the exchange stores: local (block, half) = (1, 0), (1, 1), (0, 1)
    int xw_off[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int b = c < 2 ? 1 : 0, hf = c == 0 ? 0 : 1;
        const int owner = 2 * (b ^ wr) + (hf ^ wp);                          // wave-uniform
        xw_off[c] = lds0 + L::XCH_OFF + lane * 16 + ((owner * 3 + ((wave - owner - 1) & 3)) * 2) * 1024;
    }
    const int xr_off = lds0 + L::XCH_OFF + lane * 16 + wave * 3 * 2 * 1024;
    const int nr_off = lds0 + L::NORM_OFF + (wave * 64 + 16 * wp + 4 * h) * 4;
    const int ew_off = lds0 + L::EBUF_OFF + (wp * 2 + wr) * 1024 + lane * 16;
    const int er_off = lds0 + L::EBUF_OFF + lane * 16;
    // `b`: the step buffer (and norm / E buffer) of that parity; the exchange buffer in use is always the OTHER parity's
    // the addresses the Gram side of an interval uses (step buffer, exchange buffer and norms of parity b) ...
    auto gram_bases = [&](int b) __attribute__((always_inline)) -> Bases {
        Bases r{};
        const int xb0 = lds0 + (int)xbuf_byte(b);
#pragma unroll
        for (int i = 0; i < 2; ++i) { r.g[i] = xb0 + gram_off[i]; asm volatile("" : "+v"(r.g[i])); }
#pragma unroll
        for (int c = 0; c < 3; ++c) { r.xw[c] = xw_off[c] + b * L::XCH_ONE_B; asm volatile("" : "+v"(r.xw[c])); }
        r.xr = xr_off + b * L::XCH_ONE_B; asm volatile("" : "+v"(r.xr));
        r.nr = nr_off + b * 4 * 64 * 4; asm volatile("" : "+v"(r.nr));
        return r;
    };
    // ... and the gradient side (step buffer read transposed, Q pieces of parity b; g: slots 0 .. GD-1 of the next Gram)
    auto grad_bases = [&](int b) __attribute__((always_inline)) -> Bases {
        Bases r{};
        const int xb0 = lds0 + (int)xbuf_byte(b);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) { r.t[kk][j] = xb0 + tr_off[kk][j]; asm volatile("" : "+v"(r.t[kk][j])); }
        r.ew = ew_off + b * L::EBUF_ONE_B; asm volatile("" : "+v"(r.ew));
        r.er = er_off + b * L::EBUF_ONE_B; asm volatile("" : "+v"(r.er));
#pragma unroll
        for (int i = 0; i < 2; ++i) r.g[i] = xb0 + gram_off[i];
        return r;
    };
    f32x16 S[2];                                              // partial Gram of the step in flight, local (rotated) order
    q_f16x8 ab[RING];                                         // slots 0 .. GD-1 of the NEXT Gram are read at the end of an interval
    // reduce-scatter: local (block b', half hf') is actual (b' ^ wr, hf' ^ wp), finished by wave 2 (b' ^ wr) + (hf' ^ wp);
    // one 16-byte store = registers 8 hf + 4 q .. + 3
    auto xch_store = [&](const Bases& B, int c, int q) __attribute__((always_inline)) {
        const int b = c < 2 ? 1 : 0, hf = c == 0 ? 0 : 1;
        Q32_LDS(f32x4, B.xw[c] + q * 1024) = f32x4{S[b][8 * hf + 4 * q], S[b][8 * hf + 4 * q + 1], S[b][8 * hf + 4 * q + 2], S[b][8 * hf + 4 * q + 3]};
    };
    float xv[8];                                              // x = -t of the eight S' entries this lane finishes
    // the three remote partial sums and the point norms of registers 4 q .. 4 q + 3 (q = 0, 1: fetched and finished one
    // after the other -- 16 registers in flight instead of 32)
    f32x4 part[3], cn;
    auto xch_fetch = [&](const Bases& B, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int w = 0; w < 3; ++w) part[w] = Q32_LDS(const f32x4, B.xr + (w * 2 + q) * 1024);
        // register r = 4 q + i of this lane is point 16 wp + 8 q + 4 h + i
        cn = Q32_LDS(const f32x4, B.nr + 8 * q * 4);
    };
    // finishing registers 4 q .. 4 q + 3 is 20 operations, operation-major (idx = 4 op + i): op 0: the norm term, ops 1-3: own
    // partial (still in registers) + the three others in a fixed order, op 4: x
    float fsv[4], fcn[4];
    auto xch_finish_op = [&](int q, int idx) __attribute__((always_inline)) {
        const int i = idx & 3, op = idx >> 2, r = 4 * q + i;
        if (op == 0) fcn[i] = fmaf(cn[i], -a.inv_d2, rnx);
        else if (op == 1) fsv[i] = S[0][r] + part[0][i];
        else if (op == 2) fsv[i] = fsv[i] + part[1][i];
        else if (op == 3) fsv[i] = fsv[i] + part[2][i];
        else xv[r] = fmaf(c1, fsv[i], fcn[i]);
    };

    double sum0 = 0.0, sum1 = 0.0, rowsum = 0.0;              // per-lane fp64 partials: Q sums by column set, this lane's row sum

    // Gram operand of slot t of the step in buffer b (pass t / NK, k-step t % NK).  Pass 0 reads a block for the first time:
    // counted wait for its LDS-DMA (`slack` = DMA instructions issued after this step's)
    auto gram_rd = [&](const Bases& B, int ks, int slack) __attribute__((always_inline)) {
        if ((ks & 1) == 0) wait_vmcnt_n(2 * (KB - 1 - (ks >> 1)) + slack);
        ab[ks % RING] = Q32_LDS(const q_f16x8, B.g[ks & 1] + 2048 * (ks >> 1));
    };

    // gradient operands: transposed reads of block blk (read rd = 2 kk + j), the Q pieces
    q_f16x8 Bq[2][2];                                         // [kk][actual rbk]
    q_s16x4 gh[KB][2][2];
    auto grad_rd = [&](const Bases& B, int blk, int rd) __attribute__((always_inline)) {
        gh[blk][rd >> 1][rd & 1] = tr_read(B.t[rd >> 1][rd & 1] + 2048 * blk);
    };
    constexpr int TDB = 1;                                    // feature blocks whose operands are read ahead of the barrier
    constexpr int FH = 2 * KB;                                // gradient slot in which the second quad of partial sums is fetched
    constexpr int FW0 = (20 + FH - 2) / (FH - 1), FW1 = (20 + 4 * KB - FH - 3) / (4 * KB - FH - 2);      // finishing operations per slot
    static_assert(1 + 19 / FW0 < FH && FH + 2 + 19 / FW1 < 4 * KB, "finishing schedule");

    // Q work of a step, as 32 single operations: Horner steps of
    //     esc Q(x) = x^2 (k0 + k1 x + k2 x^2),   k = esc (1/2, 1/6, 1/24),   esc = +-2^12
    // (second order, no cancellation).  The kernel only runs when the device flag bounds every t of the call by 1/16
    // (mmd.hip, QFlag): the next term, x^3 / 120, is below 4e-6 of Q there -- 1.3e-7 of E, Q being t / 2 of it -- and 3e-10 of
    // Q at the t ~ 2 / D of the shapes that matter.  The fp16 scale of the Q pieces and the sign of prior columns (s_p = -1)
    // ride in the coefficients, which are scalars picked per step: no multiply, no sign flip afterwards.
    constexpr int QOPS = 32;
    float pe[8];
    float kq0 = 0.f, kq1 = 0.f, kq2 = 0.f;
    // OPERATION-major order (idx = 8 op + el): neighbouring instructions belong to different elements -- a dependent vector
    // instruction on a lone wave costs ~13 cycles against 8 for an independent one (the first slot version ran the
    // Horner steps of one element back to back: 84 cycles per slot instead of 40)
    auto e_op = [&](int idx) __attribute__((always_inline)) {
        const int el = idx % 8, op = idx / 8;
        const float x = xv[el];
        if (op == 0) pe[el] = fmaf(kq2, x, kq1);
        else if (op == 1) pe[el] = fmaf(pe[el], x, kq0);
        else pe[el] = pe[el] * x;
    };

    Src Xn2_c{};                                              // the refill's stream: step xv_c (interval k refills with step k + 2)
    int xv_c = 0, c_set_c = 0;

    // ---- phase A: Gram of the step in buffer b  ||  Q of the previous step (column set x_set) into E buffer ebuf; partial sums
    // of the Gram into exchange buffer xbuf.
    // No edge masks in the loop: rows past n_rows only ever feed their own (discarded) gradient rows and are kept out of the
    // sums by one select; the zero padding points past n_cols are taken out of the sums in closed form after the loop.
    auto phase_a = [&](const Bases& Bg, const Bases& Bt, int x_set, int knext) __attribute__((always_inline)) {
        constexpr bool WITH_E = true;
        constexpr int slack = PF_SLACK;                       // behind this step's LDS-DMA: only the L2 prefetches of the last interval
        constexpr int ESL = NS - 4;                           // Q operations spread over slots 0 .. ESL-1 (two per slot at D = 768)
        constexpr int EW = (QOPS + ESL - 1) / ESL;
        {
            // sign by integer arithmetic on the (wave-uniform) set index: stays in scalar registers
            const unsigned sbit = (unsigned)__builtin_amdgcn_readfirstlane(x_set) << 31;
            kq0 = __uint_as_float(__float_as_uint(SPLIT_E_SCALE / 2.f) ^ sbit);
            kq1 = __uint_as_float(__float_as_uint(SPLIT_E_SCALE / 6.f) ^ sbit);
            kq2 = __uint_as_float(__float_as_uint(SPLIT_E_SCALE / 24.f) ^ sbit);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            constexpr int S1_DONE = Sched::first(NK - 1);     // slot of block 1's last matrix instruction
            const int lb = Sched::lb(s), ks = Sched::ks(s);
            if (s == 0 || s == NK) {
                // pin the gradient accumulators to their accumulator registers through the Gram: left alone, hipcc parks two
                // of the blocks in vector registers for this phase and gives their registers to S (64 moves per step)
#pragma unroll
                for (int x = 0; x < KB; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) asm volatile("" : "+a"(acc[x][y]));
            }
            S[lb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab[ks % RING], zreg[lb][ks], ks == 0 ? zero16 : S[lb], 0, 0, 0);
            Q32_PIN();
#pragma unroll
            for (int k2 = 0; k2 < NK; ++k2)
                if (Sched::rd(k2) == s) gram_rd(Bg, k2, slack);
            if constexpr (WITH_E) {
                if (s < ESL) {
#pragma unroll
                    for (int i = 0; i < EW; ++i)
                        if (EW * s + i < QOPS) e_op(EW * s + i);
                } else if (s == ESL) {
                    // this lane's eight entries belong to ONE row: their sum serves the Q sums and the row sum.  pe = s_p 2^12 Q
                    float es = ((pe[0] + pe[1]) + (pe[2] + pe[3])) + ((pe[4] + pe[5]) + (pe[6] + pe[7]));
                    es = row_valid ? es * (1.f / SPLIT_E_SCALE) : 0.f;         // a power of two: exact
                    rowsum += (double)es;
                    // |es| is the (non-negative) Q sum of the step's set; the other set's sum gets an exact zero
                    sum0 += (double)__uint_as_float(__float_as_uint(es) & (x_set ? 0u : 0x7fffffffu));
                    sum1 += (double)__uint_as_float(__float_as_uint(es) & (x_set ? 0x7fffffffu : 0u));
                } else if (s == ESL + 1) {
                    q_f16x8 ev;
#pragma unroll
                    for (int r = 0; r < 8; ++r) ev[r] = (_Float16)pe[r];
                    Q32_LDS(q_f16x8, Bt.ew) = ev;
                }
            }
            // the foreign row block is complete after slot S1_DONE: its four stores two slots later (matrix-pipe latency)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (s == (S1_DONE + 2 + i < NS - 1 ? S1_DONE + 2 + i : NS - 1)) xch_store(Bg, i >> 1, i & 1);
            if constexpr (WITH_E) {
                // the gradient's first operand reads (wave-private step data: legal ahead of the barrier), two per slot
#pragma unroll
                for (int i = 0; i < 4 * TDB; ++i)
                    if (s == (NS - 2 * TDB + i / 2 > 0 ? NS - 2 * TDB + i / 2 : 0)) grad_rd(Bt, i >> 2, i & 3);
            }
            Q32_PIN();
        }
        if constexpr (WITH_E) {
            // the next interval's scalars, in the shadow of the last matrix instruction's latency
            c_set_c = sb + knext >= nS ? 1 : 0;
            src_next(Xn2_c, xv_c);
            Q32_PIN();
        }
        xch_store(Bg, 2, 0);
        xch_store(Bg, 2, 1);
    };

    // ---- phase C: gradient of the step in buffer b (Q pieces of ebuf; each 32-feature block refilled in place with step Xn
    // behind its matrix instructions)  ||  reduction of the next step's partial sums (exchange buffer xbuf, norms nbuf)
    auto phase_c = [&](const Bases& Bt, const Bases& Bg, int b, const Src& Xn) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rbk = 0; rbk < 2; ++rbk) Bq[kk][rbk] = Q32_LDS(const q_f16x8, Bt.er + (kk * 2 + rbk) * 1024);
        xch_fetch(Bg, 0);
        Q32_PIN();
        stage_norms(Xn, b);
        Q32_PIN();
#pragma unroll
        for (int s = 0; s < 4 * KB; ++s) {
            const int blk = s >> 2, kk = (s >> 1) & 1, rbk = s & 1;
            const q_f16x8 A = __builtin_bit_cast(q_f16x8, __builtin_shufflevector(gh[blk][kk][0], gh[blk][kk][1], 0, 1, 2, 3, 4, 5, 6, 7));
            acc[blk][rbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, Bq[kk][rbk], acc[blk][rbk], 0, 0, 0);
            Q32_PIN();
            if (blk + TDB < KB) grad_rd(Bt, blk + TDB, s & 3);
            // block blk - 1 is dead (its matrix instructions are issued, their operand reads have completed): refill it
            if ((s & 3) == 0 && blk >= 1) dma_block(Xn, b, blk - 1);
            // the 20 finishing operations of registers 0-3 in slots 1 .. H-1, the second quad's partial sums fetched in slot H,
            // its 20 operations in slots H+2 .. (two per slot at D = 768)
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                if (s == 1 + i / FW0) xch_finish_op(0, i);
                if (s == FH + 2 + i / FW1) xch_finish_op(1, i);
            }
            if (s == FH) xch_fetch(Bg, 1);
            Q32_PIN();
        }
        dma_block(Xn, b, KB - 1);
        // slots 0 .. GD-1 of the next interval's Gram: the step just refilled into this buffer
#pragma unroll
        for (int t = 0; t < PRE; ++t) gram_rd(Bt, t, 0);
        Q32_PIN();
        l2_prefetch(Xn);
        Q32_PIN();
    };

#ifdef TVAE_PHASE_TIMING
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    auto interval = [&](int k) __attribute__((always_inline)) {
        PHASE_MARK(5);
        const int cur = k & 1, nxt = cur ^ 1;
        // this interval's scalars were worked out in the previous one
        const Src Xn2 = Xn2_c;
        const int c_set = c_set_c;
        const Bases Bn = gram_bases(nxt), Bc = grad_bases(cur);
        phase_a(Bn, Bc, c_set, k + 1);
        PHASE_MARK(0);
        // the barrier needs the WRITES (Q piece, partial sums) retired: LDS operations retire in order, the gradient's
        // operand reads were issued before the last stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PHASE_MARK(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PHASE_MARK(2);
        phase_c(Bc, Bn, cur, Xn2);
        PHASE_MARK(4);
    };

    if (nsteps > 0) {
        // ---- prologue = interval -1 of the SAME loop body (one instance of the code: a second, E-less instance of phase A got
        // its own accumulator registers and the loop's then time-shared theirs with two gradient blocks, 64 register moves
        // per step).  Step sb is loaded into BOTH buffers: buffer 0 for the Gram, buffer 1 as the "previous step" whose
        // gradient interval -1 forms from Q = 0 (xv = 0: every Q, sum and E piece of it is an exact zero; a finite step
        // times zero adds exact zeros to the accumulators); its refill brings step sb + 1 into buffer 1.
        {
            const Src X0 = step_src(sb);
            stage_norms(X0, 1);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X0, 1, s);
            stage_norms(X0, 0);
#pragma unroll
            for (int s = 0; s < KB; ++s) dma_block(X0, 0, s);
#pragma unroll
            for (int r = 0; r < 8; ++r) xv[r] = 0.f;
            {
                const Bases B0 = grad_bases(0);
#pragma unroll
                for (int t = 0; t < PRE; ++t) gram_rd(B0, t, 0);
            }
            wait_vmcnt<0>();                                  // buffer 1 is read transposed at the end of phase A without a counted wait
            Q32_PIN();
            c_set_c = 0;
            xv_c = sb;
            Xn2_c = X0;
            src_next(Xn2_c, xv_c);                            // step sb + 1 (or sb again, if that is the last)
        }
        for (int k = -1; k < nsteps; ++k) interval(k);
        // The zero padding points of this workgroup's column range (points [n_cols, n_pad) of either set) went through the
        // loop like any other: x = -|z_i|^2 / D^2 exactly (zero Gram, zero norm), the same Q for every one of them, nothing
        // into the gradient (x_p = 0).  Take them out of the sums: this lane's row saw (its half of) the padding points of
        // each 16-point half it finishes -- count them per set and subtract count * Q(x).
        {
            int npad[2] = {0, 0};
#pragma unroll 1
            for (int v = sb; v < se; ++v) {
                int set, p0;
                step_meta(v, set, p0);
                if (p0 + P <= a.n_cols) continue;
#pragma unroll
                for (int el = 0; el < 8; ++el) npad[set] += (p0 + 16 * wp + 8 * (el >> 2) + 4 * h + (el & 3) >= a.n_cols) ? 1 : 0;
            }
            if ((npad[0] | npad[1]) != 0 && row_valid) {
                const double qp = (double)em1_split(-rnx).q;
                sum0 -= npad[0] * qp;
                sum1 -= npad[1] * qp;
                rowsum -= (npad[0] - npad[1]) * qp;
            }
        }
    }
    wait_vmcnt<0>();
    asm volatile("" : "+v"(pf_sink));          // the prefetch sink stays allocated until every load has returned
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

// ------------------------------------------------------------------------------------------------------------------
// Sum-only companion (the y-rows launch of a gradient call: y-y, each unordered pair of 64-point blocks once when the row
// range allows; both problems of a call without gradient), the Q-form member of ITS launch pair: same one-product 32x32x16
// Gram, same LDS image, exchange and slot discipline as mmd_q32_kernel; no gradient phase, so an interval is
//
//     Gram(k+1), two passes  ||  finish the partial sums of step k, Q, sums        -> stores of S'(k+1) -> ONE barrier
//
// Two step buffers: the one under the Gram is refilled block by block behind pass 1 with step k + 3 (the other holds
// step k + 2), so every LDS-DMA has more than an interval to land.  Norms in a ring of four.  The symmetric enumeration,
// column splits and outputs are those of mmd_pipe_sum_kernel (mmd_pipe.hip).
// ------------------------------------------------------------------------------------------------------------------
template <int KB>
struct Q32SumLds {
    static constexpr int SLICE_B = KB * 2048;
    static constexpr int XBUF_B = 4 * SLICE_B;
    static constexpr int NORM_OFF = 2 * XBUF_B;               // [4 ring][4 waves][64] floats
    static constexpr int XCH_OFF = NORM_OFF + 4 * 4 * 64 * 4;
    static constexpr int XCH_ONE_B = 4 * 3 * 2 * 1024;
    static constexpr int BYTES = XCH_OFF + 2 * XCH_ONE_B;
    static_assert(BYTES <= 160 * 1024, "q32 sum kernel exceeds the 160 KiB LDS");
};

template <int KB>
__global__ __launch_bounds__(256, 1) void mmd_q32_sum_kernel(MmdArgs a) {
    if (__builtin_amdgcn_readfirstlane((int)*a.qflag) == 0) return;
    using L = Q32SumLds<KB>;
    constexpr int P = 32, SW = 32 * KB, DPAD = 4 * SW, NK = 2 * KB, NS = 2 * NK, SPB = 64 / P;
    // every k-step's operand is read ONCE for both row blocks; block 1 runs LAG k-steps ahead (see mmd_q32_kernel)
    constexpr int LAG = NK < TVAE_Q32_SUM_LAG ? NK : TVAE_Q32_SUM_LAG;
    constexpr int GD = 3, RING = LAG + 1 < NK ? LAG + 2 : NK;
    struct Sched {
        static constexpr int lb(int s) { return s < LAG ? 1 : (s < 2 * NK - LAG ? ((s - LAG) & 1) : 0); }
        static constexpr int ks(int s) { return s < LAG ? s : (s < 2 * NK - LAG ? (((s - LAG) & 1) ? LAG + (s - LAG) / 2 : (s - LAG) / 2) : s - NK); }
        static constexpr int first(int k) { return k < LAG ? k : LAG + 2 * (k - LAG) + 1; }
        static constexpr int last(int k) { return k < NK - LAG ? LAG + 2 * k : NK + k; }
        static constexpr int rd(int k) { return first(k) - GD; }
        static constexpr bool ring_ok() {
            for (int k = 0; k + RING < NK; ++k)
                if (rd(k + RING) < last(k)) return false;
            return true;
        }
        static constexpr int pre() { int n = 0; for (int k = 0; k < NK; ++k) n += rd(k) < 0 ? 1 : 0; return n; }
        // the last operand read that waits on this step's own LDS-DMA (even k-steps do): the interval's own DMA goes out behind it
        static constexpr int rd_last() { const int r = rd(NK - 2); return r < 0 ? 0 : r; }
        // refill of block blk: behind block 1's use of its second k-step (the operand reads have returned) and behind rd_last
        static constexpr int refill(int blk) { const int a = first(2 * blk + 1), b = rd_last() + 1 + blk; return a > b ? a : b; }
    };
    static_assert(Sched::ring_ok() && Sched::refill(KB - 1) < 2 * NK, "slot schedule");
    constexpr int PRE = Sched::pre();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wp = wave & 1;
    const int n32 = lane & 31, h = lane >> 5;
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lds0 = (int)smem_byte;

    const int per_problem = a.row_blocks * a.col_splits;
    const int problem = a.problem_base + (int)blockIdx.x / per_problem;
    const int rem = (int)blockIdx.x - (problem - a.problem_base) * per_problem;
    const int rb = a.sp_major ? rem % a.row_blocks : rem / a.col_splits;
    const int sp = a.sp_major ? rem / a.row_blocks : rem - rb * a.col_splits;
    const int i0 = rb * 64;
    const long long bid = (long long)a.sums_base + blockIdx.x;

    const _Float16* rows = reinterpret_cast<const _Float16*>(a.cols[problem]) + (long long)a.row_offset * DPAD;
    const float* rnorm = a.cnorm[problem] + a.row_offset;
    q_f16x8 zreg[2][NK];                                      // rotated block order: see mmd_q32_kernel
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
    const float c1 = 2.f * ss.inv_s2 * a.inv_d2;
    float rnx = row_valid ? -rnorm[my_row] * a.inv_d2 : 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) asm volatile("" : "+v"(zreg[b][ks]));
    asm volatile("" : "+v"(rnx));
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    asm volatile("" : "+v"(zero16));

    // steps of 32 points (mmd_pipe_sum_kernel: problem 0 = z rows against both sets; problem 1 = y rows against the priors,
    // all of them or, sym, each unordered pair of 64-point blocks once with weight 2 off the diagonal)
    const int nS = a.tiles_per_set / 2;
    const int nJ = nS / SPB;
    const bool sym = problem == 1 && a.sym;
    const int blkI = (a.row_offset >> 6) + rb;
    const int jlo = (int)((long long)sp * nJ / a.col_splits), jhi = (int)((long long)(sp + 1) * nJ / a.col_splits);
    const int aend = blkI < jhi ? blkI : jhi;
    const int ja0 = jlo + ((jlo ^ (blkI + 1)) & 1);
    const int nA = aend > ja0 ? (aend - ja0 + 1) >> 1 : 0;
    const int nB = (blkI >= jlo && blkI < jhi) ? 1 : 0;
    const int cbeg = blkI + 1 > jlo ? blkI + 1 : jlo;
    const int jc0 = cbeg + ((cbeg ^ blkI) & 1);
    const int nC = jhi > jc0 ? (jhi - jc0 + 1) >> 1 : 0;
    const long long total = problem == 0 ? 2ll * nS : (long long)nS;
    const long long base = problem == 0 ? 0 : nS;
    const int sb = sym ? 0 : (int)(base + sp * total / a.col_splits);
    const int se = sym ? (nA + nB + nC) * SPB : (int)(base + (sp + 1) * total / a.col_splits);
    const int nsteps = se - sb;

    struct Meta { int set, p0; float w; };
    auto step_meta = [&](int v) __attribute__((always_inline)) -> Meta {
        const int u = v < se ? v : se - 1;
        if (sym) {
            const int blk = u / SPB, sub = u - blk * SPB;
            const int J = blk < nA ? ja0 + 2 * blk : (blk < nA + nB ? blkI : jc0 + 2 * (blk - nA - nB));
            return Meta{1, J * 64 + sub * P, J != blkI ? 2.f : 1.f};
        }
        const int set = u >= nS ? 1 : 0;
        return Meta{set, (u - (set ? nS : 0)) * P, 1.f};
    };
    struct Src { const _Float16* x; const float* nrm; };
    const _Float16* const cbase0 = reinterpret_cast<const _Float16*>(a.cols[0]) + wave * SW;
    const _Float16* const cbase1 = reinterpret_cast<const _Float16*>(a.cols[1]) + wave * SW;
    const float* const nbase0 = a.cnorm[0];
    const float* const nbase1 = a.cnorm[1];
    auto step_src = [&](int v) __attribute__((always_inline)) -> Src {
        const Meta m = step_meta(v);
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(m.p0) * (unsigned)DPAD;
        return Src{(m.set ? cbase1 : cbase0) + off, (m.set ? nbase1 : nbase0) + m.p0};
    };

    unsigned dma_voff0, dma_voff1;
    {
        const int r = lane >> 2, pos = lane & 3;
        dma_voff0 = (unsigned)(r * DPAD + 8 * (pos ^ q32_swz(r >> 2))) * 2u;
        dma_voff1 = (unsigned)((16 + r) * DPAD + 8 * (pos ^ q32_swz(4 + (r >> 2)))) * 2u - 1024u;
    }
    int gram_off[2];
    {
        const int pa = n32 ^ (16 * wp);
#pragma unroll
        for (int par = 0; par < 2; ++par)
            gram_off[par] = 1024 * (pa >> 4) + 64 * (pa & 15) + 16 * ((2 * par + h) ^ q32_swz(pa >> 2));
    }
    auto xbuf_byte = [&](int b) __attribute__((always_inline)) -> unsigned { return (unsigned)(b * L::XBUF_B + wave * L::SLICE_B); };
    auto stage_norms = [&](const Src& X, int ring) __attribute__((always_inline)) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(L::NORM_OFF + (ring * 4 + wave) * 64 * 4));
        dma4(uniform_ptr(X.nrm), (unsigned)lane * 4u, dst);
    };
    auto dma_block = [&](const Src& X, int b, int s) __attribute__((always_inline)) {
        dma16_pair(reinterpret_cast<const float*>(X.x + 32 * s), dma_voff0, dma_voff1, smem_byte + xbuf_byte(b) + 2048u * s);
    };
    int xw_off[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int b = c < 2 ? 1 : 0, hf = c == 0 ? 0 : 1;
        const int owner = 2 * (b ^ wr) + (hf ^ wp);
        xw_off[c] = lds0 + L::XCH_OFF + lane * 16 + ((owner * 3 + ((wave - owner - 1) & 3)) * 2) * 1024;
    }
    const int xr_off = lds0 + L::XCH_OFF + lane * 16 + wave * 3 * 2 * 1024;
    const int nr_off = lds0 + L::NORM_OFF + (wave * 64 + 16 * wp + 4 * h) * 4;

    f32x16 S[2];
    q_f16x8 ab[RING];
    float xv[8], fsv[8], fcn[8], pe[8];
    f32x4 part[2][3], cn[2];
    double sum0 = 0.0, sum1 = 0.0;
    float fq0 = 0.f, fq1 = 0.f;
    auto flush_sums = [&]() __attribute__((always_inline)) {
        sum0 += (double)(row_valid ? fq0 : 0.f); sum1 += (double)(row_valid ? fq1 : 0.f);
        fq0 = 0.f; fq1 = 0.f;
    };

    // one interval: Gram of the step in buffer b (tile k + 1), refilled behind pass 1 with step Xr (norms into ring slot
    // ring_r); the partial sums of step k (exchange buffer xb_rd, norms in ring slot ring_e, meta m) are finished and summed
    // between the matrix instructions; the partial sums of the Gram go to exchange buffer xb_wr.
    // idx list of the vector work, operation-major over the eight entries of a lane:
    //   [0, 40)  finishing (op 0: norm term; 1-3: own + three remote partials; 4: x)
    //   [40, 72) Q(x) = y (1/2 + x/6 + y/24), y = x^2, SUMMED as it is formed: four fused-multiply-add chains of two entries
    //            (only the sum over the lane's entries is needed here; truncation as in mmd_q32_kernel)
    constexpr int VOPS = 72;
    float e4[4];
    auto v_op = [&](int idx) __attribute__((always_inline)) {
        if (idx < 40) {
            const int r = idx & 7, op = idx >> 3;
            if (op == 0) fcn[r] = fmaf(cn[r >> 2][r & 3], -a.inv_d2, rnx);
            else if (op == 1) fsv[r] = S[0][r] + part[r >> 2][0][r & 3];
            else if (op == 2) fsv[r] = fsv[r] + part[r >> 2][1][r & 3];
            else if (op == 3) fsv[r] = fsv[r] + part[r >> 2][2][r & 3];
            else xv[r] = fmaf(c1, fsv[r], fcn[r]);
        } else {
            const int el = (idx - 40) & 7, op = (idx - 40) >> 3;
            const float x = xv[el];
            if (op == 0) pe[el] = fmaf(1.f / 6.f, x, 0.5f);
            else if (op == 1) fcn[el] = x * x;
            else if (op == 2) pe[el] = fmaf(1.f / 24.f, fcn[el], pe[el]);
            else e4[el & 3] = el < 4 ? pe[el] * fcn[el] : fmaf(pe[el], fcn[el], e4[el & 3]);
        }
    };
    // the own partial S[0][0..7] of step k is consumed by op 1 above; the Gram of step k + 1 overwrites S[0] in pass 1 (slots
    // NK ..): op 1 has to be placed before slot NK -- the list puts idx 8 .. 15 into slots <= 5
    constexpr int V0 = 0;                                     // first slot with vector work (the partial sums' LDS round trip before it)
    constexpr int VSL = NS - 3 - V0;                          // vector work spread over slots V0 .. V0 + VSL - 1
    constexpr int VW = (VOPS + VSL - 1) / VSL;
    static_assert(V0 + (15 / VW) < LAG, "the own partial sums are read before block 0's first matrix instruction overwrites them");
    auto interval = [&](int b, const Src& Xr, int ring_r, int xb_rd, int xb_wr, int ring_e, const Meta& m, bool with_e, int slack) __attribute__((always_inline)) {
        const int gb0 = lds0 + (int)xbuf_byte(b);
        int gbase[2] = {gb0 + gram_off[0], gb0 + gram_off[1]};
        asm volatile("" : "+v"(gbase[0]), "+v"(gbase[1]));
        int xw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { xw[c] = xw_off[c] + xb_wr * L::XCH_ONE_B; asm volatile("" : "+v"(xw[c])); }
        auto gram_rd = [&](int ks) __attribute__((always_inline)) {
            if ((ks & 1) == 0) wait_vmcnt_n(2 * (KB - 1 - (ks >> 1)) + slack);
            ab[ks % RING] = Q32_LDS(const q_f16x8, gbase[ks & 1] + 2048 * (ks >> 1));
        };
        {
            const int xr = xr_off + xb_rd * L::XCH_ONE_B, nr = nr_off + ring_e * 4 * 64 * 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int w = 0; w < 3; ++w) part[q][w] = Q32_LDS(const f32x4, xr + (w * 2 + q) * 1024);
                cn[q] = Q32_LDS(const f32x4, nr + 8 * q * 4);
            }
        }
#pragma unroll
        for (int t = 0; t < PRE; ++t) gram_rd(t);
        Q32_PIN();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            constexpr int S1_DONE = Sched::first(NK - 1);
            const int lb = Sched::lb(s), ks = Sched::ks(s);
            S[lb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab[ks % RING], zreg[lb][ks], ks == 0 ? zero16 : S[lb], 0, 0, 0);
            Q32_PIN();
#pragma unroll
            for (int k2 = 0; k2 < NK; ++k2)
                if (Sched::rd(k2) == s) gram_rd(k2);
            if (s == Sched::rd_last()) stage_norms(Xr, ring_r);      // behind the last counted wait: the counts above stay exact
            if (s >= V0 && s < V0 + VSL) {
#pragma unroll
                for (int i = 0; i < VW; ++i)
                    if (VW * (s - V0) + i < VOPS) v_op(VW * (s - V0) + i);
            } else if (s == V0 + VSL) {
                // fp32 partials over eight intervals, then fp64 (flush_sums); the weight and the set are wave-uniform
                // (interval -1 finishes whatever the exchange buffer held: a select, not a zero weight, keeps a NaN out)
                const float es = with_e ? (e4[0] + e4[1]) + (e4[2] + e4[3]) : 0.f;
                fq0 = fmaf(es, m.set ? 0.f : m.w, fq0);
                fq1 = fmaf(es, m.set ? m.w : 0.f, fq1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (s == (S1_DONE + 2 + i < NS - 1 ? S1_DONE + 2 + i : NS - 1))
                    Q32_LDS(f32x4, xw[i >> 1] + (i & 1) * 1024) = f32x4{S[1][8 * (i >> 1) + 4 * (i & 1)], S[1][8 * (i >> 1) + 4 * (i & 1) + 1],
                                                                          S[1][8 * (i >> 1) + 4 * (i & 1) + 2], S[1][8 * (i >> 1) + 4 * (i & 1) + 3]};
            // both k-steps of block blk are in registers: refill it with the same block of step Xr
#pragma unroll
            for (int blk = 0; blk < KB; ++blk)
                if (Sched::refill(blk) == s) dma_block(Xr, b, blk);
            Q32_PIN();
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            Q32_LDS(f32x4, xw[2] + q * 1024) = f32x4{S[0][8 + 4 * q], S[0][8 + 4 * q + 1], S[0][8 + 4 * q + 2], S[0][8 + 4 * q + 3]};
    };

    if (nsteps > 0) {
        // prologue: steps sb, sb + 1 into the two buffers (norm ring slots 0, 1), then interval "-1": Gram of step sb with
        // nothing to finish (with_e false: its sums are discarded; the registers it reads hold zeros)
        const Src X0 = step_src(sb), X1 = step_src(sb + 1);
        stage_norms(X0, 0);
#pragma unroll
        for (int s = 0; s < KB; ++s) dma_block(X0, 0, s);
        stage_norms(X1, 1);
#pragma unroll
        for (int s = 0; s < KB; ++s) dma_block(X1, 1, s);
#pragma unroll
        for (int r = 0; r < 8; ++r) { xv[r] = 0.f; pe[r] = 0.f; fsv[r] = 0.f; fcn[r] = 0.f; e4[r & 3] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[0][r] = 0.f; S[1][r] = 0.f; }
        wait_vmcnt<0>();
        __syncthreads();
        // k = -1: buffer 0 holds step sb; its refill is step sb + 2 (ring slot 2)
        // interval k: Gram of step k + 1 in buffer (k + 1) & 1, refill with step k + 3, finishing step k
        for (int k = -1; k < nsteps; ++k) {
            const int b = (k + 1) & 1;
            const Src Xr = step_src(sb + k + 3);
            const Meta m = step_meta(sb + (k < 0 ? 0 : k));
            // DMA in flight on entry beside this step's own (older): the norms + 2 KB pieces of the step in the other buffer
            interval(b, Xr, (k + 3) & 3, k & 1, (k + 1) & 1, k & 3, m, k >= 0, 1 + 2 * KB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if ((k & 7) == 7) flush_sums();
        }
        flush_sums();
        // the zero padding points of this workgroup's column range: out of the sums in closed form (see mmd_q32_kernel)
        {
            float npad[2] = {0.f, 0.f};
#pragma unroll 1
            for (int v = sb; v < se; ++v) {
                const Meta m = step_meta(v);
                if (m.p0 + P <= a.n_cols) continue;
#pragma unroll
                for (int el = 0; el < 8; ++el) npad[m.set] += (m.p0 + 16 * wp + 8 * (el >> 2) + 4 * h + (el & 3) >= a.n_cols) ? m.w : 0.f;
            }
            if ((npad[0] != 0.f || npad[1] != 0.f) && row_valid) {
                const float x = rnx;
                const float y = x * x;
                const double qp = (double)(fmaf(1.f / 24.f, y, fmaf(1.f / 6.f, x, 0.5f)) * y);
                sum0 -= (double)npad[0] * qp;
                sum1 -= (double)npad[1] * qp;
            }
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

template <int KB>
int launch_q32_sum(const MmdArgs& a, int nblocks, hipStream_t s) {
    // the 128-row member without a K-split (mmd_q32s.hip) when the symmetric choice is the same for both block sizes
    // (a.sym2: mmd.hip) and it has more than half a chip of workgroups: it takes 1.05x the 64-row kernel's time per
    // workgroup for twice the rows, so it wins as soon as the 64-row launch would need a second round of workgroups
    const int problems = nblocks / (a.row_blocks * a.col_splits);
    const int rb2 = (a.n_rows + 127) / 128;
    if (TVAE_Q32_SUM_ROWS128 && a.sym2 == a.sym && (long long)rb2 * a.col_splits * problems > 128) {
        MmdArgs b = a;
        b.launch_blocks = a.row_blocks * a.col_splits;       // entries per problem the finalize pass reads
        b.row_blocks = rb2;
        return launch_mmd_q32_sum3(KB, b, rb2 * a.col_splits * problems, s);
    }
    constexpr int lds_bytes = Q32SumLds<KB>::BYTES;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_q32_sum_kernel<KB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   lds_bytes);
    }));
    hipLaunchKernelGGL((mmd_q32_sum_kernel<KB>), dim3(nblocks), dim3(256), lds_bytes, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
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

// the Q-form member of the sum-only launch pair
int launch_mmd_q32_sum(int kb, const MmdArgs& a, int nblocks, hipStream_t s) {
    switch (kb) {
        case 2: return launch_q32_sum<2>(a, nblocks, s);
        case 3: return launch_q32_sum<3>(a, nblocks, s);
        case 4: return launch_q32_sum<4>(a, nblocks, s);
        case 5: return launch_q32_sum<5>(a, nblocks, s);
        case 6: return launch_q32_sum<6>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no q32 sum kernel for %d feature blocks", kb);
}

}  // namespace tvae
