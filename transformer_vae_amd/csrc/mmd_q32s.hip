// Sum-only member of the 32-point-step kernels without a K-split (see below).  A file of its own: it is built with
// -mllvm -amdgpu-mfma-vgpr-form, so that the matrix instructions accumulate in VECTOR registers -- the finishing arithmetic
// reads every accumulator entry, and from accumulation registers each read is one more vector instruction in a loop whose
// budget is three per matrix instruction; mmd_q32.hip's gradient kernel wants the opposite.
#include <type_traits>

#include "mmd_common.h"
#include "mmd_q32.h"

namespace tvae {
namespace {

#ifndef TVAE_S3_ZA
#define TVAE_S3_ZA 20      // sum3: k-steps (of 48 at D = 768) of the row operand kept in accumulation registers
#endif
#ifndef TVAE_S3_GD
#define TVAE_S3_GD 4
#endif

// ------------------------------------------------------------------------------------------------------------------
// Sum-only launches, 128 rows per workgroup, NO K-split: wave w owns rows 32 w .. 32 w + 31 with ALL their features in
// registers (4 KB blocks: 192 registers at D = 768 -- there are no gradient accumulators to make room for here), and every
// wave multiplies them with the whole 32-point step, so a step is one chain of 4 NK matrix instructions into one 32x32
// accumulator per wave: no exchange of partial sums, no store/barrier/load round trip per step, and the finishing of step
// k - 1 (16 entries per lane, six operations each) rides in the slots of step k's chain (two accumulators take turns).
// The step image in LDS is the same as above (wave w's LDS-DMA fills feature slice w); every wave READS all four slices,
// its own first: its own LDS-DMA is known to have landed after a counted vmcnt wait, so the one barrier per step sits behind
// the first slice's matrix instructions, where the waves are in step anyway, and says two things at once: the other
// slices of this step have landed (each wave waited for its own before arriving), and every wave has left the previous
// step's buffer, which is refilled behind it with step k + 2 (three step buffers: each LDS-DMA has more than a step to
// land).  Norms: a private copy per wave in a ring of four.  Problems, column splits and the closed-form padding correction
// are those of mmd_q32_sum_kernel (mmd_q32.hip); the symmetric enumeration runs over 128-row / 128-point blocks (the set's
// image is padded to whole 128-point blocks: mmd.hip, make_plan), and the workgroup writes one wg_sums entry and zeroes the
// entries its 64-row partner in the launch pair would have written beyond it.
// ------------------------------------------------------------------------------------------------------------------
template <int KB>
struct Q32Sum3Lds {
    static constexpr int SLICE_B = KB * 2048;
    static constexpr int XBUF_B = 4 * SLICE_B;
    static constexpr int NBUF = KB <= 6 ? 3 : 2;              // step buffers (64 KB each at D = 1 024: two)
    static constexpr int NORM_OFF = NBUF * XBUF_B;            // [4 ring][4 waves][64] floats
    static constexpr int RED_OFF = NORM_OFF + 4 * 4 * 64 * 4;
    static constexpr int BYTES = RED_OFF + 8 * 8;
    static_assert(BYTES <= 160 * 1024, "q32 sum3 kernel exceeds the 160 KiB LDS");
};

template <int KB>
__global__ __launch_bounds__(256, 1) void mmd_q32_sum3_kernel(MmdArgs a) {
    if (__builtin_amdgcn_readfirstlane((int)*a.qflag) == 0) return;
    using L = Q32Sum3Lds<KB>;
    constexpr int P = 32, SW = 32 * KB, DPAD = 4 * SW, NK = 2 * KB, NS = 4 * NK, SPB = 128 / P;
    constexpr int GD = TVAE_S3_GD < NK ? TVAE_S3_GD : NK - 1, RING = 8;       // an operand is read GD slots ahead of its matrix instruction; NS is a multiple of RING: the ring runs on across steps
    static_assert(GD + 2 <= RING && NS % RING == 0, "operand ring");
    constexpr int BAR = NK - GD - 1;                          // the barrier closes this slot: the first read of a foreign slice is issued in the next one
    static_assert(BAR >= 0, "slot schedule");
    constexpr int NG = (KB + 1) / 2;                          // refill: LDS-DMA groups of two 32-feature blocks (4 KiB per wave) behind the barrier, then the norms
    constexpr int DLAST = NS - GD - 2;                        // all of it ahead of the counted wait of slot NS - GD - 1
    constexpr int NBUF = L::NBUF, AHEAD = NBUF - 1;           // the refill brings step k + AHEAD
    // three buffers: the refill has two steps to land, spread it; two buffers: it is read at the end of THIS step -- out at once
    constexpr int DSTRIDE = NBUF == 2 ? 1 : (DLAST - BAR - 1) / NG;
    static_assert(DSTRIDE >= 1 && BAR + 1 + NG * DSTRIDE + (NBUF == 2 ? 1 : 0) <= DLAST, "slot schedule");
    // two buffers: the refill has most of one step (~1 us) to land, an L2 hit's worth; each wave touches the lines of ITS
    // slice of the step one BLOCK further on with plain loads into a register nothing reads (mmd_q32_kernel, L2 prefetch)
    constexpr int PF_OPS = NBUF == 2 ? (32 * ((SW * 2 + 127) / 128) + 63) / 64 : 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sm = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, h = lane >> 5;
    const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lds0 = (int)smem_byte;

    const int per_problem = a.row_blocks * a.col_splits;
    const int problem = a.problem_base + (int)blockIdx.x / per_problem;
    const int rem = (int)blockIdx.x - (problem - a.problem_base) * per_problem;
    const int rb = a.sp_major ? rem % a.row_blocks : rem / a.col_splits;
    const int sp = a.sp_major ? rem / a.row_blocks : rem - rb * a.col_splits;
    const int i0 = rb * 128;
    // a.launch_blocks: workgroups per problem of the 64-row member of this launch pair -- the finalize pass reads that many
    // entries per problem; this member fills the first per_problem of them and zeroes the rest
    const long long bid = (long long)a.sums_base + (long long)(problem - a.problem_base) * a.launch_blocks + rem;

    const _Float16* rows = reinterpret_cast<const _Float16*>(a.cols[problem]) + (long long)a.row_offset * DPAD;
    const float* rnorm = a.cnorm[problem] + a.row_offset;
    // the wave's 32 rows, all features: zreg[j] is feature slice (wave + j) % 4 -- the order the step's slices are read in
    q_f16x8 zreg[4][NK];
    const int my_row = i0 + 32 * wave + n32;
    const bool row_valid = my_row < a.n_rows;
    {
        const _Float16* zp = rows + (long long)my_row * DPAD + 8 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sl = (wave + j) & 3;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                if (row_valid) zreg[j][ks] = *reinterpret_cast<const q_f16x8*>(zp + sl * SW + 16 * ks);
                else zreg[j][ks] = q_f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    float rnx = row_valid ? -rnorm[my_row] * a.inv_d2 : 0.f;
    const SplitScales ss = split_scales(*a.split_amax);
    const float c1 = 2.f * ss.inv_s2 * a.inv_d2;
    // register budget: the row operand alone is 8 NK registers of the 256 vector registers the finishing arithmetic can
    // address; its last ZA k-steps live in accumulation registers (a matrix instruction takes its B operand from either
    // file), the two accumulators in vector registers (-amdgpu-mfma-vgpr-form: head of the file)
    constexpr int ZA = KB <= 6 ? TVAE_S3_ZA * NK / 12 : 4 * NK - 24;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (j * NK + ks >= 4 * NK - ZA) asm volatile("" : "+a"(zreg[j][ks]));
            else asm volatile("" : "+v"(zreg[j][ks]));
        }
    asm volatile("" : "+v"(rnx));
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    asm volatile("" : "+v"(zero16));

    // steps of 32 points (problem 0 = z rows against both sets; problem 1 = y rows against the priors, all of them or, sym,
    // each unordered pair of 128-point blocks once with weight 2 off the diagonal)
    const int nS = a.tiles_per_set / 2;
    const int nJ = nS / SPB;
    const bool sym = problem == 1 && a.sym;
    const int blkI = (a.row_offset >> 7) + rb;
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
    // column ranges on whole SPB-step blocks in both enumerations (this kernel's own choice: the ranges tile [0, total) either way)
    const int sb = sym ? 0 : (int)(base + SPB * (sp * (total / SPB) / a.col_splits));
    const int se = sym ? (nA + nB + nC) * SPB : (int)(base + SPB * ((sp + 1) * (total / SPB) / a.col_splits));
    const int nsteps = se - sb;

    struct Meta { int set, p0; float w; };
    auto step_meta = [&](int v) __attribute__((always_inline)) -> Meta {
        const int u = v < se ? v : se - 1;
        if (sym) {
            const int blk = u / SPB, sub = u - blk * SPB;
            const int J = blk < nA ? ja0 + 2 * blk : (blk < nA + nB ? blkI : jc0 + 2 * (blk - nA - nB));
            return Meta{1, J * 128 + sub * P, J != blkI ? 2.f : 1.f};
        }
        const int set = u >= nS ? 1 : 0;
        return Meta{set, (u - (set ? nS : 0)) * P, 1.f};
    };
    // The column stream in BLOCKS of SPB = 4 consecutive steps with one set and one weight (a visited 128-point block of
    // the symmetric enumeration; four steps of [sb, se) otherwise -- nS and sb are multiples of SPB, so a block never
    // straddles the sets).  The loop below runs one block per iteration with NO branch but its back edge: a taken branch
    // costs this lone wave ~100 cycles of instruction fetch (phase timer), the general step_meta above ~50 scalar
    // instructions at four cycles each.  Column pointers carry the wave's feature slice and the -3 KiB bias of dvo.
    const _Float16* const cbase0 = reinterpret_cast<const _Float16*>(a.cols[0]) + wave * SW - 1536;
    const _Float16* const cbase1 = reinterpret_cast<const _Float16*>(a.cols[1]) + wave * SW - 1536;
    const float* const nbase0 = a.cnorm[0];
    const float* const nbase1 = a.cnorm[1];
    const int nblk = nsteps / SPB;
    struct Blk { const _Float16* x; const float* nrm; int set; float w; };
    auto blk_desc = [&](int b_) __attribute__((always_inline)) -> Blk {
        const int b = b_ < nblk ? b_ : nblk - 1;              // past the end: the last block again (its Gram is discarded)
        int Ja = ja0 + 2 * b, Jc = jc0 + 2 * (b - nA - nB);
        asm volatile("" : "+s"(Ja), "+s"(Jc));                // both formed, then selected: no branch
        const int J = b < nA ? Ja : (b < nA + nB ? blkI : Jc);
        const int v0 = sb + SPB * b;
        const int set = sym ? 1 : (v0 >= nS ? 1 : 0);
        const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane(sym ? J * 128 : (v0 - (set ? nS : 0)) * P);
        return Blk{(set ? cbase1 : cbase0) + p0 * (unsigned)DPAD, (set ? nbase1 : nbase0) + p0, set, (sym && J != blkI) ? 2.f : 1.f};
    };

    // per-lane byte offsets of the LDS-DMA of (32-feature block b, 16-point piece hh): lane = (point r, chunk position pos);
    // + 3 KiB bias - the instruction offset the piece is issued with (q32_dma_quad)
    unsigned dvo[KB][2];
    {
        const int r = lane >> 2, pos = lane & 3;
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                dvo[b][hh] = (unsigned)((16 * hh + r) * DPAD + 32 * b + 8 * (pos ^ q32_swz(4 * hh + (r >> 2)))) * 2u + 3072u - (unsigned)(2048 * (b & 1) + 1024 * hh);
    }
    const unsigned nvo = (unsigned)lane * 4u;
    unsigned pf_voff[PF_OPS > 0 ? PF_OPS : 1];
#pragma unroll
    for (int i = 0; i < PF_OPS; ++i) {
        constexpr int LPP = (SW * 2 + 127) / 128;             // 128-byte lines per point in a wave's slice
        int idx = 64 * i + lane;
        if (idx >= 32 * LPP) idx = 32 * LPP - 1;
        pf_voff[i] = (unsigned)((idx / LPP) * DPAD * 2 + (idx % LPP) * 128) + 3072u;
    }
    float pf_sink = 0.f;
    auto l2_prefetch = [&](const _Float16* x) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PF_OPS; ++i)
            asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_sink) : "v"(pf_voff[i]), "s"(x) : "memory");
    };
    // operand address of (slice order j, k-step parity) within a step buffer
    int gram_off[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int par = 0; par < 2; ++par)
            gram_off[j][par] = lds0 + ((wave + j) & 3) * L::SLICE_B + 1024 * (n32 >> 4) + 64 * (n32 & 15) + 16 * ((2 * par + h) ^ q32_swz(n32 >> 2));
    const unsigned dma_lds0 = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(wave * L::SLICE_B));
    const unsigned nrm_lds0 = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)(L::NORM_OFF + wave * 64 * 4));
    // step `sub` of block X
    auto stage_norms = [&](const Blk& X, int sub, int ring) __attribute__((always_inline)) {
        q32_dma_dword(X.nrm + sub * P, nvo, nrm_lds0 + (unsigned)(ring * 4 * 64 * 4));
    };
    auto dma_group = [&](const Blk& X, int sub, int b, int g) __attribute__((always_inline)) {
        const unsigned dst = dma_lds0 + (unsigned)(b * L::XBUF_B) + 4096u * g;
        const _Float16* x = X.x + sub * (P * DPAD);
        if (2 * g + 1 < KB) q32_dma_quad(x, dvo[2 * g][0], dvo[2 * g][1], dvo[2 * g + 1][0], dvo[2 * g + 1][1], dst);
        else q32_dma_pair(x, dvo[2 * g][0], dvo[2 * g][1], dst);
    };
    const int nr_off = lds0 + L::NORM_OFF + (wave * 64 + 4 * h) * 4;

    f32x16 S[2];
    q_f16x8 ab[RING];
    f32x4 cn[4];
    float fv[4], xv[4], pe[4], yv[4], e4[4];
    double sum0 = 0.0, sum1 = 0.0;
    float fq0 = 0.f, fq1 = 0.f;
    auto flush_sums = [&]() __attribute__((always_inline)) {
        sum0 += (double)fq0; sum1 += (double)fq1;
        fq0 = 0.f; fq1 = 0.f;
    };
    // vector work of a step: the 16 entries of the previous step's accumulator in four groups of four, operation-major
    // within a group: norm term, x, then Q(x) = y (1/2 + x/6 + y/24), y = x^2, summed as it is formed (mmd_q32_sum_kernel)
#ifdef TVAE_PHASE_TIMING
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
    constexpr int VOPS = 96;
    constexpr int VW = (VOPS + NS - 1) / NS;                  // KB = 6: two per slot -- with the accumulator read-out and the addresses, three vector instructions at most ride in a slot for free

    // one step, number U of block B (k = SPB B + U): Gram of step k (buffer buf) into S[U & 1]; the step in the other
    // accumulator (step k - 1) is finished and summed between the matrix instructions; behind the barrier buffer bufr is
    // refilled with step k + 2.  The operand ring runs on ACROSS steps: the last GD slots read the first operands of step
    // k + 1 (its own slice in buffer bufn, behind the counted wait on this wave's LDS-DMA), so no step starts with an LDS
    // round trip.  Everything scalar is branch-free.
    int gb0[2];                                               // operand addresses of slice order 0, for the NEXT reads of it
    Blk prv, cur, nxt, nw;                                    // blocks B - 1 (being finished), B, B + 1 (being fetched), B + 2
    auto step = [&](auto U_, int k, int buf, int bufr, int bufn) __attribute__((always_inline)) {
        constexpr int U = decltype(U_)::value, H = U & 1;
        constexpr int ring_r = (U + AHEAD) & 3, ring_e = (U + 3) & 3, sub_r = (U + AHEAD) & 3;
        const bool with_e = k > 0 && k <= nsteps;
        int gb[4][2];
        gb[0][0] = gb0[0]; gb[0][1] = gb0[1];
        auto gb_set = [&](int j) __attribute__((always_inline)) {
#pragma unroll
            for (int par = 0; par < 2; ++par) { gb[j][par] = gram_off[j][par] + buf * L::XBUF_B; asm volatile("" : "+v"(gb[j][par])); }
        };
        auto gram_rd = [&](int s) __attribute__((always_inline)) {
            const int j = s / NK, ks = s - j * NK;
            ab[s % RING] = Q32_LDS(const q_f16x8, gb[j][ks & 1] + 2048 * (ks >> 1));
        };
        // idx: group g of four entries (24 operations), operation-major within the group
        auto v_op = [&](int idx) __attribute__((always_inline)) {
            const int g = idx / 24, op = (idx - 24 * g) >> 2, el = idx & 3, r = 4 * g + el;
            // every result is pinned where it is formed: left alone hipcc sinks the whole list into the slot that uses the sums
            if (op == 0) { fv[el] = fmaf(cn[g][el], -a.inv_d2, rnx); asm volatile("" : "+v"(fv[el])); }
            else if (op == 1) { xv[el] = fmaf(c1, S[H ^ 1][r], fv[el]); asm volatile("" : "+v"(xv[el])); }
            else if (op == 2) { pe[el] = fmaf(1.f / 6.f, xv[el], 0.5f); asm volatile("" : "+v"(pe[el])); }
            else if (op == 3) { yv[el] = xv[el] * xv[el]; asm volatile("" : "+v"(yv[el])); }
            else if (op == 4) { pe[el] = fmaf(1.f / 24.f, yv[el], pe[el]); asm volatile("" : "+v"(pe[el])); }
            else { e4[el] = (g == 0 && U == 1) ? pe[el] * yv[el] : fmaf(pe[el], yv[el], e4[el]); asm volatile("" : "+v"(e4[el])); }   // four chains per lane over the four steps of a block
        };
        const Blk& RB = U + AHEAD < 4 ? cur : nxt;            // the block of step k + AHEAD
        const Blk& PB = U + AHEAD < 4 ? nxt : nw;             // ... and the one behind it (nw: worked out in step 2 of this block)
        PHASE_MARK(5);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int j = s / NK, ks = s - j * NK;
            S[H] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab[s % RING], zreg[j][ks], s == 0 ? zero16 : S[H], 0, 0, 0);
            Q32_PIN();
            if (s < 3) {                                      // the other slices' addresses and norms: first used a slice / a group later
                gb_set(s + 1);
                cn[s + 1] = Q32_LDS(const f32x4, nr_off + ring_e * 4 * 64 * 4 + 8 * (s + 1) * 4);
            }
            if (s == NS - GD - 1) {
                // step k + 1's own slice has landed: everything but the LDS-DMA issued in this step (2 KB + 1 instructions) is done
                // (two buffers: this step's LDS-DMA IS step k + 1 -- everything but the prefetches behind it)
                wait_vmcnt_n(NBUF == 2 ? PF_OPS : 2 * KB + 1);
#pragma unroll
                for (int par = 0; par < 2; ++par) { gb0[par] = gram_off[0][par] + bufn * L::XBUF_B; asm volatile("" : "+v"(gb0[par])); }
            }
            if (s + GD < NS) gram_rd(s + GD);
            else {
                const int t = s + GD - NS;                    // slot t of step k + 1
                ab[t % RING] = Q32_LDS(const q_f16x8, gb0[t & 1] + 2048 * (t >> 1));
            }
#pragma unroll
            for (int i = 0; i < VW; ++i)
                if (VW * s + i < VOPS) v_op(VW * s + i);
            if (U == 2 && s == 4) nw = blk_desc(k / SPB + 2);  // k / SPB = B
            if (U == 0 && s == 4) flush_sums();
            if (s == BAR - 1) PHASE_MARK(1);
            if (s == BAR) __builtin_amdgcn_s_barrier();
            if (s == BAR) PHASE_MARK(2);
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (s == BAR + 1 + g * DSTRIDE) dma_group(RB, sub_r, bufr, g);
            if (s == BAR + 1 + NG * DSTRIDE) stage_norms(RB, sub_r, ring_r);
            if (PF_OPS > 0 && s == BAR + 2 + NG * DSTRIDE) l2_prefetch(PB.x + sub_r * (P * DPAD));
            if (s == 2 * NK - 1) PHASE_MARK(3);
            if (s == 0) PHASE_MARK(7);
            if (s == 2) PHASE_MARK(8);
            if (s == NS - 4) PHASE_MARK(6);
            if (s == NS - 2) PHASE_MARK(4);
            if (s == NS - 2) cn[0] = Q32_LDS(const f32x4, nr_off + U * 4 * 64 * 4);      // step k's norms, for the next step's first group
            if (s == NS - 1 && U == 0) {
                // step k - 1 was the last of block B - 1: its four steps' sums take the block's weight and go to the set's
                // fp32 partial (flushed to fp64 in this step's slot 4 next time round); weight and set are wave-uniform; a
                // select, not a zero weight, keeps the first trip's garbage (and an invalid row) out
                const float es = (with_e && row_valid) ? (e4[0] + e4[1]) + (e4[2] + e4[3]) : 0.f;
                fq0 = fmaf(es, prv.set ? 0.f : prv.w, fq0);
                fq1 = fmaf(es, prv.set ? prv.w : 0.f, fq1);
            }
            if (s == NS - 1 && U == 3) { prv = cur; cur = nxt; nxt = nw; }
            Q32_PIN();
        }
    };

    if (nsteps > 0) {
        // prologue: steps sb, sb + 1 into buffers 0, 1 (norm ring slots 0, 1)
        cur = blk_desc(0); nxt = blk_desc(1); prv = cur; nw = nxt;
        stage_norms(cur, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) dma_group(cur, 0, 0, g);
        if (AHEAD > 1) {
            stage_norms(cur, 1, 1);
#pragma unroll
            for (int g = 0; g < NG; ++g) dma_group(cur, 1, 1, g);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { xv[r] = 0.f; pe[r] = 0.f; fv[r] = 0.f; yv[r] = 0.f; e4[r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[0][r] = 0.f; S[1][r] = 0.f; }
        wait_vmcnt<0>();
        __syncthreads();
        // the first operands of step 0, and the (unused) norms its finishing reads
#pragma unroll
        for (int par = 0; par < 2; ++par) gb0[par] = gram_off[0][par];
#pragma unroll
        for (int t = 0; t < GD; ++t) ab[t % RING] = Q32_LDS(const q_f16x8, gb0[t & 1] + 2048 * (t >> 1));
        cn[0] = Q32_LDS(const f32x4, nr_off);
        // iteration k forms the Gram of step k (buffer k % 3) and finishes step k - 1; one block of SPB steps per trip, and
        // one trip more than there are blocks: its first step finishes the last real step, the rest is discarded
        int buf = 0, bufr = NBUF - 1, bufn = 1;
        auto run = [&](auto U_, int k) __attribute__((always_inline)) {
            step(U_, k, buf, bufr, bufn);
            buf = buf == NBUF - 1 ? 0 : buf + 1;
            bufr = bufr == NBUF - 1 ? 0 : bufr + 1;
            bufn = bufn == NBUF - 1 ? 0 : bufn + 1;
        };
        for (int k = 0; k <= nsteps; k += SPB) {
            run(std::integral_constant<int, 0>{}, k);
            run(std::integral_constant<int, 1>{}, k + 1);
            run(std::integral_constant<int, 2>{}, k + 2);
            run(std::integral_constant<int, 3>{}, k + 3);
        }
        flush_sums();
#ifdef TVAE_PHASE_TIMING
        if (blockIdx.x == 0 && lane == 0)
            printf("sum3 wave %d steps %d: 0: %llu  1-2: %llu  ..%d: %llu  %d: %llu  ..23: %llu  ..44: %llu  45-46: %llu  47+loop: %llu\n", wave,
                   nsteps, ph[7] / nsteps, ph[8] / nsteps, BAR - 1, ph[1] / nsteps, BAR, ph[2] / nsteps, ph[3] / nsteps, ph[6] / nsteps, ph[4] / nsteps, ph[5] / nsteps);
#endif
        // the zero padding points of this workgroup's column range: out of the sums in closed form (see mmd_q32_kernel)
        {
            float npad[2] = {0.f, 0.f};
#pragma unroll 1
            for (int v = sb; v < se; ++v) {
                const Meta m = step_meta(v);
                if (m.p0 + P <= a.n_cols) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) npad[m.set] += (m.p0 + 8 * (r >> 2) + 4 * h + (r & 3) >= a.n_cols) ? m.w : 0.f;
            }
            if (row_valid && (npad[0] != 0.f || npad[1] != 0.f)) {
                const float x = rnx;
                const float y = x * x;
                const double qp = (double)(fmaf(1.f / 24.f, y, fmaf(1.f / 6.f, x, 0.5f)) * y);
                sum0 -= (double)npad[0] * qp;
                sum1 -= (double)npad[1] * qp;
            }
        }
    }
    wait_vmcnt<0>();
    asm volatile("" : "+v"(pf_sink));          // the prefetch sink stays allocated until every load has returned

    __syncthreads();
    double* red = reinterpret_cast<double*>(sm + L::RED_OFF);
    const double w0 = wave_sum(sum0), w1 = wave_sum(sum1);
    if (lane == 0) { red[wave * 2 + 0] = w0; red[wave * 2 + 1] = w1; }
    __syncthreads();
    if (tid == 0) {
        a.wg_sums[2 * bid + 0] = (red[0] + red[2]) + (red[4] + red[6]);
        a.wg_sums[2 * bid + 1] = (red[1] + red[3]) + (red[5] + red[7]);
        // the entries only the other (64- or 32-row) member of the launch pair would have written
        for (int e = rem + per_problem; e < a.launch_blocks; e += per_problem) {
            a.wg_sums[2 * (bid - rem + e) + 0] = 0.0;
            a.wg_sums[2 * (bid - rem + e) + 1] = 0.0;
        }
    }
}

template <int KB>
int launch_q32_sum3(const MmdArgs& a, int nblocks, hipStream_t s) {
    constexpr int lds_bytes = Q32Sum3Lds<KB>::BYTES;
    static PerDeviceOnce once;
    TVAE_CHECK_HIP(per_device_once(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_q32_sum3_kernel<KB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   lds_bytes);
    }));
    hipLaunchKernelGGL((mmd_q32_sum3_kernel<KB>), dim3(nblocks), dim3(256), lds_bytes, s, a);
    TVAE_CHECK_HIP(hipGetLastError());
    return TVAE_OK;
}

}  // namespace

// a: the arguments of the 64-row launch with row_blocks / launch_blocks set for 128-row workgroups (mmd_q32.hip, launch_q32_sum)
int launch_mmd_q32_sum3(int kb, const MmdArgs& a, int nblocks, hipStream_t s) {
    switch (kb) {
        case 2: return launch_q32_sum3<2>(a, nblocks, s);
        case 3: return launch_q32_sum3<3>(a, nblocks, s);
        case 4: return launch_q32_sum3<4>(a, nblocks, s);
        case 5: return launch_q32_sum3<5>(a, nblocks, s);
        case 6: return launch_q32_sum3<6>(a, nblocks, s);
        case 7: return launch_q32_sum3<7>(a, nblocks, s);
        case 8: return launch_q32_sum3<8>(a, nblocks, s);
    }
    return fail(TVAE_ERR_ARG, "mmd: no q32 sum3 kernel for %d feature blocks", kb);
}

}  // namespace tvae
