// Shared pieces of the MMD tile kernels (mmd.hip: fp32 matrix cores; mmd_bf16.hip: bf16 matrix cores).
#pragma once
#include "tvae_common.h"

namespace tvae {

constexpr int ROWS_PER_WG = 32;
constexpr int TILE_P = 16;

struct MmdArgs {
    const float* cols[2];    // padded fp32 copies: [0] = z_all, [1] = y_all, row stride DPAD
    const float* cols_lo[2]; // split-bf16 kernels: the lo copies (cols[] are then the hi copies); else unused
    const unsigned* split_amax;   // split kernels: bits of max |value| over both sets (see split_scales)
    const unsigned* qflag;        // split pipelined gradient launch, nullable: non-zero = accumulate Q = E + t (one gradient product; mmd.hip, QFlag)
    const float* cnorm[2];   // squared norms (0 for padding rows)
    int n_rows, n_cols, row_offset;
    int row_blocks, col_splits, tiles_per_set, n_rows_pad;
    int problem_base;        // first problem of this launch (0: rows = z, 1: rows = y)
    int sums_base;           // bf16 kernels: index of this launch's first workgroup in wg_sums
    int sp_major;            // workgroup order: 0 = the splits of a row block are neighbours, 1 = all row blocks of split 0 first
    // soft rendezvous of the gradient launch (null: off): see mmd.hip "SOFT RENDEZVOUS"
    unsigned* sync_ctr;      // [groups][sync_epochs] arrival counters, zeroed before the launch
    int sync_epochs, sync_group, launch_blocks;
    int sym;                 // 1: the y-y problem visits each unordered pair of 32-row blocks once (weight 2)
    int sym2;                // the same choice for 128-row blocks (mmd_q32_sum3_kernel)
    float inv_d2;
    double* wg_sums;         // [2 * row_blocks * col_splits][2]
    double* wg_rowsum;       // [col_splits][n_rows_pad]
    float* wg_grad;          // [col_splits][n_rows_pad][DPAD]
};

// expm1(-t) for t >= 0.  Below 1/8 a degree-6 Taylor polynomial (truncation < 1e-10);
// otherwise the library expm1f.  The slow branch is skipped when no lane of the wave needs it.
// E = expm1(-t) = -t + Q(t),  Q(t) = t^2/2 - t^3/6 + ... >= 0.
// The kernels need E (gradient, row sums) AND Q: in  sum E(y,y) + sum E(z,z) - 2 sum E(y,z)  the terms linear
// in t cancel ANALYTICALLY down to 2 |sum z - sum y|^2 / D^2, so accumulating E leaves a result ~1/D of the
// summands (measured: the three fp32-accumulated sums carry relative biases that differ by ~2e-9, which is
// 1.5e-5 of the MMD at D = 768, N = 32 768).  The tile kernels therefore accumulate Q, and the finalize
// kernels add the linear part in closed form from norms and column sums in fp64.
// x = -t;  Q = x^2 r(x),  r(x) = 1/2 + x/6 + x^2/24 + x^3/120 + x^4/720  (|t| < 1/8: truncation < 1e-10 of Q)
struct EQ { float e, q; };
__device__ __forceinline__ EQ em1_split(float t) {
    const float x = -t;
    float p = 1.f / 720.f;
    p = fmaf(p, x, 1.f / 120.f);
    p = fmaf(p, x, 1.f / 24.f);
    p = fmaf(p, x, 1.f / 6.f);
    p = fmaf(p, x, 0.5f);
    const float px = p * x;
    return EQ{fmaf(px, x, x), px * x};
}
// 0 <= t < 1/256 (wide latents: t ~ 1/D): r(x) = 1/2 + x/6 + x^2/24, truncation < 1e-9 of Q
__device__ __forceinline__ EQ em1_split_tiny(float t) {
    const float x = -t;
    float p = 1.f / 24.f;
    p = fmaf(p, x, 1.f / 6.f);
    p = fmaf(p, x, 0.5f);
    const float px = p * x;
    return EQ{fmaf(px, x, x), px * x};
}
constexpr float EM1_TINY = 1.f / 256.f;
constexpr float EM1_POLY_MAX = 0.125f;     // beyond: library exp; |E| >= 0.1175 there, so Q = E + t keeps 20 bits

// Split-fp16 mode: fp32 values are held as fp16 pairs hi + lo of s * v, s = 2^(11 - floor(log2 amax)) so that the
// largest magnitude of the call lands in [2^11, 2^12) (fp16 keeps 11 bits there and its smallest steps, 2^-24, are
// 2^-35 of it); E in (-1, 0] is scaled by 2^12.  All scales are powers of two: exact to apply and to undo.
constexpr float SPLIT_E_SCALE = 4096.f;
// Products per Gram entry in split mode.  3: X_hi Z_hi + X_hi Z_lo + X_lo Z_hi (2^-22 of a product dropped).
// 2 (default): X_lo Z_hi dropped as well -- the ROW side Z (this workgroup's rows, in registers) stays exact to 22 bits,
// the streamed COLUMN side X enters the Gram with its 11-bit hi part only.  A column's rounding error is zero-mean and
// independent of every other column's, and each row's sums run over all columns, so the errors average out instead of
// adding up: measured against fp64 (tools/mmd_modes.py) the gradient error moves from 1.39e-7 to 1.49e-7 at N = 32 768,
// D = 768 and stays below 1.5e-7 at N = 600 .. 4 096, the loss error does not move, and every parity test and fuzz sweep
// passes unchanged.  (1 -- hi x hi only -- measured 1e-7 .. 3e-7 and another 7 % faster, but its row-side rounding is the
// same for every column of a row and has no such argument behind it: not used.)  The gradient products keep all three.
#ifndef TVAE_GRAM_PRODUCTS
#define TVAE_GRAM_PRODUCTS 2
#endif
#ifndef TVAE_GRAD_PRODUCTS
#define TVAE_GRAD_PRODUCTS 3      // gradient entry: X_hi E_hi + X_lo E_hi + X_hi E_lo (2: X_lo E_hi dropped, experiment)
#endif
struct SplitScales { float s, inv_s2, grad_unscale; };
__device__ __forceinline__ SplitScales split_scales(unsigned amax_bits) {
    const float amax = __uint_as_float(amax_bits);
    int e = 0;
    if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &e);      // amax = m 2^e, m in [0.5, 1)
    const float s = ldexpf(1.f, 12 - e);                            // s amax in [2^11, 2^12)
    return SplitScales{s, ldexpf(1.f, 2 * (e - 12)), ldexpf(1.f, e - 12) / SPLIT_E_SCALE};
}

// LDS-DMA: every lane moves 16 B (dma16) or 4 B (dma4) global -> LDS at m0_base + lane*size.
// Issued from inline asm so that hipcc neither counts it in its vmcnt bookkeeping nor drains it
// early; M0 (the LDS base of the DMA) is written and restored inside the one statement (guide 5.7).
// Source = wave-uniform SGPR base + per-lane 32-bit byte offset + immediate (saddr form): the tile
// walk costs scalar adds only, and one VGPR holds the per-lane offset for the whole kernel.
// (No instruction immediate: on LDS-DMA the offset field is added to the LDS address as well.)
// The scalar operands go through readfirstlane: a no-op when hipcc already holds them in SGPRs, and
// the only way to get them there when register pressure made it keep a wave-uniform value in VGPRs.
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma16(const float* sbase, unsigned voff, unsigned lds_byte) {
    sbase = uniform_ptr(sbase);
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma4(const float* sbase, unsigned voff, unsigned lds_byte) {
    sbase = uniform_ptr(sbase);
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

// Two LDS-DMA instructions under one M0 set-up: the second lands 1 KiB further in LDS through the
// instruction offset, which the hardware adds to the GLOBAL address as well -- voff_b is the caller's
// per-lane offset with that 1 KiB already taken off.
__device__ __forceinline__ void dma16_pair(const float* sbase, unsigned voff_a, unsigned voff_b, unsigned lds_byte) {
    sbase = uniform_ptr(sbase);
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_a), "v"(voff_b), "s"(sbase), "s"(lds_byte) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// n is a constant after loop unrolling; the switch folds to one s_waitcnt
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<1>(); break;   case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;   case 4: wait_vmcnt<4>(); break;   case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;   case 7: wait_vmcnt<7>(); break;   case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;   case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break; case 13: wait_vmcnt<13>(); break; case 14: wait_vmcnt<14>(); break;
        case 15: wait_vmcnt<15>(); break; case 16: wait_vmcnt<16>(); break; case 17: wait_vmcnt<17>(); break;
        case 18: wait_vmcnt<18>(); break; case 19: wait_vmcnt<19>(); break; case 20: wait_vmcnt<20>(); break;
        case 21: wait_vmcnt<21>(); break; case 22: wait_vmcnt<22>(); break; case 23: wait_vmcnt<23>(); break;
        case 24: wait_vmcnt<24>(); break; case 25: wait_vmcnt<25>(); break; case 26: wait_vmcnt<26>(); break;
        default: wait_vmcnt<0>(); break;
    }
}

// Debug aid (build with -DTVAE_PHASE_TIMING): workgroup 0 prints the cycles its wave 0 spent per phase.
#ifdef TVAE_PHASE_TIMING
#define PHASE_MARK(i) do { const unsigned long long _t = __builtin_readcyclecounter(); ph[i] += _t - ph_last; ph_last = _t; } while (0)
#else
#define PHASE_MARK(i) do { } while (0)
#endif

// bf16 matrix-core tile kernels (mmd_bf16.hip); kb = padded dimension / 128 (2..8)
// split: fp32 values as scaled hi + lo fp16 pairs, two or three products per fp32 product; rt = 16-row tiles per workgroup (2, or 4
// for the split gradient launch with kb <= 6)
int launch_mmd_bf16(bool grad, bool split, int kb, int rt, const MmdArgs& a, int nblocks, hipStream_t s);
// software-pipelined gradient kernel of the 16-bit matrix-core modes, 64 rows per workgroup (mmd_pipe.hip); kb = 2..6
int launch_mmd_pipe(bool grad, bool split, int kb, const MmdArgs& a, int nblocks, hipStream_t s);
// Q-form member of the split gradient launch pair: one fp16 product, 32-point steps, 32x32x16 everywhere (mmd_q32.hip); kb = 2..6
int launch_mmd_q32(int kb, const MmdArgs& a, int nblocks, hipStream_t s);
int launch_mmd_q32_sum(int kb, const MmdArgs& a, int nblocks, hipStream_t s);
// its 128-row member without a K-split (mmd_q32s.hip); a: row_blocks / launch_blocks already set for 128-row workgroups
int launch_mmd_q32_sum3(int kb, const MmdArgs& a, int nblocks, hipStream_t s);
// D x D second moments of the rows [row0, row0 + n_rows) of a split-fp16 image on the 16-bit matrix cores (gram16.hip):
// C (row stride ld) = sum x x^T with the image's scale undone (amax_bits: the scale word of the call)
int64_t gram16_slab_floats(int64_t DPAD, int64_t n_rows);
int gram16_launch(const void* img_hi, long long lo_off, int64_t row0, int64_t n_rows, int64_t DPAD, int64_t D, int64_t ld,
                  const unsigned* amax_bits, float* C, float* slab, int64_t slab_floats, hipStream_t s);
// fp32 kernel for narrow latents, 64 rows per single-wave workgroup (mmd_narrow.hip); nb = padded dimension / 16 (1..5)
int launch_mmd_narrow(bool grad, int nb, const MmdArgs& a, int nblocks, hipStream_t s);
// split-fp16 kernel for narrow latents, one wave per workgroup (mmd_n16.hip); kbn = padded dimension / 32 (1..6).
// nwg0 workgroups of problem 0 (z rows, both column sets, gradient when grad), then nwg1 of problem 1 (y rows, priors)
int mmd_n16_rows(int kbn);      // rows per workgroup
int mmd_n16_wg_per_cu(int kbn); // resident workgroups per CU (registers)
int launch_mmd_n16(bool grad, int kbn, const MmdArgs& a, int nwg0, int nwg1, hipStream_t s);

}  // namespace tvae
