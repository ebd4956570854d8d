// extern "C" entry points of libtvae_hip.so (declared in include/tvae_hip.h) for the encoder
// and decoder; the MMD entry points live next to their kernels in mmd.hip.
#include <cstdarg>
#include <cstdio>

#include "tvae_common.h"

namespace tvae {

std::string& last_error_ref() {
    static thread_local std::string msg;
    return msg;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

namespace {

// bump allocator over the caller's workspace (256-B aligned pieces)
struct Arena {
    char* base; int64_t size; int64_t off = 0; bool ok = true;
    Arena(void* p, int64_t n) : base(static_cast<char*>(p)), size(n) {}
    void* take(int64_t bytes) {
        const int64_t o = off;
        off = align_up(off + bytes, 256);
        if (bytes > 0 && (base == nullptr || off > size)) ok = false;
        return ok && bytes > 0 ? base + o : nullptr;
    }
};
// Sizing passes run the same code with a null arena.
struct Sizer {
    int64_t off = 0;
    void take(int64_t bytes) { off = align_up(off + bytes, 256); }
};

inline bool dtype_ok(int dtype) { return dtype == TVAE_F32 || dtype == TVAE_BF16; }

// The bias gradient rides in the weight-gradient product (GemmDesc::colsum_out: one virtual column of ones)
// when that column is cheap: it shares the last 64-column tile of the weight gradient, or the product is small
// (at the reference's batch sizes a separate column reduction costs two more launches than it has work for).
inline bool fuse_bias_grad(int64_t n_cols, int64_t k_rows) { return n_cols % 64 != 0 || k_rows <= 16384; }
// split-K slabs of a weight-gradient product that may carry the bias column (the split count depends on the
// number of column tiles, so cover both widths)
inline int64_t wgrad_slab_floats(int64_t M, int64_t N, int64_t K) {
    const int64_t a = gemm_slab_floats((int)M, (int)N, (int)K), b = gemm_slab_floats((int)M, (int)N + 1, (int)K);
    return a > b ? a : b;
}

}  // namespace
}  // namespace tvae

using namespace tvae;

extern "C" int tvae_version(void) { return 1000; }

extern "C" const char* tvae_last_error(void) { return last_error_ref().c_str(); }

// ------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------
extern "C" int64_t tvae_encode_fwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L) {
    return align_up(gemm_slab_floats((int)(B * n_eff), (int)L, (int)d) * 4, 256);
}

extern "C" int tvae_encode_fwd(const void* h, int64_t h_stride_b, int64_t h_stride_s, const void* We, const void* be,
                               void* latent, int64_t B, int64_t n_eff, int64_t d, int64_t L, int dtype,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    TVAE_REQUIRE(dtype_ok(dtype), "encode_fwd: unknown dtype %d", dtype);
    TVAE_REQUIRE(h && We && be && latent, "encode_fwd: null pointer argument");
    TVAE_REQUIRE(B > 0 && n_eff > 0 && d > 0 && L > 0, "encode_fwd: empty shape B=%lld n=%lld d=%lld L=%lld",
                 (long long)B, (long long)n_eff, (long long)d, (long long)L);
    TVAE_REQUIRE(B * n_eff < (1ll << 31), "encode_fwd: too many rows");
    GemmDesc g{};
    g.A = h; g.B = We; g.C = latent; g.bias = be;
    g.M = (int)(B * n_eff); g.N = (int)L; g.K = (int)d;
    g.a_kmajor = false; g.b_kmajor = false;
    g.amap = RowMap{h_stride_b, h_stride_s, (int)n_eff};
    g.bmap = dense_rows(d);
    g.cmap = dense_rows(L);
    g.epilogue = EPI_BIAS_TANH; g.dtype = dtype;
    g.slab = static_cast<float*>(workspace); g.slab_floats = workspace ? workspace_bytes / 4 : 0;
    return gemm_launch(g, static_cast<hipStream_t>(stream));
}

extern "C" int64_t tvae_encode_bwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L) {
    const int64_t rows = B * n_eff;
    Sizer s;
    s.take(rows * L * 4);                                              // d_pre
    s.take(wgrad_slab_floats(L, d, rows) * 4);                         // dWe split-K slabs (+ the dbe column)
    s.take(gemm_slab_floats((int)rows, (int)d, (int)L) * 4);           // dh slabs (normally 0)
    s.take(colreduce_scratch_floats(rows, L) * 4);                     // dbe partials
    return s.off;
}

namespace {
// d_latent_extra / extra_scale: nullable pair; the latent cotangent is d_latent + *extra_scale * d_latent_extra
int encode_bwd_impl(const void* d_latent, const void* d_latent_extra, const float* extra_scale, const void* latent,
                    const void* h, int64_t h_stride_b, int64_t h_stride_s, const void* We, void* dh, void* dWe, void* dbe,
                    int64_t B, int64_t S, int64_t n_eff, int64_t d, int64_t L, int dtype, void* workspace,
                    int64_t workspace_bytes, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    TVAE_REQUIRE(dtype_ok(dtype), "encode_bwd: unknown dtype %d", dtype);
    TVAE_REQUIRE(d_latent && latent && We, "encode_bwd: null pointer argument");
    TVAE_REQUIRE(B > 0 && n_eff > 0 && d > 0 && L > 0 && S >= n_eff, "encode_bwd: bad shape");
    TVAE_REQUIRE(!(dWe) || h, "encode_bwd: dWe requested without h");
    const int64_t rows = B * n_eff;
    TVAE_REQUIRE(rows < (1ll << 31), "encode_bwd: too many rows");
    Arena ar(workspace, workspace_bytes);
    void* dpre = ar.take(rows * L * 4);
    const int64_t slab_we = wgrad_slab_floats(L, d, rows);
    float* slab1 = static_cast<float*>(ar.take(slab_we * 4));
    const int64_t slab_dh = gemm_slab_floats((int)rows, (int)d, (int)L);
    float* slab2 = static_cast<float*>(ar.take(slab_dh * 4));
    float* colscratch = static_cast<float*>(ar.take(colreduce_scratch_floats(rows, L) * 4));
    if (!ar.ok)
        return fail(TVAE_ERR_WORKSPACE, "encode_bwd: workspace %lld B < required %lld B", (long long)workspace_bytes,
                    (long long)ar.off);

    if (d_latent_extra) TVAE_PROPAGATE(tanh_bwd_sum(d_latent, d_latent_extra, extra_scale, latent, dpre, rows * L, dtype, s));
    else TVAE_PROPAGATE(tanh_bwd(d_latent, latent, dpre, rows * L, dtype, s));
    GemmDesc gh{}, gw{};
    if (dh) {
        // rows >= n_eff of every sequence: exact zeros (slice backward); rows < n_eff: d_pre @ We
        if (!dWe) TVAE_PROPAGATE(zero_tail_rows(dh, B, S, n_eff, d, dtype, s));     // (else it rides with the pair below)
        GemmDesc& g = gh;
        g.A = dpre; g.B = We; g.C = dh; g.M = (int)rows; g.N = (int)d; g.K = (int)L;
        g.a_kmajor = false; g.b_kmajor = true;
        g.amap = dense_rows(L); g.bmap = dense_rows(d);
        g.cmap = RowMap{S * d, d, (int)n_eff};
        g.epilogue = EPI_NONE; g.dtype = dtype; g.slab = slab2; g.slab_floats = slab_dh;
    }
    if (dWe) {
        GemmDesc& g = gw;
        g.A = dpre; g.B = h; g.C = dWe; g.M = (int)L; g.N = (int)d; g.K = (int)rows;
        g.a_kmajor = true; g.b_kmajor = true;
        g.amap = dense_rows(L);
        g.bmap = RowMap{h_stride_b, h_stride_s, (int)n_eff};
        g.cmap = dense_rows(d);
        g.epilogue = EPI_NONE; g.dtype = dtype; g.slab = slab1; g.slab_floats = slab_we;
        if (dbe && fuse_bias_grad(d, rows)) { g.colsum_out = dbe; dbe = nullptr; }   // dbe = dpre^T 1
    }
    if (dh && dWe) {
        const ZeroTail zt = zero_tail_plan(dh, B, S, n_eff, d, dtype);
        bool zt_done = false;
        TVAE_PROPAGATE(gemm_launch_pair(gh, gw, s, &zt, &zt_done));
        if (!zt_done) TVAE_PROPAGATE(zero_tail_rows(dh, B, S, n_eff, d, dtype, s));   // large products: its own launch
    } else if (dh) TVAE_PROPAGATE(gemm_launch(gh, s));
    else if (dWe) TVAE_PROPAGATE(gemm_launch(gw, s));
    if (dbe) TVAE_PROPAGATE(colsum(dpre, rows, L, dbe, dtype, colscratch, s));
    return TVAE_OK;
}
}  // namespace

extern "C" int tvae_encode_bwd(const void* d_latent, const void* latent, const void* h, int64_t h_stride_b,
                               int64_t h_stride_s, const void* We, void* dh, void* dWe, void* dbe, int64_t B,
                               int64_t S, int64_t n_eff, int64_t d, int64_t L, int dtype, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return encode_bwd_impl(d_latent, nullptr, nullptr, latent, h, h_stride_b, h_stride_s, We, dh, dWe, dbe, B, S, n_eff, d,
                           L, dtype, workspace, workspace_bytes, stream);
}

extern "C" int tvae_encode_bwd_sum(const void* d_latent, const void* d_latent_extra, const float* extra_scale,
                                   const void* latent, const void* h, int64_t h_stride_b, int64_t h_stride_s,
                                   const void* We, void* dh, void* dWe, void* dbe, int64_t B, int64_t S, int64_t n_eff,
                                   int64_t d, int64_t L, int dtype, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    TVAE_REQUIRE(d_latent_extra && extra_scale, "encode_bwd_sum: null extra cotangent or scale");
    return encode_bwd_impl(d_latent, d_latent_extra, extra_scale, latent, h, h_stride_b, h_stride_s, We, dh, dWe, dbe, B, S,
                           n_eff, d, L, dtype, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------
// decoder
// ------------------------------------------------------------------------------------------
extern "C" int64_t tvae_decode_fwd_workspace_bytes(int64_t N, int64_t L, int64_t d) {
    return align_up(gemm_slab_floats((int)N, (int)d, (int)L) * 4, 256);
}

extern "C" int tvae_decode_fwd(const void* latent, const void* Wd, const void* bd, const void* gamma, const void* beta,
                               float eps, void* recon, void* pre_norm, float* mean, float* rstd, int64_t N, int64_t L,
                               int64_t d, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    TVAE_REQUIRE(dtype_ok(dtype), "decode_fwd: unknown dtype %d", dtype);
    TVAE_REQUIRE(latent && recon, "decode_fwd: null pointer argument");
    TVAE_REQUIRE(N > 0 && L > 0 && d > 0 && N < (1ll << 31), "decode_fwd: bad shape");
    TVAE_REQUIRE(Wd || L == d, "decode_fwd: identity decoder needs latent_size == d_model (%lld vs %lld)",
                 (long long)L, (long long)d);
    TVAE_REQUIRE(!Wd || bd, "decode_fwd: Wd without bd");
    TVAE_REQUIRE(!gamma || (beta && mean && rstd), "decode_fwd: LayerNorm needs beta, mean and rstd buffers");
    TVAE_REQUIRE(!(gamma && Wd) || pre_norm, "decode_fwd: LayerNorm after a linear needs the pre_norm buffer");
    const void* ln_in = latent;
    if (Wd) {
        GemmDesc g{};
        g.A = latent; g.B = Wd; g.C = gamma ? pre_norm : recon; g.bias = bd;
        g.M = (int)N; g.N = (int)d; g.K = (int)L;
        g.a_kmajor = false; g.b_kmajor = false;
        g.amap = dense_rows(L); g.bmap = dense_rows(L); g.cmap = dense_rows(d);
        g.epilogue = EPI_BIAS; g.dtype = dtype;
        g.slab = static_cast<float*>(workspace); g.slab_floats = workspace ? workspace_bytes / 4 : 0;
        if (gamma) {        // large batches: linear + bias + LayerNorm by one launch (bit-identical to the two below)
            bool fused = false;
            TVAE_PROPAGATE(gemm_bias_layernorm(g, gamma, beta, eps, recon, mean, rstd, s, &fused));
            if (fused) return TVAE_OK;
        }
        TVAE_PROPAGATE(gemm_launch(g, s));
        ln_in = pre_norm;
    }
    if (gamma) return layernorm_fwd(ln_in, gamma, beta, eps, recon, mean, rstd, N, d, dtype, s);
    if (!Wd && recon != latent) {
        const int64_t esz = dtype == TVAE_F32 ? 4 : 2;
        TVAE_CHECK_HIP(hipMemcpyAsync(recon, latent, N * d * esz, hipMemcpyDeviceToDevice, s));
    }
    return TVAE_OK;
}

extern "C" int64_t tvae_decode_bwd_workspace_bytes(int64_t N, int64_t L, int64_t d) {
    Sizer s;
    s.take(N * d * 4);                                              // dr
    s.take(wgrad_slab_floats(d, L, N) * 4);                         // dWd slabs (+ the dbd column)
    s.take(gemm_slab_floats((int)N, (int)L, (int)d) * 4);           // d_latent slabs
    s.take(layernorm_bwd_scratch_floats(N, d) * 4);                 // dgamma/dbeta or dbd partials
    return s.off;
}

extern "C" int tvae_decode_bwd(const void* d_recon, const void* latent, const void* Wd, const void* gamma,
                               const void* pre_norm, const float* mean, const float* rstd, void* d_latent, void* dWd,
                               void* dbd, void* dgamma, void* dbeta, int64_t N, int64_t L, int64_t d, int dtype,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    TVAE_REQUIRE(dtype_ok(dtype), "decode_bwd: unknown dtype %d", dtype);
    TVAE_REQUIRE(d_recon && latent, "decode_bwd: null pointer argument");
    TVAE_REQUIRE(N > 0 && L > 0 && d > 0 && N < (1ll << 31), "decode_bwd: bad shape");
    TVAE_REQUIRE(Wd || L == d, "decode_bwd: identity decoder needs latent_size == d_model");
    TVAE_REQUIRE(!gamma || (mean && rstd), "decode_bwd: LayerNorm backward needs mean and rstd");
    const void* ln_in = Wd ? pre_norm : latent;
    TVAE_REQUIRE(!gamma || ln_in, "decode_bwd: LayerNorm backward needs the pre-norm activations");
    TVAE_REQUIRE((!dgamma && !dbeta) || (gamma && dgamma && dbeta), "decode_bwd: dgamma/dbeta come as a pair, with gamma");
    Arena ar(workspace, workspace_bytes);
    void* dr_buf = ar.take(N * d * 4);
    const int64_t slab_wd = wgrad_slab_floats(d, L, N);
    float* slab1 = static_cast<float*>(ar.take(slab_wd * 4));
    const int64_t slab_dz = gemm_slab_floats((int)N, (int)L, (int)d);
    float* slab2 = static_cast<float*>(ar.take(slab_dz * 4));
    float* colscratch = static_cast<float*>(ar.take(layernorm_bwd_scratch_floats(N, d) * 4));
    if (!ar.ok)
        return fail(TVAE_ERR_WORKSPACE, "decode_bwd: workspace %lld B < required %lld B", (long long)workspace_bytes,
                    (long long)ar.off);

    const void* dr = d_recon;
    if (gamma) {
        void* dst = (!Wd && d_latent) ? d_latent : dr_buf;   // identity decoder: LN backward IS d_latent
        if (dgamma && layernorm_bwd_can_fuse(d_recon, ln_in, gamma, dst, d, dtype)) {
            TVAE_PROPAGATE(layernorm_bwd_fused(d_recon, ln_in, mean, rstd, gamma, dst, dgamma, dbeta, N, d, dtype,
                                               colscratch, s));
        } else {
            if (dgamma) TVAE_PROPAGATE(layernorm_bwd_params(d_recon, ln_in, mean, rstd, dgamma, dbeta, N, d, dtype, colscratch, s));
            TVAE_PROPAGATE(layernorm_bwd_rows(d_recon, ln_in, mean, rstd, gamma, dst, N, d, dtype, s));
        }
        dr = dst;
    }
    if (!Wd) {
        if (d_latent && dr != d_latent) {
            const int64_t esz = dtype == TVAE_F32 ? 4 : 2;
            TVAE_CHECK_HIP(hipMemcpyAsync(d_latent, dr, N * d * esz, hipMemcpyDeviceToDevice, s));
        }
        return TVAE_OK;
    }
    GemmDesc gz{}, gw{};
    if (d_latent) {
        GemmDesc& g = gz;
        g.A = dr; g.B = Wd; g.C = d_latent; g.M = (int)N; g.N = (int)L; g.K = (int)d;
        g.a_kmajor = false; g.b_kmajor = true;
        g.amap = dense_rows(d); g.bmap = dense_rows(L); g.cmap = dense_rows(L);
        g.epilogue = EPI_NONE; g.dtype = dtype; g.slab = slab2; g.slab_floats = slab_dz;
    }
    if (dWd) {
        GemmDesc& g = gw;
        g.A = dr; g.B = latent; g.C = dWd; g.M = (int)d; g.N = (int)L; g.K = (int)N;
        g.a_kmajor = true; g.b_kmajor = true;
        g.amap = dense_rows(d); g.bmap = dense_rows(L); g.cmap = dense_rows(L);
        g.epilogue = EPI_NONE; g.dtype = dtype; g.slab = slab1; g.slab_floats = slab_wd;
        if (dbd && fuse_bias_grad(L, N)) { g.colsum_out = dbd; dbd = nullptr; }      // dbd = dr^T 1
    }
    if (d_latent && dWd) TVAE_PROPAGATE(gemm_launch_pair(gz, gw, s));
    else if (d_latent) TVAE_PROPAGATE(gemm_launch(gz, s));
    else if (dWd) TVAE_PROPAGATE(gemm_launch(gw, s));
    if (dbd) TVAE_PROPAGATE(colsum(dr, N, d, dbd, dtype, colscratch, s));
    return TVAE_OK;
}
