// Module-level entry points: PerceiverResampler and GatedCrossAttentionBlock, forward and hand-written backward,
// composed from the kernels in ff_gemm / ff_rowwise / ff_attention.  No allocation, no synchronisation: the caller
// provides `saved` (activations kept for backward) and `scratch` buffers sized by the *_bytes() queries.
#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include "ff_common.h"
#include "ff_internal.h"

namespace ff {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#ifdef FF_DEBUG
int dbg_switch(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return FF_ERR_LAUNCH;
    }
    return FF_OK;
}


static LnArgs ln_args(int dtype, int rows, int cols, RowMap x, RowMap y, RowMap dx) {
    LnArgs a;
    a.dtype = dtype; a.rows = rows; a.cols = cols;
    a.x_map = x; a.y_map = y; a.dx_map = dx;
    a.add_rows_per_seg = 0; a.add_div = 0; a.eps = 1e-5f; a.stats_given = 0;
    return a;
}

// =====================================================================================================
// PerceiverResampler
// =====================================================================================================
struct RsDims {
    int dt, Bn, T, v, D, F, q, H, dh, inner, ffi, R, depth, nte, act;
    size_t es;
    float scale;
    explicit RsDims(const ff_resampler_desc& d) {
        dt = d.dtype; Bn = d.batch; T = d.n_frames; v = d.n_tokens; D = d.dim; F = T * v; q = d.num_latents;
        H = d.heads; dh = d.dim_head; inner = H * dh; ffi = d.ff_mult * D; R = F + q; depth = d.depth;
        nte = d.num_time_embeds; act = d.act; es = dtype_size(dt); scale = 1.0f / sqrtf((float)dh);
    }
};
struct RsLayerSaved {
    void *x_in, *kv_in, *Qs, *K, *V, *O, *x_mid, *xn_f, *Hpre, *Aact;
    float *mean_l, *rstd_l, *lse, *mean_f, *rstd_f;
};
constexpr int kMaxDepth = 64;
struct RsSaved {
    float *mean_m, *rstd_m, *mean_o, *rstd_o;
    void* x_last;
    RsLayerSaved L[kMaxDepth];
};
static size_t rs_saved_layout(const RsDims& s, void* base, size_t cap, RsSaved& o) {
    Arena a(base, cap);
    const size_t rows_q = (size_t)s.Bn * s.q, rows_kv = (size_t)s.Bn * s.R;
    o.mean_m = a.take<float>((size_t)s.Bn * s.F * 4);
    o.rstd_m = a.take<float>((size_t)s.Bn * s.F * 4);
    for (int l = 0; l < s.depth; l++) {
        RsLayerSaved& L = o.L[l];
        L.x_in = l == 0 ? nullptr : a.take(rows_q * s.D * s.es);
        L.mean_l = a.take<float>(rows_q * 4);
        L.rstd_l = a.take<float>(rows_q * 4);
        L.kv_in = a.take(rows_kv * s.D * s.es);
        L.Qs = a.take(rows_q * s.inner * s.es);
        L.K = a.take(rows_kv * s.inner * s.es);
        L.V = a.take(rows_kv * s.inner * s.es);
        L.lse = a.take<float>((size_t)s.Bn * s.H * s.q * 4);
        L.O = a.take(rows_q * s.inner * s.es);
        L.x_mid = a.take(rows_q * s.D * s.es);
        L.mean_f = a.take<float>(rows_q * 4);
        L.rstd_f = a.take<float>(rows_q * 4);
        L.xn_f = a.take(rows_q * s.D * s.es);
        L.Hpre = a.take(rows_q * s.ffi * s.es);
        L.Aact = a.take(rows_q * s.ffi * s.es);
    }
    o.x_last = a.take(rows_q * s.D * s.es);
    o.mean_o = a.take<float>(rows_q * 4);
    o.rstd_o = a.take<float>(rows_q * 4);
    return align_up(a.used);
}

// Backward keeps what the weight-gradient GEMMs need (d x at each layer's output, d x_mid, d H, d Qs, d K, d V) for EVERY layer:
// the weight gradients are not on the critical path, so they run after the data-gradient chain as grouped launches over up to
// kGemmMaxZ layers (same shapes, different operands) - one chip-filling launch instead of four that each fill a third of it.
struct RsLayerStash {
    void *dx_out, *dx_mid, *dH, *dQs, *dK, *dV;
};
struct RsScratch {
    void *dx0, *dxn, *dO, *dkv, *dln, *dxf, *ws;
    RsLayerStash L[kMaxDepth];
    float* lnp[3 * kMaxDepth + 1];     // partials of the postponed LayerNorm-backward reductions (null: not deferrable at this width)
    size_t ws_bytes;
};
static size_t rs_ws_bytes(const RsDims& s) {
    const int Mq = s.Bn * s.q, Mkv = s.Bn * s.R;
    size_t w = 0;
    auto g = [&](int M, int N, int K, int nz) { w = std::max(w, gemm_workspace_bytes(s.dt, M, N, K, nz, 0)); };
    g(Mq, s.inner, s.D, 1); g(Mkv, s.inner, s.D, 2); g(Mq, s.D, s.inner, 1); g(Mq, s.ffi, s.D, 1); g(Mq, s.D, s.ffi, 1);   // fwd
    for (int nz = 1; nz <= 4; nz++) {                                                                                       // grouped wgrad (kRsGroup)
        g(s.D, s.ffi, Mq, nz); g(s.ffi, s.D, Mq, nz); g(s.D, s.inner, Mq, nz); g(s.inner, s.D, Mq, nz); g(s.inner, s.D, Mkv, nz);
    }
    g(Mkv, s.D, s.inner, 1);
    w = std::max(w, layernorm_bwd_workspace(s.Bn * s.F, s.D));
    w = std::max(w, layernorm_bwd_workspace(Mq, s.D));
    w = std::max(w, rows_reduce_workspace(Mq, s.D, s.q, 1));
    w = std::max(w, rows_reduce_workspace(s.Bn * s.F, s.D, s.F, s.v));
    return align_up(w) + align_up((size_t)s.Bn * s.H * s.q * 4);
}
static size_t rs_scratch_layout(const RsDims& s, void* base, size_t cap, bool bwd, RsScratch& o) {
    Arena a(base, cap);
    const size_t rows_q = (size_t)s.Bn * s.q, rows_kv = (size_t)s.Bn * s.R;
    o.ws_bytes = rs_ws_bytes(s);
    o.ws = a.take(o.ws_bytes);
    if (bwd) {
        o.dx0 = a.take(rows_q * s.D * s.es);        // d (latents broadcast over the batch)
        o.dxn = a.take(rows_q * s.D * s.es);
        o.dO = a.take(rows_q * s.inner * s.es);
        o.dkv = a.take(rows_kv * s.D * s.es);
        o.dln = a.take(rows_q * s.D * s.es);
        o.dxf = a.take((size_t)s.Bn * s.F * s.D * s.es);
        for (int l = 0; l < s.depth; l++) {
            RsLayerStash& L = o.L[l];
            L.dx_out = a.take(rows_q * s.D * s.es);
            L.dx_mid = a.take(rows_q * s.D * s.es);
            L.dH = a.take(rows_q * s.ffi * s.es);
            L.dQs = a.take(rows_q * s.inner * s.es);
            L.dK = a.take(rows_kv * s.inner * s.es);
            L.dV = a.take(rows_kv * s.inner * s.es);
        }
        static const int defer_ln = dbg_switch("FF_DEFER_LN", 1);
        const bool lnd = defer_ln != 0 && layernorm_bwd_deferrable(s.dt, s.D);
        for (int i = 0; i < 3 * s.depth + 1; i++)       // [3l] ff norm, [3l+1] norm_latents, [3l+2] norm_media of layer l; [3 depth] final norm
            o.lnp[i] = lnd ? a.take<float>(layernorm_bwd_partial_bytes(i % 3 == 2 ? s.Bn * s.F : (int)rows_q, s.D)) : nullptr;
    }
    return align_up(a.used);
}

static int rs_check(const ff_resampler_desc* d) {
    FF_CHECK(d, FF_ERR_SHAPE, "resampler: null descriptor");
    FF_CHECK(d->dtype == FF_DTYPE_F32 || d->dtype == FF_DTYPE_BF16, FF_ERR_UNSUPPORTED, "resampler: dtype %d", d->dtype);
    FF_CHECK(d->batch > 0 && d->n_frames > 0 && d->n_tokens > 0 && d->dim > 0 && d->depth > 0 && d->depth <= kMaxDepth && d->heads > 0 &&
                 d->dim_head > 0 && d->num_latents > 0 && d->ff_mult > 0,
             FF_ERR_SHAPE, "resampler: bad dimensions");
    FF_CHECK(d->n_frames <= d->num_time_embeds, FF_ERR_SHAPE, "resampler: %d frames > num_time_embeds %d (perceiver_resampler.py:166)",
             d->n_frames, d->num_time_embeds);
    FF_CHECK(d->act >= FF_ACT_GELU && d->act <= FF_ACT_RELU, FF_ERR_SHAPE, "resampler: act %d", d->act);
    return FF_OK;
}

static ff_attn_desc rs_attn_desc(const RsDims& s) {
    ff_attn_desc a = {};
    a.dtype = s.dt; a.batch = s.Bn; a.heads = s.H; a.dim_head = s.dh; a.n_q = s.q; a.n_kv = s.R; a.mode = FF_ATTN_DENSE;
    const ff_strides sq = {(long long)s.q * s.inner, s.inner, s.dh}, sk = {(long long)s.R * s.inner, s.inner, s.dh};
    a.q = sq; a.o = sq; a.dq = sq; a.dout = sq;
    a.k = sk; a.v = sk; a.dk = sk; a.dv = sk;
    return a;
}

static int resampler_fwd(const ff_resampler_desc* d, const void* x_f, const void* const* P, void* out, void* saved, size_t saved_bytes,
                         void* scratch, size_t scratch_bytes, hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(x_f && P && out && saved && scratch, FF_ERR_SHAPE, "resampler_fwd: null argument");
    const RsDims s(*d);
    RsSaved S;
    RsScratch W;
    FF_CHECK(rs_saved_layout(s, saved, saved_bytes, S) <= saved_bytes, FF_ERR_WORKSPACE, "resampler_fwd: saved buffer too small");
    FF_CHECK(rs_scratch_layout(s, scratch, scratch_bytes, false, W) <= scratch_bytes, FF_ERR_WORKSPACE, "resampler_fwd: scratch too small");
    const int Mq = s.Bn * s.q, Mkv = s.Bn * s.R, Mf = s.Bn * s.F;
    const RowMap pD = plain_rows(s.D), pI = plain_rows(s.inner), pF = plain_rows(s.ffi);
    const RowMap lat_bcast = RowMap{s.D, 0, s.q};                                   // latents (q, D) repeated over the batch (:179)
    const RowMap kv_media = RowMap{s.D, (long long)s.R * s.D, s.F};                // media rows inside kv_in
    const RowMap kv_lat = RowMap{s.D, (long long)s.R * s.D, s.q};                  // latent rows inside kv_in (base + F*D)
    const void *latents = P[0], *tpe = P[1];

    {   // statistics of x_f + time_pos_emb, shared by the norm_media of every layer (:166, :52)
        LnArgs a = ln_args(s.dt, Mf, s.D, pD, pD, pD);
        a.add_rows_per_seg = s.F; a.add_div = s.v;
        FF_TRY(layernorm_fwd(a, x_f, tpe, nullptr, nullptr, nullptr, S.mean_m, S.rstd_m, st));
    }
    for (int l = 0; l < s.depth; l++) {
        const void* const* p = P + FF_RESAMPLER_GLOBAL_PARAMS + FF_RESAMPLER_LAYER_PARAMS * l;
        RsLayerSaved& L = S.L[l];
        const void* x_in = l == 0 ? latents : L.x_in;
        const RowMap x_map = l == 0 ? lat_bcast : pD;
        void* x_next = l + 1 < s.depth ? S.L[l + 1].x_in : S.x_last;
        char* kv_lat_base = (char*)L.kv_in + (size_t)s.F * s.D * s.es;
        {   // norm_media -> media rows of kv_in (:52,65)
            LnArgs a = ln_args(s.dt, Mf, s.D, pD, kv_media, pD);
            a.add_rows_per_seg = s.F; a.add_div = s.v; a.stats_given = 1;
            FF_TRY(layernorm_fwd(a, x_f, tpe, p[0], p[1], L.kv_in, S.mean_m, S.rstd_m, st));
        }
        {   // norm_latents -> latent rows of kv_in (:53,65)
            LnArgs a = ln_args(s.dt, Mq, s.D, x_map, kv_lat, pD);
            FF_TRY(layernorm_fwd(a, x_in, nullptr, p[2], p[3], kv_lat_base, L.mean_l, L.rstd_l, st));
        }
        // q = to_q(latents) * scale (:57,79)
        FF_TRY(Gemm(s.dt, Mq, s.inner, s.D).a(0, kv_lat).b(0, pD).c(pI).scale(s.scale).problem(kv_lat_base, p[4], L.Qs).run(W.ws, W.ws_bytes, st));
        // k, v = to_k / to_v (cat(media, latents)) (:69-70), one grouped launch
        FF_TRY(Gemm(s.dt, Mkv, s.inner, s.D).a(0, pD).b(0, pD).c(pI).problem(L.kv_in, p[5], L.K).problem(L.kv_in, p[6], L.V).run(W.ws, W.ws_bytes, st));
        FF_TRY(attention_fwd(rs_attn_desc(s), L.Qs, L.K, L.V, nullptr, L.O, L.lse, st));   // :85-92
        // x = x + to_out(o) (:96, :182)
        FF_TRY(Gemm(s.dt, Mq, s.D, s.inner).a(0, pI).b(0, pI).c(pD).res_map(x_map).problem(L.O, p[7], L.x_mid, nullptr, nullptr, x_in).run(W.ws, W.ws_bytes, st));
        // x = x + ffw(x) (:183; utils.py:45-50)
        FF_TRY(layernorm_fwd(ln_args(s.dt, Mq, s.D, pD, pD, pD), L.x_mid, nullptr, p[8], p[9], L.xn_f, L.mean_f, L.rstd_f, st));
        FF_TRY(Gemm(s.dt, Mq, s.ffi, s.D).a(0, pD).b(0, pD).c(pF).act(s.act).problem(L.xn_f, p[10], L.Aact, L.Hpre).run(W.ws, W.ws_bytes, st));
        FF_TRY(Gemm(s.dt, Mq, s.D, s.ffi).a(0, pF).b(0, pF).c(pD).problem(L.Aact, p[11], x_next, nullptr, nullptr, L.x_mid).run(W.ws, W.ws_bytes, st));
    }
    return layernorm_fwd(ln_args(s.dt, Mq, s.D, pD, pD, pD), S.x_last, nullptr, P[2], P[3], out, S.mean_o, S.rstd_o, st);   // :187
}

static int resampler_bwd(const ff_resampler_desc* d, const void* x_f, const void* const* P, const void* dout, const void* saved,
                         size_t saved_bytes, void* const* G, void* dx_f, void* scratch, size_t scratch_bytes, hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(x_f && P && dout && saved && G && scratch, FF_ERR_SHAPE, "resampler_bwd: null argument");
    const RsDims s(*d);
    RsSaved S;
    RsScratch W;
    FF_CHECK(rs_saved_layout(s, (void*)saved, saved_bytes, S) <= saved_bytes, FF_ERR_WORKSPACE, "resampler_bwd: saved buffer too small");
    FF_CHECK(rs_scratch_layout(s, scratch, scratch_bytes, true, W) <= scratch_bytes, FF_ERR_WORKSPACE, "resampler_bwd: scratch too small");
    const int Mq = s.Bn * s.q, Mkv = s.Bn * s.R, Mf = s.Bn * s.F;
    const RowMap pD = plain_rows(s.D), pI = plain_rows(s.inner), pF = plain_rows(s.ffi);
    const RowMap lat_bcast = RowMap{s.D, 0, s.q};
    const RowMap kv_media = RowMap{s.D, (long long)s.R * s.D, s.F};
    const RowMap kv_lat = RowMap{s.D, (long long)s.R * s.D, s.q};
    const void *latents = P[0], *tpe = P[1];
    void* dxf = dx_f ? dx_f : W.dxf;
    // attention workspace lives behind the shared workspace
    float* attn_ws = (float*)((char*)W.ws + W.ws_bytes - align_up((size_t)s.Bn * s.H * s.q * 4));
    const size_t gws = W.ws_bytes - align_up((size_t)s.Bn * s.H * s.q * 4);

    // ---- data-gradient chain (critical path); weight-gradient operands are left in W.L[l] ----
    // The final reductions of the LayerNorm backwards (d gamma / d beta) are not on that chain either: their partials stay in W.lnp and
    // one launch per kLnFinishMax of them finishes all 3 depth + 1 after the chain.
    LnPending ln_sets[3 * kMaxDepth + 1];
    auto ln_bwd = [&](int slot, const LnArgs& a, const void* dy_, const void* x_, const void* add_, const void* gamma_, const float* mean_,
                      const float* rstd_, void* dx_, const void* dx_res_, void* dg_, void* db_) -> int {
        if (W.lnp[slot])
            return layernorm_bwd(a, dy_, x_, add_, gamma_, mean_, rstd_, dx_, dx_res_, dg_, db_, W.lnp[slot], layernorm_bwd_partial_bytes(a.rows, a.cols), st,
                                 nullptr, &ln_sets[slot]);       // (leaves ln_sets[slot].partial null if it had to run the unfused path)
        return layernorm_bwd(a, dy_, x_, add_, gamma_, mean_, rstd_, dx_, dx_res_, dg_, db_, W.ws, gws, st);
    };
    // final norm (:187): d x at the output of the last layer
    FF_TRY(ln_bwd(3 * s.depth, ln_args(s.dt, Mq, s.D, pD, pD, pD), dout, S.x_last, nullptr, P[2], S.mean_o, S.rstd_o, W.L[s.depth - 1].dx_out, nullptr,
                  G[2], G[3]));
    for (int l = s.depth - 1; l >= 0; l--) {
        const void* const* p = P + FF_RESAMPLER_GLOBAL_PARAMS + FF_RESAMPLER_LAYER_PARAMS * l;
        void* const* g = G + FF_RESAMPLER_GLOBAL_PARAMS + FF_RESAMPLER_LAYER_PARAMS * l;
        const RsLayerSaved& L = S.L[l];
        const RsLayerStash& T = W.L[l];
        const void* x_in = l == 0 ? latents : L.x_in;
        const RowMap x_map = l == 0 ? lat_bcast : pD;
        char* dkv_lat_base = (char*)W.dkv + (size_t)s.F * s.D * s.es;
        void* dx_below = l == 0 ? W.dx0 : W.L[l - 1].dx_out;                                   // d x at this layer's input
        // ---- FeedForward backward (x_next = x_mid + W3 act(W1 LN(x_mid))) ----
        FF_TRY(Gemm(s.dt, Mq, s.ffi, s.D).a(0, pD).b(1, pF).c(pF).act_bwd(s.act).problem(T.dx_out, p[11], T.dH, nullptr, L.Hpre).run(W.ws, gws, st));
        FF_TRY(Gemm(s.dt, Mq, s.D, s.ffi).a(0, pF).b(1, pD).c(pD).problem(T.dH, p[10], W.dxn).run(W.ws, gws, st));
        FF_TRY(ln_bwd(3 * l, ln_args(s.dt, Mq, s.D, pD, pD, pD), W.dxn, L.x_mid, nullptr, p[8], L.mean_f, L.rstd_f, T.dx_mid, T.dx_out, g[8], g[9]));   // T.dx_mid = d x_mid
        // ---- attention backward (x_mid = x_in + Wo O) ----
        FF_TRY(Gemm(s.dt, Mq, s.inner, s.D).a(0, pD).b(1, pI).c(pI).problem(T.dx_mid, p[7], W.dO).run(W.ws, gws, st));
        FF_TRY(attention_bwd(rs_attn_desc(s), L.Qs, L.K, L.V, nullptr, L.O, W.dO, L.lse, T.dQs, T.dK, T.dV, attn_ws,
                             (size_t)s.Bn * s.H * s.q * 4, st));
        // d kv_in = dK Wk + dV Wv
        FF_TRY(Gemm(s.dt, Mkv, s.D, s.inner).a(0, pI).b(1, pD).c(pD).problem(T.dK, p[5], W.dkv).run(W.ws, gws, st));
        FF_TRY(Gemm(s.dt, Mkv, s.D, s.inner).a(0, pI).b(1, pD).c(pD).problem(T.dV, p[6], W.dkv, nullptr, nullptr, W.dkv).run(W.ws, gws, st));
        // d LN(latents) = scale * dQs Wq + d kv_in[latent rows]
        FF_TRY(Gemm(s.dt, Mq, s.D, s.inner).a(0, pI).b(1, pD).c(pD).res_map(kv_lat).scale(s.scale)
                   .problem(T.dQs, p[4], W.dln, nullptr, nullptr, dkv_lat_base).run(W.ws, gws, st));
        // norm_latents backward, accumulated onto the residual path: d x_in
        FF_TRY(ln_bwd(3 * l + 1, ln_args(s.dt, Mq, s.D, x_map, pD, pD), W.dln, x_in, nullptr, p[2], L.mean_l, L.rstd_l, dx_below, T.dx_mid, g[2], g[3]));
        {   // norm_media backward: d x_f accumulates over the layers (needed for d time_pos_emb even with CLIP frozen)
            LnArgs a = ln_args(s.dt, Mf, s.D, pD, kv_media, pD);
            a.add_rows_per_seg = s.F; a.add_div = s.v;
            FF_TRY(ln_bwd(3 * l + 2, a, W.dkv, x_f, tpe, p[0], S.mean_m, S.rstd_m, dxf, l == s.depth - 1 ? nullptr : dxf, g[0], g[1]));
        }
    }
    FF_TRY(layernorm_bwd_finish(s.dt, ln_sets, 3 * s.depth + 1, st));
    // ---- weight gradients, grouped over up to kGemmMaxZ layers per launch ----
    constexpr int kRsGroup = 4;      // layers per grouped weight-gradient launch (their tile counts are whole rounds of the chip already)
    for (int l0 = 0; l0 < s.depth; l0 += kRsGroup) {
        const int l1 = std::min(s.depth, l0 + kRsGroup);
        Gemm g3(s.dt, s.D, s.ffi, Mq), g1(s.dt, s.ffi, s.D, Mq), go(s.dt, s.D, s.inner, Mq), gq(s.dt, s.inner, s.D, Mq);
        g3.a(1, pD).b(1, pF).c(pF);                       // d W3 = d x_out^T . act(H)
        g1.a(1, pF).b(1, pD).c(pD);                       // d W1 = d H^T . LN(x_mid)
        go.a(1, pD).b(1, pI).c(pI);                       // d Wo = d x_mid^T . O
        gq.a(1, pI).b(1, kv_lat).c(pD).scale(s.scale);    // d Wq = scale * d Qs^T . LN(latents)   (the latent rows of kv_in)
        for (int l = l0; l < l1; l++) {
            void* const* g = G + FF_RESAMPLER_GLOBAL_PARAMS + FF_RESAMPLER_LAYER_PARAMS * l;
            const RsLayerSaved& L = S.L[l];
            const RsLayerStash& T = W.L[l];
            g3.problem(T.dx_out, L.Aact, g[11]);
            g1.problem(T.dH, L.xn_f, g[10]);
            go.problem(T.dx_mid, L.O, g[7]);
            gq.problem(T.dQs, (const char*)L.kv_in + (size_t)s.F * s.D * s.es, g[4]);
        }
        FF_TRY(g3.run(W.ws, gws, st));
        FF_TRY(g1.run(W.ws, gws, st));
        FF_TRY(go.run(W.ws, gws, st));
        FF_TRY(gq.run(W.ws, gws, st));
    }
    {   // d Wk, d Wv = d K^T / d V^T . kv_in: two problems per layer
        Gemm gkv(s.dt, s.inner, s.D, Mkv);
        gkv.a(1, pI).b(1, pD).c(pD);
        for (int l = 0; l < s.depth; l++) {
            void* const* g = G + FF_RESAMPLER_GLOBAL_PARAMS + FF_RESAMPLER_LAYER_PARAMS * l;
            gkv.problem(W.L[l].dK, S.L[l].kv_in, g[5]);
            gkv.problem(W.L[l].dV, S.L[l].kv_in, g[6]);
            if (gkv.P.nz == kRsGroup || l == s.depth - 1) {
                FF_TRY(gkv.run(W.ws, gws, st));
                gkv.P.nz = 0;
            }
        }
    }
    void* dx_in = W.dx0;
    // d latents = sum over the batch of d x_0 (:179);  d time_pos_emb[t] = sum_{b, n} d x_f[b, t, n] (:166)
    FF_TRY(rows_reduce(s.dt, Mq, s.D, pD, s.q, 1, dx_in, G[0], W.ws, gws, st));
    if (s.nte > s.T) {
        hipError_t e = hipMemsetAsync((char*)G[1] + (size_t)s.T * s.D * s.es, 0, (size_t)(s.nte - s.T) * s.D * s.es, st);
        FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "resampler_bwd: memset: %s", hipGetErrorString(e));
    }
    return rows_reduce(s.dt, Mf, s.D, pD, s.F, s.v, dxf, G[1], W.ws, gws, st);
}

// =====================================================================================================
// PerceiverResampler, one layer per call (SURVEY 8-b2's minimum export set: ff_resampler_layer_fwd/bwd + prologue / epilogue)
// perceiver_resampler.py:181-183 is a per-layer loop; a caller that drives it layer by layer gets every layer's parameter gradients FINAL
// when that layer's backward call returns - one data-parallel bucket per layer, leaving while the layer below runs backward - at the price
// of what only the stack-level call can do (weight gradients of four layers per launch, LayerNorm finals of all layers in three launches).
// =====================================================================================================
struct RsProSaved { float *mean_m, *rstd_m; };
static size_t rs_pro_layout(const RsDims& s, void* base, size_t cap, RsProSaved& o) {
    Arena a(base, cap);
    o.mean_m = a.take<float>((size_t)s.Bn * s.F * 4);
    o.rstd_m = a.take<float>((size_t)s.Bn * s.F * 4);
    return align_up(a.used);
}
static size_t rs_layer_saved_layout(const RsDims& s, void* base, size_t cap, RsLayerSaved& L) {
    Arena a(base, cap);
    const size_t rows_q = (size_t)s.Bn * s.q, rows_kv = (size_t)s.Bn * s.R;
    L.x_in = nullptr;                           // the caller's tensor
    L.mean_l = a.take<float>(rows_q * 4);
    L.rstd_l = a.take<float>(rows_q * 4);
    L.kv_in = a.take(rows_kv * s.D * s.es);
    L.Qs = a.take(rows_q * s.inner * s.es);
    L.K = a.take(rows_kv * s.inner * s.es);
    L.V = a.take(rows_kv * s.inner * s.es);
    L.lse = a.take<float>((size_t)s.Bn * s.H * s.q * 4);
    L.O = a.take(rows_q * s.inner * s.es);
    L.x_mid = a.take(rows_q * s.D * s.es);
    L.mean_f = a.take<float>(rows_q * 4);
    L.rstd_f = a.take<float>(rows_q * 4);
    L.xn_f = a.take(rows_q * s.D * s.es);
    L.Hpre = a.take(rows_q * s.ffi * s.es);
    L.Aact = a.take(rows_q * s.ffi * s.es);
    return align_up(a.used);
}
struct RsLayerScratch {
    void *dxn, *dO, *dkv, *dln, *dx_mid, *dH, *dQs, *dK, *dV, *ws;
    float* lnp[3];
    size_t ws_bytes;
};
static size_t rs_layer_scratch_layout(const RsDims& s, void* base, size_t cap, RsLayerScratch& o) {
    Arena a(base, cap);
    const size_t rows_q = (size_t)s.Bn * s.q, rows_kv = (size_t)s.Bn * s.R;
    o.ws_bytes = rs_ws_bytes(s);
    o.ws = a.take(o.ws_bytes);
    o.dxn = a.take(rows_q * s.D * s.es);
    o.dO = a.take(rows_q * s.inner * s.es);
    o.dkv = a.take(rows_kv * s.D * s.es);
    o.dln = a.take(rows_q * s.D * s.es);
    o.dx_mid = a.take(rows_q * s.D * s.es);
    o.dH = a.take(rows_q * s.ffi * s.es);
    o.dQs = a.take(rows_q * s.inner * s.es);
    o.dK = a.take(rows_kv * s.inner * s.es);
    o.dV = a.take(rows_kv * s.inner * s.es);
    const bool lnd = layernorm_bwd_deferrable(s.dt, s.D);
    for (int i = 0; i < 3; i++)       // [0] ff norm, [1] norm_latents, [2] norm_media
        o.lnp[i] = lnd ? a.take<float>(layernorm_bwd_partial_bytes(i == 2 ? s.Bn * s.F : (int)rows_q, s.D)) : nullptr;
    return align_up(a.used);
}

static int rs_prologue_fwd(const ff_resampler_desc* d, const void* x_f, const void* tpe, void* pro, size_t pro_bytes, hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(x_f && tpe && pro, FF_ERR_SHAPE, "resampler_prologue_fwd: null argument");
    const RsDims s(*d);
    RsProSaved S;
    FF_CHECK(rs_pro_layout(s, pro, pro_bytes, S) <= pro_bytes, FF_ERR_WORKSPACE, "resampler_prologue_fwd: saved buffer too small");
    LnArgs a = ln_args(s.dt, s.Bn * s.F, s.D, plain_rows(s.D), plain_rows(s.D), plain_rows(s.D));
    a.add_rows_per_seg = s.F; a.add_div = s.v;       // statistics of x_f + time_pos_emb, shared by the norm_media of every layer (:166, :52)
    return layernorm_fwd(a, x_f, tpe, nullptr, nullptr, nullptr, S.mean_m, S.rstd_m, st);
}

static int rs_layer_fwd(const ff_resampler_desc* d, const void* x_f, const void* tpe, const void* pro, size_t pro_bytes, const void* x_in,
                        int x_is_latents, const void* const* p, void* x_out, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                        hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(x_f && tpe && pro && x_in && p && x_out && saved && scratch, FF_ERR_SHAPE, "resampler_layer_fwd: null argument");
    const RsDims s(*d);
    RsProSaved Pr;
    RsLayerSaved L;
    RsLayerScratch W;
    FF_CHECK(rs_pro_layout(s, (void*)pro, pro_bytes, Pr) <= pro_bytes, FF_ERR_WORKSPACE, "resampler_layer_fwd: prologue buffer too small");
    FF_CHECK(rs_layer_saved_layout(s, saved, saved_bytes, L) <= saved_bytes, FF_ERR_WORKSPACE, "resampler_layer_fwd: saved buffer too small");
    FF_CHECK(rs_layer_scratch_layout(s, scratch, scratch_bytes, W) <= scratch_bytes, FF_ERR_WORKSPACE, "resampler_layer_fwd: scratch too small");
    const int Mq = s.Bn * s.q, Mkv = s.Bn * s.R, Mf = s.Bn * s.F;
    const RowMap pD = plain_rows(s.D), pI = plain_rows(s.inner), pF = plain_rows(s.ffi);
    const RowMap x_map = x_is_latents ? RowMap{s.D, 0, s.q} : pD;                    // latents (q, D) repeated over the batch (:179)
    const RowMap kv_media = RowMap{s.D, (long long)s.R * s.D, s.F}, kv_lat = RowMap{s.D, (long long)s.R * s.D, s.q};
    char* kv_lat_base = (char*)L.kv_in + (size_t)s.F * s.D * s.es;
    {   // norm_media -> media rows of kv_in (:52,65)
        LnArgs a = ln_args(s.dt, Mf, s.D, pD, kv_media, pD);
        a.add_rows_per_seg = s.F; a.add_div = s.v; a.stats_given = 1;
        FF_TRY(layernorm_fwd(a, x_f, tpe, p[0], p[1], L.kv_in, Pr.mean_m, Pr.rstd_m, st));
    }
    FF_TRY(layernorm_fwd(ln_args(s.dt, Mq, s.D, x_map, kv_lat, pD), x_in, nullptr, p[2], p[3], kv_lat_base, L.mean_l, L.rstd_l, st));   // :53,65
    FF_TRY(Gemm(s.dt, Mq, s.inner, s.D).a(0, kv_lat).b(0, pD).c(pI).scale(s.scale).problem(kv_lat_base, p[4], L.Qs).run(W.ws, W.ws_bytes, st));   // :57,79
    FF_TRY(Gemm(s.dt, Mkv, s.inner, s.D).a(0, pD).b(0, pD).c(pI).problem(L.kv_in, p[5], L.K).problem(L.kv_in, p[6], L.V).run(W.ws, W.ws_bytes, st));   // :69-70
    FF_TRY(attention_fwd(rs_attn_desc(s), L.Qs, L.K, L.V, nullptr, L.O, L.lse, st));                                                  // :85-92
    FF_TRY(Gemm(s.dt, Mq, s.D, s.inner).a(0, pI).b(0, pI).c(pD).res_map(x_map).problem(L.O, p[7], L.x_mid, nullptr, nullptr, x_in).run(W.ws, W.ws_bytes, st));   // :96, :182
    FF_TRY(layernorm_fwd(ln_args(s.dt, Mq, s.D, pD, pD, pD), L.x_mid, nullptr, p[8], p[9], L.xn_f, L.mean_f, L.rstd_f, st));          // :183; utils.py:45-50
    FF_TRY(Gemm(s.dt, Mq, s.ffi, s.D).a(0, pD).b(0, pD).c(pF).act(s.act).problem(L.xn_f, p[10], L.Aact, L.Hpre).run(W.ws, W.ws_bytes, st));
    return Gemm(s.dt, Mq, s.D, s.ffi).a(0, pF).b(0, pF).c(pD).problem(L.Aact, p[11], x_out, nullptr, nullptr, L.x_mid).run(W.ws, W.ws_bytes, st);
}

static int rs_layer_bwd(const ff_resampler_desc* d, const void* x_f, const void* tpe, const void* pro, size_t pro_bytes, const void* x_in,
                        int x_is_latents, const void* const* p, const void* dx_out, const void* saved, size_t saved_bytes, void* const* g,
                        void* dx_in, void* dx_f, int dx_f_accumulate, void* scratch, size_t scratch_bytes, hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(x_f && tpe && pro && x_in && p && dx_out && saved && g && dx_in && dx_f && scratch, FF_ERR_SHAPE, "resampler_layer_bwd: null argument");
    const RsDims s(*d);
    RsProSaved Pr;
    RsLayerSaved L;
    RsLayerScratch W;
    FF_CHECK(rs_pro_layout(s, (void*)pro, pro_bytes, Pr) <= pro_bytes, FF_ERR_WORKSPACE, "resampler_layer_bwd: prologue buffer too small");
    FF_CHECK(rs_layer_saved_layout(s, (void*)saved, saved_bytes, L) <= saved_bytes, FF_ERR_WORKSPACE, "resampler_layer_bwd: saved buffer too small");
    FF_CHECK(rs_layer_scratch_layout(s, scratch, scratch_bytes, W) <= scratch_bytes, FF_ERR_WORKSPACE, "resampler_layer_bwd: scratch too small");
    const int Mq = s.Bn * s.q, Mkv = s.Bn * s.R, Mf = s.Bn * s.F;
    const RowMap pD = plain_rows(s.D), pI = plain_rows(s.inner), pF = plain_rows(s.ffi);
    const RowMap x_map = x_is_latents ? RowMap{s.D, 0, s.q} : pD;
    const RowMap kv_media = RowMap{s.D, (long long)s.R * s.D, s.F}, kv_lat = RowMap{s.D, (long long)s.R * s.D, s.q};
    float* attn_ws = (float*)((char*)W.ws + W.ws_bytes - align_up((size_t)s.Bn * s.H * s.q * 4));
    const size_t gws = W.ws_bytes - align_up((size_t)s.Bn * s.H * s.q * 4);
    char* dkv_lat_base = (char*)W.dkv + (size_t)s.F * s.D * s.es;
    const char* kv_lat_base = (const char*)L.kv_in + (size_t)s.F * s.D * s.es;
    LnPending ln_sets[3];
    auto ln_bwd = [&](int slot, const LnArgs& a, const void* dy_, const void* x_, const void* add_, const void* gamma_, const float* mean_,
                      const float* rstd_, void* dx_, const void* dx_res_, void* dg_, void* db_) -> int {
        if (W.lnp[slot])
            return layernorm_bwd(a, dy_, x_, add_, gamma_, mean_, rstd_, dx_, dx_res_, dg_, db_, W.lnp[slot], layernorm_bwd_partial_bytes(a.rows, a.cols), st,
                                 nullptr, &ln_sets[slot]);
        return layernorm_bwd(a, dy_, x_, add_, gamma_, mean_, rstd_, dx_, dx_res_, dg_, db_, W.ws, gws, st);
    };
    // ---- FeedForward backward (x_out = x_mid + W3 act(W1 LN(x_mid))) ----
    FF_TRY(Gemm(s.dt, Mq, s.ffi, s.D).a(0, pD).b(1, pF).c(pF).act_bwd(s.act).problem(dx_out, p[11], W.dH, nullptr, L.Hpre).run(W.ws, gws, st));
    FF_TRY(Gemm(s.dt, Mq, s.D, s.ffi).a(0, pF).b(1, pD).c(pD).problem(W.dH, p[10], W.dxn).run(W.ws, gws, st));
    FF_TRY(ln_bwd(0, ln_args(s.dt, Mq, s.D, pD, pD, pD), W.dxn, L.x_mid, nullptr, p[8], L.mean_f, L.rstd_f, W.dx_mid, dx_out, g[8], g[9]));
    // ---- attention backward (x_mid = x_in + Wo O) ----
    FF_TRY(Gemm(s.dt, Mq, s.inner, s.D).a(0, pD).b(1, pI).c(pI).problem(W.dx_mid, p[7], W.dO).run(W.ws, gws, st));
    FF_TRY(attention_bwd(rs_attn_desc(s), L.Qs, L.K, L.V, nullptr, L.O, W.dO, L.lse, W.dQs, W.dK, W.dV, attn_ws, (size_t)s.Bn * s.H * s.q * 4, st));
    FF_TRY(Gemm(s.dt, Mkv, s.D, s.inner).a(0, pI).b(1, pD).c(pD).problem(W.dK, p[5], W.dkv).run(W.ws, gws, st));
    FF_TRY(Gemm(s.dt, Mkv, s.D, s.inner).a(0, pI).b(1, pD).c(pD).problem(W.dV, p[6], W.dkv, nullptr, nullptr, W.dkv).run(W.ws, gws, st));
    FF_TRY(Gemm(s.dt, Mq, s.D, s.inner).a(0, pI).b(1, pD).c(pD).res_map(kv_lat).scale(s.scale)
               .problem(W.dQs, p[4], W.dln, nullptr, nullptr, dkv_lat_base).run(W.ws, gws, st));
    FF_TRY(ln_bwd(1, ln_args(s.dt, Mq, s.D, x_map, pD, pD), W.dln, x_in, nullptr, p[2], L.mean_l, L.rstd_l, dx_in, W.dx_mid, g[2], g[3]));
    {   // norm_media backward: d x_f accumulates over the layers (needed for d time_pos_emb even with CLIP frozen)
        LnArgs a = ln_args(s.dt, Mf, s.D, pD, kv_media, pD);
        a.add_rows_per_seg = s.F; a.add_div = s.v;
        FF_TRY(ln_bwd(2, a, W.dkv, x_f, tpe, p[0], Pr.mean_m, Pr.rstd_m, dx_f, dx_f_accumulate ? dx_f : nullptr, g[0], g[1]));
    }
    FF_TRY(layernorm_bwd_finish(s.dt, ln_sets, 3, st));
    // ---- this layer's weight gradients, final when the call returns ----
    FF_TRY(Gemm(s.dt, s.D, s.ffi, Mq).a(1, pD).b(1, pF).c(pF).problem(dx_out, L.Aact, g[11]).run(W.ws, gws, st));          // d W3 = d x_out^T . act(H)
    FF_TRY(Gemm(s.dt, s.ffi, s.D, Mq).a(1, pF).b(1, pD).c(pD).problem(W.dH, L.xn_f, g[10]).run(W.ws, gws, st));           // d W1 = d H^T . LN(x_mid)
    FF_TRY(Gemm(s.dt, s.D, s.inner, Mq).a(1, pD).b(1, pI).c(pI).problem(W.dx_mid, L.O, g[7]).run(W.ws, gws, st));         // d Wo = d x_mid^T . O
    FF_TRY(Gemm(s.dt, s.inner, s.D, Mq).a(1, pI).b(1, kv_lat).c(pD).scale(s.scale).problem(W.dQs, kv_lat_base, g[4]).run(W.ws, gws, st));   // d Wq
    return Gemm(s.dt, s.inner, s.D, Mkv).a(1, pI).b(1, pD).c(pD).problem(W.dK, L.kv_in, g[5]).problem(W.dV, L.kv_in, g[6]).run(W.ws, gws, st);  // d Wk, d Wv
}

static int rs_prologue_bwd(const ff_resampler_desc* d, const void* dx0, const void* dx_f, void* d_latents, void* d_tpe, void* scratch,
                           size_t scratch_bytes, hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(dx0 && dx_f && d_latents && d_tpe && scratch, FF_ERR_SHAPE, "resampler_prologue_bwd: null argument");
    const RsDims s(*d);
    RsLayerScratch W;
    FF_CHECK(rs_layer_scratch_layout(s, scratch, scratch_bytes, W) <= scratch_bytes, FF_ERR_WORKSPACE, "resampler_prologue_bwd: scratch too small");
    const RowMap pD = plain_rows(s.D);
    // d latents = sum over the batch of d x_0 (:179);  d time_pos_emb[t] = sum_{b, n} d x_f[b, t, n] (:166)
    FF_TRY(rows_reduce(s.dt, s.Bn * s.q, s.D, pD, s.q, 1, dx0, d_latents, W.ws, W.ws_bytes, st));
    if (s.nte > s.T) {
        hipError_t e = hipMemsetAsync((char*)d_tpe + (size_t)s.T * s.D * s.es, 0, (size_t)(s.nte - s.T) * s.D * s.es, st);
        FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "resampler_prologue_bwd: memset: %s", hipGetErrorString(e));
    }
    return rows_reduce(s.dt, s.Bn * s.F, s.D, pD, s.F, s.v, dx_f, d_tpe, W.ws, W.ws_bytes, st);
}

static int rs_epilogue_fwd(const ff_resampler_desc* d, const void* x_last, const void* gamma, const void* beta, void* out, void* epi, size_t epi_bytes,
                           hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(x_last && gamma && beta && out && epi, FF_ERR_SHAPE, "resampler_epilogue_fwd: null argument");
    const RsDims s(*d);
    const size_t rows_q = (size_t)s.Bn * s.q;
    FF_CHECK(epi_bytes >= 2 * align_up(rows_q * 4), FF_ERR_WORKSPACE, "resampler_epilogue_fwd: saved buffer too small");
    float* mean = (float*)epi;
    float* rstd = (float*)((char*)epi + align_up(rows_q * 4));
    return layernorm_fwd(ln_args(s.dt, (int)rows_q, s.D, plain_rows(s.D), plain_rows(s.D), plain_rows(s.D)), x_last, nullptr, gamma, beta, out, mean, rstd, st);   // :187
}
static int rs_epilogue_bwd(const ff_resampler_desc* d, const void* dout, const void* x_last, const void* gamma, const void* epi, size_t epi_bytes,
                           void* dx_last, void* dgamma, void* dbeta, void* scratch, size_t scratch_bytes, hipStream_t st) {
    FF_TRY(rs_check(d));
    FF_CHECK(dout && x_last && gamma && epi && dx_last && dgamma && dbeta && scratch, FF_ERR_SHAPE, "resampler_epilogue_bwd: null argument");
    const RsDims s(*d);
    const size_t rows_q = (size_t)s.Bn * s.q;
    FF_CHECK(epi_bytes >= 2 * align_up(rows_q * 4), FF_ERR_WORKSPACE, "resampler_epilogue_bwd: saved buffer too small");
    RsLayerScratch W;
    FF_CHECK(rs_layer_scratch_layout(s, scratch, scratch_bytes, W) <= scratch_bytes, FF_ERR_WORKSPACE, "resampler_epilogue_bwd: scratch too small");
    const float* mean = (const float*)epi;
    const float* rstd = (const float*)((const char*)epi + align_up(rows_q * 4));
    return layernorm_bwd(ln_args(s.dt, (int)rows_q, s.D, plain_rows(s.D), plain_rows(s.D), plain_rows(s.D)), dout, x_last, nullptr, gamma, mean, rstd, dx_last,
                         nullptr, dgamma, dbeta, W.ws, W.ws_bytes, st);
}

// =====================================================================================================
// GatedCrossAttentionBlock
// =====================================================================================================
struct XaDims {
    int dt, b, L, d, dv, Nm, nv, Nk, H, dh, inner, ffi, act;
    size_t es;
    float scale;
    bool ext_kv;     // keys / values come from outside the block (cached decode, ff_kv_project_fwd): no K/V region in `saved`
    explicit XaDims(const ff_xattn_desc& x) {
        ext_kv = x.cached_k.sr != 0;
        dt = x.dtype; b = x.batch; L = x.n_tokens; d = x.dim; dv = x.dim_visual; Nm = x.n_media; nv = x.n_visual; Nk = Nm * nv;
        H = x.heads; dh = x.dim_head; inner = H * dh; ffi = x.ff_mult * d; act = x.act; es = dtype_size(dt);
        scale = 1.0f / sqrtf((float)dh);
    }
};
struct XaSaved {
    float *mean_a, *rstd_a, *lse, *mean_f, *rstd_f;
    void *yn, *Qs, *KV, *O, *attn_out, *y1, *xn_f, *Hpre, *Aact, *ffw_out;
    size_t kv_offset;
};
static size_t xa_saved_layout(const XaDims& s, void* base, size_t cap, XaSaved& o) {
    Arena a(base, cap);
    const size_t M = (size_t)s.b * s.L;
    o.KV = s.ext_kv ? nullptr : a.take((size_t)s.b * s.Nk * 2 * s.inner * s.es);   // first: its offset is part of the ABI (ff_xattn_kv_offset)
    o.kv_offset = 0;
    o.mean_a = a.take<float>(M * 4);
    o.rstd_a = a.take<float>(M * 4);
    o.yn = a.take(M * s.d * s.es);
    o.Qs = a.take(M * s.inner * s.es);
    o.lse = a.take<float>((size_t)s.b * s.H * s.L * 4);
    o.O = a.take(M * s.inner * s.es);
    o.attn_out = a.take(M * s.d * s.es);
    o.y1 = a.take(M * s.d * s.es);
    o.mean_f = a.take<float>(M * 4);
    o.rstd_f = a.take<float>(M * 4);
    o.xn_f = a.take(M * s.d * s.es);
    o.Hpre = a.take(M * s.ffi * s.es);
    o.Aact = a.take(M * s.ffi * s.es);
    o.ffw_out = a.take(M * s.d * s.es);
    return align_up(a.used);
}
struct XaScratch {
    void *dy1, *dH, *dxn, *dO, *dQs, *dKV, *dyn, *ws;
    size_t ws_bytes;
    float* lnx;      // exchange buffer of the resident kernels' phase 3 (per-row pairs of every (sample, head) workgroup)
};
// What the four weight-gradient GEMMs of a block read besides `saved` and d y_out: kept in a caller-owned `stash` when they are
// deferred (ff_xattn_block_bwd_kv_data -> ff_xattn_wgrad_grouped), in `scratch` otherwise.
struct XaStash {
    void *dy1, *dH, *dQs;
    float *lnp_f = nullptr, *lnp_a = nullptr;   // per-workgroup partials of the two LayerNorm backwards (their final reduction is postponed too)
};
// The ~5 us final reductions of the two LayerNorm backwards (d gamma, d beta, both gate gradients) are off the data-gradient chain as
// well: in the deferred mode their partials stay in the stash and ff_xattn_wgrad_grouped finishes up to 8 of them with one launch.
static bool xa_ln_deferrable(const XaDims& s) {      // development builds: FF_DEFER_LN=0 finishes every LayerNorm backward on the spot (A/B timing)
    static const int on = dbg_switch("FF_DEFER_LN", 1);
    return on != 0 && layernorm_bwd_deferrable(s.dt, s.d);
}
static size_t xa_stash_layout(const XaDims& s, void* base, size_t cap, XaStash& o) {
    Arena a(base, cap);
    const size_t M = (size_t)s.b * s.L;
    o.dy1 = a.take(M * s.d * s.es);
    o.dH = a.take(M * s.ffi * s.es);
    o.dQs = a.take(M * s.inner * s.es);
    if (xa_ln_deferrable(s)) {
        o.lnp_f = a.take<float>(layernorm_bwd_partial_bytes((int)M, s.d));
        o.lnp_a = a.take<float>(layernorm_bwd_partial_bytes((int)M, s.d));
    }
    return align_up(a.used);
}
static size_t xa_ws_bytes(const XaDims& s) {
    const int M = s.b * s.L, Mk = s.b * s.Nk;
    size_t w = 0;
    auto g = [&](int m, int n, int k) { w = std::max(w, gemm_workspace_bytes(s.dt, m, n, k, 1, 0)); };
    g(M, s.inner, s.d); g(Mk, 2 * s.inner, s.dv); g(M, s.d, s.inner); g(M, s.ffi, s.d); g(M, s.d, s.ffi);
    g(s.d, s.ffi, M); g(s.ffi, s.d, M); g(s.d, s.inner, M); g(s.inner, s.d, M); g(2 * s.inner, s.dv, Mk); g(Mk, s.dv, 2 * s.inner);
    w = std::max(w, layernorm_bwd_workspace(M, s.d));
    w = std::max(w, gate_grad_workspace(M, s.d));
    if (decode_ffw_supported(s.dt, M, s.d, s.ffi)) w = std::max(w, decode_ffw_workspace_bytes(s.d, s.ffi));
    return align_up(w) + align_up((size_t)s.b * s.H * s.L * 4);
}
static size_t xa_scratch_layout(const XaDims& s, void* base, size_t cap, bool bwd, XaScratch& o) {
    Arena a(base, cap);
    const size_t M = (size_t)s.b * s.L;
    o.ws_bytes = xa_ws_bytes(s);
    o.ws = a.take(o.ws_bytes);
    o.lnx = a.take<float>(xa_ln3_part_bytes(s.b, s.H));
    if (bwd) {
        o.dy1 = a.take(M * s.d * s.es);
        o.dH = a.take(M * s.ffi * s.es);
        o.dxn = a.take(M * s.d * s.es);
        o.dO = a.take(M * s.inner * s.es);
        o.dQs = a.take(M * s.inner * s.es);
        o.dKV = s.ext_kv ? nullptr : a.take((size_t)s.b * s.Nk * 2 * s.inner * s.es);
        o.dyn = a.take(M * s.d * s.es);
    }
    return align_up(a.used);
}
static int xa_check(const ff_xattn_desc* d) {
    FF_CHECK(d, FF_ERR_SHAPE, "xattn: null descriptor");
    FF_CHECK(d->dtype == FF_DTYPE_F32 || d->dtype == FF_DTYPE_BF16, FF_ERR_UNSUPPORTED, "xattn: dtype %d", d->dtype);
    FF_CHECK(d->batch > 0 && d->n_tokens > 0 && d->dim > 0 && d->dim_visual > 0 && d->n_media > 0 && d->n_visual > 0 && d->heads > 0 &&
                 d->dim_head > 0 && d->ff_mult > 0,
             FF_ERR_SHAPE, "xattn: bad dimensions");
    FF_CHECK(d->act >= FF_ACT_GELU && d->act <= FF_ACT_RELU, FF_ERR_SHAPE, "xattn: act %d", d->act);
    return FF_OK;
}
static ff_attn_desc xa_attn_desc(const ff_xattn_desc& x, const XaDims& s, bool cached) {
    ff_attn_desc a = {};
    a.dtype = s.dt; a.batch = s.b; a.heads = s.H; a.dim_head = s.dh; a.n_q = s.L; a.n_kv = s.Nk;
    a.mode = FF_ATTN_MEDIA; a.n_visual = s.nv; a.tt_stride = x.tt_stride; a.tt_offset = x.tt_offset;
    const ff_strides sq = {(long long)s.L * s.inner, s.inner, s.dh};
    const ff_strides sk = {(long long)s.Nk * 2 * s.inner, 2LL * s.inner, s.dh};
    a.q = sq; a.o = sq; a.dq = sq; a.dout = sq;
    a.k = cached ? x.cached_k : sk;
    a.v = cached ? x.cached_v : sk;
    a.dk = sk; a.dv = sk;
    return a;
}

// development builds: FF_XATTN_FUSED=0 falls back to the separate LayerNorm / projection GEMM / attention launches (A/B timing)
static bool xa_fused_enabled() {
    static const int v = dbg_switch("FF_XATTN_FUSED", 1);
    return v != 0;
}
// development builds: FF_XATTN_LN3=0 keeps the LayerNorm behind phase 2 (forward: LN(y1); backward: the backward of LN(y)) as a launch of its own
static bool xa_ln3_enabled() {
    static const int v = dbg_switch("FF_XATTN_LN3", 1);
    return v != 0;
}
static XaFusedArgs xa_fused_args(const ff_xattn_desc& x, const XaDims& s, bool ext_kv) {
    XaFusedArgs a = {};
    a.batch = s.b; a.heads = s.H; a.n_q = s.L; a.n_kv = s.Nk; a.n_visual = s.nv; a.tt_stride = x.tt_stride; a.tt_offset = x.tt_offset;
    a.dim = s.d; a.inner = s.inner; a.scale = s.scale; a.eps = 1e-5f;
    static const int xcd_split = dbg_switch("FF_XATTN_XCD_SPLIT", 0);
    a.xcd_split = xcd_split;
    const ff_strides sk = {(long long)s.Nk * 2 * s.inner, 2LL * s.inner, s.dh};
    a.k = ext_kv ? x.cached_k : sk;
    a.v = ext_kv ? x.cached_v : sk;
    a.dk = sk; a.dv = sk;
    return a;
}

// Whether xattn_bwd runs the backward of LN(y) inside the fused attention-backward launch (phase 3).  A function of the descriptor alone:
// xattn_wgrad_grouped must know how many partial blocks that pass left in the stash (one per SAMPLE then, not one per 4 rows).
static bool xa_ln3_in_bwd(const ff_xattn_desc& x, const XaDims& s) {
    if (!xa_ln3_enabled() || !xa_fused_enabled() || !xa_fused_supported(s.dt, s.dh, s.d, s.inner) || x.sync == nullptr) return false;
    const XaFusedArgs fa = xa_fused_args(x, s, s.ext_kv);
    const int M = s.b * s.L;
    return xa_out_fusable(fa, s.dt, s.dh) && s.b <= layernorm_bwd_partial_blocks(M) && (size_t)s.b * (2 * (size_t)s.d + 2) * sizeof(float) <= layernorm_bwd_workspace(M, s.d);
}
static int xattn_fwd(const ff_xattn_desc* d, const void* y, const void* vf, const int* tt, const void* const* P, const void* ck,
                     const void* cv, void* y_out, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, hipStream_t st) {
    FF_TRY(xa_check(d));
    const bool cached = ck != nullptr;
    FF_CHECK(y && tt && P && y_out && saved && scratch && (cached ? cv != nullptr : vf != nullptr), FF_ERR_SHAPE, "xattn_fwd: null argument");
    const XaDims s(*d);
    FF_CHECK(s.ext_kv == cached, FF_ERR_SHAPE, "xattn_fwd: external K / V need their strides in the descriptor (cached_k / cached_v) and vice versa");
    XaSaved S;
    XaScratch W;
    FF_CHECK(xa_saved_layout(s, saved, saved_bytes, S) <= saved_bytes, FF_ERR_WORKSPACE, "xattn_fwd: saved buffer too small");
    FF_CHECK(xa_scratch_layout(s, scratch, scratch_bytes, false, W) <= scratch_bytes, FF_ERR_WORKSPACE, "xattn_fwd: scratch too small");
    const int M = s.b * s.L, Mk = s.b * s.Nk;
    const RowMap pd = plain_rows(s.d), pI = plain_rows(s.inner), pF = plain_rows(s.ffi), pV = plain_rows(s.dv), pKV = plain_rows(2 * s.inner);
    const void *Kp, *Vp;
    if (!cached) {  // k, v = to_kv(flatten(visual_features)).chunk(2) (:84-86): K = columns [0, inner), V = [inner, 2 inner)
        FF_TRY(Gemm(s.dt, Mk, 2 * s.inner, s.dv).a(0, pV).b(0, pV).c(pKV).problem(vf, P[5], S.KV).run(W.ws, W.ws_bytes, st));
        Kp = S.KV;
        Vp = (const char*)S.KV + (size_t)s.inner * s.es;
    } else {
        Kp = ck; Vp = cv;
    }
    bool out_fused = false, ln_fused = false;
    if (xa_fused_enabled() && xa_fused_supported(s.dt, s.dh, s.d, s.inner)) {
        // y = norm(y); q = to_q(y) * scale; masked softmax(q k^T) v (:74-78, :95-123) in ONE launch per block
        // ... and, with a sync buffer at the training / decode shape, y1 = y + tanh(alpha_attn) * to_out(o) (:126, :180) in the same launch
        const XaFusedArgs fa = xa_fused_args(*d, s, cached);
        XaOutArgs oa = {(const bf16*)P[6], (const bf16*)P[0], (bf16*)S.y1, (bf16*)S.attn_out, (unsigned*)d->sync};
        // (not at the decode shape, <= 32 rows in all: there `to_out` is a 3.7 us weight-streaming launch and the in-launch exchange costs more
        // than that launch and its boundary - measured, r5s1: phase 2 adds ~10 us to the fused launch)
        out_fused = d->sync != nullptr && xa_out_fusable(fa, s.dt, s.dh) && !decode_ffw_supported(s.dt, M, s.d, s.ffi);
        if (out_fused && xa_ln3_enabled()) {      // ... and LN(y1) of the feed-forward (utils.py:46) behind it, still in the same launch (phase 3)
            oa.ln_g = (const bf16*)P[7]; oa.ln_b = (const bf16*)P[8]; oa.ln_out = (bf16*)S.xn_f; oa.ln_mean = S.mean_f; oa.ln_rstd = S.rstd_f;
            oa.ln_part = W.lnx;
            ln_fused = true;
        }
        FF_TRY(xa_qattn_fwd(fa, s.dt, s.dh, y, P[2], P[3], P[4], Kp, Vp, tt, S.yn, S.Qs, S.O, S.mean_a, S.rstd_a, S.lse, st, out_fused ? &oa : nullptr));
    } else {
        // y = norm(y); q = to_q(y) * scale (:74-78)
        FF_TRY(layernorm_fwd(ln_args(s.dt, M, s.d, pd, pd, pd), y, nullptr, P[2], P[3], S.yn, S.mean_a, S.rstd_a, st));
        FF_TRY(Gemm(s.dt, M, s.inner, s.d).a(0, pd).b(0, pd).c(pI).scale(s.scale).problem(S.yn, P[4], S.Qs).run(W.ws, W.ws_bytes, st));
        FF_TRY(attention_fwd(xa_attn_desc(*d, s, cached), S.Qs, Kp, Vp, tt, S.O, S.lse, st));           // :95-123
    }
    // y = y + tanh(alpha_attn) * to_out(o) (:126, :180)
    if (!out_fused) FF_TRY(Gemm(s.dt, M, s.d, s.inner).a(0, pI).b(0, pI).c(pd).problem(S.O, P[6], S.y1, S.attn_out, nullptr, y, P[0]).run(W.ws, W.ws_bytes, st));
    // y = y + tanh(alpha_ffw) * ffw(y) (:182)
    if (decode_ffw_supported(s.dt, M, s.d, s.ffi))      // <= 32 rows (the cached decode step): LayerNorm + up-projection and down-projection + gate + residual, two launches
        return decode_ffw(M, s.d, s.ffi, s.act, 1e-5f, S.y1, P[7], P[8], P[9], P[10], P[1], S.xn_f, S.mean_f, S.rstd_f, S.Hpre, S.Aact, S.ffw_out, y_out,
                          W.ws, W.ws_bytes, st);
    if (!ln_fused) FF_TRY(layernorm_fwd(ln_args(s.dt, M, s.d, pd, pd, pd), S.y1, nullptr, P[7], P[8], S.xn_f, S.mean_f, S.rstd_f, st));
    FF_TRY(Gemm(s.dt, M, s.ffi, s.d).a(0, pd).b(0, pd).c(pF).act(s.act).problem(S.xn_f, P[9], S.Aact, S.Hpre).run(W.ws, W.ws_bytes, st));
    return Gemm(s.dt, M, s.d, s.ffi).a(0, pF).b(0, pF).c(pd).problem(S.Aact, P[10], y_out, S.ffw_out, nullptr, S.y1, P[1]).run(W.ws, W.ws_bytes, st);
}

// ext_k / ext_v != null: the keys / values were projected outside the block (ff_kv_project_fwd, strides d->cached_k / cached_v);
// the block then hands d K / d V to `dkv_out` (same (b, n_kv, 2, heads, dim_head) layout as the projection's output) instead of
// computing d to_kv.weight and d visual_features itself.
// stash != null: the four weight-gradient GEMMs (d ffw.3, d ffw.1, d to_out, d to_q) are NOT run; their operands d y1, d H, d Qs are
// left in `stash` for xattn_wgrad_grouped.  The LayerNorm / gate gradients (G[0..3], G[7], G[8]) are always produced here.
static int xattn_bwd(const ff_xattn_desc* d, const void* y, const void* vf, const int* tt, const void* const* P, const void* dy2,
                     const void* saved, size_t saved_bytes, void* const* G, void* dy, void* dvf, void* scratch, size_t scratch_bytes,
                     hipStream_t st, const void* ext_k = nullptr, const void* ext_v = nullptr, void* dkv_out = nullptr,
                     void* stash = nullptr, size_t stash_bytes = 0) {
    FF_TRY(xa_check(d));
    const bool hoisted = ext_k != nullptr;
    FF_CHECK(y && tt && P && dy2 && saved && G && dy && scratch && (hoisted ? (ext_v && dkv_out) : vf != nullptr), FF_ERR_SHAPE,
             "xattn_bwd: null argument");
    const XaDims s(*d);
    FF_CHECK(s.ext_kv == hoisted, FF_ERR_SHAPE, "xattn_bwd: external K / V need their strides in the descriptor (cached_k / cached_v) and vice versa");
    XaSaved S;
    XaScratch W;
    FF_CHECK(xa_saved_layout(s, (void*)saved, saved_bytes, S) <= saved_bytes, FF_ERR_WORKSPACE, "xattn_bwd: saved buffer too small");
    FF_CHECK(xa_scratch_layout(s, scratch, scratch_bytes, true, W) <= scratch_bytes, FF_ERR_WORKSPACE, "xattn_bwd: scratch too small");
    XaStash T{W.dy1, W.dH, W.dQs};
    const bool defer = stash != nullptr;
    if (defer) FF_CHECK(xa_stash_layout(s, stash, stash_bytes, T) <= stash_bytes, FF_ERR_WORKSPACE, "xattn_bwd: stash too small");
    const bool defer_ln = defer && xa_ln_deferrable(s);
    if (defer_ln)   // xattn_wgrad_grouped cannot know what happened here, so the one-pass LayerNorm backward must apply: rows 16-byte aligned
        FF_CHECK(((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)P[2] | (uintptr_t)P[7]) % 16 == 0, FF_ERR_SHAPE,
                 "xattn_bwd (deferred weight gradients): y, d y_out, d y and the LayerNorm weights must be 16-byte aligned");
    LnPending pend_f, pend_a;
    const int M = s.b * s.L, Mk = s.b * s.Nk;
    const RowMap pd = plain_rows(s.d), pI = plain_rows(s.inner), pF = plain_rows(s.ffi), pV = plain_rows(s.dv), pKV = plain_rows(2 * s.inner);
    const size_t attn_ws_bytes = align_up((size_t)s.b * s.H * s.L * 4);
    const size_t gws = W.ws_bytes - attn_ws_bytes;
    float* attn_ws = (float*)((char*)W.ws + gws);

    // ---- y2 = y1 + tanh(alpha_ffw) * ffw(y1) ----
    if (!defer) FF_TRY(Gemm(s.dt, s.d, s.ffi, M).a(1, pd).b(1, pF).c(pF).problem(dy2, S.Aact, G[10], nullptr, nullptr, nullptr, P[1]).run(W.ws, gws, st));
    FF_TRY(Gemm(s.dt, M, s.ffi, s.d).a(0, pd).b(1, pF).c(pF).act_bwd(s.act).problem(dy2, P[10], T.dH, nullptr, S.Hpre, nullptr, P[1]).run(W.ws, gws, st));
    if (!defer) FF_TRY(Gemm(s.dt, s.ffi, s.d, M).a(1, pF).b(1, pd).c(pd).problem(T.dH, S.xn_f, G[9]).run(W.ws, gws, st));
    // d LN(y1) = d H . W1: a split-K product at the benchmark's shape.  In the deferred mode its fp32 slabs are summed by the LayerNorm backward
    // that consumes them (which keeps its partials in the stash, not in W.ws) instead of by a reduce launch of their own (round 4)
    static const int fold_reduce = dbg_switch("FF_FOLD_SPLITK_LN", 1);
    int dxn_splits = 1;
    FF_TRY(Gemm(s.dt, M, s.d, s.ffi).a(0, pF).b(1, pd).c(pd).problem(T.dH, P[9], W.dxn).run(W.ws, gws, st, defer_ln && fold_reduce ? &dxn_splits : nullptr));
    {   // LN(y1) backward -> dy1, with both gate gradients folded in: d alpha_ffw = sum(dy2 . ffw_out), d alpha_attn = sum(dy1 . attn_out)
        LnDots dots;
        dots.a = S.ffw_out; dots.alpha_a = P[1]; dots.out_a = G[1];
        dots.b = S.attn_out; dots.alpha_b = P[0]; dots.out_b = G[0];
        if (defer_ln) {
            LnArgs la = ln_args(s.dt, M, s.d, pd, pd, pd);
            if (dxn_splits > 1) { la.dy_splits = dxn_splits; la.dy_slab = (long long)M * s.d; }
            FF_TRY(layernorm_bwd(la, dxn_splits > 1 ? W.ws : W.dxn, S.y1, nullptr, P[7], S.mean_f, S.rstd_f, T.dy1, dy2, G[7], G[8], T.lnp_f,
                                 layernorm_bwd_partial_bytes(M, s.d), st, &dots, &pend_f));
            FF_CHECK(pend_f.partial, FF_ERR_SHAPE, "xattn_bwd: the one-pass LayerNorm backward did not apply in the deferred mode");
        } else FF_TRY(layernorm_bwd(ln_args(s.dt, M, s.d, pd, pd, pd), W.dxn, S.y1, nullptr, P[7], S.mean_f, S.rstd_f, T.dy1, dy2, G[7], G[8], W.ws, gws, st, &dots));
    }
    // ---- y1 = y + tanh(alpha_attn) * to_out(attention) ----
    if (!defer) FF_TRY(Gemm(s.dt, s.d, s.inner, M).a(1, pd).b(1, pI).c(pI).problem(T.dy1, S.O, G[6], nullptr, nullptr, nullptr, P[0]).run(W.ws, gws, st));
    char* dK = hoisted ? (char*)dkv_out : (char*)W.dKV;
    char* dV = dK + (size_t)s.inner * s.es;
    const void* Kp = hoisted ? ext_k : S.KV;
    const void* Vp = hoisted ? ext_v : (const void*)((const char*)S.KV + (size_t)s.inner * s.es);
    bool dyn_fused = false, lnb_fused = false;
    float* lnb_partial = nullptr;
    if (xa_fused_enabled() && xa_fused_supported(s.dt, s.dh, s.d, s.inner)) {
        // d o = tanh(alpha_attn) * d y1 . Wo and the attention backward in one launch (two when the queries of a sample span several tiles)
        // ... and, with a sync buffer at the training shape, d LN(y) = scale * d Qs . Wq in the same launch
        int single = 0;
        const XaFusedArgs fa = xa_fused_args(*d, s, hoisted);
        XaOutArgs oa = {(const bf16*)P[4], nullptr, (bf16*)W.dyn, nullptr, (unsigned*)d->sync};
        dyn_fused = d->sync != nullptr && xa_out_fusable(fa, s.dt, s.dh);
        if (dyn_fused && xa_ln3_in_bwd(*d, s)) {      // ... and the backward of LN(y) behind it (phase 3): d y leaves this launch, d LN(y) is never stored
            lnb_fused = true;
            lnb_partial = defer_ln ? T.lnp_a : (float*)W.ws;
            oa.out = nullptr;
            oa.ln_g = (const bf16*)P[2]; oa.ln_x = (const bf16*)y; oa.ln_res = (const bf16*)T.dy1; oa.ln_out = (bf16*)dy;
            oa.ln_mean = S.mean_a; oa.ln_rstd = S.rstd_a; oa.ln_part = W.lnx; oa.ln_wpart = lnb_partial;
        }
        FF_TRY(xa_dattn_bwd(fa, s.dt, s.dh, T.dy1, P[6], P[0], S.Qs, Kp, Vp, tt, S.O, S.lse, W.dO, T.dQs, dK, dV, attn_ws, &single, st,
                            dyn_fused ? &oa : nullptr));
        if (!single) FF_TRY(attention_bwd_dkv(xa_attn_desc(*d, s, hoisted), S.Qs, Kp, Vp, tt, W.dO, S.lse, attn_ws, dK, dV, st));
        if (lnb_fused && !defer_ln) {     // the per-sample column sums sit in W.ws, which the products below may use for their split-K slabs: reduce them now
            LnPending q;
            q.partial = lnb_partial; q.nblk = s.b; q.cols = s.d; q.dgamma = G[2]; q.dbeta = G[3];
            FF_TRY(layernorm_bwd_finish(s.dt, &q, 1, st));
        }
    } else {
        FF_TRY(Gemm(s.dt, M, s.inner, s.d).a(0, pd).b(1, pI).c(pI).problem(T.dy1, P[6], W.dO, nullptr, nullptr, nullptr, P[0]).run(W.ws, gws, st));
        FF_TRY(attention_bwd(xa_attn_desc(*d, s, hoisted), S.Qs, Kp, Vp, tt, S.O, W.dO, S.lse, T.dQs, dK, dV, attn_ws, (size_t)s.b * s.H * s.L * 4, st));
    }
    if (!defer) FF_TRY(Gemm(s.dt, s.inner, s.d, M).a(1, pI).b(1, pd).c(pd).scale(s.scale).problem(T.dQs, S.yn, G[4]).run(W.ws, gws, st));
    if (!hoisted) {
        FF_TRY(Gemm(s.dt, 2 * s.inner, s.dv, Mk).a(1, pKV).b(1, pV).c(pV).problem(W.dKV, vf, G[5]).run(W.ws, gws, st));
        if (dvf) FF_TRY(Gemm(s.dt, Mk, s.dv, 2 * s.inner).a(0, pKV).b(1, pV).c(pV).problem(W.dKV, P[5], dvf).run(W.ws, gws, st));
    }
    if (!dyn_fused) FF_TRY(Gemm(s.dt, M, s.d, s.inner).a(0, pI).b(1, pd).c(pd).scale(s.scale).problem(T.dQs, P[4], W.dyn).run(W.ws, gws, st));
    if (lnb_fused) return FF_OK;     // d y left the fused launch; the column sums (d gamma, d beta) were reduced above, or wait in T.lnp_a - s.b blocks, see
                                     // xa_ln3_in_bwd - for xattn_wgrad_grouped
    if (defer_ln) {
        FF_TRY(layernorm_bwd(ln_args(s.dt, M, s.d, pd, pd, pd), W.dyn, y, nullptr, P[2], S.mean_a, S.rstd_a, dy, T.dy1, G[2], G[3], T.lnp_a,
                             layernorm_bwd_partial_bytes(M, s.d), st, nullptr, &pend_a));
        FF_CHECK(pend_a.partial, FF_ERR_SHAPE, "xattn_bwd: the one-pass LayerNorm backward did not apply in the deferred mode");
        return FF_OK;
    }
    return layernorm_bwd(ln_args(s.dt, M, s.d, pd, pd, pd), W.dyn, y, nullptr, P[2], S.mean_a, S.rstd_a, dy, T.dy1, G[2], G[3], W.ws, gws, st);
}

// Weight gradients of n <= kGemmMaxZ same-shaped blocks whose data-gradient pass ran with a stash: four grouped launches
// (d ffw.3.weight, d ffw.1.weight, d attn.to_out.weight, d attn.to_q.weight), each covering all n blocks.
static size_t xa_wgrad_ws_bytes(const XaDims& s) {
    const int M = s.b * s.L;
    size_t w = 0;
    for (int nz = 1; nz <= kGemmMaxZ; nz++)
        w = std::max({w, gemm_workspace_bytes(s.dt, s.d, s.ffi, M, nz, 0), gemm_workspace_bytes(s.dt, s.ffi, s.d, M, nz, 0),
                      gemm_workspace_bytes(s.dt, s.d, s.inner, M, nz, 0), gemm_workspace_bytes(s.dt, s.inner, s.d, M, nz, 0)});
    return align_up(w);
}
static int xattn_wgrad_grouped(const ff_xattn_desc* d, int n, const void* const* dy2, const void* const* saved, size_t saved_bytes,
                               const void* const* stash, size_t stash_bytes, const void* const* params, void* const* grads, void* ws,
                               size_t ws_bytes, hipStream_t st) {
    FF_TRY(xa_check(d));
    FF_CHECK(n >= 1 && n <= kGemmMaxZ && dy2 && saved && stash && params && grads, FF_ERR_SHAPE, "xattn_wgrad_grouped: 1..%d blocks per call", kGemmMaxZ);
    const XaDims s(*d);
    FF_CHECK(ws_bytes >= xa_wgrad_ws_bytes(s) && (ws || !xa_wgrad_ws_bytes(s)), FF_ERR_WORKSPACE, "xattn_wgrad_grouped: workspace too small");
    const int M = s.b * s.L;
    const RowMap pd = plain_rows(s.d), pI = plain_rows(s.inner), pF = plain_rows(s.ffi);
    LnPending ln_sets[2 * kGemmMaxZ];
    const bool finish_ln = xa_ln_deferrable(s);
    const int ln_blocks = layernorm_bwd_partial_blocks(M);
    Gemm g3(s.dt, s.d, s.ffi, M), g1(s.dt, s.ffi, s.d, M), go(s.dt, s.d, s.inner, M), gq(s.dt, s.inner, s.d, M);
    g3.a(1, pd).b(1, pF).c(pF);                     // d ffw.3 = tanh(alpha_ffw) * d y2^T . act(H)
    g1.a(1, pF).b(1, pd).c(pd);                     // d ffw.1 = d H^T . LN(y1)
    go.a(1, pd).b(1, pI).c(pI);                     // d to_out = tanh(alpha_attn) * d y1^T . O
    gq.a(1, pI).b(1, pd).c(pd).scale(s.scale);      // d to_q = scale * d Qs^T . LN(y)
    for (int i = 0; i < n; i++) {
        XaSaved S;
        XaStash T;
        FF_CHECK(dy2[i] && saved[i] && stash[i], FF_ERR_SHAPE, "xattn_wgrad_grouped: null buffer of block %d", i);
        FF_CHECK(xa_saved_layout(s, (void*)saved[i], saved_bytes, S) <= saved_bytes, FF_ERR_WORKSPACE, "xattn_wgrad_grouped: saved buffer too small");
        FF_CHECK(xa_stash_layout(s, (void*)stash[i], stash_bytes, T) <= stash_bytes, FF_ERR_WORKSPACE, "xattn_wgrad_grouped: stash too small");
        const void* const* P = params + (size_t)i * FF_XATTN_PARAMS;
        void* const* G = grads + (size_t)i * FF_XATTN_PARAMS;
        FF_CHECK(G[10] && G[9] && G[6] && G[4] && P[0] && P[1], FF_ERR_SHAPE, "xattn_wgrad_grouped: null parameter / gradient of block %d", i);
        g3.problem(dy2[i], S.Aact, G[10], nullptr, nullptr, nullptr, P[1]);
        g1.problem(T.dH, S.xn_f, G[9]);
        go.problem(T.dy1, S.O, G[6], nullptr, nullptr, nullptr, P[0]);
        gq.problem(T.dQs, S.yn, G[4]);
        if (finish_ln) {    // what xattn_bwd left open: LN(y1) backward with both gate gradients, LN(y) backward
            FF_CHECK(G[0] && G[1] && G[2] && G[3] && G[7] && G[8], FF_ERR_SHAPE, "xattn_wgrad_grouped: null LayerNorm / gate gradient of block %d", i);
            LnPending& f = ln_sets[2 * i];
            f.partial = T.lnp_f; f.nblk = ln_blocks; f.cols = s.d; f.dgamma = G[7]; f.dbeta = G[8];
            f.alpha_a = P[1]; f.out_a = G[1]; f.alpha_b = P[0]; f.out_b = G[0];
            LnPending& q = ln_sets[2 * i + 1];
            q.partial = T.lnp_a; q.nblk = xa_ln3_in_bwd(*d, s) ? s.b : ln_blocks; q.cols = s.d; q.dgamma = G[2]; q.dbeta = G[3];
        }
    }
    if (finish_ln) FF_TRY(layernorm_bwd_finish(s.dt, ln_sets, 2 * n, st));
    FF_TRY(g3.run(ws, ws_bytes, st));
    FF_TRY(g1.run(ws, ws_bytes, st));
    FF_TRY(go.run(ws, ws_bytes, st));
    return gq.run(ws, ws_bytes, st);
}

// =====================================================================================================
// Key / value projection of ALL cross-attention layers at once (gated_cross_attention.py:84-86 runs `to_kv` on the same
// visual features in every layer, with different weights): grouped launches of up to kGemmMaxZ same-shape problems fill the
// chip where the per-layer 2048 x 1024 x 1024 products do not, and d(visual features) is summed once instead of by 36 autograd adds.
// =====================================================================================================
static int kvp_check(const ff_kvproj_desc* d) {
    FF_CHECK(d, FF_ERR_SHAPE, "kv_project: null descriptor");
    FF_CHECK(d->dtype == FF_DTYPE_F32 || d->dtype == FF_DTYPE_BF16, FF_ERR_UNSUPPORTED, "kv_project: dtype %d", d->dtype);
    FF_CHECK(d->n_layers > 0 && d->rows > 0 && d->dim_visual > 0 && d->kv_dim > 0 && (long long)d->rows * d->dim_visual < (1LL << 31), FF_ERR_SHAPE,
             "kv_project: bad dimensions");
    return FF_OK;
}
static size_t kvp_gemm_ws(const ff_kvproj_desc* d) {
    size_t w = 0;
    for (int z = 1; z <= kKvProjGroup; z++)      // the last group may hold fewer layers (and then pick a split-K plan)
        w = std::max({w, gemm_workspace_bytes(d->dtype, d->rows, d->kv_dim, d->dim_visual, z, 0),
                      gemm_workspace_bytes(d->dtype, d->kv_dim, d->dim_visual, d->rows, z, 0),
                      gemm_workspace_bytes(d->dtype, d->rows, d->dim_visual, d->kv_dim, z, 0)});
    return align_up(w);
}
static size_t kvp_ws_bytes(const ff_kvproj_desc* d, bool with_dvf) {
    const size_t es = d->dtype == FF_DTYPE_BF16 ? 2 : 4;
    size_t w = kvp_gemm_ws(d);
    if (with_dvf)
        w += align_up((size_t)d->n_layers * d->rows * d->dim_visual * es) +
             align_up(rows_reduce_workspace(d->n_layers, d->rows * d->dim_visual, d->n_layers, d->n_layers));
    return w;
}
static int kv_project_fwd(const ff_kvproj_desc* d, const void* vf, const void* const* w_kv, void* const* kv_out, void* ws, size_t ws_bytes,
                          hipStream_t st) {
    FF_TRY(kvp_check(d));
    FF_CHECK(vf && w_kv && kv_out && ws && ws_bytes >= kvp_gemm_ws(d), FF_ERR_WORKSPACE, "kv_project_fwd: null argument or workspace too small");
    const RowMap pV = plain_rows(d->dim_visual), pKV = plain_rows(d->kv_dim);
    for (int l0 = 0; l0 < d->n_layers; l0 += kKvProjGroup) {
        Gemm g(d->dtype, d->rows, d->kv_dim, d->dim_visual);
        g.a(0, pV).b(0, pV).c(pKV);
        for (int l = l0; l < std::min(d->n_layers, l0 + kKvProjGroup); l++) g.problem(vf, w_kv[l], kv_out[l]);
        FF_TRY(g.run(ws, ws_bytes, st));
    }
    return FF_OK;
}
static int kv_project_bwd(const ff_kvproj_desc* d, const void* vf, const void* const* w_kv, const void* const* dkv, void* const* dw_kv,
                          void* dvf, void* ws, size_t ws_bytes, hipStream_t st) {
    FF_TRY(kvp_check(d));
    FF_CHECK(vf && w_kv && dkv && dw_kv && ws && ws_bytes >= kvp_ws_bytes(d, dvf != nullptr), FF_ERR_WORKSPACE,
             "kv_project_bwd: null argument or workspace too small");
    const size_t es = d->dtype == FF_DTYPE_BF16 ? 2 : 4, gws = kvp_gemm_ws(d);
    const RowMap pV = plain_rows(d->dim_visual), pKV = plain_rows(d->kv_dim);
    char* terms = (char*)ws + gws;                                             // [n_layers][rows][dim_visual]: one product per layer
    const size_t term_bytes = (size_t)d->rows * d->dim_visual * es;
    for (int l0 = 0; l0 < d->n_layers; l0 += kKvProjGroup) {
        const int l1 = std::min(d->n_layers, l0 + kKvProjGroup);
        Gemm gw(d->dtype, d->kv_dim, d->dim_visual, d->rows);                  // d W_l = d KV_l^T . vf
        gw.a(1, pKV).b(1, pV).c(pV);
        for (int l = l0; l < l1; l++) gw.problem(dkv[l], vf, dw_kv[l]);
        FF_TRY(gw.run(ws, gws, st));
        if (dvf) {
            Gemm gx(d->dtype, d->rows, d->dim_visual, d->kv_dim);              // d vf contribution of layer l = d KV_l . W_l
            gx.a(0, pKV).b(1, pV).c(pV);
            for (int l = l0; l < l1; l++) gx.problem(dkv[l], w_kv[l], terms + (size_t)l * term_bytes);
            FF_TRY(gx.run(ws, gws, st));
        }
    }
    if (dvf) {   // sum the n_layers products in fp32
        const int cols = d->rows * d->dim_visual;
        FF_TRY(rows_reduce(d->dtype, d->n_layers, cols, plain_rows(cols), d->n_layers, d->n_layers, terms, dvf,
                           terms + align_up((size_t)d->n_layers * term_bytes), rows_reduce_workspace(d->n_layers, cols, d->n_layers, d->n_layers), st));
    }
    return FF_OK;
}

}  // namespace ff

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" int ff_version(void) { return 4; }
extern "C" const char* ff_arch(void) { return "gfx950"; }
extern "C" const char* ff_last_error(void) { return ff::g_err; }

extern "C" size_t ff_resampler_saved_bytes(const ff_resampler_desc* d) {
    if (ff::rs_check(d) != FF_OK) return 0;
    ff::RsSaved S;
    return ff::rs_saved_layout(ff::RsDims(*d), nullptr, 0, S);
}
extern "C" size_t ff_resampler_scratch_bytes(const ff_resampler_desc* d) {
    if (ff::rs_check(d) != FF_OK) return 0;
    ff::RsScratch W;
    return ff::rs_scratch_layout(ff::RsDims(*d), nullptr, 0, true, W);
}
extern "C" int ff_resampler_fwd(const ff_resampler_desc* d, const void* x_f, const void* const* params, void* out, void* saved,
                                size_t saved_bytes, void* scratch, size_t scratch_bytes, ff_stream_t stream) {
    return ff::resampler_fwd(d, x_f, params, out, saved, saved_bytes, scratch, scratch_bytes, (hipStream_t)stream);
}
extern "C" int ff_resampler_bwd(const ff_resampler_desc* d, const void* x_f, const void* const* params, const void* dout,
                                const void* saved, size_t saved_bytes, void* const* grads, void* dx_f, void* scratch,
                                size_t scratch_bytes, ff_stream_t stream) {
    return ff::resampler_bwd(d, x_f, params, dout, saved, saved_bytes, grads, dx_f, scratch, scratch_bytes, (hipStream_t)stream);
}

extern "C" size_t ff_resampler_prologue_saved_bytes(const ff_resampler_desc* d) {
    if (ff::rs_check(d) != FF_OK) return 0;
    ff::RsProSaved S;
    return ff::rs_pro_layout(ff::RsDims(*d), nullptr, 0, S);
}
extern "C" size_t ff_resampler_layer_saved_bytes(const ff_resampler_desc* d) {
    if (ff::rs_check(d) != FF_OK) return 0;
    ff::RsLayerSaved L;
    return ff::rs_layer_saved_layout(ff::RsDims(*d), nullptr, 0, L);
}
extern "C" size_t ff_resampler_layer_scratch_bytes(const ff_resampler_desc* d) {
    if (ff::rs_check(d) != FF_OK) return 0;
    ff::RsLayerScratch W;
    return ff::rs_layer_scratch_layout(ff::RsDims(*d), nullptr, 0, W);
}
extern "C" size_t ff_resampler_epilogue_saved_bytes(const ff_resampler_desc* d) {
    if (ff::rs_check(d) != FF_OK) return 0;
    return 2 * ff::align_up((size_t)d->batch * d->num_latents * 4);
}
extern "C" int ff_resampler_prologue_fwd(const ff_resampler_desc* d, const void* x_f, const void* time_pos_emb, void* saved_pro, size_t saved_pro_bytes,
                                         hipStream_t stream) {
    return ff::rs_prologue_fwd(d, x_f, time_pos_emb, saved_pro, saved_pro_bytes, stream);
}
extern "C" int ff_resampler_layer_fwd(const ff_resampler_desc* d, const void* x_f, const void* time_pos_emb, const void* saved_pro, size_t saved_pro_bytes,
                                      const void* x_in, int x_in_is_latents, const void* const* layer_params, void* x_out, void* saved, size_t saved_bytes,
                                      void* scratch, size_t scratch_bytes, hipStream_t stream) {
    return ff::rs_layer_fwd(d, x_f, time_pos_emb, saved_pro, saved_pro_bytes, x_in, x_in_is_latents, layer_params, x_out, saved, saved_bytes, scratch,
                            scratch_bytes, stream);
}
extern "C" int ff_resampler_layer_bwd(const ff_resampler_desc* d, const void* x_f, const void* time_pos_emb, const void* saved_pro, size_t saved_pro_bytes,
                                      const void* x_in, int x_in_is_latents, const void* const* layer_params, const void* dx_out, const void* saved,
                                      size_t saved_bytes, void* const* layer_grads, void* dx_in, void* dx_f, int dx_f_accumulate, void* scratch,
                                      size_t scratch_bytes, hipStream_t stream) {
    return ff::rs_layer_bwd(d, x_f, time_pos_emb, saved_pro, saved_pro_bytes, x_in, x_in_is_latents, layer_params, dx_out, saved, saved_bytes, layer_grads,
                            dx_in, dx_f, dx_f_accumulate, scratch, scratch_bytes, stream);
}
extern "C" int ff_resampler_prologue_bwd(const ff_resampler_desc* d, const void* dx0, const void* dx_f, void* d_latents, void* d_time_pos_emb, void* scratch,
                                         size_t scratch_bytes, hipStream_t stream) {
    return ff::rs_prologue_bwd(d, dx0, dx_f, d_latents, d_time_pos_emb, scratch, scratch_bytes, stream);
}
extern "C" int ff_resampler_epilogue_fwd(const ff_resampler_desc* d, const void* x_last, const void* norm_weight, const void* norm_bias, void* out,
                                         void* saved_epi, size_t saved_epi_bytes, hipStream_t stream) {
    return ff::rs_epilogue_fwd(d, x_last, norm_weight, norm_bias, out, saved_epi, saved_epi_bytes, stream);
}
extern "C" int ff_resampler_epilogue_bwd(const ff_resampler_desc* d, const void* dout, const void* x_last, const void* norm_weight, const void* saved_epi,
                                         size_t saved_epi_bytes, void* dx_last, void* d_norm_weight, void* d_norm_bias, void* scratch, size_t scratch_bytes,
                                         hipStream_t stream) {
    return ff::rs_epilogue_bwd(d, dout, x_last, norm_weight, saved_epi, saved_epi_bytes, dx_last, d_norm_weight, d_norm_bias, scratch, scratch_bytes, stream);
}

extern "C" size_t ff_xattn_sync_bytes(void) { return (size_t)(4 * FF_XATTN_SYNC_SLOTS + 64) * sizeof(unsigned); }
extern "C" int ff_xattn_sync_status(const void* sync, hipStream_t stream) {
    FF_CHECK(sync, FF_ERR_SHAPE, "ff_xattn_sync_status: null buffer");
    unsigned flag = 0;
    hipError_t e = hipMemcpyAsync(&flag, (const unsigned*)sync + 2 * FF_XATTN_SYNC_SLOTS, sizeof(flag), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "ff_xattn_sync_status: %s", hipGetErrorString(e));
    return flag ? 1 : 0;
}
extern "C" size_t ff_xattn_saved_bytes(const ff_xattn_desc* d) {
    if (ff::xa_check(d) != FF_OK) return 0;
    ff::XaSaved S;
    return ff::xa_saved_layout(ff::XaDims(*d), nullptr, 0, S);
}
extern "C" size_t ff_xattn_scratch_bytes(const ff_xattn_desc* d) {
    if (ff::xa_check(d) != FF_OK) return 0;
    ff::XaScratch W;
    return ff::xa_scratch_layout(ff::XaDims(*d), nullptr, 0, true, W);
}
extern "C" size_t ff_xattn_kv_offset(const ff_xattn_desc* d) {
    (void)d;
    return 0;
}
extern "C" int ff_xattn_block_fwd(const ff_xattn_desc* d, const void* y, const void* visual_features, const int* text_time,
                                  const void* const* params, const void* cached_k, const void* cached_v, void* y_out, void* saved,
                                  size_t saved_bytes, void* scratch, size_t scratch_bytes, ff_stream_t stream) {
    return ff::xattn_fwd(d, y, visual_features, text_time, params, cached_k, cached_v, y_out, saved, saved_bytes, scratch, scratch_bytes,
                         (hipStream_t)stream);
}
extern "C" int ff_xattn_block_bwd(const ff_xattn_desc* d, const void* y, const void* visual_features, const int* text_time,
                                  const void* const* params, const void* dy_out, const void* saved, size_t saved_bytes, void* const* grads,
                                  void* dy, void* dvisual_features, void* scratch, size_t scratch_bytes, ff_stream_t stream) {
    return ff::xattn_bwd(d, y, visual_features, text_time, params, dy_out, saved, saved_bytes, grads, dy, dvisual_features, scratch,
                         scratch_bytes, (hipStream_t)stream);
}

extern "C" int ff_xattn_block_bwd_kv(const ff_xattn_desc* d, const void* y, const void* k, const void* v, const int* text_time,
                                     const void* const* params, const void* dy_out, const void* saved, size_t saved_bytes,
                                     void* const* grads, void* dy, void* dkv, void* scratch, size_t scratch_bytes, ff_stream_t stream) {
    using namespace ff;
    FF_CHECK(k && v && dkv, FF_ERR_SHAPE, "ff_xattn_block_bwd_kv: null K / V / dKV");
    return xattn_bwd(d, y, nullptr, text_time, params, dy_out, saved, saved_bytes, grads, dy, nullptr, scratch, scratch_bytes,
                     (hipStream_t)stream, k, v, dkv);
}
extern "C" size_t ff_xattn_wgrad_stash_bytes(const ff_xattn_desc* d) {
    if (ff::xa_check(d) != FF_OK) return 0;
    ff::XaStash T;
    return ff::xa_stash_layout(ff::XaDims(*d), nullptr, 0, T);
}
extern "C" size_t ff_xattn_wgrad_workspace_bytes(const ff_xattn_desc* d) {
    if (ff::xa_check(d) != FF_OK) return 0;
    return ff::xa_wgrad_ws_bytes(ff::XaDims(*d));
}
extern "C" int ff_xattn_block_bwd_kv_data(const ff_xattn_desc* d, const void* y, const void* k, const void* v, const int* text_time,
                                          const void* const* params, const void* dy_out, const void* saved, size_t saved_bytes,
                                          void* const* grads, void* dy, void* dkv, void* stash, size_t stash_bytes, void* scratch,
                                          size_t scratch_bytes, ff_stream_t stream) {
    using namespace ff;
    FF_CHECK(k && v && dkv && stash, FF_ERR_SHAPE, "ff_xattn_block_bwd_kv_data: null K / V / dKV / stash");
    return xattn_bwd(d, y, nullptr, text_time, params, dy_out, saved, saved_bytes, grads, dy, nullptr, scratch, scratch_bytes,
                     (hipStream_t)stream, k, v, dkv, stash, stash_bytes);
}
extern "C" int ff_xattn_wgrad_grouped(const ff_xattn_desc* d, int n_blocks, const void* const* dy_out, const void* const* saved,
                                      size_t saved_bytes, const void* const* stash, size_t stash_bytes, const void* const* params,
                                      void* const* grads, void* workspace, size_t workspace_bytes, ff_stream_t stream) {
    return ff::xattn_wgrad_grouped(d, n_blocks, dy_out, saved, saved_bytes, stash, stash_bytes, params, grads, workspace, workspace_bytes,
                                   (hipStream_t)stream);
}
extern "C" size_t ff_kv_project_workspace_bytes(const ff_kvproj_desc* d, int with_dvf) {
    if (ff::kvp_check(d) != FF_OK) return 0;
    return ff::kvp_ws_bytes(d, with_dvf != 0);
}
extern "C" int ff_kv_project_fwd(const ff_kvproj_desc* d, const void* visual_features, const void* const* w_kv, void* const* kv_out,
                                 void* workspace, size_t workspace_bytes, ff_stream_t stream) {
    return ff::kv_project_fwd(d, visual_features, w_kv, kv_out, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int ff_kv_project_bwd(const ff_kvproj_desc* d, const void* visual_features, const void* const* w_kv, const void* const* dkv,
                                 void* const* dw_kv, void* dvisual_features, void* workspace, size_t workspace_bytes, ff_stream_t stream) {
    return ff::kv_project_bwd(d, visual_features, w_kv, dkv, dw_kv, dvisual_features, workspace, workspace_bytes, (hipStream_t)stream);
}
