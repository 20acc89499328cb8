// Internal (C++) interfaces between the translation units of libflamingo_fusion.
#pragma once
#include <algorithm>
#include "ff_common.h"

namespace ff {

// ---- GEMM ----------------------------------------------------------------------------------
constexpr int kGemmMaxZ = 12;     // same-shape problems per grouped launch (capacity; the callers choose how many they batch)
constexpr int kKvProjGroup = 4;   // layers per K / V projection launch: 4 x 128 tiles of 128 x 128 = one full round of 2 workgroups per CU
struct GemmProblem {
    const void* A;
    const void* B;
    void* C;
    void* aux_out;
    const void* aux_in;
    const void* residual;
    const void* gate;
};
struct GemmParams {
    int M, N, K;
    int a_layout, b_layout;
    RowMap a_map, b_map, c_map, r_map;  // r_map addresses `residual` (defaults to c_map)
    float scale;
    int act, act_bwd;
    int split_k, k_per_split;
    int tile;   // block tile chosen by the host (bf16: 128 = 128x128, 64 = 64x64, 6412 = 64x128, 128002 / 128160 = producer / consumer kernels)
    int force_tile, force_stages;   // explicit plan override from the descriptor (0 = automatic)
    int xcd_ms, xcd_ns;   // XCD partition of the tile grid (ms * ns sub-grids, one per XCD)
    float* partial;
    int a_vec_ok, b_vec_ok;
    int c_vec8;   // bf16 epilogue may use 16-byte accesses on C / aux / residual
    int nz;
    GemmProblem p[kGemmMaxZ];
};
int gemm_pick_split(int dtype, int M, int N, int K, int nz);
size_t gemm_workspace_bytes(int dtype, int M, int N, int K, int nz, int split_k);
// leave_partial != null: a split-K problem with a plain epilogue (no scale / aux / gate / activation / residual, one problem) skips its
// reduce launch and leaves the fp32 slabs [split][M][N] at the start of `workspace` for the consumer to sum (layernorm_bwd: LnArgs.dy_splits);
// *leave_partial = the number of slabs, or 1 when C was written as usual.
int gemm_launch(GemmParams P, int dtype, void* workspace, size_t ws_bytes, hipStream_t st, int* leave_partial = nullptr);

// Convenience builder used by the module-level code.
struct Gemm {
    GemmParams P;
    int dtype;
    Gemm(int dtype_, int M, int N, int K) : dtype(dtype_) {
        P = GemmParams{};
        P.M = M; P.N = N; P.K = K;
        P.scale = 1.f; P.act = FF_ACT_NONE; P.act_bwd = FF_ACT_NONE; P.split_k = 0; P.nz = 0;
        P.a_map = plain_rows(K); P.b_map = plain_rows(K); P.c_map = plain_rows(N); P.r_map = P.c_map;
    }
    // A stored [M][K] (layout 0) or [K][M] (layout 1); same for B with N.
    Gemm& a(int layout, RowMap m) { P.a_layout = layout; P.a_map = m; return *this; }
    Gemm& b(int layout, RowMap m) { P.b_layout = layout; P.b_map = m; return *this; }
    Gemm& c(RowMap m) { P.c_map = m; P.r_map = m; return *this; }
    Gemm& res_map(RowMap m) { P.r_map = m; return *this; }
    Gemm& scale(float s) { P.scale = s; return *this; }
    Gemm& act(int a) { P.act = a; return *this; }
    Gemm& act_bwd(int a) { P.act_bwd = a; return *this; }
    Gemm& problem(const void* A, const void* B, void* C, void* aux_out = nullptr, const void* aux_in = nullptr,
                  const void* residual = nullptr, const void* gate = nullptr) {
        P.p[P.nz++] = GemmProblem{A, B, C, aux_out, aux_in, residual, gate};
        return *this;
    }
    size_t workspace() const { return gemm_workspace_bytes(dtype, P.M, P.N, P.K, std::max(P.nz, 1), P.split_k); }
    int run(void* ws, size_t ws_bytes, hipStream_t st, int* leave_partial = nullptr) const { return gemm_launch(P, dtype, ws, ws_bytes, st, leave_partial); }
};

// launch timing hooks (ff_gemm_profile_*): tile < 0 marks the attention kernels (-1 fwd, -2 dQ, -3 dK/dV; M = n_q, N = n_kv,
// K = dim_head, nz = batch * heads, split_k = mode) and the fused projection + attention kernels of the cross-attention block
// (-4 forward, -5 backward; M = n_q, N = n_kv, K = model dim, nz = batch * heads, a_layout = heads, split_k = dim_head; -6 / -7: the same launches with
// phase 2 - to_out + gate + residual resp. d LN(y) - inside; -8 / -9: with phase 3 - the LayerNorm behind that output - inside as well)
int profile_begin(int dtype, int tile, int a_layout, int b_layout, int M, int N, int K, int nz, int split_k, hipStream_t st);
void profile_end(int i, hipStream_t st);

// ---- row-wise kernels ------------------------------------------------------------------------
struct LnArgs {
    int dtype;
    int rows, cols;
    RowMap x_map, y_map, dx_map;
    int add_rows_per_seg, add_div;
    float eps;
    int stats_given;
    // backward only: dy_splits > 0 = `dy` points at that many fp32 slabs [rows][cols] (slab stride dy_slab elements) whose SUM is dy - the
    // partial tiles of a split-K GEMM, combined on the way into the kernel instead of by a reduce launch of their own
    int dy_splits = 0;
    long long dy_slab = 0;
};
int layernorm_fwd(const LnArgs& a, const void* x, const void* add, const void* gamma, const void* beta, void* y,
                  float* mean, float* rstd, hipStream_t st);
size_t layernorm_bwd_workspace(int rows, int cols);
// Optional tanh-gate gradients fused into a LayerNorm backward (same shape / row map as dx):
//   out_a = (1 - tanh(alpha_a)^2) * sum(dx_residual .* a),   out_b = (1 - tanh(alpha_b)^2) * sum(dx .* b)
struct LnDots {
    const void* a = nullptr; const void* alpha_a = nullptr; void* out_a = nullptr;
    const void* b = nullptr; const void* alpha_b = nullptr; void* out_b = nullptr;
};
// A LayerNorm backward whose cross-workgroup reduction (d gamma, d beta, the gate gradients) is postponed: the ~5 us `final` launch
// is not on the data-gradient chain, so callers collect these and finish up to kLnFinishMax of them with ONE launch
// (layernorm_bwd_finish).  `partial` is the caller's buffer (layernorm_bwd_partial_bytes) and must stay alive until then.
struct LnPending {
    const float* partial = nullptr;     // null: nothing was postponed (the one-pass kernel did not apply; everything ran immediately)
    int nblk = 0, cols = 0;
    void *dgamma = nullptr, *dbeta = nullptr;
    const void* alpha_a = nullptr; void* out_a = nullptr;
    const void* alpha_b = nullptr; void* out_b = nullptr;
};
constexpr int kLnFinishMax = 8;
bool layernorm_bwd_deferrable(int dtype, int cols);          // shape part of the condition (pointers must also be 16-byte aligned)
size_t layernorm_bwd_partial_bytes(int rows, int cols);
int layernorm_bwd_partial_blocks(int rows);
// pending != null: `ws` receives the partials (>= layernorm_bwd_partial_bytes) and the final reduction is left to layernorm_bwd_finish
int layernorm_bwd(const LnArgs& a, const void* dy, const void* x, const void* add, const void* gamma, const float* mean,
                  const float* rstd, void* dx, const void* dx_residual, void* dgamma, void* dbeta, void* ws,
                  size_t ws_bytes, hipStream_t st, const LnDots* dots = nullptr, LnPending* pending = nullptr);
int layernorm_bwd_finish(int dtype, const LnPending* sets, int n, hipStream_t st);
size_t rows_reduce_workspace(int rows, int cols, int rows_per_batch, int rows_per_group);
int rows_reduce(int dtype, int rows, int cols, RowMap x_map, int rows_per_batch, int rows_per_group, const void* x,
                void* out, void* ws, size_t ws_bytes, hipStream_t st);
size_t gate_grad_workspace(int rows, int cols);
int gate_grad(int dtype, int rows, int cols, const void* a, const void* b, const void* alpha, void* dalpha, void* ws,
              size_t ws_bytes, hipStream_t st);
int text_time(int batch, int n_tokens, const void* ml, int elem_bytes, int* out, hipStream_t st);

// ---- attention -------------------------------------------------------------------------------
size_t attention_bwd_workspace(const ff_attn_desc& d);
int attention_fwd(const ff_attn_desc& d, const void* Q, const void* K, const void* V, const int* tt, void* O, float* lse,
                  hipStream_t st);
int attention_bwd(const ff_attn_desc& d, const void* Q, const void* K, const void* V, const int* tt, const void* O,
                  const void* dO, const float* lse, void* dQ, void* dK, void* dV, void* ws, size_t ws_bytes,
                  hipStream_t st);

// d K / d V only (own rows = keys), for callers that already hold d O and Dsum[b][h][q] = sum_d dO * O
int attention_bwd_dkv(const ff_attn_desc& d, const void* Q, const void* K, const void* V, const int* tt, const void* dO, const float* lse,
                      const float* Dsum, void* dK, void* dV, hipStream_t st);

// ---- fused projection + attention kernels of the gated cross-attention block (ff_xattn_fused.hip) ----
// Q / O / dQ / dO live as [batch * n_q][inner] rows (head h = columns h * dim_head ..); K / V / dK / dV through strides.
struct XaFusedArgs {
    int batch, heads, n_q, n_kv, n_visual, tt_stride, tt_offset;
    int dim, inner;          // model width (the projection's contraction length), heads * dim_head
    float scale, eps;
    int xcd_split;           // resident kernels: an XCD takes 4 heads x batch / 4 samples instead of 8 heads x batch / 8 samples (see res_work_item)
    ff_strides k, v, dk, dv;
};
// Phase 2 of the resident kernels (round 5): the product over all heads of a sample inside the same launch.  Forward: W = to_out.weight
// [dim][inner], out = y1 = y + tanh(*gate) * O . W^T, aux = O . W^T.  Backward: W = to_q.weight [inner][dim], out = scale * dQs . W (d LN(y)).
// sync: the caller-owned counters of ff_xattn_desc.sync.
// Phase 3 (optional, ln_out != null): the LayerNorm that consumes phase 2's output, in the same launch - a workgroup holds 1 / heads of every row, so the
// rows' statistics (forward: mean / M2 of the slice, combined like Chan et al.; backward: the two sums of the LayerNorm backward) make one more trip
// through a second bank of arrival counters, two floats per row and workgroup (`ln_part`).
//   forward : ln_out = LN(y1) with ln_g / ln_b (the feed-forward's LayerNorm, utils.py:46), ln_mean / ln_rstd WRITTEN (saved for its backward)
//   backward: ln_out = d y = LayerNorm-backward(d LN(y); ln_x = y, ln_g, ln_mean / ln_rstd READ) + ln_res (d y1), and the sample's column sums
//             (d gamma | d beta) of the workgroup's slice go to ln_wpart[sample][2 dim + 2] for layernorm_bwd_finish; `out` (d LN(y)) may be null
struct XaOutArgs {
    const bf16* W;
    const bf16* gate;
    bf16* out;
    bf16* aux;
    unsigned* sync;
    const bf16* ln_g;
    const bf16* ln_b;
    bf16* ln_out;
    float* ln_mean;
    float* ln_rstd;
    const bf16* ln_x;
    const bf16* ln_res;
    float* ln_part;      // [batch][heads][32][2] fp32, scratch
    float* ln_wpart;     // backward: [batch][2 dim + 2] fp32
};
size_t xa_ln3_part_bytes(int batch, int heads);
// whether xa_qattn_fwd / xa_dattn_bwd will take the resident kernels WITH phase 2 for this problem when handed a sync buffer
bool xa_out_fusable(const XaFusedArgs& a, int dtype, int dim_head);
bool xa_fused_supported(int dtype, int dim_head, int dim, int inner);
// LayerNorm(y) -> q = to_q * scale -> masked attention; writes Qs, O, lse, the LayerNorm statistics and (optionally) the normalised rows yn
int xa_qattn_fwd(const XaFusedArgs& a, int dtype, int dim_head, const void* y, const void* gamma, const void* beta, const void* Wq, const void* K,
                 const void* V, const int* tt, void* yn, void* Qs, void* O, float* mean, float* rstd, float* lse, hipStream_t st,
                 const XaOutArgs* out = nullptr);
// d O = tanh(*gate) * d y1 . Wo -> d Q (and d K / d V when *single_tile comes back 1; otherwise d O and Dsum are left for attention_bwd_dkv)
int xa_dattn_bwd(const XaFusedArgs& a, int dtype, int dim_head, const void* dy1, const void* Wo, const void* gate, const void* Qs, const void* K,
                 const void* V, const int* tt, const void* O, const float* lse, void* dO, void* dQ, void* dK, void* dV, float* Dsum,
                 int* single_tile, hipStream_t st, const XaOutArgs* out = nullptr);

// ---- decode-shaped feed-forward (ff_decode.hip): <= 32 rows, activations resident in LDS, weights streamed HBM -> VGPR -> MFMA ----
bool decode_ffw_supported(int dtype, int M, int d, int ffi);
size_t decode_ffw_workspace_bytes(int d, int ffi);
int decode_ffw(int M, int d, int ffi, int act, float eps, const void* y1, const void* gamma, const void* beta, const void* W1, const void* W3,
               const void* alpha, void* xn, float* mean, float* rstd, void* Hpre, void* Aact, void* ffw_out, void* y_out, void* ws, size_t ws_bytes,
               hipStream_t st);

// ---- bump allocator over a caller-provided buffer ----------------------------------------------
struct Arena {
    unsigned char* base;
    size_t cap, used;
    Arena(void* p, size_t c) : base((unsigned char*)p), cap(c), used(0) {}
    template <typename T = void> T* take(size_t bytes) {
        size_t at = align_up(used);
        used = at + bytes;
        return (T*)(base ? base + at : nullptr);  // base == nullptr: sizing pass
    }
    bool ok() const { return used <= cap; }
};

}  // namespace ff
