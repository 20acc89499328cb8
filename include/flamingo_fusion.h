/*
 * libflamingo_fusion — C ABI of the MI355X (gfx950) Flamingo fusion path.
 *
 * This is the drop-in boundary below the Python modules: every entry point takes plain device pointers,
 * sizes and a HIP stream; no torch / C++ types; kernels are enqueued on the caller's stream and never
 * synchronise or allocate.  All tensors are caller-owned, 16-byte aligned, innermost dimension contiguous.
 * Return value: FF_OK or a negative error code; ff_last_error() gives a thread-local message.
 * State: none that is shared.  The entry points may be used from several threads on distinct streams / buffers; no environment variable is
 * read (the A/B switches of development builds exist only under -DFF_DEBUG, build.py --debug); every plan override travels in a
 * descriptor (ff_gemm_desc.split_k / tile / stages).  ff_last_error() is per thread, like errno.  The one process-wide object is the
 * opt-in launch-timing log of ff_gemm_profile_* (off by default): it has to see the launches of every thread - PyTorch issues backward
 * from its autograd worker thread - so recording into it is thread-safe (atomic slot counter), while enable / read / disable belong to
 * one controlling thread with no launch in flight.
 *
 * What each entry point replaces in the reference (dhansmair/flamingo-mini, paths relative to its root):
 *   ff_resampler_fwd/bwd     PerceiverResampler.forward + its autograd   flamingo_mini/perceiver_resampler.py:143-188
 *                            (PerceiverAttentionLayer.forward :32-96, FeedForward flamingo_mini/utils.py:22-50)
 *   ff_xattn_block_fwd/bwd   GatedCrossAttentionBlock.forward + autograd  flamingo_mini/gated_cross_attention.py:160-184
 *                            (MaskedCrossAttention.forward :42-131, cached K/V path :88-92,102-104)
 *   ff_text_time             media_locations.cumsum(dim=-1)               flamingo_mini/gated_cross_attention.py:97
 *   ff_quick_gelu_fwd/bwd    CLIP MLP activation (QuickGELU)               transformers CLIPMLP via modeling_flamingo.py:70-78
 *   ff_shifted_ce_fwd/bwd    shifted F.cross_entropy on the logits        flamingo_mini/modeling_flamingo.py:288-298
 *   ff_adamw_step            torch AdamW over parameters_trainable()      training/train.sh:10-13, modeling_flamingo.py:132-138
 * The primitive entry points (ff_gemm, ff_layernorm_*, ff_attention_*, ff_rows_reduce, ff_gate_grad) are the
 * kernels those are built from; they are exported so each can be parity-tested on its own.
 */
#ifndef FLAMINGO_FUSION_H
#define FLAMINGO_FUSION_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the declarations of this header are exported (tests/test_cabi.py compares the
 * dynamic symbol table of the built .so with them). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct ihipStream_t* ff_stream_t; /* == hipStream_t */

enum { FF_OK = 0, FF_ERR_SHAPE = -1, FF_ERR_UNSUPPORTED = -2, FF_ERR_WORKSPACE = -3, FF_ERR_LAUNCH = -4 };
enum { FF_DTYPE_F32 = 0, FF_DTYPE_BF16 = 1 };
enum { FF_ACT_NONE = -1, FF_ACT_GELU = 0, FF_ACT_SQRELU = 1, FF_ACT_RELU = 2 }; /* utils.py:26-30 */

int ff_version(void);          /* ABI version, bumped on any signature change (3: ff_gemm_desc.tile / .stages replace ff_gemm_set_tuning;
                                * 4: ff_xattn_desc.sync, ff_xattn_sync_bytes / _status, ff_resampler_layer_* / _prologue_* / _epilogue_*) */
const char* ff_arch(void);     /* "gfx950" */
const char* ff_last_error(void);

/* Row addressing shared by the row-wise kernels and the GEMM operands:
 * logical row r lives at  base + (r / rows_per_seg) * seg_stride + (r % rows_per_seg) * ld   (elements).
 * rows_per_seg <= 0: plain row-major with leading dimension ld.  seg_stride == 0 broadcasts one segment. */
typedef struct ff_rowmap {
    long long ld;
    long long seg_stride;
    int rows_per_seg;
    int reserved;
} ff_rowmap;

/* ------------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   v = scale * sum_k A[m,k] * B[k,n]
 *                             aux_out[m,n] = v                      (if aux_out)
 *                             v *= tanh(*gate)                      (if gate: device scalar of `dtype`)
 *                             v  = act(v)                           (if act >= 0)
 *                             v *= act'(aux_in[m,n])                (if act_bwd >= 0)
 *                             C[m,n] = v + residual[m,n]            (residual optional)
 * a_layout 0: A stored [M][K] (nn.Linear input);  1: A stored [K][M] (used for weight gradients, A = dY^T)
 * b_layout 0: B stored [N][K] (nn.Linear weight (out,in));  1: B stored [K][N]
 * a_map/b_map address the STORED rows (M or K rows of A; N or K rows of B); c_map addresses C, aux_*, residual.
 * split_k > 1 needs workspace of ff_gemm_workspace_bytes().
 * ------------------------------------------------------------------------------------------------------ */
typedef struct ff_gemm_desc {
    int dtype;
    int M, N, K;
    int a_layout, b_layout;
    ff_rowmap a_map, b_map, c_map;
    float scale;
    int act;      /* FF_ACT_* or FF_ACT_NONE */
    int act_bwd;  /* FF_ACT_* or FF_ACT_NONE */
    int split_k;  /* 0 = choose automatically */
    int tile;     /* bf16 block tile: 0 = choose automatically; 128 = 128x128 (4 waves), 6412 = 64x128, 64 = 64x64,
                   * 64002 / 128002 = 64x64 / 128x128 producer/consumer (8 waves), 128160 = 128x160 producer/consumer (A K-major only),
                   * 128168 = the same 128x160 tile with eight MFMA waves (4 x 2) + four DMA waves,
                   * 256128 = 256x128 producer/consumer, eight MFMA + eight DMA waves (chosen automatically for products with >= 4096 rows and a K-major A; with an
                   *          M-major A - weight gradients - eight MFMA + four DMA waves, selectable only),
                   * 256256 = 256x256 on sixteen waves that both issue the DMA and run the MFMAs (both operands K-major; `stages` ignored; chosen
                   *          automatically for such products with >= 4096 rows, >= 2048 contraction elements and at least one tile per CU),
                   * 3264 = 32x64 producer/consumer (decode: M <= 32 rows; both operands K-major),
                   * 3216 = weight-streaming kernel for M <= 32 rows (16 output columns per workgroup, no LDS staging; K-major operands, K % 32 == 0) */
    int stages;   /* depth of the LDS operand ring: 0 = default, 2..4 */
} ff_gemm_desc;

size_t ff_gemm_workspace_bytes(const ff_gemm_desc* d);
int ff_gemm(const ff_gemm_desc* d, const void* A, const void* B, void* C, void* aux_out, const void* aux_in,
            const void* residual, const void* gate, void* workspace, size_t workspace_bytes, ff_stream_t stream);


/* Optional measurement aid (bench.py's roofline leg): while enabled, every GEMM main-kernel launch is bracketed by two
 * HIP events on its own stream.  ff_gemm_profile_read() waits for the recorded launches, fills `out` (returns the count)
 * and clears the log.  ff_gemm_profile_enable(0) switches it off.  Recording is thread-safe (launches of any thread on any stream claim
 * their slot atomically); enable / read / disable belong to ONE controlling thread, called while no launch is in flight. */
typedef struct ff_gemm_profile_record {
    int dtype, tile, a_layout, b_layout;
    int M, N, K, nz, split_k;
    float ms;
} ff_gemm_profile_record;
int ff_gemm_profile_enable(int max_records);
int ff_gemm_profile_read(ff_gemm_profile_record* out, int max_records);
/* Introspection: block tile and split-K factor ff_gemm would choose for this problem (no device access). */
int ff_gemm_plan(const ff_gemm_desc* d, int* bm, int* bn, int* split_k);

/* ------------------------------------------------------------------------------------------------------
 * LayerNorm over the last axis (eps inside the sqrt, biased variance: torch.nn.LayerNorm).
 * Optional broadcast addend fused in front: x[r] += add[((r % add_rows_per_seg) / add_div)]  — the
 * time_pos_emb add of perceiver_resampler.py:166.
 * fwd: stats_given != 0 reuses mean/rstd instead of recomputing; y == NULL computes the statistics only.
 * bwd: dx = dx_residual + LN'(dy) (dx_residual optional, may alias dx); dgamma/dbeta are written, not accumulated.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct ff_ln_desc {
    int dtype;
    int rows, cols;
    ff_rowmap x_map;  /* x (fwd, bwd) */
    ff_rowmap y_map;  /* y (fwd) / dy (bwd) */
    ff_rowmap dx_map; /* dx and dx_residual (bwd) */
    int add_rows_per_seg, add_div;
    float eps;
    int stats_given;
} ff_ln_desc;

int ff_layernorm_fwd(const ff_ln_desc* d, const void* x, const void* add, const void* gamma, const void* beta,
                     void* y, float* mean, float* rstd, ff_stream_t stream);
size_t ff_layernorm_bwd_workspace_bytes(const ff_ln_desc* d);
int ff_layernorm_bwd(const ff_ln_desc* d, const void* dy, const void* x, const void* add, const void* gamma,
                     const float* mean, const float* rstd, void* dx, const void* dx_residual, void* dgamma,
                     void* dbeta, void* workspace, size_t workspace_bytes, ff_stream_t stream);

/* out[g][c] = sum over rows r with group(r) == g of x[r][c],  group(r) = (r % rows_per_batch) / rows_per_group.
 * (d latents: rows_per_batch = q, rows_per_group = 1;  d time_pos_emb: rows_per_batch = T*v, rows_per_group = v) */
typedef struct ff_reduce_desc {
    int dtype;
    int rows, cols;
    ff_rowmap x_map;
    int rows_per_batch, rows_per_group;
} ff_reduce_desc;
size_t ff_rows_reduce_workspace_bytes(const ff_reduce_desc* d);
int ff_rows_reduce(const ff_reduce_desc* d, const void* x, void* out, void* workspace, size_t workspace_bytes,
                   ff_stream_t stream);

/* d alpha = (1 - tanh(alpha)^2) * sum(a .* b) over rows x cols  (gated_cross_attention.py:180,182 backward) */
size_t ff_gate_grad_workspace_bytes(int rows, int cols);
int ff_gate_grad(int dtype, int rows, int cols, const void* a, const void* b, const void* alpha, void* dalpha,
                 void* workspace, size_t workspace_bytes, ff_stream_t stream);

/* text_time[b][i] = sum_{j<=i} media_locations[b][j]   (media_locations: int64 / int32 / uint8(bool)) */
int ff_text_time(int batch, int n_tokens, const void* media_locations, int elem_bytes, int* text_time,
                 ff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Attention core  O = softmax(Q K^T) V  over row ranges, never materialising the score matrix.
 *   mode FF_ATTN_DENSE : every query attends to all n_kv keys (resampler, perceiver_resampler.py:85-92)
 *   mode FF_ATTN_MEDIA : query i attends to the n_visual keys of image text_time[b][i]
 *                        (gated_cross_attention.py:97-123): text_time == 0 -> output row 0 and no gradient;
 *                        text_time > n_kv / n_visual -> uniform average over all keys (fully masked row).
 * Q is expected pre-scaled.  Element (b, row, h, d) of a tensor lives at base + b*sb + row*sr + h*sh + d.
 * lse[b][h][row] (fp32) is saved by fwd and consumed by bwd.
 * ------------------------------------------------------------------------------------------------------ */
enum { FF_ATTN_DENSE = 0, FF_ATTN_MEDIA = 1 };
typedef struct ff_strides {
    long long sb, sr, sh;
} ff_strides;
typedef struct ff_attn_desc {
    int dtype;
    int batch, heads, dim_head;
    int n_q, n_kv;
    int mode, n_visual;
    int tt_stride;   /* row stride of text_time (elements); query i of batch b reads text_time[b*tt_stride + tt_offset + i] */
    int tt_offset;
    ff_strides q, k, v, o;      /* forward tensors; dq/dk/dv/do use dq, dk, dv, dout strides below */
    ff_strides dq, dk, dv, dout;
} ff_attn_desc;

int ff_attention_fwd(const ff_attn_desc* d, const void* Q, const void* K, const void* V, const int* text_time,
                     void* O, float* lse, ff_stream_t stream);
size_t ff_attention_bwd_workspace_bytes(const ff_attn_desc* d);
int ff_attention_bwd(const ff_attn_desc* d, const void* Q, const void* K, const void* V, const int* text_time,
                     const void* O, const void* dO, const float* lse, void* dQ, void* dK, void* dV,
                     void* workspace, size_t workspace_bytes, ff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * PerceiverResampler (perceiver_resampler.py:99-188).
 * x_f: (batch, n_frames, n_tokens, dim) contiguous; out: (batch, num_latents, dim).
 * params / grads: arrays of device pointers in this order (names = reference state_dict keys):
 *   [0] latents  [1] time_pos_emb  [2] norm.weight  [3] norm.bias
 *   then per layer i, base = 4 + 12*i:
 *   +0 layers.i.0.norm_media.weight  +1 .norm_media.bias  +2 .norm_latents.weight  +3 .norm_latents.bias
 *   +4 layers.i.0.to_q.weight  +5 .to_k.weight  +6 .to_v.weight  +7 .to_out.weight
 *   +8 layers.i.1.0.weight  +9 layers.i.1.0.bias  +10 layers.i.1.1.weight  +11 layers.i.1.3.weight
 * `saved` persists fwd -> bwd (activations, statistics); `scratch` is transient.
 * bwd writes every gradient (no accumulation); dx_f may be NULL (CLIP frozen) — d time_pos_emb is still exact.
 * This is the whole stack in one call, like the reference's own call site (modeling_flamingo.py:176 calls the module once): only the
 * stack-level call can group the weight gradients of four layers per launch and finish all 3 * depth + 1 LayerNorm-backward reductions
 * with three launches.  SURVEY.md 8-b2's per-layer export set (ff_resampler_layer_fwd/bwd + prologue / epilogue) follows below; it is what
 * a data-parallel caller uses (one gradient bucket per layer).
 * ------------------------------------------------------------------------------------------------------ */
#define FF_RESAMPLER_GLOBAL_PARAMS 4
#define FF_RESAMPLER_LAYER_PARAMS 12
typedef struct ff_resampler_desc {
    int dtype;
    int batch, n_frames, n_tokens, dim;
    int depth, heads, dim_head, num_latents, num_time_embeds, ff_mult, act;
} ff_resampler_desc;
size_t ff_resampler_saved_bytes(const ff_resampler_desc* d);
size_t ff_resampler_scratch_bytes(const ff_resampler_desc* d);
int ff_resampler_fwd(const ff_resampler_desc* d, const void* x_f, const void* const* params, void* out,
                     void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, ff_stream_t stream);
int ff_resampler_bwd(const ff_resampler_desc* d, const void* x_f, const void* const* params, const void* dout,
                     const void* saved, size_t saved_bytes, void* const* grads, void* dx_f, void* scratch,
                     size_t scratch_bytes, ff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * PerceiverResampler, ONE LAYER PER CALL (ABI 4; SURVEY.md 8-b2's minimum export set; perceiver_resampler.py:181-183 is a per-layer loop).
 *   forward :  prologue_fwd (statistics of x_f + time_pos_emb, :166 / :52, shared by every layer's norm_media)
 *              -> depth x layer_fwd (x = x + attn(x_f, x); x = x + ffw(x), :182-183; layer 0 takes the latents (num_latents, dim) with
 *                 x_in_is_latents = 1: they are broadcast over the batch, :179)
 *              -> epilogue_fwd (final LayerNorm, :187)
 *   backward:  epilogue_bwd -> depth x layer_bwd (last layer first) -> prologue_bwd
 * layer_params / layer_grads: the 12 pointers of ONE layer in the order of ff_resampler_fwd's per-layer block (+0 norm_media.weight ... +11
 * layers.i.1.3.weight).  layer_bwd writes every one of the layer's parameter gradients - final when the call returns: a data-parallel
 * caller gets one gradient bucket per layer, which can leave while the layer below runs backward - and d x_in (batch, num_latents, dim) (for
 * layer 0: the gradient of the BROADCAST latents; prologue_bwd sums it over the batch), and adds its share of d x_f to `dx_f`
 * (dx_f_accumulate = 0 for the first layer_bwd call of a pass, which overwrites).  prologue_bwd: d latents, d time_pos_emb (complete).
 * ff_resampler_desc.depth is ignored by these calls.  `saved` buffers persist fwd -> bwd; every `scratch` is ff_resampler_layer_scratch_bytes().
 * The stack-level ff_resampler_fwd / _bwd above remain the single-GPU fast path (they group the weight gradients of four layers per
 * launch and finish all LayerNorm-backward reductions with three launches).
 * ------------------------------------------------------------------------------------------------------ */
size_t ff_resampler_prologue_saved_bytes(const ff_resampler_desc* d);
size_t ff_resampler_layer_saved_bytes(const ff_resampler_desc* d);
size_t ff_resampler_layer_scratch_bytes(const ff_resampler_desc* d);
size_t ff_resampler_epilogue_saved_bytes(const ff_resampler_desc* d);
int ff_resampler_prologue_fwd(const ff_resampler_desc* d, const void* x_f, const void* time_pos_emb, void* saved_pro, size_t saved_pro_bytes,
                              ff_stream_t stream);
int ff_resampler_layer_fwd(const ff_resampler_desc* d, const void* x_f, const void* time_pos_emb, const void* saved_pro, size_t saved_pro_bytes,
                           const void* x_in, int x_in_is_latents, const void* const* layer_params, void* x_out, void* saved, size_t saved_bytes,
                           void* scratch, size_t scratch_bytes, ff_stream_t stream);
int ff_resampler_epilogue_fwd(const ff_resampler_desc* d, const void* x_last, const void* norm_weight, const void* norm_bias, void* out,
                              void* saved_epi, size_t saved_epi_bytes, ff_stream_t stream);
int ff_resampler_epilogue_bwd(const ff_resampler_desc* d, const void* dout, const void* x_last, const void* norm_weight, const void* saved_epi,
                              size_t saved_epi_bytes, void* dx_last, void* d_norm_weight, void* d_norm_bias, void* scratch, size_t scratch_bytes,
                              ff_stream_t stream);
int ff_resampler_layer_bwd(const ff_resampler_desc* d, const void* x_f, const void* time_pos_emb, const void* saved_pro, size_t saved_pro_bytes,
                           const void* x_in, int x_in_is_latents, const void* const* layer_params, const void* dx_out, const void* saved,
                           size_t saved_bytes, void* const* layer_grads, void* dx_in, void* dx_f, int dx_f_accumulate, void* scratch,
                           size_t scratch_bytes, ff_stream_t stream);
int ff_resampler_prologue_bwd(const ff_resampler_desc* d, const void* dx0, const void* dx_f, void* d_latents, void* d_time_pos_emb, void* scratch,
                              size_t scratch_bytes, ff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * GatedCrossAttentionBlock (gated_cross_attention.py:135-184).
 * y: (batch, n_tokens, dim); visual_features: (batch, n_media, n_visual, dim_visual); text_time from ff_text_time.
 * params / grads order:
 *   [0] alpha_attn [1] alpha_ffw [2] attn.norm.weight [3] attn.norm.bias [4] attn.to_q.weight
 *   [5] attn.to_kv.weight [6] attn.to_out.weight [7] ffw.0.weight [8] ffw.0.bias [9] ffw.1.weight [10] ffw.3.weight
 * Cached decode (previous_kv): pass cached_k/cached_v (+ strides) and visual_features = NULL; n_media is then
 * n_kv / n_visual and text_time rows are read at tt_offset (the last n_tokens entries, :102-104).
 * Uncached: K/V are produced in `saved` at ff_xattn_kv_offset() as (batch, n_media*n_visual, 2, heads, dim_head).
 * Non-zero cached_k / cached_v strides in the descriptor <=> K / V come from outside (cached decode or ff_kv_project_fwd); `saved`
 * then holds no K/V region (size queries and calls must use the same descriptor).
 * `sync` (ABI 4, optional): ff_xattn_sync_bytes() bytes of device memory that were ZERO when first handed to the library and that only the
 * library writes afterwards.  With it, and at the training / decode shape of the published configurations (bf16, 8 heads of 64, at most 32
 * tokens and 64 keys per sample, dim a multiple of 256 up to 1280 - at 1536 the backward kernel's resident rows + ring + tiles are 165 184 B
 * of LDS, over the CU's 160 KiB, and the block keeps its separate launches -, batch <= FF_XATTN_SYNC_SLOTS, and a device that reports 8 XCDs
 * of at least 16 CUs each: see "Co-residency" in csrc/ff_xattn_fused.hip), `to_out` + tanh gate + residual
 * (gated_cross_attention.py:124-126,180) run INSIDE the fused LayerNorm -> to_q -> attention launch, and d LN(y) = d q . Wq inside the fused
 * attention-backward launch: the eight (sample, head) workgroups of a sample exchange their tiles through per-sample arrival counters in
 * `sync` instead of through a kernel boundary (two 64 x 64-tile GEMM launches per block and step less).  At the training shape the
 * LayerNorm behind each of those outputs runs in the same launch as well - forward: LN(y1) of the feed-forward (utils.py:46); backward: the
 * LayerNorm backward of LN(y) - with the rows' statistics making one more trip through a second bank of counters (two more launches per block
 * and step less).  Calls that share a `sync` buffer must be ordered on one stream (the counters are per sample, not per call); NULL keeps the
 * separate launches.  Results are the same either way up to the rounding of fp32 sums taken in another order; the word behind the first two
 * banks is an error flag (ff_xattn_sync_status; non-zero: an arrival wait timed out - a launch was denied co-residency of a sample's eight
 * workgroups - and that call's output is invalid).  A caller MUST read that flag before trusting results computed with a `sync` buffer: the
 * Python layer does so after warm-up steps, every N graph replays, at every eager optimizer step (non-blocking) and at the end of bench.py.
 * ------------------------------------------------------------------------------------------------------ */
#define FF_XATTN_PARAMS 11
#define FF_XATTN_SYNC_SLOTS 1024
typedef struct ff_xattn_desc {
    int dtype;
    int batch, n_tokens, dim, dim_visual;
    int n_media, n_visual, heads, dim_head, ff_mult, act;
    int tt_stride, tt_offset;
    ff_strides cached_k, cached_v; /* used only when cached_k != NULL */
    void* sync;                    /* see above; NULL = none */
} ff_xattn_desc;
size_t ff_xattn_sync_bytes(void);              /* (4 * FF_XATTN_SYNC_SLOTS + 64) * 4: two banks of per-sample counters each way + a status word */
int ff_xattn_sync_status(const void* sync, ff_stream_t stream);   /* synchronises `stream`; 0 = no wait ever timed out, 1 = one did, < 0 = error */
size_t ff_xattn_saved_bytes(const ff_xattn_desc* d);
size_t ff_xattn_scratch_bytes(const ff_xattn_desc* d);
size_t ff_xattn_kv_offset(const ff_xattn_desc* d);
int ff_xattn_block_fwd(const ff_xattn_desc* d, const void* y, const void* visual_features, const int* text_time,
                       const void* const* params, const void* cached_k, const void* cached_v, void* y_out,
                       void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, ff_stream_t stream);
int ff_xattn_block_bwd(const ff_xattn_desc* d, const void* y, const void* visual_features, const int* text_time,
                       const void* const* params, const void* dy_out, const void* saved, size_t saved_bytes,
                       void* const* grads, void* dy, void* dvisual_features, void* scratch, size_t scratch_bytes,
                       ff_stream_t stream);


/* ------------------------------------------------------------------------------------------------------
 * Key / value projection hoisted out of the cross-attention layers.  gated_cross_attention.py:84-86 applies every layer's
 * `to_kv` to the SAME visual features; projecting for all layers in grouped launches (and summing d visual_features once)
 * replaces n_layers small GEMMs per direction.  w_kv[l] (kv_dim, dim_visual) = layer l's to_kv.weight; kv_out[l] / dkv[l]
 * (rows, kv_dim) with rows = batch * n_media * n_visual, K = columns [0, kv_dim/2), V = the rest - the layout
 * ff_xattn_block_fwd expects for cached_k / cached_v with strides {n_kv * kv_dim, kv_dim, dim_head}.
 * Training step:  ff_kv_project_fwd -> per layer ff_xattn_block_fwd(cached_k = kv_out[l], cached_v = kv_out[l] + kv_dim/2)
 *                 ... per layer ff_xattn_block_bwd_kv(... dkv[l]) -> ff_kv_project_bwd (d to_kv.weight of every layer, d visual_features).
 * ------------------------------------------------------------------------------------------------------ */
typedef struct ff_kvproj_desc {
    int dtype;
    int n_layers;
    int rows;         /* batch * n_media * n_visual */
    int dim_visual;
    int kv_dim;       /* 2 * heads * dim_head */
} ff_kvproj_desc;
size_t ff_kv_project_workspace_bytes(const ff_kvproj_desc* d, int with_dvisual_features);
int ff_kv_project_fwd(const ff_kvproj_desc* d, const void* visual_features, const void* const* w_kv, void* const* kv_out,
                      void* workspace, size_t workspace_bytes, ff_stream_t stream);
int ff_kv_project_bwd(const ff_kvproj_desc* d, const void* visual_features, const void* const* w_kv, const void* const* dkv,
                      void* const* dw_kv, void* dvisual_features, void* workspace, size_t workspace_bytes, ff_stream_t stream);
/* Backward of a block whose K / V came from ff_kv_project_fwd (d->cached_k / cached_v = their strides): writes d K / d V into
 * `dkv` (same layout as kv_out[l]); grads[5] (to_kv.weight) is not written. */
int ff_xattn_block_bwd_kv(const ff_xattn_desc* d, const void* y, const void* k, const void* v, const int* text_time,
                          const void* const* params, const void* dy_out, const void* saved, size_t saved_bytes, void* const* grads,
                          void* dy, void* dkv, void* scratch, size_t scratch_bytes, ff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Deferred, grouped weight gradients.  The four weight-gradient GEMMs of a block (d ffw.3.weight, d ffw.1.weight,
 * d attn.to_out.weight, d attn.to_q.weight = 1/3 of its FLOPs) are not on the critical path of backward and a single one
 * fills a third of the chip.  ff_xattn_block_bwd_kv_data is ff_xattn_block_bwd_kv without them: it leaves their operands
 * (d y1, d H, d Qs) in the caller-owned `stash`; ff_xattn_wgrad_grouped then computes them for up to FF_WGRAD_GROUP_MAX
 * same-shaped blocks per call in four grouped launches.  grads[5] (to_kv) is never written.  The LayerNorm / gate gradients
 * (grads[0..3], [7], [8]) are COMPLETE only after the grouped call as well: the data pass runs the one-pass LayerNorm backward and
 * leaves its per-workgroup partial sums in the stash; the grouped call reduces those of all its blocks with one launch (72 -> 9
 * launches per step, none of them on the data-gradient chain, at flamingo-mini's size).  For that the data pass requires y, dy_out, dy and the two
 * LayerNorm weight vectors to be 16-byte aligned (FF_ERR_SHAPE otherwise; use ff_xattn_block_bwd_kv for odd views).
 * `params` / `grads` of the grouped call: n_blocks * FF_XATTN_PARAMS pointers, block after block, the SAME gradient pointers the
 * data pass received; dy_out / saved / stash: one pointer per block (the buffers of that block's data pass).
 * ------------------------------------------------------------------------------------------------------ */
#define FF_WGRAD_GROUP_MAX 12
size_t ff_xattn_wgrad_stash_bytes(const ff_xattn_desc* d);
size_t ff_xattn_wgrad_workspace_bytes(const ff_xattn_desc* d);
int ff_xattn_block_bwd_kv_data(const ff_xattn_desc* d, const void* y, const void* k, const void* v, const int* text_time,
                               const void* const* params, const void* dy_out, const void* saved, size_t saved_bytes, void* const* grads,
                               void* dy, void* dkv, void* stash, size_t stash_bytes, void* scratch, size_t scratch_bytes, ff_stream_t stream);
int ff_xattn_wgrad_grouped(const ff_xattn_desc* d, int n_blocks, const void* const* dy_out, const void* const* saved, size_t saved_bytes,
                           const void* const* stash, size_t stash_bytes, const void* const* params, void* const* grads, void* workspace,
                           size_t workspace_bytes, ff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW over the trainable parameters (the reference trains with `--optim adamw_torch`,
 * training/train.sh:10-13; parameters_trainable() modeling_flamingo.py:132-138).  All tensors of one call share `dtype`
 * (params, grads and both moment buffers); math is fp32.  step is the 1-based step count AFTER incrementing
 * (bias corrections 1 - beta^step).  grad_scale multiplies every gradient first (0 = 1.0; e.g. 1/world after a SUM all-reduce).
 * ------------------------------------------------------------------------------------------------------ */
typedef struct ff_adamw_desc {
    int dtype;
    int n_tensors;
    int step;
    float lr, beta1, beta2, eps, weight_decay, grad_scale;
    const float* step_dev;   /* optional device scalar holding the step count (>= 1): used instead of `step`, so the launch can be
                              * replayed from a captured HIP graph while the caller advances the counter on the device */
} ff_adamw_desc;
int ff_adamw_step(const ff_adamw_desc* d, void* const* params, const void* const* grads, void* const* exp_avg,
                  void* const* exp_avg_sq, const long long* numels, ff_stream_t stream);
/* Mixed-precision variant (the reference trains with `--fp16` autocast, i.e. fp32 master weights and fp32 Adam moments,
 * training/train.sh:24): parameters / gradients in d->dtype, the two moments in `state_dtype` (d->dtype or FF_DTYPE_F32), and -
 * for bf16 parameters - optional fp32 `master` copies: the update is applied to the master copy and the bf16 parameter is its
 * rounding, written by the same kernel.  `lr_dev` (optional device scalar) replaces d->lr, so a learning-rate schedule stays
 * effective when the launch is replayed from a captured HIP graph. */
int ff_adamw_step_mixed(const ff_adamw_desc* d, int state_dtype, void* const* params, const void* const* grads, void* const* exp_avg,
                        void* const* exp_avg_sq, float* const* master, const float* lr_dev, const long long* numels, ff_stream_t stream);


/* ------------------------------------------------------------------------------------------------------
 * Shifted next-token cross-entropy (modeling_flamingo.py:288-298): position i predicts labels[i+1].
 * logits (batch, seq, vocab) contiguous; labels (batch, seq) int64; rows = batch * (seq - 1).
 * fwd: loss_row[r] = logsumexp(logits[b,i,:]) - logits[b,i,labels[b,i+1]] (0 where the label == ignore_index), lse[r] saved.
 * bwd: dlogits[b,i,:] = (softmax - onehot) * grad_row[r];  dlogits[b,seq-1,:] = 0.   The reduction (mean over the
 * non-ignored rows / sum / none) is applied by the caller on the `rows` values.
 * ------------------------------------------------------------------------------------------------------ */
int ff_shifted_ce_fwd(int dtype, int batch, int seq, int vocab, const void* logits, const long long* labels, long long ignore_index,
                      float* loss_row, float* lse, ff_stream_t stream);
int ff_shifted_ce_bwd(int dtype, int batch, int seq, int vocab, const void* logits, const long long* labels, long long ignore_index,
                      const float* lse, const float* grad_row, void* dlogits, ff_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * QuickGELU of the CLIP vision tower, y = x * sigmoid(1.702 x) (transformers.activations.QuickGELUActivation, selected by
 * the openai/clip-vit-* configs the reference loads in modeling_flamingo.py:70-78), and its derivative; n contiguous elements.
 * ------------------------------------------------------------------------------------------------------ */
int ff_quick_gelu_fwd(int dtype, long long n, const void* x, void* y, ff_stream_t stream);
int ff_quick_gelu_bwd(int dtype, long long n, const void* x, const void* dy, void* dx, ff_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FLAMINGO_FUSION_H */
