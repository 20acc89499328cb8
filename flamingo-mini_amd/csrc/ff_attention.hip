// Attention core of the fusion path:  O = softmax(Q K^T) V  without materialising the score matrix.
// Serves both PerceiverAttentionLayer (dense, perceiver_resampler.py:79-92) and MaskedCrossAttention
// (per-image equality mask, zero rows, uniform rows: gated_cross_attention.py:95-123).
//
// Work split: a workgroup = 4 waves = 64 "own" rows (queries in fwd / dQ, keys in dK/dV); the "other" side is
// streamed through LDS in 64-row tiles shared by the 4 waves.  Everything is kept TRANSPOSED:
//     Z[other][own]   = X_tile[other][:] . own[own][:]          (S^T = K Q^T  resp.  S = Q K^T)
//     Acc^T[d][own]  += Y_tile[other][d] * Z'[other][own]       (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS)
// so that (a) the softmax statistics of an own row are lane-local (lane & 15 = own row), (b) Z' leaves the first
// MFMA in exactly the register layout the second MFMA wants as its B operand (no LDS round trip for P), and
// (c) the strided operand of the second product comes from the row-major LDS tile via ds_read_b64_tr_b16 (bf16)
// or plain ds_read_b32 (fp32).  Wave64 reductions: 2 shuffles (xor 16, 32) per row maximum / sum.
#include "ff_common.h"
#include "ff_internal.h"
#include "ff_attention_core.h"

namespace ff {

// =====================================================================================================
// forward
// =====================================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const ff_attn_desc d_in, const T* __restrict__ Q, const T* __restrict__ K,
                                                       const T* __restrict__ V, const int* __restrict__ tt, T* __restrict__ O,
                                                       float* __restrict__ lse) {
    const ff_attn_desc d = fetch_args(d_in);
    pin_args(Q, K, V, tt, O, lse);
    constexpr int LD = DH + AttnCfg<T>::pad;
    __shared__ __attribute__((aligned(16))) T sK[kTile * LD];
    __shared__ __attribute__((aligned(16))) T sV[kTile * LD];
    __shared__ int sh[8];
    const int b = blockIdx.z, h = blockIdx.y;
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4, w = threadIdx.x >> 6;
    const int q = blockIdx.x * kTile + w * 16 + c;
    const RowRange rr = row_range(d, tt, b, q);
    int blo, bhi;
    block_range(rr.lo, rr.hi, sh, blo, bhi);

    OwnFrag<T, DH> fq;
    fq.load(q < d.n_q ? Q + b * d.q.sb + (long long)q * d.q.sr + h * d.q.sh : nullptr, g);

    f32x4 acc[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; dt++) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = kNegBig, lsum = 0.f;

    const T* Kb = K + b * d.k.sb + h * d.k.sh;
    const T* Vb = V + b * d.v.sb + h * d.v.sh;
    attn_fwd_loop<T, DH, PrefetchStage<T, DH>>(d, fq, rr, blo, bhi, Kb, Vb, sK, sV, acc, m, lsum);
    lsum = group_sum(lsum);
    if (q < d.n_q) {
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        store_acc_row<T, DH>(O + b * d.o.sb + (long long)q * d.o.sr + h * d.o.sh, acc, inv, g);
        if (g == 0 && lse) lse[((long long)b * d.heads + h) * d.n_q + q] = lsum > 0.f ? m + __logf(lsum) : kPosBig;
    }
}

// =====================================================================================================
// backward, dQ (own rows = queries).  Also emits Dsum[b][h][q] = sum_d dO*O for the dK/dV kernel.
// =====================================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const ff_attn_desc d_in, const T* __restrict__ Q, const T* __restrict__ K,
                                                          const T* __restrict__ V, const int* __restrict__ tt, const T* __restrict__ O,
                                                          const T* __restrict__ dO, const float* __restrict__ lse, T* __restrict__ dQ,
                                                          float* __restrict__ Dsum) {
    const ff_attn_desc d = fetch_args(d_in);
    pin_args(Q, K, V, tt, O, dO, lse, dQ, Dsum);
    constexpr int LD = DH + AttnCfg<T>::pad;
    __shared__ __attribute__((aligned(16))) T sK[kTile * LD];
    __shared__ __attribute__((aligned(16))) T sV[kTile * LD];
    __shared__ int sh[8];
    const int b = blockIdx.z, h = blockIdx.y;
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4, w = threadIdx.x >> 6;
    const int q = blockIdx.x * kTile + w * 16 + c;
    const bool qok = q < d.n_q;
    RowRange rr = row_range(d, tt, b, q);

    OwnFrag<T, DH> fq, fdo, fo;
    fq.load(qok ? Q + b * d.q.sb + (long long)q * d.q.sr + h * d.q.sh : nullptr, g);
    fdo.load(qok ? dO + b * d.dout.sb + (long long)q * d.dout.sr + h * d.dout.sh : nullptr, g);
    fo.load(qok ? O + b * d.o.sb + (long long)q * d.o.sr + h * d.o.sh : nullptr, g);
    const float Dq = group_sum(fdo.dot(fo));
    const long long sidx = ((long long)b * d.heads + h) * d.n_q + q;
    if (qok && g == 0) Dsum[sidx] = Dq;
    const float L = qok ? lse[sidx] : kPosBig;
    if (!rr.softmax) rr.lo = rr.hi = 0;  // zero / uniform rows: no gradient reaches the scores
    int blo, bhi;
    block_range(rr.lo, rr.hi, sh, blo, bhi);

    f32x4 acc[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; dt++) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* Kb = K + b * d.k.sb + h * d.k.sh;
    const T* Vb = V + b * d.v.sb + h * d.v.sh;
    attn_dq_loop<T, DH, PrefetchStage<T, DH>>(d, fq, fdo, rr, L, Dq, blo, bhi, Kb, Vb, sK, sV, acc);
    if (qok) store_acc_row<T, DH>(dQ + b * d.dq.sb + (long long)q * d.dq.sr + h * d.dq.sh, acc, 1.f, g);
}

// =====================================================================================================
// backward, dK / dV (own rows = keys)
// =====================================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const ff_attn_desc d_in, const T* __restrict__ Q, const T* __restrict__ K,
                                                           const T* __restrict__ V, const int* __restrict__ tt, const T* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ Dsum,
                                                           T* __restrict__ dK, T* __restrict__ dV) {
    const ff_attn_desc d = fetch_args(d_in);
    pin_args(Q, K, V, tt, dO, lse, Dsum, dK, dV);
    constexpr int LD = DH + AttnCfg<T>::pad;
    __shared__ __attribute__((aligned(16))) T sQ[kTile * LD];
    __shared__ __attribute__((aligned(16))) T sDO[kTile * LD];
    __shared__ int s_lo[kTile], s_hi[kTile], s_flag[kTile];
    __shared__ float s_lse[kTile], s_D[kTile];
    const int b = blockIdx.z, h = blockIdx.y;
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4, w = threadIdx.x >> 6;
    const int key = blockIdx.x * kTile + w * 16 + c;
    const bool kok = key < d.n_kv;

    OwnFrag<T, DH> fk, fv;
    fk.load(kok ? K + b * d.k.sb + (long long)key * d.k.sr + h * d.k.sh : nullptr, g);
    fv.load(kok ? V + b * d.v.sb + (long long)key * d.v.sr + h * d.v.sh : nullptr, g);

    f32x4 acc_k[DH / 16], acc_v[DH / 16];
#pragma unroll
    for (int dt = 0; dt < DH / 16; dt++) { acc_k[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_v[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const T* Qb = Q + b * d.q.sb + h * d.q.sh;
    const T* dOb = dO + b * d.dout.sb + h * d.dout.sh;
    // query tile t + 1 (Q, dO and the per-query statistics) is requested into registers as soon as tile t is published: its latency runs under
    // tile t's MFMAs (round 4; before, every tile's loads sat in front of its products)
    TileRegs<T, DH> rq, rdo;
    RowRange rr_n = {0, 0, 0, 0};
    float lse_n = kPosBig, D_n = 0.f;
    auto issue = [&](int q0) {
        stage_issue<T, DH>(rq, Qb, d.q.sr, q0, d.n_q);
        stage_issue<T, DH>(rdo, dOb, d.dout.sr, q0, d.n_q);
        if (threadIdx.x < kTile) {
            const int q = q0 + threadIdx.x;
            rr_n = row_range(d, tt, b, q);
            const long long sidx = ((long long)b * d.heads + h) * d.n_q + q;
            lse_n = q < d.n_q ? lse[sidx] : kPosBig;
            D_n = q < d.n_q ? Dsum[sidx] : 0.f;
        }
    };
    issue(0);
    for (int q0 = 0; q0 < d.n_q; q0 += kTile) {
        __syncthreads();
        stage_commit<T, DH>(rq, sQ);
        stage_commit<T, DH>(rdo, sDO);
        if (threadIdx.x < kTile) {
            s_lo[threadIdx.x] = rr_n.lo;
            s_hi[threadIdx.x] = rr_n.hi;
            s_flag[threadIdx.x] = rr_n.softmax | (rr_n.uniform << 1);
            s_lse[threadIdx.x] = lse_n;
            s_D[threadIdx.x] = D_n;
        }
        __syncthreads();
        if (q0 + kTile < d.n_q) issue(q0 + kTile);
        attn_dkv_step<T, DH>(key, fk, fv, sQ, sDO, s_lo, s_hi, s_flag, s_lse, s_D, acc_k, acc_v);
    }
    if (kok) {
        store_acc_row<T, DH>(dK + b * d.dk.sb + (long long)key * d.dk.sr + h * d.dk.sh, acc_k, 1.f, g);
        store_acc_row<T, DH>(dV + b * d.dv.sb + (long long)key * d.dv.sr + h * d.dv.sh, acc_v, 1.f, g);
    }
}

// =====================================================================================================
// host side
// =====================================================================================================
static int check_desc(const ff_attn_desc& d, bool bwd) {
    FF_CHECK(d.batch > 0 && d.heads > 0 && d.n_q > 0 && d.n_kv > 0, FF_ERR_SHAPE, "attention: bad shape b=%d h=%d nq=%d nkv=%d", d.batch,
             d.heads, d.n_q, d.n_kv);
    FF_CHECK(d.mode == FF_ATTN_DENSE || (d.mode == FF_ATTN_MEDIA && d.n_visual > 0 && d.n_kv % d.n_visual == 0), FF_ERR_SHAPE,
             "attention: mode %d with n_visual=%d n_kv=%d", d.mode, d.n_visual, d.n_kv);
    const int vec = d.dtype == FF_DTYPE_BF16 ? 8 : 4;
    auto ok = [&](const ff_strides& s) { return s.sb % vec == 0 && s.sr % vec == 0 && s.sh % vec == 0; };
    bool aligned = ok(d.q) && ok(d.k) && ok(d.v) && ok(d.o);
    if (bwd) aligned = aligned && ok(d.dq) && ok(d.dk) && ok(d.dv) && ok(d.dout);
    FF_CHECK(aligned, FF_ERR_UNSUPPORTED, "attention: strides must be multiples of %d elements", vec);
    return FF_OK;
}

#define FF_ATTN_DISPATCH(d, ...)                                                                                  \
    do {                                                                                                           \
        if ((d).dtype == FF_DTYPE_BF16) {                                                                          \
            typedef bf16 T;                                                                                        \
            if ((d).dim_head == 64) { constexpr int DH = 64; __VA_ARGS__; }                                               \
            else if ((d).dim_head == 32) { constexpr int DH = 32; __VA_ARGS__; }                                          \
            else if ((d).dim_head == 128) { constexpr int DH = 128; __VA_ARGS__; }                                        \
            else FF_CHECK(false, FF_ERR_UNSUPPORTED, "attention: bf16 dim_head %d (supported 32/64/128)", (d).dim_head); \
        } else if ((d).dtype == FF_DTYPE_F32) {                                                                    \
            typedef float T;                                                                                       \
            if ((d).dim_head == 64) { constexpr int DH = 64; __VA_ARGS__; }                                               \
            else if ((d).dim_head == 16) { constexpr int DH = 16; __VA_ARGS__; }                                          \
            else if ((d).dim_head == 32) { constexpr int DH = 32; __VA_ARGS__; }                                          \
            else if ((d).dim_head == 128) { constexpr int DH = 128; __VA_ARGS__; }                                        \
            else FF_CHECK(false, FF_ERR_UNSUPPORTED, "attention: fp32 dim_head %d (supported 16/32/64/128)", (d).dim_head); \
        } else FF_CHECK(false, FF_ERR_UNSUPPORTED, "attention: dtype %d", (d).dtype);                              \
    } while (0)

int attention_fwd(const ff_attn_desc& d, const void* Q, const void* K, const void* V, const int* tt, void* O, float* lse, hipStream_t st) {
    FF_TRY(check_desc(d, false));
    FF_CHECK(Q && K && V && O && (d.mode == FF_ATTN_DENSE || tt), FF_ERR_SHAPE, "attention_fwd: null argument");
    const dim3 grid(cdiv(d.n_q, kTile), d.heads, d.batch);
    const int pid = profile_begin(d.dtype, -1, 0, 0, d.n_q, d.n_kv, d.dim_head, d.batch * d.heads, d.mode, st);
    FF_ATTN_DISPATCH(d, attn_fwd_kernel<T, DH><<<grid, dim3(256), 0, st>>>(d, (const T*)Q, (const T*)K, (const T*)V, tt, (T*)O, lse));
    profile_end(pid, st);
    return check_launch("attn_fwd");
}

size_t attention_bwd_workspace(const ff_attn_desc& d) { return (size_t)d.batch * d.heads * d.n_q * sizeof(float); }

int attention_bwd_dkv(const ff_attn_desc& d, const void* Q, const void* K, const void* V, const int* tt, const void* dO, const float* lse,
                      const float* Dsum, void* dK, void* dV, hipStream_t st) {
    FF_TRY(check_desc(d, true));
    FF_CHECK(Q && K && V && dO && lse && Dsum && dK && dV && (d.mode == FF_ATTN_DENSE || tt), FF_ERR_SHAPE, "attention_bwd_dkv: null argument");
    const dim3 gk(cdiv(d.n_kv, kTile), d.heads, d.batch);
    const int pid = profile_begin(d.dtype, -3, 0, 0, d.n_q, d.n_kv, d.dim_head, d.batch * d.heads, d.mode, st);
    FF_ATTN_DISPATCH(d, attn_bwd_dkv_kernel<T, DH><<<gk, dim3(256), 0, st>>>(d, (const T*)Q, (const T*)K, (const T*)V, tt, (const T*)dO, lse, Dsum, (T*)dK, (T*)dV));
    profile_end(pid, st);
    return check_launch("attn_bwd_dkv");
}

int attention_bwd(const ff_attn_desc& d, const void* Q, const void* K, const void* V, const int* tt, const void* O, const void* dO,
                  const float* lse, void* dQ, void* dK, void* dV, void* ws, size_t ws_bytes, hipStream_t st) {
    FF_TRY(check_desc(d, true));
    FF_CHECK(Q && K && V && O && dO && lse && dQ && dK && dV && (d.mode == FF_ATTN_DENSE || tt), FF_ERR_SHAPE, "attention_bwd: null argument");
    FF_CHECK(ws && ws_bytes >= attention_bwd_workspace(d), FF_ERR_WORKSPACE, "attention_bwd workspace: need %zu have %zu",
             attention_bwd_workspace(d), ws_bytes);
    float* Dsum = (float*)ws;
    const dim3 gq(cdiv(d.n_q, kTile), d.heads, d.batch), gk(cdiv(d.n_kv, kTile), d.heads, d.batch);
    int pid = profile_begin(d.dtype, -2, 0, 0, d.n_q, d.n_kv, d.dim_head, d.batch * d.heads, d.mode, st);
    FF_ATTN_DISPATCH(d, attn_bwd_dq_kernel<T, DH><<<gq, dim3(256), 0, st>>>(d, (const T*)Q, (const T*)K, (const T*)V, tt, (const T*)O, (const T*)dO, lse, (T*)dQ, Dsum));
    profile_end(pid, st);
    FF_TRY(check_launch("attn_bwd_dq"));
    pid = profile_begin(d.dtype, -3, 0, 0, d.n_q, d.n_kv, d.dim_head, d.batch * d.heads, d.mode, st);
    FF_ATTN_DISPATCH(d, attn_bwd_dkv_kernel<T, DH><<<gk, dim3(256), 0, st>>>(d, (const T*)Q, (const T*)K, (const T*)V, tt, (const T*)dO, lse, (const float*)Dsum, (T*)dK, (T*)dV));
    profile_end(pid, st);
    return check_launch("attn_bwd_dkv");
}

}  // namespace ff

extern "C" int ff_attention_fwd(const ff_attn_desc* d, const void* Q, const void* K, const void* V, const int* text_time, void* O,
                                float* lse, ff_stream_t stream) {
    return ff::attention_fwd(*d, Q, K, V, text_time, O, lse, (hipStream_t)stream);
}
extern "C" size_t ff_attention_bwd_workspace_bytes(const ff_attn_desc* d) { return ff::attention_bwd_workspace(*d); }
extern "C" int ff_attention_bwd(const ff_attn_desc* d, const void* Q, const void* K, const void* V, const int* text_time, const void* O,
                                const void* dO, const float* lse, void* dQ, void* dK, void* dV, void* workspace, size_t workspace_bytes,
                                ff_stream_t stream) {
    return ff::attention_bwd(*d, Q, K, V, text_time, O, dO, lse, dQ, dK, dV, workspace, workspace_bytes, (hipStream_t)stream);
}
