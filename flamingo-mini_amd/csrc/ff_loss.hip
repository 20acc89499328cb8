// Shifted next-token cross-entropy, forward and backward fused (reference modeling_flamingo.py:288-298:
// shift_logits = logits[..., :-1, :], shift_labels = labels[..., 1:], F.cross_entropy(..., reduction)).
// The reference materialises the shifted copy, a log-softmax and its backward (~10 kernels over batch*seq*vocab elements);
// here the forward reads the logits once (online max / sum-exp per row, one workgroup per token) and the backward reads them
// once more and writes d logits directly in the unshifted layout (last position = 0).  fp32 math on fp32 / bf16 logits.
#include "ff_common.h"
#include "ff_internal.h"

namespace ff {

// Visit the V elements of a row whose start is only element-aligned (vocab 50258: rows start 4 bytes off a 16-byte
// boundary): scalar head up to the next 16-byte boundary, 16-byte vectors for the body, scalar tail.
template <typename T, typename F>
FF_DEV void for_row_vectors(const T* row, int V, F&& f) {   // f(column, value)
    constexpr int N = Vec<T>::N;
    int head = (int)(((16u - (unsigned)((unsigned long long)row & 15u)) & 15u) / sizeof(T));
    head = head < V ? head : V;
    const int nvec = (V - head) / N, tail0 = head + nvec * N;
    if ((int)threadIdx.x < head) f((int)threadIdx.x, to_f32(row[threadIdx.x]));
    for (int v = threadIdx.x; v < nvec; v += 256) {
        float x[N];
        Vec<T>::load(row + head + v * N, x);
#pragma unroll
        for (int e = 0; e < N; e++) f(head + v * N + e, x[e]);
    }
    for (int c = tail0 + threadIdx.x; c < V; c += 256) f(c, to_f32(row[c]));
}

// loss_row[b*(L-1)+i] = logsumexp(logits[b,i,:]) - logits[b,i,labels[b,i+1]]   (0 where the label == ignore_index)
template <typename T>
__global__ __launch_bounds__(256) void shifted_ce_fwd_kernel(int L, int V, const T* __restrict__ logits, const long long* __restrict__ labels,
                                                             long long ignore_index, float* __restrict__ loss_row, float* __restrict__ lse) {
    __shared__ float sm[4], ss[4];
    pin_args(L, V, logits, labels, ignore_index, loss_row, lse);
    const int r = blockIdx.x, b = r / (L - 1), i = r - b * (L - 1);
    const T* row = logits + ((long long)b * L + i) * V;
    float m = -INFINITY, s = 0.f;
    for_row_vectors(row, V, [&](int, float x) {
        if (x > m) { s = s * __expf(m - x) + 1.f; m = x; }
        else s += __expf(x - m);
    });
    // combine (m, s) pairs: wave, then block
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
        m = mn;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = m; ss[w] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), S = 0.f;
        for (int k = 0; k < 4; k++) S += ss[k] * __expf(sm[k] - M);
        const float l = M + logf(S);
        lse[r] = l;
        const long long tgt = labels[(long long)b * L + i + 1];
        // a label outside [0, V) that is not ignore_index (e.g. the <EOC> id with an un-resized embedding) is an error: torch's
        // cross_entropy asserts on the device; here the row's loss becomes NaN (no out-of-bounds read), which no reduction can hide
        loss_row[r] = tgt == ignore_index ? 0.f : ((tgt < 0 || tgt >= V) ? __builtin_nanf("") : l - to_f32(row[tgt]));
    }
}

// dlogits[b,i,:] = (softmax(logits[b,i,:]) - onehot(label)) * g[b*(L-1)+i]   for i < L-1;   dlogits[b,L-1,:] = 0
template <typename T>
__global__ __launch_bounds__(256) void shifted_ce_bwd_kernel(int L, int V, const T* __restrict__ logits, const long long* __restrict__ labels,
                                                             long long ignore_index, const float* __restrict__ lse, const float* __restrict__ g,
                                                             T* __restrict__ dlogits) {
    pin_args(L, V, logits, labels, ignore_index, lse, g, dlogits);
    const int b = blockIdx.x / L, i = blockIdx.x - b * L;
    const long long off = ((long long)b * L + i) * V;
    constexpr int N = Vec<T>::N;
    const T* row = logits + off;
    T* drow = dlogits + off;                                   // same element offset => same alignment phase as `row`
    const bool last = i == L - 1;
    const int r = b * (L - 1) + i;
    const long long tgt = last ? -1 : labels[(long long)b * L + i + 1];
    const float gr = (last || tgt == ignore_index) ? 0.f : ((tgt < 0 || tgt >= V) ? __builtin_nanf("") : g[r]), l = last ? 0.f : lse[r];   // bad label: NaN row
    auto dval = [&](int c, float x) { return gr == 0.f ? 0.f : (__expf(x - l) - (c == tgt ? 1.f : 0.f)) * gr; };
    int head = (int)(((16u - (unsigned)((unsigned long long)row & 15u)) & 15u) / sizeof(T));
    head = head < V ? head : V;
    const int nvec = (V - head) / N, tail0 = head + nvec * N;
    if ((int)threadIdx.x < head) drow[threadIdx.x] = from_f32<T>(dval(threadIdx.x, to_f32(row[threadIdx.x])));
    for (int v = threadIdx.x; v < nvec; v += 256) {
        const int c0 = head + v * N;
        float x[N];
        Vec<T>::load(row + c0, x);
#pragma unroll
        for (int e = 0; e < N; e++) x[e] = dval(c0 + e, x[e]);
        Vec<T>::store(drow + c0, x);
    }
    for (int c = tail0 + threadIdx.x; c < V; c += 256) drow[c] = from_f32<T>(dval(c, to_f32(row[c])));
}

}  // namespace ff

extern "C" int ff_shifted_ce_fwd(int dtype, int batch, int seq, int vocab, const void* logits, const long long* labels, long long ignore_index,
                                 float* loss_row, float* lse, ff_stream_t stream) {
    using namespace ff;
    FF_CHECK(batch > 0 && seq > 1 && vocab > 0 && logits && labels && loss_row && lse, FF_ERR_SHAPE, "ff_shifted_ce_fwd: bad arguments");
    const dim3 grid(batch * (seq - 1));
    if (dtype == FF_DTYPE_BF16) shifted_ce_fwd_kernel<bf16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(seq, vocab, (const bf16*)logits, labels, ignore_index, loss_row, lse);
    else if (dtype == FF_DTYPE_F32) shifted_ce_fwd_kernel<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>(seq, vocab, (const float*)logits, labels, ignore_index, loss_row, lse);
    else FF_CHECK(false, FF_ERR_UNSUPPORTED, "ff_shifted_ce_fwd: dtype %d", dtype);
    return check_launch("shifted_ce_fwd");
}

extern "C" int ff_shifted_ce_bwd(int dtype, int batch, int seq, int vocab, const void* logits, const long long* labels, long long ignore_index,
                                 const float* lse, const float* grad_row, void* dlogits, ff_stream_t stream) {
    using namespace ff;
    FF_CHECK(batch > 0 && seq > 1 && vocab > 0 && logits && labels && lse && grad_row && dlogits, FF_ERR_SHAPE, "ff_shifted_ce_bwd: bad arguments");
    FF_CHECK((((unsigned long long)logits ^ (unsigned long long)dlogits) & 15u) == 0, FF_ERR_UNSUPPORTED, "ff_shifted_ce_bwd: logits and dlogits must share their 16-byte alignment phase");
    const dim3 grid(batch * seq);
    if (dtype == FF_DTYPE_BF16) shifted_ce_bwd_kernel<bf16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(seq, vocab, (const bf16*)logits, labels, ignore_index, lse, grad_row, (bf16*)dlogits);
    else if (dtype == FF_DTYPE_F32) shifted_ce_bwd_kernel<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>(seq, vocab, (const float*)logits, labels, ignore_index, lse, grad_row, (float*)dlogits);
    else FF_CHECK(false, FF_ERR_UNSUPPORTED, "ff_shifted_ce_bwd: dtype %d", dtype);
    return check_launch("shifted_ce_bwd");
}
