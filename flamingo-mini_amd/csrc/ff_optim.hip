// Fused multi-tensor AdamW for the trainable parameters (resampler, gated xattn blocks, token embedding): SURVEY.md 8f2.
// One pass over param / grad / exp_avg / exp_avg_sq per step, fp32 math, 16-byte vector streams, up to 32 tensors per launch
// (the table travels in the kernel argument).  Semantics = torch.optim.AdamW (decoupled weight decay, bias correction):
//     p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Reference training recipe: `--optim adamw_torch`, lr 1e-4 (training/train.sh:10-13).  HBM-bound: 7 streams per element.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ff_common.h"
#include "ff_internal.h"

namespace ff {

constexpr int kAdamTensors = 32;
constexpr int kAdamChunk = 256 * 128;  // elements per workgroup (16 sweeps of 8-element vectors: the per-workgroup table lookup is amortised)

struct AdamTable {
    void* p[kAdamTensors];
    const void* g[kAdamTensors];
    void* m[kAdamTensors];
    void* v[kAdamTensors];
    float* w[kAdamTensors];  // fp32 master copies of the parameters (mixed-precision mode), else unused
    long long n[kAdamTensors];
    int block_start[kAdamTensors + 1];
    int count;
    float lr, beta1, beta2, eps, decay, bc1, bc2_sqrt, grad_scale;
    const float* step_dev;   // capturable mode: the step count lives on the device (HIP-graph replays cannot change kernel arguments)
    const float* lr_dev;     // ... and so does the learning rate, when a scheduler is to stay effective under replay
};

// T: storage type of the parameters' compute copy and of the gradients; ST: storage type of the two moments; MASTER: the update
// is applied to an fp32 master copy (t.w) and the compute copy is its rounding - what `--fp16` / bf16 autocast training keeps
// (training/train.sh:24), so that steps far below bf16 resolution of a weight (lr 1e-4) are not lost.
// MODE 1 (default for the bf16-state kernel; FF_ADAMW_MODE=0 for A/B): gradients and moments, touched once per step, are streamed with
// nontemporal accesses so that they do not evict what the next kernels read (37.52 -> 37.40 ms/step at config B in a same-box A/B)
template <typename T, typename ST, bool MASTER, int VEC, int MODE = 0>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamTable t) {
    int ti = 0;
#pragma unroll 1
    while (ti + 1 < t.count && (int)blockIdx.x >= t.block_start[ti + 1]) ti++;
    const long long n = t.n[ti];
    const long long base = (long long)((int)blockIdx.x - t.block_start[ti]) * kAdamChunk;
    T* p = (T*)t.p[ti];
    const T* g = (const T*)t.g[ti];
    ST* m = (ST*)t.m[ti];
    ST* v = (ST*)t.v[ti];
    float* w = t.w[ti];
    float bc1 = t.bc1, bc2_sqrt = t.bc2_sqrt;
    if (t.step_dev) {
        const float step = *t.step_dev;
        bc1 = 1.f - powf(t.beta1, step);
        bc2_sqrt = sqrtf(1.f - powf(t.beta2, step));
    }
    const float lr = t.lr_dev ? *t.lr_dev : t.lr;
    const float step_size = lr / bc1, keep = 1.f - lr * t.decay;
    bool vec = VEC > 1 && n % VEC == 0 && ((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0;
    if (MASTER) vec = vec && (uintptr_t)w % 16 == 0;
    auto ldv = [&](auto* q, long long i, float (&o)[VEC], auto stream) {      // VEC consecutive elements of any storage type as floats
        typedef std::remove_cv_t<std::remove_pointer_t<decltype(q)>> Q;
        constexpr int QN = Vec<Q>::N;
#pragma unroll
        for (int c = 0; c < VEC / QN; c++) {
            float part[QN];
            if constexpr (decltype(stream)::value && MODE >= 1) Vec<Q>::load_nt(q + i + c * QN, part);
            else Vec<Q>::load(q + i + c * QN, part);
#pragma unroll
            for (int e = 0; e < QN; e++) o[c * QN + e] = part[e];
        }
    };
    auto stv = [&](auto* q, long long i, const float (&o)[VEC], auto stream) {
        typedef std::remove_pointer_t<decltype(q)> Q;
        constexpr int QN = Vec<Q>::N;
#pragma unroll
        for (int c = 0; c < VEC / QN; c++) {
            float part[QN];
#pragma unroll
            for (int e = 0; e < QN; e++) part[e] = o[c * QN + e];
            if constexpr (decltype(stream)::value && MODE >= 1) Vec<Q>::store_nt(q + i + c * QN, part);
            else Vec<Q>::store(q + i + c * QN, part);
        }
    };
    const long long end = min(n, base + (long long)kAdamChunk);
    auto update = [&](float (&pf)[VEC], const float (&gf)[VEC], float (&mf)[VEC], float (&vf)[VEC]) {
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const float gr = gf[e] * t.grad_scale;
            pf[e] *= keep;
            mf[e] = t.beta1 * mf[e] + (1.f - t.beta1) * gr;
            vf[e] = t.beta2 * vf[e] + (1.f - t.beta2) * gr * gr;
            pf[e] -= step_size * mf[e] / (sqrtf(vf[e]) / bc2_sqrt + t.eps);
        }
    };
    if (vec) {
        constexpr std::true_type S{};
        constexpr std::false_type K{};
        auto store = [&](long long i, const float (&pf)[VEC], const float (&mf)[VEC], const float (&vf)[VEC]) {
            stv(p, i, pf, K); stv(m, i, mf, S); stv(v, i, vf, S);      // the updated parameters are what the next forward reads: kept cacheable
            if (MASTER) stv(w, i, pf, S);
        };
        long long i = base + (long long)threadIdx.x * VEC;
        // two pieces per thread and pass: eight 16-byte loads are in flight before the first one is needed (the tensors may alias as far as
        // the compiler knows, so it would not hoist the second piece's loads above the first piece's stores by itself)
        for (; i + 256 * VEC < end; i += 512 * VEC) {
            const long long j = i + 256 * VEC;
            float p0[VEC], g0[VEC], m0[VEC], v0[VEC], p1[VEC], g1[VEC], m1[VEC], v1[VEC];
            if (MASTER) { ldv(w, i, p0, S); ldv(w, j, p1, S); } else { ldv(p, i, p0, K); ldv(p, j, p1, K); }
            ldv(g, i, g0, S); ldv(g, j, g1, S);
            ldv(m, i, m0, S); ldv(m, j, m1, S);
            ldv(v, i, v0, S); ldv(v, j, v1, S);
            update(p0, g0, m0, v0);
            update(p1, g1, m1, v1);
            store(i, p0, m0, v0);
            store(j, p1, m1, v1);
        }
        if (i < end) {
            float pf[VEC], gf[VEC], mf[VEC], vf[VEC];
            if (MASTER) ldv(w, i, pf, S); else ldv(p, i, pf, K);
            ldv(g, i, gf, S); ldv(m, i, mf, S); ldv(v, i, vf, S);
            update(pf, gf, mf, vf);
            store(i, pf, mf, vf);
        }
        return;
    }
    for (long long i = base + (long long)threadIdx.x * VEC; i < end; i += 256 * VEC) {
        float pf[VEC], gf[VEC], mf[VEC], vf[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const bool ok = i + e < n;
            pf[e] = ok ? (MASTER ? w[i + e] : to_f32(p[i + e])) : 0.f; gf[e] = ok ? to_f32(g[i + e]) : 0.f;
            mf[e] = ok ? to_f32(m[i + e]) : 0.f; vf[e] = ok ? to_f32(v[i + e]) : 0.f;
        }
        update(pf, gf, mf, vf);
#pragma unroll
        for (int e = 0; e < VEC; e++)
            if (i + e < n) {
                p[i + e] = from_f32<T>(pf[e]); m[i + e] = from_f32<ST>(mf[e]); v[i + e] = from_f32<ST>(vf[e]);
                if (MASTER) w[i + e] = pf[e];
            }
    }
}

static int adamw_launch(const ff_adamw_desc* d, int state_dtype, void* const* params, const void* const* grads, void* const* exp_avg,
                        void* const* exp_avg_sq, float* const* master, const float* lr_dev, const long long* numels, hipStream_t stream) {
    FF_CHECK(d && params && grads && exp_avg && exp_avg_sq && numels, FF_ERR_SHAPE, "ff_adamw_step: null argument");
    FF_CHECK(d->dtype == FF_DTYPE_F32 || d->dtype == FF_DTYPE_BF16, FF_ERR_UNSUPPORTED, "ff_adamw_step: dtype %d", d->dtype);
    FF_CHECK(state_dtype == d->dtype || state_dtype == FF_DTYPE_F32, FF_ERR_UNSUPPORTED, "ff_adamw_step: moments must be stored in the parameter dtype or in fp32");
    FF_CHECK(!master || d->dtype == FF_DTYPE_BF16, FF_ERR_UNSUPPORTED, "ff_adamw_step: fp32 master copies go with bf16 parameters");
    FF_CHECK(d->n_tensors >= 0 && (d->step >= 1 || d->step_dev), FF_ERR_SHAPE, "ff_adamw_step: n_tensors=%d step=%d", d->n_tensors, d->step);
    AdamTable t;
    t.lr = d->lr; t.beta1 = d->beta1; t.beta2 = d->beta2; t.eps = d->eps; t.decay = d->weight_decay;
    t.step_dev = d->step_dev;
    t.lr_dev = lr_dev;
    t.bc1 = 1.f - powf(d->beta1, (float)std::max(d->step, 1));
    t.bc2_sqrt = sqrtf(1.f - powf(d->beta2, (float)std::max(d->step, 1)));
    t.grad_scale = d->grad_scale == 0.f ? 1.f : d->grad_scale;
    int i = 0;
    while (i < d->n_tensors) {
        int cnt = 0, blocks = 0;
        while (i < d->n_tensors && cnt < kAdamTensors) {
            if (numels[i] > 0) {
                FF_CHECK(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && (!master || master[i]), FF_ERR_SHAPE, "ff_adamw_step: tensor %d has a null pointer", i);
                t.p[cnt] = params[i]; t.g[cnt] = grads[i]; t.m[cnt] = exp_avg[i]; t.v[cnt] = exp_avg_sq[i]; t.n[cnt] = numels[i];
                t.w[cnt] = master ? master[i] : nullptr;
                t.block_start[cnt] = blocks;
                blocks += cdiv(numels[i], kAdamChunk);
                cnt++;
            }
            i++;
        }
        if (!cnt) break;
        t.block_start[cnt] = blocks;
        t.count = cnt;
        const dim3 grid(blocks), block(256);
        if (d->dtype == FF_DTYPE_F32) adamw_kernel<float, float, false, 4><<<grid, block, 0, stream>>>(t);
        else if (master) {
            FF_CHECK(state_dtype == FF_DTYPE_F32, FF_ERR_UNSUPPORTED, "ff_adamw_step: fp32 master copies go with fp32 moments");
            adamw_kernel<bf16, float, true, 8, 1><<<grid, block, 0, stream>>>(t);       // (master copy, gradients, moments: streamed nontemporally)
        } else if (state_dtype == FF_DTYPE_F32) adamw_kernel<bf16, float, false, 8, 1><<<grid, block, 0, stream>>>(t);
        else {
            static const int mode = dbg_switch("FF_ADAMW_MODE", 1);
            if (mode >= 1) adamw_kernel<bf16, bf16, false, 8, 1><<<grid, block, 0, stream>>>(t);
            else adamw_kernel<bf16, bf16, false, 8><<<grid, block, 0, stream>>>(t);
        }
        FF_TRY(check_launch("adamw"));
    }
    return FF_OK;
}

}  // namespace ff

extern "C" int ff_adamw_step(const ff_adamw_desc* d, void* const* params, const void* const* grads, void* const* exp_avg,
                             void* const* exp_avg_sq, const long long* numels, ff_stream_t stream) {
    return ff::adamw_launch(d, d ? d->dtype : 0, params, grads, exp_avg, exp_avg_sq, nullptr, nullptr, numels, (hipStream_t)stream);
}
extern "C" int ff_adamw_step_mixed(const ff_adamw_desc* d, int state_dtype, void* const* params, const void* const* grads, void* const* exp_avg,
                                   void* const* exp_avg_sq, float* const* master, const float* lr_dev, const long long* numels,
                                   ff_stream_t stream) {
    return ff::adamw_launch(d, state_dtype, params, grads, exp_avg, exp_avg_sq, master, lr_dev, numels, (hipStream_t)stream);
}
