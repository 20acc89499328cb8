// Flamingo fusion path for MI355X (gfx950 / CDNA4) — shared device helpers.
// Wave = 64 lanes, MFMA 16x16 tiles: bf16 via v_mfma_f32_16x16x32_bf16, exact fp32 via v_mfma_f32_16x16x4_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flamingo_fusion.h"

namespace ff {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define FF_DEV __device__ __forceinline__
#define FF_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int kWave = 64;

// ---- error plumbing (host) ---------------------------------------------------------------
void set_error(const char* fmt, ...);
#define FF_CHECK(cond, code, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            ::ff::set_error(__VA_ARGS__);         \
            return (code);                        \
        }                                         \
    } while (0)
#define FF_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != FF_OK) return _rc; \
    } while (0)
int check_launch(const char* what);

// A/B switches of DEVELOPMENT builds (build.py --debug: -DFF_DEBUG, libflamingo_fusion_debug.so): read once from the environment.
// The shipped library has none - dbg_switch() is the constant default there, so no environment variable can change what it computes
// or how (no global mutable state).
#ifdef FF_DEBUG
int dbg_switch(const char* name, int dflt);
#else
constexpr int dbg_switch(const char*, int dflt) { return dflt; }
#endif

// ---- scalar conversions ------------------------------------------------------------------
FF_DEV float to_f32(float v) { return v; }
FF_DEV float to_f32(bf16 v) { return (float)v; }
template <typename T> FF_DEV T from_f32(float v);
template <> FF_DEV float from_f32<float>(float v) { return v; }
template <> FF_DEV bf16 from_f32<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32 (RNE)

// ---- 16-byte vector I/O of N consecutive elements as floats ------------------------------
// Vec<T>::N elements per 16 bytes: 4 floats or 8 bf16.
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef f32x4 raw;
    static FF_DEV void load(const float* p, float (&o)[4]) {
        f32x4 v = *(const f32x4*)p;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    static FF_DEV void store(float* p, const float (&o)[4]) {
        f32x4 v = {o[0], o[1], o[2], o[3]};
        *(f32x4*)p = v;
    }
    // streaming variants (data touched once per step: keep it out of the way of what the next kernels want cached)
    static FF_DEV void load_nt(const float* p, float (&o)[4]) {
        f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    static FF_DEV void store_nt(float* p, const float (&o)[4]) {
        f32x4 v = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(v, (f32x4*)p);
    }
};
template <> struct Vec<bf16> {
    static constexpr int N = 8;
    typedef bf16x8 raw;
    static FF_DEV void load(const bf16* p, float (&o)[8]) {
        bf16x8 v = *(const bf16x8*)p;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = (float)v[i];
    }
    static FF_DEV void store(bf16* p, const float (&o)[8]) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (bf16)o[i];
        *(bf16x8*)p = v;
    }
    static FF_DEV void load_nt(const bf16* p, float (&o)[8]) {
        bf16x8 v = __builtin_nontemporal_load((const bf16x8*)p);
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = (float)v[i];
    }
    static FF_DEV void store_nt(bf16* p, const float (&o)[8]) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (bf16)o[i];
        __builtin_nontemporal_store(v, (bf16x8*)p);
    }
};

// ---- wave / block reductions -------------------------------------------------------------
FF_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
FF_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over a block of NW waves; `red` is NW floats of LDS. Result broadcast to all threads.
template <int NW> FF_DEV float block_sum(float v, float* red) {
    v = wave_sum(v);
    if (NW == 1) return v;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) s += red[i];
    return s;
}

// ---- kernel-argument fetch ------------------------------------------------------------------
// Kernel arguments sit in HBM: a scalar load that misses costs ~0.7 us, and the compiler fetches arguments lazily, where they
// are first needed - typically 3-5 *dependent* groups before the first vector load of a kernel that itself runs 5-30 us
// (tools/gemm_timeline.py).  fetch_args copies a small argument struct with one batch of scalar loads and pins every dword in
// an SGPR; pin_args does the same for pointer / scalar parameters.
template <typename P> FF_DEV void pin1(P& p) { asm volatile("" : "+s"(p)); }
template <typename... P> FF_DEV void pin_args(P&... p) { (pin1(p), ...); }
template <typename T> FF_DEV T fetch_args(const T& a) {
    static_assert(sizeof(T) % 4 == 0 && sizeof(T) <= 320, "argument struct too large to pin");
    constexpr int N = sizeof(T) / 4;
    unsigned w[N];
    __builtin_memcpy(w, &a, sizeof(T));
#pragma unroll
    for (int i = 0; i < N; i++) pin1(w[i]);
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
}

// ---- activations (flamingo_mini/utils.py:26-30) ------------------------------------------
FF_DEV float act_fwd(float h, int act) {
    if (act == FF_ACT_GELU) return 0.5f * h * (1.f + erff(h * 0.70710678118654752440f));
    float r = fmaxf(h, 0.f);
    return act == FF_ACT_SQRELU ? r * r : r;
}
FF_DEV float act_grad(float h, int act) {  // d act / d h
    if (act == FF_ACT_GELU) {
        float cdf = 0.5f * (1.f + erff(h * 0.70710678118654752440f));
        float pdf = 0.39894228040143267794f * __expf(-0.5f * h * h);
        return cdf + h * pdf;
    }
    if (act == FF_ACT_SQRELU) return 2.f * fmaxf(h, 0.f);
    return h > 0.f ? 1.f : 0.f;
}

// The same two functions for epilogues whose result is stored in bfloat16: the exact GELU (erf form, utils.py:26 / nn.GELU()) through
// Abramowitz & Stegun 7.1.26 - erfc(x) = (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p x), |error| <= 1.5e-7, i.e. 2^-14 of a bfloat16
// ulp of any output it matters for - instead of the library erff: ~15 instead of ~45 instructions per element, one exponential shared by the
// cdf tail and the pdf.  Measured on the 1024 x 5120 x 1280 launches (tools/gemm_graph_bench.py, EPI=act / act_bwd): the erff epilogues cost
// 3.8 us (forward) and 4.5 us (data gradient) per launch on top of the same launches with the squared-ReLU epilogue.  fp32 outputs keep erff.
FF_DEV void gelu_cdf_fast(float h, float& cdf, float& e) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(h), 1.f));
    e = __expf(-0.5f * h * h);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = 0.5f * p * t * e;           // 0.5 erfc(|h| / sqrt 2): the smaller of cdf and 1 - cdf, no cancellation in the tail
    cdf = h >= 0.f ? 1.f - q : q;
}
template <typename T> FF_DEV float act_fwd_t(float h, int act) {
    if (sizeof(T) == 2 && act == FF_ACT_GELU) {
        float cdf, e;
        gelu_cdf_fast(h, cdf, e);
        return h * cdf;
    }
    return act_fwd(h, act);
}
template <typename T> FF_DEV float act_grad_t(float h, int act) {
    if (sizeof(T) == 2 && act == FF_ACT_GELU) {
        float cdf, e;
        gelu_cdf_fast(h, cdf, e);
        return fmaf(h * 0.39894228040143267794f, e, cdf);
    }
    return act_grad(h, act);
}

// ---- row addressing ----------------------------------------------------------------------
// Logical row r of a (rows x cols) matrix lives at  base + (r / rows_per_seg) * seg_stride + (r % rows_per_seg) * ld.
// rows_per_seg <= 0 means a plain row-major matrix.  seg_stride == 0 broadcasts one segment to all (latents).
struct RowMap {
    long long ld;
    long long seg_stride;
    int rows_per_seg;
    FF_DEV long long off(int r) const {
        if (rows_per_seg <= 0) return (long long)r * ld;
        int s = r / rows_per_seg;
        return (long long)s * seg_stride + (long long)(r - s * rows_per_seg) * ld;
    }
};
inline RowMap make_rowmap(const ff_rowmap& m) { return RowMap{m.ld, m.seg_stride, m.rows_per_seg}; }
inline RowMap plain_rows(long long ld) { return RowMap{ld, 0, 0}; }

// ---- MFMA wrappers -----------------------------------------------------------------------
// D(16x16) += A(16xK) * B(Kx16).  C/D layout: lane l holds D[(l>>4)*4 + r][l & 15], r = 0..3.
// bf16: K = 32, lane l supplies A[l&15][(l>>4)*8 .. +8) and B[(l>>4)*8 .. +8)[l&15].
// fp32: K = 4,  lane l supplies A[l&15][l>>4] and B[l>>4][l&15].
FF_DEV f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
FF_DEV f32x4 mfma_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS transpose read: each 16-lane group reads a [4 rows][16 cols] bf16 block (lane i supplies the 8-byte address of
// row i>>2, cols (i&3)*4..+4) and lane i receives column i of the block (4 values, one per row).
FF_DEV bf16x4 lds_read_tr16(const bf16* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(FF_LDS_PTR(s16x4, p));
    return __builtin_bit_cast(bf16x4, v);
}
FF_DEV bf16x8 cat4(bf16x4 lo, bf16x4 hi) {
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline size_t dtype_size(int dt) { return dt == FF_DTYPE_BF16 ? 2 : 4; }

}  // namespace ff
