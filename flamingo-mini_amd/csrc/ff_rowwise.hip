// Row-wise (HBM-bound) kernels of the fusion path: LayerNorm forward / backward, column reductions for the
// parameter gradients (d gamma, d beta, d latents, d time_pos_emb), the tanh-gate scalar gradient and the
// media-location cumulative sum.  One wave owns one row; 16-byte vector accesses; fp32 statistics.
#include "ff_common.h"
#include "ff_internal.h"

namespace ff {

#ifdef FF_XA_TIMELINE   // debug build (tools/build_timeline.sh): phase timestamps of ln_bwd_fused_kernel, read with ff_debug_ln_timeline_read
__device__ unsigned long long g_ln_timeline[4096 * 8];
#define FF_LTL(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_ln_timeline[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FF_LTL(i) do { } while (0)
#endif

// VEC consecutive elements as floats (VEC == Vec<T>::N -> one 16-byte access, VEC == 1 -> scalar fallback)
template <typename T, int VEC> FF_DEV void ld(const T* p, float (&o)[VEC]) {
    if constexpr (VEC == 1) o[0] = to_f32(p[0]);
    else Vec<T>::load(p, o);
}
template <typename T, int VEC> FF_DEV void st(T* p, const float (&o)[VEC]) {
    if constexpr (VEC == 1) p[0] = from_f32<T>(o[0]);
    else Vec<T>::store(p, o);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x (+add) - mean) * rstd * gamma + beta       (perceiver_resampler.py:52-53,187;
// gated_cross_attention.py:74; utils.py:46).  Two-pass statistics in fp32 like torch.
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnArgs a_in, const T* __restrict__ x, const T* __restrict__ add,
                                                     const T* __restrict__ gamma, const T* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd) {
    const LnArgs a = fetch_args(a_in);
    pin_args(x, add, gamma, beta, y, mean, rstd);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const T* xr = x + a.x_map.off(row);
    const T* ar = nullptr;
    if (add) ar = add + (long long)((row % a.add_rows_per_seg) / a.add_div) * a.cols;
    const int nchunk = a.cols / VEC;
    float mu, rs;
    if (nchunk <= 4 * 64 && VEC > 1) {
        // the row fits four 16-byte pieces per lane: it is read ONCE and stays in registers through the two statistics passes and the
        // normalisation (the general path below re-reads it from cache twice: two more dependent round trips in a kernel that is one row long)
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = lane + 64 * u;
            if (c < nchunk) {
                ld<T, VEC>(xr + c * VEC, v[u]);
                if (ar) {
                    float t[VEC];
                    ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
                    for (int i = 0; i < VEC; i++) v[u][i] += t[i];
                }
            }
        }
        if (a.stats_given) {
            mu = mean[row];
            rs = rstd[row];
        } else {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (lane + 64 * u < nchunk)
#pragma unroll
                    for (int i = 0; i < VEC; i++) s += v[u][i];
            mu = wave_sum(s) / (float)a.cols;
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (lane + 64 * u < nchunk)
#pragma unroll
                    for (int i = 0; i < VEC; i++) q += (v[u][i] - mu) * (v[u][i] - mu);
            rs = rsqrtf(wave_sum(q) / (float)a.cols + a.eps);
            if (lane == 0 && mean) {
                mean[row] = mu;
                rstd[row] = rs;
            }
        }
        if (!y) return;
        T* yr = y + a.y_map.off(row);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = lane + 64 * u;
            if (c < nchunk) {
                float g[VEC], b[VEC], o[VEC];
                ld<T, VEC>(gamma + c * VEC, g);
                ld<T, VEC>(beta + c * VEC, b);
#pragma unroll
                for (int i = 0; i < VEC; i++) o[i] = (v[u][i] - mu) * rs * g[i] + b[i];
                st<T, VEC>(yr + c * VEC, o);
            }
        }
        return;
    }
    if (a.stats_given) {
        mu = mean[row];
        rs = rstd[row];
    } else {
        float s = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float v[VEC];
            ld<T, VEC>(xr + c * VEC, v);
            if (ar) {
                float t[VEC];
                ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
                for (int i = 0; i < VEC; i++) v[i] += t[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; i++) s += v[i];
        }
        mu = wave_sum(s) / (float)a.cols;
        float q = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float v[VEC];
            ld<T, VEC>(xr + c * VEC, v);
            if (ar) {
                float t[VEC];
                ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
                for (int i = 0; i < VEC; i++) v[i] += t[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; i++) q += (v[i] - mu) * (v[i] - mu);
        }
        rs = rsqrtf(wave_sum(q) / (float)a.cols + a.eps);
        if (lane == 0 && mean) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
    if (!y) return;
    T* yr = y + a.y_map.off(row);
    for (int c = lane; c < nchunk; c += 64) {
        float v[VEC], g[VEC], b[VEC];
        ld<T, VEC>(xr + c * VEC, v);
        if (ar) {
            float t[VEC];
            ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
            for (int i = 0; i < VEC; i++) v[i] += t[i];
        }
        ld<T, VEC>(gamma + c * VEC, g);
        ld<T, VEC>(beta + c * VEC, b);
#pragma unroll
        for (int i = 0; i < VEC; i++) v[i] = (v[i] - mu) * rs * g[i] + b[i];
        st<T, VEC>(yr + c * VEC, v);
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward, data gradient:  dx = dx_residual + rstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)),
// dyh = dy * gamma.
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const LnArgs a_in, const T* __restrict__ dy, const T* __restrict__ x,
                                                        const T* __restrict__ add, const T* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, T* dx,
                                                        const T* dx_res) {
    const LnArgs a = fetch_args(a_in);
    pin_args(dy, x, add, gamma, mean, rstd, dx, dx_res);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const T* xr = x + a.x_map.off(row);
    const T* dyr = dy + a.y_map.off(row);
    const T* ar = nullptr;
    if (add) ar = add + (long long)((row % a.add_rows_per_seg) / a.add_div) * a.cols;
    const float mu = mean[row], rs = rstd[row];
    const int nchunk = a.cols / VEC;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        float v[VEC], d[VEC], g[VEC];
        ld<T, VEC>(xr + c * VEC, v);
        if (ar) {
            float t[VEC];
            ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
            for (int i = 0; i < VEC; i++) v[i] += t[i];
        }
        ld<T, VEC>(dyr + c * VEC, d);
        ld<T, VEC>(gamma + c * VEC, g);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            const float dyh = d[i] * g[i];
            s1 += dyh;
            s2 += dyh * (v[i] - mu) * rs;
        }
    }
    const float m1 = wave_sum(s1) / (float)a.cols, m2 = wave_sum(s2) / (float)a.cols;
    const long long doff = a.dx_map.off(row);
    for (int c = lane; c < nchunk; c += 64) {
        float v[VEC], d[VEC], g[VEC], o[VEC];
        ld<T, VEC>(xr + c * VEC, v);
        if (ar) {
            float t[VEC];
            ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
            for (int i = 0; i < VEC; i++) v[i] += t[i];
        }
        ld<T, VEC>(dyr + c * VEC, d);
        ld<T, VEC>(gamma + c * VEC, g);
#pragma unroll
        for (int i = 0; i < VEC; i++) o[i] = rs * (d[i] * g[i] - m1 - (v[i] - mu) * rs * m2);
        if (dx_res) {
            float q[VEC];
            ld<T, VEC>(dx_res + doff + c * VEC, q);
#pragma unroll
            for (int i = 0; i < VEC; i++) o[i] += q[i];
        }
        st<T, VEC>(dx + doff + c * VEC, o);
    }
}


// ------------------------------------------------------------------------------------------------
// Fused LayerNorm backward: one pass produces dx (+ residual), per-block partial sums of d gamma / d beta, and
// (optionally) the two tanh-gate dot products of a gated block:
//     dot_a = sum(dx_residual .* A)   (d alpha of the branch whose output was added AFTER this LayerNorm's input)
//     dot_b = sum(dx          .* B)   (d alpha of the branch added BEFORE it: dx is this kernel's own output)
// partial[block][2*cols + 2]; ln_bwd_final_kernel reduces over blocks and applies (1 - tanh(alpha)^2).
// A wave owns rows (statistics are lane-local + wave reductions), lanes own column chunks across the wave's rows.
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const LnArgs a_in, const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ add, const T* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, T* dx,
                                                           const T* dx_res, const T* __restrict__ dot_a, const T* __restrict__ dot_b,
                                                           float* __restrict__ partial, int rows_per_block) {
    FF_LTL(0);
    const LnArgs a = fetch_args(a_in);
    pin_args(dy, x, add, gamma, mean, rstd, dx, dx_res, dot_a, dot_b, partial, rows_per_block);
    extern __shared__ __attribute__((aligned(16))) float sacc[];   // [4 waves][2][cols]
    FF_LTL(1);
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchunk = a.cols / VEC;
    float accg[NCH][VEC], accb[NCH][VEC];
#pragma unroll
    for (int i = 0; i < NCH; i++)
#pragma unroll
        for (int e = 0; e < VEC; e++) accg[i][e] = accb[i][e] = 0.f;
    float da = 0.f, db = 0.f;
    const int row_end = min(a.rows, (int)(blockIdx.x + 1) * rows_per_block);
    // dy of a row chunk: one load, or - a.dy_splits > 0 - the sum of that many fp32 slabs a split-K GEMM left behind (plain [rows][cols] each)
    auto load_dy = [&](int row, const T* dyr, int c, float (&d)[VEC]) {
        if (a.dy_splits > 0) {
            const float* pp = (const float*)dy + (long long)row * a.cols + c * VEC;
#pragma unroll
            for (int e = 0; e < VEC; e++) d[e] = 0.f;
            for (int sp = 0; sp < a.dy_splits; sp++) {
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    const f32x4 q = *(const f32x4*)(pp + (long long)sp * a.dy_slab + e);
                    d[e] += q[0]; d[e + 1] += q[1]; d[e + 2] += q[2]; d[e + 3] += q[3];
                }
            }
        } else ld<T, VEC>(dyr + c * VEC, d);
    };
    for (int row = blockIdx.x * rows_per_block + w; row < row_end; row += 4) {
        const T* xr = x + a.x_map.off(row);
        const T* dyr = a.dy_splits > 0 ? dy : dy + a.y_map.off(row);
        const T* ar = add ? add + (long long)((row % a.add_rows_per_seg) / a.add_div) * a.cols : nullptr;
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        constexpr bool KEEP = NCH <= 4;                 // x-hat and dy of the row stay in registers between the two passes
        // ... and so do the operands only the second pass needs (the residual and the two gate-gradient factors): requested together with
        // the first pass's loads, they arrive while the row statistics are being reduced instead of behind a second memory round trip
        constexpr bool HOIST = KEEP && VEC > 1;
        float xk[KEEP ? NCH : 1][VEC], dk[KEEP ? NCH : 1][VEC];
        uint4 rq[HOIST ? NCH : 1], ra[HOIST ? NCH : 1], rb[HOIST ? NCH : 1];
        const long long doff = a.dx_map.off(row);
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float v[VEC], d[VEC], g[VEC];
                ld<T, VEC>(xr + c * VEC, v);
                if (ar) {
                    float t[VEC];
                    ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
                    for (int e = 0; e < VEC; e++) v[e] += t[e];
                }
                load_dy(row, dyr, c, d);
                ld<T, VEC>(gamma + c * VEC, g);
                if constexpr (HOIST) {
                    if (dx_res) rq[i] = *(const uint4*)(dx_res + doff + c * VEC);
                    if (dot_a) ra[i] = *(const uint4*)(dot_a + doff + c * VEC);
                    if (dot_b) rb[i] = *(const uint4*)(dot_b + doff + c * VEC);
                }
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    const float dyh = d[e] * g[e], xh = (v[e] - mu) * rs;
                    s1 += dyh;
                    s2 += dyh * xh;
                    if (KEEP) { xk[i][e] = xh; dk[i][e] = d[e]; }
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)a.cols, m2 = wave_sum(s2) / (float)a.cols;
        auto unpack = [](const uint4& r, float (&o)[VEC]) {
            if constexpr (VEC > 1) {
                typedef typename Vec<T>::raw raw_t;
                const raw_t x = __builtin_bit_cast(raw_t, r);
#pragma unroll
                for (int e = 0; e < VEC; e++) o[e] = to_f32(x[e]);
            }
        };
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float xh[VEC], d[VEC], g[VEC], o[VEC];
                if (KEEP) {
#pragma unroll
                    for (int e = 0; e < VEC; e++) { xh[e] = xk[i][e]; d[e] = dk[i][e]; }
                } else {
                    float v[VEC];
                    ld<T, VEC>(xr + c * VEC, v);
                    if (ar) {
                        float t[VEC];
                        ld<T, VEC>(ar + c * VEC, t);
#pragma unroll
                        for (int e = 0; e < VEC; e++) v[e] += t[e];
                    }
                    load_dy(row, dyr, c, d);
#pragma unroll
                    for (int e = 0; e < VEC; e++) xh[e] = (v[e] - mu) * rs;
                }
                ld<T, VEC>(gamma + c * VEC, g);
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    o[e] = rs * (d[e] * g[e] - m1 - xh[e] * m2);
                    accg[i][e] += d[e] * xh[e];
                    accb[i][e] += d[e];
                }
                if (dx_res) {
                    float q[VEC];
                    if constexpr (HOIST) unpack(rq[i], q);
                    else ld<T, VEC>(dx_res + doff + c * VEC, q);
                    if (dot_a) {
                        float u[VEC];
                        if constexpr (HOIST) unpack(ra[i], u);
                        else ld<T, VEC>(dot_a + doff + c * VEC, u);
#pragma unroll
                        for (int e = 0; e < VEC; e++) da += q[e] * u[e];
                    }
#pragma unroll
                    for (int e = 0; e < VEC; e++) o[e] += q[e];
                }
                if (dx) {
                    float r[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; e++) r[e] = to_f32(from_f32<T>(o[e]));   // the value the next kernel will read
                    if (dot_b) {
                        float u[VEC];
                        if constexpr (HOIST) unpack(rb[i], u);
                        else ld<T, VEC>(dot_b + doff + c * VEC, u);
#pragma unroll
                        for (int e = 0; e < VEC; e++) db += r[e] * u[e];
                    }
                    st<T, VEC>(dx + doff + c * VEC, o);
                }
            }
        }
    }
    FF_LTL(2);
    // cross-wave accumulation of the column partials: every wave parks its registers in its own LDS image [wave][2][cols] with 16-byte
    // writes, one barrier, then the block adds the four images column by column on the way out.  (The first version let the waves take
    // turns on ONE image with scalar read-modify-writes: 6.2 of the kernel's 12 us at 1024 x 1280 - tools/ln_timeline.py.)
    {
        float* mine = sacc + (size_t)w * 2 * a.cols;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    *(f32x4*)(mine + c * VEC + e) = f32x4{accg[i][e], accg[i][e + 1], accg[i][e + 2], accg[i][e + 3]};
                    *(f32x4*)(mine + a.cols + c * VEC + e) = f32x4{accb[i][e], accb[i][e + 1], accb[i][e + 2], accb[i][e + 3]};
                }
            }
        }
    }
    __syncthreads();
    FF_LTL(3);
    const long long P = 2LL * a.cols + 2;
    float* out = partial + (long long)blockIdx.x * P;
    for (int i = threadIdx.x * 4; i < 2 * a.cols; i += 256 * 4) {      // cols % 4 == 0 on this path
        const f32x4 s0 = *(const f32x4*)(sacc + i), s1 = *(const f32x4*)(sacc + 2 * a.cols + i);
        const f32x4 s2 = *(const f32x4*)(sacc + 4 * a.cols + i), s3 = *(const f32x4*)(sacc + 6 * a.cols + i);
        float* o = out + i;                                              // (the partial rows are only 8-byte aligned: P is even)
        o[0] = (s0[0] + s1[0]) + (s2[0] + s3[0]); o[1] = (s0[1] + s1[1]) + (s2[1] + s3[1]);
        o[2] = (s0[2] + s1[2]) + (s2[2] + s3[2]); o[3] = (s0[3] + s1[3]) + (s2[3] + s3[3]);
    }
    da = block_sum<4>(da, red);
    db = block_sum<4>(db, red);
    if (threadIdx.x == 0) { out[2 * a.cols] = da; out[2 * a.cols + 1] = db; }
    FF_LTL(4);
}

// out[idx] = sum over blocks of partial[block][idx]; 16 outputs x 16 block-lanes per workgroup, 8 loads in flight per thread
// (the first version - 64 x 4, 4 in flight - spent 7 us on 16 dependent round trips to read 2.6 MB)
template <typename T>
FF_DEV void ln_final_body(int nblk, int cols, const float* __restrict__ partial, T* dgamma, T* dbeta, const T* alpha_a, T* out_a, const T* alpha_b,
                          T* out_b) {
    __shared__ float red[16][17];
    const int P = 2 * cols + 2;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + tx;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (idx < P) {
        int b = ty;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] += partial[(long long)(b + 16 * u) * P + idx];
        }
        for (; b < nblk; b += 16) v[0] += partial[(long long)b * P + idx];
    }
    red[ty][tx] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    __syncthreads();
    if (ty == 0 && idx < P) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 16; u++) s += red[u][tx];
        const float v = s;
        if (idx < cols) { if (dgamma) dgamma[idx] = from_f32<T>(v); }
        else if (idx < 2 * cols) { if (dbeta) dbeta[idx - cols] = from_f32<T>(v); }
        else if (idx == 2 * cols) {
            if (out_a) { const float t = tanhf(to_f32(alpha_a[0])); out_a[0] = from_f32<T>(v * (1.f - t * t)); }
        } else if (out_b) { const float t = tanhf(to_f32(alpha_b[0])); out_b[0] = from_f32<T>(v * (1.f - t * t)); }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_final_kernel(int nblk, int cols, const float* __restrict__ partial, T* dgamma, T* dbeta,
                                                           const T* alpha_a, T* out_a, const T* alpha_b, T* out_b) {
    pin_args(nblk, cols, partial, dgamma, dbeta, alpha_a, out_a, alpha_b, out_b);
    ln_final_body<T>(nblk, cols, partial, dgamma, dbeta, alpha_a, out_a, alpha_b, out_b);
}
// the same reduction for up to kLnFinishMax postponed LayerNorm backwards in one launch: blockIdx.y = which one
struct LnFinishTable {
    LnPending s[kLnFinishMax];
};
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_final_multi_kernel(const LnFinishTable t) {
    const LnPending& e = t.s[blockIdx.y];
    if ((int)blockIdx.x * 16 >= 2 * e.cols + 2) return;      // (the grid is sized for the widest set)
    ln_final_body<T>(e.nblk, e.cols, e.partial, (T*)e.dgamma, (T*)e.dbeta, (const T*)e.alpha_a, (T*)e.out_a, (const T*)e.alpha_b, (T*)e.out_b);
}

// ------------------------------------------------------------------------------------------------
// Column reductions.  Block = 64 column-threads (VEC columns each) x 4 row-lanes.
//   MODE 0: out[g][c]  = sum_{rows of group g} x[r][c]
//   MODE 1: out0[c] = sum_r dy[r][c] * xhat[r][c],  out1[c] = sum_r dy[r][c]        (LayerNorm d gamma, d beta)
// grid = (column blocks, groups, row splits); partial[split][group][slot][cols] fp32, reduced by col_reduce_final.
// ------------------------------------------------------------------------------------------------
struct ColReduceArgs {
    int rows, cols;
    RowMap x_map, y_map;
    int rows_per_batch, rows_per_group, n_batch;
    int add_rows_per_seg, add_div;
    int rows_per_split, nsplit, ngroups;
};

template <typename T, int VEC, int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const ColReduceArgs a_in, const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ add, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ partial) {
    const ColReduceArgs a = fetch_args(a_in);
    pin_args(x, dy, add, mean, rstd, partial);
    constexpr int NS = MODE == 1 ? 2 : 1;
    __shared__ float red[4][NS][64 * VEC];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + tx) * VEC;
    const int g = blockIdx.y, split = blockIdx.z;
    float acc[NS][VEC];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int i = 0; i < VEC; i++) acc[s][i] = 0.f;
    if (col < a.cols) {
        const int n_g = a.n_batch * a.rows_per_group;
        const int i_end = min(n_g, (split + 1) * a.rows_per_split);
        for (int i = split * a.rows_per_split + ty; i < i_end; i += 4) {
            const int bi = i / a.rows_per_group, j = i - bi * a.rows_per_group;
            const int row = bi * a.rows_per_batch + g * a.rows_per_group + j;
            if (row >= a.rows) continue;
            float v[VEC];
            ld<T, VEC>(x + a.x_map.off(row) + col, v);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < VEC; e++) acc[0][e] += v[e];
            } else {
                if (add) {
                    float t[VEC];
                    ld<T, VEC>(add + (long long)((row % a.add_rows_per_seg) / a.add_div) * a.cols + col, t);
#pragma unroll
                    for (int e = 0; e < VEC; e++) v[e] += t[e];
                }
                float d[VEC];
                ld<T, VEC>(dy + a.y_map.off(row) + col, d);
                const float mu = mean[row], rs = rstd[row];
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    acc[0][e] += d[e] * (v[e] - mu) * rs;
                    acc[1][e] += d[e];
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int e = 0; e < VEC; e++) red[ty][s][tx * VEC + e] = acc[s][e];
    __syncthreads();
    if (ty == 0 && col < a.cols) {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                const float v = red[0][s][tx * VEC + e] + red[1][s][tx * VEC + e] + red[2][s][tx * VEC + e] + red[3][s][tx * VEC + e];
                partial[(((long long)split * a.ngroups + g) * NS + s) * a.cols + col + e] = v;
            }
    }
}

// out_s[g][c] = sum_split partial[split][g][s][c]
template <typename T>
__global__ __launch_bounds__(256) void col_reduce_final_kernel(int nsplit, int ngroups, int nslots, int cols,
                                                               const float* __restrict__ partial, T* out0, T* out1) {
    pin_args(nsplit, ngroups, nslots, cols, partial, out0, out1);
    const long long total = (long long)ngroups * nslots * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;     // independent chains: keep 4+ loads in flight per thread
        int s = 0;
        for (; s + 4 <= nsplit; s += 4) {
            v0 += partial[(long long)s * total + idx];
            v1 += partial[(long long)(s + 1) * total + idx];
            v2 += partial[(long long)(s + 2) * total + idx];
            v3 += partial[(long long)(s + 3) * total + idx];
        }
        for (; s < nsplit; s++) v0 += partial[(long long)s * total + idx];
        const float v = (v0 + v1) + (v2 + v3);
        const int c = (int)(idx % cols);
        const long long gs = idx / cols;
        const int slot = (int)(gs % nslots), g = (int)(gs / nslots);
        T* out = slot == 0 ? out0 : out1;
        out[(long long)g * cols + c] = from_f32<T>(v);
    }
}

// ------------------------------------------------------------------------------------------------
// d alpha = (1 - tanh(alpha)^2) * sum(a .* b)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dot_partial_kernel(long long n, const T* __restrict__ a, const T* __restrict__ b,
                                                          float* __restrict__ partial) {
    pin_args(n, a, b, partial);
    __shared__ float red[4];
    float s = 0.f;
    const long long nchunk = n / VEC;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += (long long)gridDim.x * 256) {
        float u[VEC], v[VEC];
        ld<T, VEC>(a + c * VEC, u);
        ld<T, VEC>(b + c * VEC, v);
#pragma unroll
        for (int i = 0; i < VEC; i++) s += u[i] * v[i];
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void gate_grad_final_kernel(int nblk, const float* __restrict__ partial, const T* alpha, T* dalpha) {
    pin_args(nblk, partial, alpha, dalpha);
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const float t = tanhf(to_f32(alpha[0]));
        dalpha[0] = from_f32<T>(s * (1.f - t * t));
    }
}

// text_time[b][i] = inclusive prefix sum of media_locations (gated_cross_attention.py:97)
template <typename I> __global__ void text_time_kernel(int batch, int n, const I* __restrict__ ml, int* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    int acc = 0;
    for (int i = 0; i < n; i++) {
        acc += (int)ml[(long long)b * n + i];
        out[(long long)b * n + i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int gate_grad(int dtype, int rows, int cols, const void* a, const void* b, const void* alpha, void* dalpha, void* ws,
              size_t ws_bytes, hipStream_t st_);

static bool vec_ok(int dtype, int cols, std::initializer_list<RowMap> maps, std::initializer_list<const void*> ptrs) {
    const int n = dtype == FF_DTYPE_BF16 ? 8 : 4;
    if (cols % n) return false;
    for (const RowMap& m : maps)
        if (m.ld % n || (m.rows_per_seg > 0 && m.seg_stride % n)) return false;
    for (const void* p : ptrs)
        if (p && (uintptr_t)p % 16) return false;
    return true;
}

#define FF_DISPATCH_T_VEC(dtype, vec, ...)                 \
    do {                                                    \
        if ((dtype) == FF_DTYPE_BF16) {                     \
            typedef bf16 T;                                 \
            if (vec) { constexpr int VEC = 8; __VA_ARGS__; }       \
            else { constexpr int VEC = 1; __VA_ARGS__; }           \
        } else {                                            \
            typedef float T;                                \
            if (vec) { constexpr int VEC = 4; __VA_ARGS__; }       \
            else { constexpr int VEC = 1; __VA_ARGS__; }           \
        }                                                   \
    } while (0)

int layernorm_fwd(const LnArgs& a, const void* x, const void* add, const void* gamma, const void* beta, void* y, float* mean,
                  float* rstd, hipStream_t st_) {
    FF_CHECK(a.rows > 0 && a.cols > 0 && x, FF_ERR_SHAPE, "layernorm_fwd: bad shape rows=%d cols=%d", a.rows, a.cols);
    FF_CHECK(!a.stats_given || (mean && rstd), FF_ERR_SHAPE, "layernorm_fwd: stats_given without statistics");
    FF_CHECK(!y || (gamma && beta), FF_ERR_SHAPE, "layernorm_fwd: gamma/beta missing");
    FF_CHECK(!add || (a.add_rows_per_seg > 0 && a.add_div > 0), FF_ERR_SHAPE, "layernorm_fwd: addend needs add_rows_per_seg/add_div");
    const bool v = vec_ok(a.dtype, a.cols, {a.x_map, a.y_map}, {x, add, gamma, beta, y});
    const int grid = cdiv(a.rows, 4);
    FF_DISPATCH_T_VEC(a.dtype, v, ln_fwd_kernel<T, VEC><<<dim3(grid), dim3(256), 0, st_>>>(a, (const T*)x, (const T*)add, (const T*)gamma, (const T*)beta, (T*)y, mean, rstd));
    return check_launch("ln_fwd");
}

static void split_plan(int n_rows_in_group, int col_blocks, int ngroups, int& rows_per_split, int& nsplit) {
    int target = std::max(1, 512 / std::max(1, col_blocks * ngroups));
    nsplit = std::max(1, std::min(std::min(target, 32), cdiv(n_rows_in_group, 32)));
    rows_per_split = cdiv(n_rows_in_group, nsplit);
    nsplit = cdiv(n_rows_in_group, rows_per_split);
}

static size_t col_reduce_ws(int n_rows_in_group, int cols, int ngroups, int nslots) {
    // upper bound independent of the vector width: col_blocks >= 1
    int rps, ns;
    split_plan(n_rows_in_group, 1, ngroups, rps, ns);
    return (size_t)ns * ngroups * nslots * cols * sizeof(float);
}


// fused path: supported when the row fits NCH <= 8 chunks of 64 lanes and the partial image fits LDS
static int ln_fused_blocks(int rows, int& rows_per_block) {
    rows_per_block = std::max(4, (cdiv(rows, 256) + 3) / 4 * 4);
    return cdiv(rows, rows_per_block);
}
static bool ln_fused_ok(int dtype, int cols, bool vec) {
    if (!vec) return false;
    const int n = dtype == FF_DTYPE_BF16 ? 8 : 4;
    return cdiv(cols / n, 64) <= 8 && (size_t)8 * cols * sizeof(float) <= 150 * 1024;      // four [2][cols] fp32 images in LDS
}
size_t layernorm_bwd_workspace(int rows, int cols) {
    int rpb;
    const size_t fused = (size_t)ln_fused_blocks(rows, rpb) * (2 * (size_t)cols + 2) * sizeof(float);
    return std::max(fused, col_reduce_ws(rows, cols, 1, 2));
}

bool layernorm_bwd_deferrable(int dtype, int cols) { return cols % (dtype == FF_DTYPE_BF16 ? 8 : 4) == 0 && ln_fused_ok(dtype, cols, true); }
int layernorm_bwd_partial_blocks(int rows) {
    int rpb;
    return ln_fused_blocks(rows, rpb);
}
size_t layernorm_bwd_partial_bytes(int rows, int cols) { return (size_t)layernorm_bwd_partial_blocks(rows) * (2 * (size_t)cols + 2) * sizeof(float); }

template <typename T, int VEC>
static int launch_ln_fused(const LnArgs& a, const void* dy, const void* x, const void* add, const void* gamma, const float* mean,
                           const float* rstd, void* dx, const void* dx_res, const LnDots& dots, void* dgamma, void* dbeta, float* partial,
                           hipStream_t st_, LnPending* pending) {
    int rpb;
    const int nblk = ln_fused_blocks(a.rows, rpb);
    const int nch = cdiv(a.cols / VEC, 64);
    const size_t lds = (size_t)8 * a.cols * sizeof(float);
#define FF_LN_LAUNCH(NCH)                                                                                                                 \
    do {                                                                                                                                  \
        if (lds > 48 * 1024) {                                                                                                            \
            static bool attr[64] = {};                                                                                                    \
            int dev = 0;                                                                                                                  \
            (void)hipGetDevice(&dev);                                                                                                     \
            if (dev < 0 || dev >= 64 || !attr[dev]) {                                                                                     \
                hipFuncSetAttribute((const void*)ln_bwd_fused_kernel<T, VEC, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
                if (dev >= 0 && dev < 64) attr[dev] = true;                                                                               \
            }                                                                                                                             \
        }                                                                                                                                 \
        ln_bwd_fused_kernel<T, VEC, NCH><<<dim3(nblk), dim3(256), lds, st_>>>(a, (const T*)dy, (const T*)x, (const T*)add, (const T*)gamma, \
                                                                              mean, rstd, (T*)dx, (const T*)dx_res, (const T*)dots.a,     \
                                                                              (const T*)dots.b, partial, rpb);                            \
    } while (0)
    if (nch <= 2) FF_LN_LAUNCH(2);
    else if (nch <= 4) FF_LN_LAUNCH(4);
    else FF_LN_LAUNCH(8);
#undef FF_LN_LAUNCH
    FF_TRY(check_launch("ln_bwd_fused"));
    if (pending) {      // the cross-workgroup reduction is left to layernorm_bwd_finish
        pending->partial = partial; pending->nblk = nblk; pending->cols = a.cols; pending->dgamma = dgamma; pending->dbeta = dbeta;
        pending->alpha_a = dots.alpha_a; pending->out_a = dots.out_a; pending->alpha_b = dots.alpha_b; pending->out_b = dots.out_b;
        return FF_OK;
    }
    const int P = 2 * a.cols + 2;
    ln_bwd_final_kernel<T><<<dim3(cdiv(P, 16)), dim3(256), 0, st_>>>(nblk, a.cols, partial, (T*)dgamma, (T*)dbeta, (const T*)dots.alpha_a,
                                                                     (T*)dots.out_a, (const T*)dots.alpha_b, (T*)dots.out_b);
    return check_launch("ln_bwd_final");
}

int layernorm_bwd(const LnArgs& a, const void* dy, const void* x, const void* add, const void* gamma, const float* mean,
                  const float* rstd, void* dx, const void* dx_residual, void* dgamma, void* dbeta, void* ws, size_t ws_bytes,
                  hipStream_t st_, const LnDots* dots_in, LnPending* pending) {
    FF_CHECK(a.rows > 0 && a.cols > 0 && dy && x && gamma && mean && rstd, FF_ERR_SHAPE, "layernorm_bwd: null/shape");
    if (pending) *pending = LnPending{};
    FF_CHECK(!add || (a.add_rows_per_seg > 0 && a.add_div > 0), FF_ERR_SHAPE, "layernorm_bwd: addend needs add_rows_per_seg/add_div");
    const LnDots dots = dots_in ? *dots_in : LnDots{};
    FF_CHECK(!dots.a || dx_residual, FF_ERR_SHAPE, "layernorm_bwd: dot_a is taken against dx_residual");
    FF_CHECK(!dots.b || dx, FF_ERR_SHAPE, "layernorm_bwd: dot_b is taken against dx");
    const bool v = vec_ok(a.dtype, a.cols, {a.x_map, a.y_map, a.dx_map}, {x, add, gamma, dy, dx, dx_residual, dots.a, dots.b});
    FF_CHECK(a.dy_splits == 0 || (ln_fused_ok(a.dtype, a.cols, v) && (dgamma || dots.a || dots.b) && a.cols % 4 == 0), FF_ERR_UNSUPPORTED,
             "layernorm_bwd: summing split-K slabs needs the one-pass kernel");
    if (ln_fused_ok(a.dtype, a.cols, v) && (dgamma || dots.a || dots.b)) {
        FF_CHECK(!dgamma == !dbeta, FF_ERR_SHAPE, "layernorm_bwd: dgamma and dbeta come together");
        int rpb;
        const size_t need = (size_t)ln_fused_blocks(a.rows, rpb) * (2 * (size_t)a.cols + 2) * sizeof(float);
        FF_CHECK(ws && ws_bytes >= need, FF_ERR_WORKSPACE, "layernorm_bwd workspace: need %zu have %zu", need, ws_bytes);
        if (a.dtype == FF_DTYPE_BF16) return launch_ln_fused<bf16, 8>(a, dy, x, add, gamma, mean, rstd, dx, dx_residual, dots, dgamma, dbeta, (float*)ws, st_, pending);
        return launch_ln_fused<float, 4>(a, dy, x, add, gamma, mean, rstd, dx, dx_residual, dots, dgamma, dbeta, (float*)ws, st_, pending);
    }
    if (dots.a) FF_TRY(gate_grad(a.dtype, a.rows, a.cols, dx_residual, dots.a, dots.alpha_a, dots.out_a, ws, ws_bytes, st_));   // unfused fallback (plain maps)
    if (dx) {
        const int grid = cdiv(a.rows, 4);
        FF_DISPATCH_T_VEC(a.dtype, v, ln_bwd_dx_kernel<T, VEC><<<dim3(grid), dim3(256), 0, st_>>>(a, (const T*)dy, (const T*)x, (const T*)add, (const T*)gamma, mean, rstd, (T*)dx, (const T*)dx_residual));
        FF_TRY(check_launch("ln_bwd_dx"));
    }
    if (dgamma || dbeta) {
        FF_CHECK(dgamma && dbeta, FF_ERR_SHAPE, "layernorm_bwd: dgamma and dbeta come together");
        const int vecn = v ? (a.dtype == FF_DTYPE_BF16 ? 8 : 4) : 1;
        ColReduceArgs c = {};
        c.rows = a.rows; c.cols = a.cols; c.x_map = a.x_map; c.y_map = a.y_map;
        c.rows_per_batch = a.rows; c.rows_per_group = a.rows; c.n_batch = 1; c.ngroups = 1;
        c.add_rows_per_seg = a.add_rows_per_seg; c.add_div = a.add_div;
        const int col_blocks = cdiv(a.cols, 64 * vecn);
        split_plan(a.rows, col_blocks, 1, c.rows_per_split, c.nsplit);
        const size_t need = (size_t)c.nsplit * 2 * a.cols * sizeof(float);
        FF_CHECK(ws && ws_bytes >= need, FF_ERR_WORKSPACE, "layernorm_bwd workspace: need %zu have %zu", need, ws_bytes);
        FF_DISPATCH_T_VEC(a.dtype, v, col_reduce_kernel<T, VEC, 1><<<dim3(col_blocks, 1, c.nsplit), dim3(256), 0, st_>>>(c, (const T*)x, (const T*)dy, (const T*)add, mean, rstd, (float*)ws));
        FF_TRY(check_launch("ln_bwd_param_partial"));
        const int grid = std::min(cdiv(2LL * a.cols, 256), 1024);
        if (a.dtype == FF_DTYPE_BF16)
            hipLaunchKernelGGL(col_reduce_final_kernel<bf16>, dim3(grid), dim3(256), 0, st_, c.nsplit, 1, 2, a.cols, (const float*)ws, (bf16*)dgamma, (bf16*)dbeta);
        else
            hipLaunchKernelGGL(col_reduce_final_kernel<float>, dim3(grid), dim3(256), 0, st_, c.nsplit, 1, 2, a.cols, (const float*)ws, (float*)dgamma, (float*)dbeta);
        FF_TRY(check_launch("ln_bwd_param_final"));
    }
    if (dots.b) FF_TRY(gate_grad(a.dtype, a.rows, a.cols, dx, dots.b, dots.alpha_b, dots.out_b, ws, ws_bytes, st_));
    return FF_OK;
}

int layernorm_bwd_finish(int dtype, const LnPending* sets, int n, hipStream_t st_) {
    for (int i0 = 0; i0 < n; i0 += kLnFinishMax) {
        LnFinishTable t = {};
        int cnt = 0, pmax = 0;
        for (int i = i0; i < n && i < i0 + kLnFinishMax; i++)
            if (sets[i].partial) {
                t.s[cnt++] = sets[i];
                pmax = std::max(pmax, 2 * sets[i].cols + 2);
            }
        if (!cnt) continue;
        const dim3 grid(cdiv(pmax, 16), cnt);
        if (dtype == FF_DTYPE_BF16) ln_bwd_final_multi_kernel<bf16><<<grid, dim3(256), 0, st_>>>(t);
        else ln_bwd_final_multi_kernel<float><<<grid, dim3(256), 0, st_>>>(t);
        FF_TRY(check_launch("ln_bwd_final_multi"));
    }
    return FF_OK;
}

size_t rows_reduce_workspace(int rows, int cols, int rows_per_batch, int rows_per_group) {
    if (rows_per_batch <= 0 || rows_per_group <= 0) return 0;
    const int ngroups = cdiv(rows_per_batch, rows_per_group);
    const int n_batch = cdiv(rows, rows_per_batch);
    return col_reduce_ws(n_batch * rows_per_group, cols, ngroups, 1);
}

int rows_reduce(int dtype, int rows, int cols, RowMap x_map, int rows_per_batch, int rows_per_group, const void* x, void* out,
                void* ws, size_t ws_bytes, hipStream_t st_) {
    FF_CHECK(rows > 0 && cols > 0 && rows_per_batch > 0 && rows_per_group > 0 && x && out, FF_ERR_SHAPE, "rows_reduce: bad arguments");
    FF_CHECK(rows % rows_per_batch == 0 && rows_per_batch % rows_per_group == 0, FF_ERR_SHAPE,
             "rows_reduce: rows=%d rows_per_batch=%d rows_per_group=%d do not nest", rows, rows_per_batch, rows_per_group);
    const bool v = vec_ok(dtype, cols, {x_map}, {x, out});
    const int vecn = v ? (dtype == FF_DTYPE_BF16 ? 8 : 4) : 1;
    ColReduceArgs c = {};
    c.rows = rows; c.cols = cols; c.x_map = x_map; c.y_map = x_map;
    c.rows_per_batch = rows_per_batch; c.rows_per_group = rows_per_group;
    c.n_batch = rows / rows_per_batch; c.ngroups = rows_per_batch / rows_per_group;
    const int col_blocks = cdiv(cols, 64 * vecn);
    split_plan(c.n_batch * rows_per_group, col_blocks, c.ngroups, c.rows_per_split, c.nsplit);
    const size_t need = (size_t)c.nsplit * c.ngroups * cols * sizeof(float);
    FF_CHECK(ws && ws_bytes >= need, FF_ERR_WORKSPACE, "rows_reduce workspace: need %zu have %zu", need, ws_bytes);
    FF_DISPATCH_T_VEC(dtype, v, col_reduce_kernel<T, VEC, 0><<<dim3(col_blocks, c.ngroups, c.nsplit), dim3(256), 0, st_>>>(c, (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)ws));
    FF_TRY(check_launch("rows_reduce_partial"));
    const int grid = std::min(cdiv((long long)c.ngroups * cols, 256), 1024);
    if (dtype == FF_DTYPE_BF16)
        hipLaunchKernelGGL(col_reduce_final_kernel<bf16>, dim3(grid), dim3(256), 0, st_, c.nsplit, c.ngroups, 1, cols, (const float*)ws, (bf16*)out, (bf16*)nullptr);
    else
        hipLaunchKernelGGL(col_reduce_final_kernel<float>, dim3(grid), dim3(256), 0, st_, c.nsplit, c.ngroups, 1, cols, (const float*)ws, (float*)out, (float*)nullptr);
    return check_launch("rows_reduce_final");
}

static int gate_blocks(long long n) { return (int)std::max<long long>(1, std::min<long long>(512, n / 4096)); }
size_t gate_grad_workspace(int rows, int cols) { return (size_t)gate_blocks((long long)rows * cols) * sizeof(float); }

int gate_grad(int dtype, int rows, int cols, const void* a, const void* b, const void* alpha, void* dalpha, void* ws,
              size_t ws_bytes, hipStream_t st_) {
    const long long n = (long long)rows * cols;
    FF_CHECK(n > 0 && a && b && alpha && dalpha, FF_ERR_SHAPE, "gate_grad: bad arguments");
    const int nblk = gate_blocks(n);
    FF_CHECK(ws && ws_bytes >= nblk * sizeof(float), FF_ERR_WORKSPACE, "gate_grad workspace too small");
    const bool v = vec_ok(dtype, cols, {}, {a, b});
    FF_DISPATCH_T_VEC(dtype, v, dot_partial_kernel<T, VEC><<<dim3(nblk), dim3(256), 0, st_>>>(n, (const T*)a, (const T*)b, (float*)ws));
    FF_TRY(check_launch("gate_grad_partial"));
    if (dtype == FF_DTYPE_BF16)
        hipLaunchKernelGGL(gate_grad_final_kernel<bf16>, dim3(1), dim3(256), 0, st_, nblk, (const float*)ws, (const bf16*)alpha, (bf16*)dalpha);
    else
        hipLaunchKernelGGL(gate_grad_final_kernel<float>, dim3(1), dim3(256), 0, st_, nblk, (const float*)ws, (const float*)alpha, (float*)dalpha);
    return check_launch("gate_grad_final");
}

int text_time(int batch, int n_tokens, const void* ml, int elem_bytes, int* out, hipStream_t st_) {
    FF_CHECK(batch > 0 && n_tokens > 0 && ml && out, FF_ERR_SHAPE, "text_time: bad arguments");
    const int grid = cdiv(batch, 64);
    if (elem_bytes == 8) hipLaunchKernelGGL(text_time_kernel<long long>, dim3(grid), dim3(64), 0, st_, batch, n_tokens, (const long long*)ml, out);
    else if (elem_bytes == 4) hipLaunchKernelGGL(text_time_kernel<int>, dim3(grid), dim3(64), 0, st_, batch, n_tokens, (const int*)ml, out);
    else if (elem_bytes == 1) hipLaunchKernelGGL(text_time_kernel<unsigned char>, dim3(grid), dim3(64), 0, st_, batch, n_tokens, (const unsigned char*)ml, out);
    else FF_CHECK(false, FF_ERR_UNSUPPORTED, "text_time: media_locations element size %d", elem_bytes);
    return check_launch("text_time");
}

}  // namespace ff

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static ff::LnArgs to_args(const ff_ln_desc* d) {
    ff::LnArgs a;
    a.dtype = d->dtype; a.rows = d->rows; a.cols = d->cols;
    a.x_map = ff::make_rowmap(d->x_map); a.y_map = ff::make_rowmap(d->y_map); a.dx_map = ff::make_rowmap(d->dx_map);
    a.add_rows_per_seg = d->add_rows_per_seg; a.add_div = d->add_div;
    a.eps = d->eps; a.stats_given = d->stats_given;
    return a;
}
extern "C" int ff_layernorm_fwd(const ff_ln_desc* d, const void* x, const void* add, const void* gamma, const void* beta, void* y,
                                float* mean, float* rstd, ff_stream_t stream) {
    return ff::layernorm_fwd(to_args(d), x, add, gamma, beta, y, mean, rstd, (hipStream_t)stream);
}
extern "C" size_t ff_layernorm_bwd_workspace_bytes(const ff_ln_desc* d) { return ff::layernorm_bwd_workspace(d->rows, d->cols); }
extern "C" int ff_layernorm_bwd(const ff_ln_desc* d, const void* dy, const void* x, const void* add, const void* gamma,
                                const float* mean, const float* rstd, void* dx, const void* dx_residual, void* dgamma, void* dbeta,
                                void* workspace, size_t workspace_bytes, ff_stream_t stream) {
    return ff::layernorm_bwd(to_args(d), dy, x, add, gamma, mean, rstd, dx, dx_residual, dgamma, dbeta, workspace, workspace_bytes,
                             (hipStream_t)stream, nullptr);
}
extern "C" size_t ff_rows_reduce_workspace_bytes(const ff_reduce_desc* d) {
    return ff::rows_reduce_workspace(d->rows, d->cols, d->rows_per_batch, d->rows_per_group);
}
extern "C" int ff_rows_reduce(const ff_reduce_desc* d, const void* x, void* out, void* workspace, size_t workspace_bytes,
                              ff_stream_t stream) {
    return ff::rows_reduce(d->dtype, d->rows, d->cols, ff::make_rowmap(d->x_map), d->rows_per_batch, d->rows_per_group, x, out,
                           workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" size_t ff_gate_grad_workspace_bytes(int rows, int cols) { return ff::gate_grad_workspace(rows, cols); }
extern "C" int ff_gate_grad(int dtype, int rows, int cols, const void* a, const void* b, const void* alpha, void* dalpha,
                            void* workspace, size_t workspace_bytes, ff_stream_t stream) {
    return ff::gate_grad(dtype, rows, cols, a, b, alpha, dalpha, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int ff_text_time(int batch, int n_tokens, const void* media_locations, int elem_bytes, int* text_time, ff_stream_t stream) {
    return ff::text_time(batch, n_tokens, media_locations, elem_bytes, text_time, (hipStream_t)stream);
}

#ifdef FF_XA_TIMELINE
extern "C" int ff_debug_ln_timeline_read(unsigned long long* out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff::g_ln_timeline), sizeof(unsigned long long) * 8 * n_blocks);
}
#endif
