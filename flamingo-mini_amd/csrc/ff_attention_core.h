// Device-side building blocks of the attention kernels (ff_attention.hip) and of the fused projection + attention kernels
// (ff_xattn_fused.hip): own-row fragments, LDS tile staging, the two transposed MFMA products, per-row key ranges.
#pragma once
#include <type_traits>
#include "ff_common.h"

namespace ff {

constexpr int kTile = 64;  // rows per workgroup / per LDS tile
constexpr float kNegBig = -1.0e30f, kPosBig = 1.0e30f;

template <typename T> struct AttnCfg;
template <> struct AttnCfg<bf16> { static constexpr int pad = 8; };   // LDS row padding (elements)
template <> struct AttnCfg<float> { static constexpr int pad = 4; };

// ---- LDS tile layouts ---------------------------------------------------------------------------------
// PadLayout : [row][DH + pad], filled by stage_tile (global -> VGPR -> LDS); what the stand-alone kernels use.
// SwzLayout : [row][64] bf16 with the 16-byte chunks of a row XOR-swizzled by row & 7 - the K-major operand tile of the GEMM
//             family, so it can be filled by LDS-DMA (dma_tile<64, 0>: the swizzle lives on the source address) ahead of its use.
template <typename T, int DH> struct PadLayout {
    static constexpr int LD = DH + AttnCfg<T>::pad;
    static constexpr int tile_elems = kTile * LD;
    static FF_DEV int off(int row, int col) { return row * LD + col; }
};
struct SwzLayout {
    static constexpr int tile_elems = kTile * 64;
    static FF_DEV int off(int row, int col) { return row * 64 + ((((col >> 3) ^ row) & 7) << 3) + (col & 7); }
};

// ---- fragments of the 16 own rows of a wave (B-operand style: lane (c, g) holds row c) -------------
template <typename T, int DH> struct OwnFrag;
template <int DH> struct OwnFrag<bf16, DH> {
    bf16x8 f[DH / 32];
    FF_DEV void load(const bf16* row, int g) {  // row == nullptr -> zeros
#pragma unroll
        for (int ks = 0; ks < DH / 32; ks++) {
            if (row) f[ks] = *(const bf16x8*)(row + ks * 32 + g * 8);
            else
#pragma unroll
                for (int e = 0; e < 8; e++) f[ks][e] = (bf16)0.f;
        }
    }
    template <typename L> FF_DEV void load_tile(const bf16* tile, int row, int g, bool ok) {   // row of an LDS tile in layout L
#pragma unroll
        for (int ks = 0; ks < DH / 32; ks++) {
            if (ok) f[ks] = *(const bf16x8*)(tile + L::off(row, ks * 32 + g * 8));
            else
#pragma unroll
                for (int e = 0; e < 8; e++) f[ks][e] = (bf16)0.f;
        }
    }
    FF_DEV float dot(const OwnFrag& o) const {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < DH / 32; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) s += (float)f[ks][e] * (float)o.f[ks][e];
        return s;
    }
};
template <int DH> struct OwnFrag<float, DH> {
    f32x4 f[DH / 16];
    FF_DEV void load(const float* row, int g) {
#pragma unroll
        for (int s = 0; s < DH / 16; s++) f[s] = row ? *(const f32x4*)(row + s * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    template <typename L> FF_DEV void load_tile(const float* tile, int row, int g, bool ok) {
#pragma unroll
        for (int s = 0; s < DH / 16; s++) f[s] = ok ? *(const f32x4*)(tile + L::off(row, s * 16 + g * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    FF_DEV float dot(const OwnFrag& o) const {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < DH / 16; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) s += f[i][e] * o.f[i][e];
        return s;
    }
};

// ---- global -> LDS staging of a [64][DH] tile (zero filled past n_valid rows) ----------------------
template <typename T, int DH>
FF_DEV void stage_tile(T* lds, const T* base, long long row_stride, int row0, int n_rows) {
    constexpr int LD = DH + AttnCfg<T>::pad;
    constexpr int VN = Vec<T>::N, CH = DH / VN;
    for (int idx = threadIdx.x; idx < kTile * CH; idx += 256) {
        const int r = idx / CH, ch = idx - r * CH;
        uint4 v = {0, 0, 0, 0};
        if (row0 + r < n_rows) v = *(const uint4*)(base + (long long)(row0 + r) * row_stride + ch * VN);
        *(uint4*)(lds + r * LD + ch * VN) = v;
    }
}

// The same staging split in two (round 4): `stage_issue` requests a tile's 16-byte pieces into registers, `stage_commit` writes them to LDS.
// The stand-alone kernels issue tile t + 1 right after tile t has been published, so its global-memory latency runs under tile t's MFMAs
// instead of in front of them (one workgroup per CU at the resampler's shape: nothing else was hiding it - 20 us for 25 MB).
template <typename T, int DH> struct TileRegs {
    static constexpr int N = (kTile * (DH / Vec<T>::N) + 255) / 256;
    uint4 v[N];
};
template <typename T, int DH>
FF_DEV void stage_issue(TileRegs<T, DH>& r, const T* base, long long row_stride, int row0, int n_rows) {
    constexpr int VN = Vec<T>::N, CH = DH / VN;
#pragma unroll
    for (int i = 0; i < TileRegs<T, DH>::N; i++) {
        const int idx = threadIdx.x + i * 256, rr = idx / CH, ch = idx - rr * CH;
        r.v[i] = uint4{0, 0, 0, 0};
        if (idx < kTile * CH && row0 + rr < n_rows) r.v[i] = *(const uint4*)(base + (long long)(row0 + rr) * row_stride + ch * VN);
    }
}
template <typename T, int DH> FF_DEV void stage_commit(const TileRegs<T, DH>& r, T* lds) {
    constexpr int LD = DH + AttnCfg<T>::pad;
    constexpr int VN = Vec<T>::N, CH = DH / VN;
#pragma unroll
    for (int i = 0; i < TileRegs<T, DH>::N; i++) {
        const int idx = threadIdx.x + i * 256, rr = idx / CH, ch = idx - rr * CH;
        if (idx < kTile * CH) *(uint4*)(lds + rr * LD + ch * VN) = r.v[i];
    }
}

// ---- Z[sub][r] (other row sub*16 + g*4 + r, own row c) += X_tile . own^T ---------------------------
template <int DH, typename L = PadLayout<bf16, DH>> FF_DEV void mma_k(f32x4 (&z)[4], const bf16* tile, const OwnFrag<bf16, DH>& own) {
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
#pragma unroll
    for (int sub = 0; sub < 4; sub++)
#pragma unroll
        for (int ks = 0; ks < DH / 32; ks++) {
            const bf16x8 a = *(const bf16x8*)(tile + L::off(sub * 16 + c, ks * 32 + g * 8));
            z[sub] = mfma_bf16(a, own.f[ks], z[sub]);
        }
}
template <int DH, typename L = PadLayout<float, DH>> FF_DEV void mma_k(f32x4 (&z)[4], const float* tile, const OwnFrag<float, DH>& own) {
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
#pragma unroll
    for (int sub = 0; sub < 4; sub++)
#pragma unroll
        for (int s = 0; s < DH / 16; s++) {
            const f32x4 a = *(const f32x4*)(tile + L::off(sub * 16 + c, s * 16 + g * 4));
#pragma unroll
            for (int j = 0; j < 4; j++) z[sub] = mfma_f32(a[j], own.f[s][j], z[sub]);
        }
}

// ---- Acc^T[dt][r] (d = dt*16 + g*4 + r, own row c) += Y_tile^T . Z' --------------------------------
template <int DH, typename L = PadLayout<bf16, DH>> FF_DEV void mma_t(f32x4 (&acc)[DH / 16], const bf16* tile, const f32x4 (&z)[4]) {
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
#pragma unroll
    for (int s = 0; s < 2; s++) {  // k-step of 32 other rows = sub-tiles 2s, 2s+1
        bf16x8 b;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            b[e] = (bf16)z[2 * s][e];
            b[4 + e] = (bf16)z[2 * s + 1][e];
        }
        const int row = (2 * s) * 16 + g * 4 + (c >> 2), col = (c & 3) * 4;
#pragma unroll
        for (int dt = 0; dt < DH / 16; dt++) {
            const bf16x8 a = cat4(lds_read_tr16(tile + L::off(row, col + dt * 16)), lds_read_tr16(tile + L::off(row + 16, col + dt * 16)));
            acc[dt] = mfma_bf16(a, b, acc[dt]);
        }
    }
}
template <int DH, typename L = PadLayout<float, DH>> FF_DEV void mma_t(f32x4 (&acc)[DH / 16], const float* tile, const f32x4 (&z)[4]) {
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
#pragma unroll
    for (int sub = 0; sub < 4; sub++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int dt = 0; dt < DH / 16; dt++) acc[dt] = mfma_f32(tile[L::off(sub * 16 + g * 4 + r, dt * 16 + c)], z[sub][r], acc[dt]);
        }
}

// ---- which keys a query row may see ------------------------------------------------------------------
struct RowRange {
    int lo, hi;   // attended keys [lo, hi)
    int softmax;  // 1: ordinary softmax row (gradient flows to the scores); 0: zero row or uniform row
    int uniform;  // 1: fully masked row -> scores are all equal (gated_cross_attention.py:112-115)
};
FF_DEV RowRange row_range(const ff_attn_desc& d, const int* tt, int b, int q) {
    RowRange r;
    if (q >= d.n_q) { r.lo = r.hi = 0; r.softmax = 0; r.uniform = 0; return r; }
    if (d.mode == FF_ATTN_DENSE) { r.lo = 0; r.hi = d.n_kv; r.softmax = 1; r.uniform = 0; return r; }
    const int t = tt[(long long)b * d.tt_stride + d.tt_offset + q];
    const int n_media = d.n_kv / d.n_visual;
    if (t <= 0) { r.lo = r.hi = 0; r.softmax = 0; r.uniform = 0; }                              // :119-121 zeroed row
    else if (t <= n_media) { r.lo = (t - 1) * d.n_visual; r.hi = t * d.n_visual; r.softmax = 1; r.uniform = 0; }  // :111
    else { r.lo = 0; r.hi = d.n_kv; r.softmax = 0; r.uniform = 1; }                            // all masked -> uniform
    return r;
}

template <typename T, int DH> FF_DEV void store_acc_row(T* row, const f32x4 (&acc)[DH / 16], float scale, int g) {
    typedef __attribute__((ext_vector_type(4))) T vec4;
#pragma unroll
    for (int dt = 0; dt < DH / 16; dt++) {
        vec4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = from_f32<T>(acc[dt][r] * scale);
        *(vec4*)(row + dt * 16 + g * 4) = v;
    }
}

FF_DEV float group_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16, 64)), fmaxf(__shfl_xor(v, 32, 64), __shfl_xor(v, 48, 64))); }
FF_DEV float group_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }

// block-wide [min lo, max hi) over the 64 own rows, to skip key tiles nobody attends to
FF_DEV void block_range(int lo, int hi, int* sh, int& blo, int& bhi) {
    if (hi <= lo) { lo = 0x7fffffff; hi = 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = lo; sh[4 + w] = hi; }
    __syncthreads();
    blo = min(min(sh[0], sh[1]), min(sh[2], sh[3]));
    bhi = max(max(sh[4], sh[5]), max(sh[6], sh[7]));
}


// =====================================================================================================
// Loop bodies shared by the stand-alone attention kernels and the fused projection + attention kernels.
// All of them use the workgroup's 4 waves: wave w owns "own" rows w*16 .. w*16+15 of a 64-row tile.
// =====================================================================================================
// forward: online softmax over the key tiles [blo, bhi) for the wave's 16 own queries (fragment fq, per-lane key range rr)
// Staging policy of the K / V tiles: St::L = LDS layout, St::stage2(...) = bring rows row0.. of two matrices into two tiles (complete,
// but not yet barrier-published, on return).  `staged_k0`: key tile the caller has already staged (and published) itself, or -1.
template <typename T, int DH> struct SyncStage {
    typedef PadLayout<T, DH> L;
    static FF_DEV void stage2(T* s0, const T* b0, long long sr0, T* s1, const T* b1, long long sr1, int row0, int n_rows) {
        stage_tile<T, DH>(s0, b0, sr0, row0, n_rows);
        stage_tile<T, DH>(s1, b1, sr1, row0, n_rows);
    }
};
// ... and the register-prefetched form of it (the stand-alone kernels): issue2 early, commit2 behind the barrier
template <typename T, int DH> struct PrefetchStage : SyncStage<T, DH> {
    struct Regs { TileRegs<T, DH> a, b; };
    static FF_DEV void issue2(Regs& r, const T* b0, long long sr0, const T* b1, long long sr1, int row0, int n_rows) {
        stage_issue<T, DH>(r.a, b0, sr0, row0, n_rows);
        stage_issue<T, DH>(r.b, b1, sr1, row0, n_rows);
    }
    static FF_DEV void commit2(const Regs& r, T* s0, T* s1) {
        stage_commit<T, DH>(r.a, s0);
        stage_commit<T, DH>(r.b, s1);
    }
};
// register sets kept in flight.  One: the next tile's latency runs under the current tile's products (resampler forward 19.7 -> 15.2 us, dQ
// 26.8 -> 17.1 us).  Three were measured too (r4s7b): no further gain - after the first overlap the loop is bound by its two barriers and the
// LDS round trip per tile, not by memory latency - so the ring stays one deep.
template <typename T, int DH> struct PrefetchDepth { static constexpr int value = 1; };
template <typename St> struct StageSplit { static constexpr bool value = false; };
template <typename T, int DH> struct StageSplit<PrefetchStage<T, DH>> { static constexpr bool value = true; };

template <typename T, int DH, typename St = SyncStage<T, DH>>
FF_DEV void attn_fwd_loop(const ff_attn_desc& d, const OwnFrag<T, DH>& fq, const RowRange& rr, int blo, int bhi, const T* Kb, const T* Vb,
                          T* sK, T* sV, f32x4 (&acc)[DH / 16], float& m, float& lsum, int staged_k0 = -1) {
    typedef typename St::L L;
    const int g = (threadIdx.x & 63) >> 4;
    const int k_first = (blo / kTile) * kTile;
    constexpr bool SPLIT = StageSplit<St>::value;
    constexpr int D = SPLIT ? PrefetchDepth<T, DH>::value : 1;           // key tiles requested ahead (statically indexed register sets)
    static_assert(D == 1, "deeper rings were measured without gain and are not covered by the parity suite");
    [[maybe_unused]] typename std::conditional<SPLIT, typename PrefetchStage<T, DH>::Regs, int>::type rg[D];
    if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < D; j++)
            if (k_first + j * kTile < bhi && k_first + j * kTile != staged_k0) St::issue2(rg[j], Kb, d.k.sr, Vb, d.v.sr, k_first + j * kTile, d.n_kv);
    }
    auto tile = [&](int k0, auto& regs) {
        if (k0 != staged_k0) {
            __syncthreads();
            if constexpr (SPLIT) St::commit2(regs, sK, sV);
            else St::stage2(sK, Kb, d.k.sr, sV, Vb, d.v.sr, k0, d.n_kv);
            __syncthreads();
        }
        if constexpr (SPLIT)
            if (k0 + D * kTile < bhi) St::issue2(regs, Kb, d.k.sr, Vb, d.v.sr, k0 + D * kTile, d.n_kv);      // in flight under the next D tiles' MFMAs
        f32x4 z[4];
#pragma unroll
        for (int s = 0; s < 4; s++) z[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_k<DH, L>(z, sK, fq);
        float tmax = kNegBig;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = k0 + s * 16 + g * 4 + r;
                const bool ok = key >= rr.lo && key < rr.hi;
                const float v = rr.uniform ? 0.f : z[s][r];
                z[s][r] = ok ? v : kNegBig;
                tmax = fmaxf(tmax, z[s][r]);
            }
        tmax = group_max(tmax);
        const float m_new = fmaxf(m, tmax);
        const float alpha = __expf(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float p = z[s][r] > 0.5f * kNegBig ? __expf(z[s][r] - m_new) : 0.f;
                z[s][r] = p;
                psum += p;
            }
        lsum = lsum * alpha + psum;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < DH / 16; dt++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[dt][r] *= alpha;
        mma_t<DH, L>(acc, sV, z);
    };
    for (int k0 = k_first; k0 < bhi; k0 += D * kTile) {
#pragma unroll
        for (int j = 0; j < D; j++)
            if (k0 + j * kTile < bhi) tile(k0 + j * kTile, rg[j]);
    }
}

// backward, own rows = queries: dQ^T += K^T dS^T over the key tiles [blo, bhi); L = saved log-sum-exp, Dq = sum_d dO * O of the own row
template <typename T, int DH, typename St = SyncStage<T, DH>>
FF_DEV void attn_dq_loop(const ff_attn_desc& d, const OwnFrag<T, DH>& fq, const OwnFrag<T, DH>& fdo, const RowRange& rr, float L, float Dq,
                         int blo, int bhi, const T* Kb, const T* Vb, T* sK, T* sV, f32x4 (&acc)[DH / 16], int staged_k0 = -1) {
    typedef typename St::L LY;
    const int g = (threadIdx.x & 63) >> 4;
    const int k_first = (blo / kTile) * kTile;
    constexpr bool SPLIT = StageSplit<St>::value;
    constexpr int D = SPLIT ? PrefetchDepth<T, DH>::value : 1;
    static_assert(D == 1, "deeper rings were measured without gain and are not covered by the parity suite");
    [[maybe_unused]] typename std::conditional<SPLIT, typename PrefetchStage<T, DH>::Regs, int>::type rg[D];
    if constexpr (SPLIT) {
#pragma unroll
        for (int j = 0; j < D; j++)
            if (k_first + j * kTile < bhi && k_first + j * kTile != staged_k0) St::issue2(rg[j], Kb, d.k.sr, Vb, d.v.sr, k_first + j * kTile, d.n_kv);
    }
    auto tile = [&](int k0, auto& regs) {
        if (k0 != staged_k0) {
            __syncthreads();
            if constexpr (SPLIT) St::commit2(regs, sK, sV);
            else St::stage2(sK, Kb, d.k.sr, sV, Vb, d.v.sr, k0, d.n_kv);
            __syncthreads();
        }
        if constexpr (SPLIT)
            if (k0 + D * kTile < bhi) St::issue2(regs, Kb, d.k.sr, Vb, d.v.sr, k0 + D * kTile, d.n_kv);
        f32x4 z[4], dp[4];
#pragma unroll
        for (int s = 0; s < 4; s++) { z[s] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_k<DH, LY>(z, sK, fq);     // S^T
        mma_k<DH, LY>(dp, sV, fdo);   // dP^T = V dO^T
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = k0 + s * 16 + g * 4 + r;
                const bool ok = key >= rr.lo && key < rr.hi;
                const float p = ok ? __expf(z[s][r] - L) : 0.f;
                z[s][r] = p * (dp[s][r] - Dq);  // dS^T
            }
        mma_t<DH, LY>(acc, sK, z);    // dQ^T += K^T dS^T
    };
    for (int k0 = k_first; k0 < bhi; k0 += D * kTile) {
#pragma unroll
        for (int j = 0; j < D; j++)
            if (k0 + j * kTile < bhi) tile(k0 + j * kTile, rg[j]);
    }
}

// backward, own rows = keys: one 64-query tile already staged in LDS (sQ, sDO) with its per-query key ranges / flags / statistics
// (s_lo, s_hi, s_flag = softmax | uniform << 1, s_lse, s_D); accumulates dV^T += dO^T P and dK^T += Q^T dS for the wave's 16 own keys.
template <typename T, int DH, typename LY = PadLayout<T, DH>>
FF_DEV void attn_dkv_step(int key, const OwnFrag<T, DH>& fk, const OwnFrag<T, DH>& fv, const T* sQ, const T* sDO, const int* s_lo,
                          const int* s_hi, const int* s_flag, const float* s_lse, const float* s_D, f32x4 (&acc_k)[DH / 16],
                          f32x4 (&acc_v)[DH / 16]) {
    const int g = (threadIdx.x & 63) >> 4;
    f32x4 z[4], dp[4];
#pragma unroll
    for (int s = 0; s < 4; s++) { z[s] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    mma_k<DH, LY>(z, sQ, fk);     // S[q][key]
    mma_k<DH, LY>(dp, sDO, fv);   // dP[q][key]
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int qi = s * 16 + g * 4 + r;
            const bool ok = key >= s_lo[qi] && key < s_hi[qi];
            const int flag = s_flag[qi];
            const float sc = (flag & 2) ? 0.f : z[s][r];
            const float p = ok ? __expf(sc - s_lse[qi]) : 0.f;
            z[s][r] = p;                                                   // P
            dp[s][r] = (flag & 1) ? p * (dp[s][r] - s_D[qi]) : 0.f;        // dS
        }
    mma_t<DH, LY>(acc_v, sDO, z);   // dV^T += dO^T P
    mma_t<DH, LY>(acc_k, sQ, dp);   // dK^T += Q^T dS
}

}  // namespace ff
