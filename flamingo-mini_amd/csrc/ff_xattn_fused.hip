// Fused cross-attention kernels of the gated xattn block (gated_cross_attention.py:74-126 and its autograd):
//
//   xa_qattn_fwd_kernel   one workgroup per (query tile, head, sample):
//                         LayerNorm(y) -> q = to_q(.) * scale (head slice of the weight) -> masked softmax(q K^T) V -> O
//                         i.e. what used to be a LayerNorm launch, a split-K GEMM (+ its reduce launch) and an attention launch.
//                         LayerNorm is the GEMM's prologue: raw y tiles travel global -> LDS by DMA and are normalised in the
//                         A fragments (statistics of the tile's rows are computed once by the workgroup itself).
//   xa_dattn_bwd_kernel   the mirror image: d O = tanh(alpha) * d y1 . Wo (head slice) -> attention backward (d Q; and d K, d V too
//                         when a sample's queries fit one tile, which is the training shape: 32 tokens) - instead of a GEMM
//                         (+ split-K reduce) and two attention launches.
//
// bf16: operand tiles through the LDS-DMA ring of the GEMM family (ff_gemm_tiles.h), v_mfma_f32_16x16x32_bf16; each wave owns 16
// rows of the tile for the projection AND the attention, so Q / d O reach the attention MFMAs through a 9 KiB LDS tile.
// fp32 (verification precision): the projection reads its operands straight from global memory (v_mfma_f32_16x16x4_f32).
#include <atomic>
#include "ff_common.h"
#include "ff_internal.h"
#include "ff_gemm_tiles.h"
#include "ff_attention_core.h"

namespace ff {

#ifdef FF_XA_TIMELINE   // debug build: per-workgroup phase timestamps (100 MHz constant clock), read with ff_debug_xa_timeline_read
__device__ unsigned long long g_xa_timeline[4096 * 16];
#define FF_XTL(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_xa_timeline[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FF_XTL(i) do { } while (0)
#endif

namespace {

template <typename T> struct IsBf16 { static constexpr bool value = false; };
template <> struct IsBf16<bf16> { static constexpr bool value = true; };

// bf16 with 64-wide heads (every published Flamingo configuration): the attention tiles are K-major swizzled [64][64] tiles filled by
// LDS-DMA, so the first key tile (and, in backward, the Q and O tiles) are in flight while the projection runs.
template <typename T, int DH> struct Fast { static constexpr bool value = IsBf16<T>::value && DH == 64; };

struct DmaStage64 {
    typedef SwzLayout L;
    static FF_DEV void issue(bf16* s, const bf16* base, long long sr, int row0, int n_rows) {
        const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        dma_tile<64, 0>(r, s, RowMap{sr, 0, 0}, row0, n_rows, 0, kBK, w, l);
    }
    static FF_DEV void stage2(bf16* s0, const bf16* b0, long long sr0, bf16* s1, const bf16* b1, long long sr1, int row0, int n_rows) {
        issue(s0, b0, sr0, row0, n_rows);
        issue(s1, b1, sr1, row0, n_rows);
        wait_vmcnt<0>();
    }
};
// 16 bytes held in registers -> Vec<T>::N floats
FF_DEV void unpack16(const uint4& r, float (&v)[8], bf16) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, r);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (float)x[e];
}
FF_DEV void unpack16(const uint4& r, float (&v)[4], float) {
    const f32x4 x = __builtin_bit_cast(f32x4, r);
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = x[e];
}

template <typename T, int DH, bool FAST> struct StageOf { typedef SyncStage<T, DH> type; };
template <> struct StageOf<bf16, 64, true> { typedef DmaStage64 type; };

// ring depth: as many 64-deep k-steps in flight as ~96 KiB of LDS hold (one workgroup per CU, its DMA queue is what hides the latency)
template <int BM, int DH> constexpr int ring_stages() {
    constexpr int stage = (BM + DH) * kBK * 2, n = 96 * 1024 / stage;
    return n > 8 ? 8 : (n < 2 ? 2 : n);
}

// LDS carve-up (bytes).  region A: the projection's operand ring, later the tiles parked by the workgroup itself (Q resp. dO) and,
// on the synchronous path, the K / V (/ Q) tiles; region B (fast path only): tiles prefetched by DMA while the projection runs.
template <typename T, int DH, int BM, int NS, int SELF_TILES, int SYNC_TILES, int PRE_TILES> struct Carve {
    static constexpr bool FAST = Fast<T, DH>::value;
    static constexpr size_t tile = FAST ? (size_t)SwzLayout::tile_elems * sizeof(bf16) : (size_t)PadLayout<T, DH>::tile_elems * sizeof(T);
    static constexpr size_t ring = IsBf16<T>::value ? (size_t)NS * (BM + DH) * kBK * sizeof(bf16) : 0;
    static constexpr size_t self = (size_t)(SELF_TILES + (FAST ? 0 : SYNC_TILES)) * tile;
    static constexpr size_t regionA = ((ring > self ? ring : self) + 255) / 256 * 256;
    static constexpr size_t regionB = FAST ? (size_t)PRE_TILES * tile : 0;
    static constexpr size_t fixed = regionA + regionB;
};

FF_DEV ff_attn_desc attn_view(const XaFusedArgs& a, int dim_head) {
    ff_attn_desc d = {};
    d.batch = a.batch; d.heads = a.heads; d.dim_head = dim_head; d.n_q = a.n_q; d.n_kv = a.n_kv;
    d.mode = FF_ATTN_MEDIA; d.n_visual = a.n_visual; d.tt_stride = a.tt_stride; d.tt_offset = a.tt_offset;
    d.k = a.k; d.v = a.v; d.dk = a.dk; d.dv = a.dv;
    return d;
}

// tile kt must have landed; up to `younger` (wave-uniform, <= MAXY) younger tiles of PER_TILE DMA instructions each may stay in flight
template <int PER_TILE, int MAXY> FF_DEV void wait_tiles(int younger) {
    if constexpr (MAXY == 0) wait_vmcnt<0>();
    else {
        if (younger >= MAXY) wait_vmcnt<MAXY * PER_TILE>();
        else wait_tiles<PER_TILE, MAXY - 1>(younger);
    }
}

// D[n][m] += B_tile[n][k] * A_tile[m][k] over k = [0, dim) (dim % 64 == 0) for the wave's 16 rows m of the A tile and all DH columns n,
// operands through the LDS-DMA ring.  BL: layout of the B operand (0: stored [DH][dim], 1: stored [dim][ldb], columns n_base..).
// `pro` is the A-operand prologue (the forward kernel's LayerNorm; NoPrologue otherwise).
// A 64-row tile: all four waves compute 16 rows each and share the DMA issue; the prologue is applied to the A fragments.
// A 32-row tile has MFMA work for two waves only: those two compute, the other two are the tile's DMA engine (issuing a tile piece
// costs an in-order wave 100+ cycles, which would otherwise sit on the compute waves' critical path) - and they also apply the
// prologue, in place in LDS, one k-step ahead of the compute waves.
struct NoPrologue {
    static constexpr bool active = false;
    FF_DEV void frag(bf16x8&, int, int) const {}
    template <int BM> FF_DEV void tile(bf16*, int, int, int) const {}
};

template <int DH, int BM, int NS, int BL, typename P>
FF_DEV void project_bf16(bf16* ring, const bf16* Ab, long long lda, int row0, int row_lim, const bf16* Bb, long long ldb, int n_base, int n_lim,
                         int dim, int w, int l, f32x4 (&acc)[DH / 16], const P& pro) {
    constexpr int A_ELEMS = BM * kBK, B_ELEMS = DH * kBK, STAGE = A_ELEMS + B_ELEMS, NT = DH / 16;
    constexpr int NWC = BM / 16;                            // computing waves (16 rows each)
    constexpr bool SPLIT = NWC != 4;                        // producer / consumer waves
    constexpr int NWP = SPLIT ? 4 - NWC : 4;                // issuing waves
    constexpr int PA = BM / (NWP * 8), PB = DH / (NWP * 8), PER_TILE = PA + PB;      // DMA instructions per issuing wave per k-step
    static_assert(PER_TILE * (NS - 1) <= 63, "vmcnt overflow");
    const bool computes = w < NWC, issues = !SPLIT || w >= NWC;            // wave-uniform
    const int wp = SPLIT ? w - NWC : w;
    const RowMap a_map{lda, 0, 0}, b_map{ldb, 0, 0};
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, 0x7fffffff, 0x00020000);
    const int nk = dim / kBK;
    unsigned va[PA], vb[PB];
    if (issues) {
        dma_prepare<BM, 0, NWP>(a_map, row0, row_lim, wp, l, va);
        dma_prepare<DH, BL, NWP>(b_map, n_base, n_lim, wp, l, vb);
    }
    const unsigned b_step = BL == 0 ? 2u : (unsigned)ldb * 2u;             // bytes per unit of k
    auto issue = [&](int tile) {
        bf16* st = ring + (tile % NS) * STAGE;
        const unsigned k0 = (unsigned)tile * kBK;
        dma_tile_fast<BM, 0, NWP>(ra, st, va, k0 * 2u, wp);
        dma_tile_fast<DH, BL, NWP>(rb, st + A_ELEMS, vb, k0 * b_step, wp);
    };
    auto compute = [&](int kt, bool frag_prologue) {
        const bf16* sA = ring + (kt % NS) * STAGE;
        const bf16* sB = sA + A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < kBK / 32; ks++) {
            bf16x8 fa = frag_read2<BM, 0>(sA, w * 16, ks);
            if (frag_prologue) pro.frag(fa, kt * kBK + ks * 32 + ((l >> 4) << 3), w * 16 + (l & 15));
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] = mfma_bf16(frag_read2<DH, BL>(sB, j * 16, ks), fa, acc[j]);   // D[n][m]
        }
    };
    if constexpr (!SPLIT) {
#pragma unroll
        for (int s = 0; s < NS - 1; s++)
            if (s < nk) issue(s);
        for (int kt = 0; kt < nk; kt++) {
            wait_tiles<PER_TILE, NS - 2>(min(nk - 1 - kt, NS - 2));
            __builtin_amdgcn_s_barrier();       // tile kt is in LDS for everyone; stage (kt - 1) % NS is free
            if (kt + NS - 1 < nk) issue(kt + NS - 1);
            compute(kt, P::active);
        }
    } else {
        // barrier B(kt) closes step kt: the compute waves are done with tile kt (its stage may be refilled) and the issuing waves
        // have tile kt + 1 landed and run through the prologue.
        if (issues) {
#pragma unroll
            for (int s = 0; s < NS; s++)
                if (s < nk) issue(s);
            wait_tiles<PER_TILE, NS - 1>(min(nk - 1, NS - 1));
            if (P::active) pro.template tile<BM>(ring, 0, wp, l);
        }
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; kt++) {
            if (computes) compute(kt, false);
            else if (kt + 1 < nk) {
                wait_tiles<PER_TILE, NS - 2>(min(nk - 2 - kt, NS - 2));
                if (P::active) pro.template tile<BM>(ring + ((kt + 1) % NS) * STAGE, (kt + 1) * kBK, wp, l);
            }
            __builtin_amdgcn_s_barrier();
            if (issues && kt + NS < nk) issue(kt + NS);
        }
        wait_vmcnt<0>();
    }
}

// LayerNorm as the prologue of the Q projection: statistics and gamma / beta come from LDS (written by the workgroup beforehand)
        struct LnPrologue {
            static constexpr bool active = true;
            const bf16 *s_g, *s_b;
            const float *s_mean, *s_rstd;
            // fragment form: 8 consecutive k (from kk) of tile row m, applied by the computing wave that owns the row
            FF_DEV void frag(bf16x8& fa, int kk, int m) const {
                const float mu = s_mean[m], rs = s_rstd[m];
                float gv[8], bv[8];
                Vec<bf16>::load(s_g + kk, gv);
                Vec<bf16>::load(s_b + kk, bv);
#pragma unroll
                for (int e = 0; e < 8; e++) fa[e] = (bf16)(((float)fa[e] - mu) * rs * gv[e] + bv[e]);
            }
            // tile form: the [BMT][64] K-major swizzled A tile of k-step k0, normalised in place by the 2 issuing waves (128 threads)
            template <int BMT> FF_DEV void tile(bf16* sA, int k0, int wp, int l) const {
                const int tid = wp * 64 + l;
#pragma unroll
                for (int it = 0; it < BMT * 8 / 128; it++) {
                    const int i = tid + it * 128, row = i >> 3, slot = i & 7;
                    const int kk = k0 + ((slot ^ (row & 7)) << 3);
                    const float mu = s_mean[row], rs = s_rstd[row];
                    bf16* p = sA + row * kBK + slot * 8;
                    float v[8], gv[8], bv[8];
                    Vec<bf16>::load(p, v);
                    Vec<bf16>::load(s_g + kk, gv);
                    Vec<bf16>::load(s_b + kk, bv);
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = (v[e] - mu) * rs * gv[e] + bv[e];
                    Vec<bf16>::store(p, v);
                }
            }
        };

// acc[j][r] = value (row m = w*16 + c, column j*16 + g*4 + r) -> LDS tile in layout L
template <typename T, int DH, typename L> FF_DEV void park_rows(T* tile, const f32x4 (&acc)[DH / 16], float scale, int w, int c, int g) {
    typedef __attribute__((ext_vector_type(4))) T vec4;
#pragma unroll
    for (int j = 0; j < DH / 16; j++) {
        vec4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = from_f32<T>(acc[j][r] * scale);
        *(vec4*)(tile + L::off(w * 16 + c, j * 16 + g * 4)) = v;
    }
}

// rows [0, n_rows) of an LDS tile -> global rows (16-byte row-contiguous pieces)
template <typename T, int DH, typename L> FF_DEV void tile_to_global(const T* tile, T* dst, long long row_stride, int n_rows) {
    constexpr int VN = Vec<T>::N, CH = DH / VN;
    for (int idx = threadIdx.x; idx < n_rows * CH; idx += 256) {
        const int r = idx / CH, ch = idx - r * CH;
        *(uint4*)(dst + (long long)r * row_stride + ch * VN) = *(const uint4*)(tile + L::off(r, ch * VN));
    }
}

}  // namespace

// =====================================================================================================
// forward: LayerNorm -> Q projection (one head) -> masked attention
// =====================================================================================================
template <typename T, int DH, int BM>
__global__ __launch_bounds__(256) void xa_qattn_fwd_kernel(const XaFusedArgs a_in, const T* __restrict__ y, const T* __restrict__ gamma,
                                                           const T* __restrict__ beta, const T* __restrict__ Wq, const T* __restrict__ K,
                                                           const T* __restrict__ V, const int* __restrict__ tt, T* __restrict__ yn,
                                                           T* __restrict__ Qs, T* __restrict__ O, float* __restrict__ mean,
                                                           float* __restrict__ rstd, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FF_XTL(0);
    const XaFusedArgs a = fetch_args(a_in);
    constexpr bool FAST = Fast<T, DH>::value;
    constexpr int NS = ring_stages<BM, DH>(), NT = DH / 16, VN = Vec<T>::N;
    typedef typename StageOf<T, DH, FAST>::type St;
    typedef typename St::L L;
    typedef Carve<T, DH, BM, NS, 1, 2, 2> CV;        // parks Q; K, V tiles
    // work list = (sample, query tile, head) with the head fastest: the workgroups that read the same rows of y / dy1 sit on one XCD
    const int n_qt = (a.n_q + BM - 1) / BM;
    const int lin = xcd_remap(blockIdx.x, n_qt * a.heads * a.batch);
    const int h = lin % a.heads, qt = (lin / a.heads) % n_qt, b = lin / (a.heads * n_qt);
    const int t = threadIdx.x, l = t & 63, c = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row0 = qt * BM;
    const int n_rows = min(BM, a.n_q - row0);
    const int dimp = (a.dim + kBK - 1) / kBK * kBK;
    T* sQ = (T*)smem;
    T* sK = FAST ? (T*)(smem + CV::regionA) : sQ + L::tile_elems;
    T* sV = sK + L::tile_elems;
    T* s_g = (T*)(smem + CV::fixed);
    T* s_b = s_g + dimp;
    float* s_mean = (float*)(s_b + dimp);
    float* s_rstd = s_mean + BM;
    int* sh = (int*)(s_rstd + BM);

    // ---- every load the prologue needs is issued before the first wait: the rows of y (ONE pass, NB 16-byte pieces per thread in
    // flight), gamma / beta, and text_time for the key ranges ----
    constexpr int TPR = 256 / BM, NB = 24;    // threads per row (adjacent lanes of one wave), pieces in flight per thread
    const int r = t / TPR, sub = t % TPR;
    const bool rok = r < n_rows;
    const long long grow = (long long)b * a.n_q + row0 + (rok ? r : 0);
    const T* yr = y + grow * a.dim;
    const int nchunk = a.dim / VN;
    const bool one_batch = nchunk <= TPR * NB;
    const int ppt = nchunk / TPR;             // pieces per thread: the same for every thread of a row (dim % (8 * VN) == 0)
    uint4 raw[NB];
#pragma unroll
    for (int u = 0; u < NB; u++)
        if (u < ppt) raw[u] = *(const uint4*)(yr + (sub + u * TPR) * VN);
    uint4 gb_raw[2];
    const int gb_chunks = dimp / VN;          // gamma pieces, then beta pieces; zero past `dim`
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int i = t + it * 256;
        const int ch = i < gb_chunks ? i : i - gb_chunks;
        gb_raw[it] = uint4{0, 0, 0, 0};
        if (i < 2 * gb_chunks && ch < nchunk) gb_raw[it] = *(const uint4*)((i < gb_chunks ? gamma : beta) + ch * VN);
    }
    for (int i = t + 512; i < 2 * gb_chunks; i += 256) {      // very wide models only: the rest goes global -> LDS piece by piece
        const int ch = i < gb_chunks ? i : i - gb_chunks;
        uint4 v = {0, 0, 0, 0};
        if (ch < nchunk) v = *(const uint4*)((i < gb_chunks ? gamma : beta) + ch * VN);
        *(uint4*)((i < gb_chunks ? s_g : s_b) + ch * VN) = v;
    }
    const ff_attn_desc d = attn_view(a, DH);
    const int m_own = w * 16 + c;                      // the lane's own row of the tile
    const bool own_ok = m_own < n_rows;                // (false for the helper waves of a 32-row tile)
    const int q = row0 + m_own;
    RowRange rr = row_range(d, tt, b, q);
    if (!own_ok) { rr.lo = rr.hi = 0; rr.softmax = 0; rr.uniform = 0; }
    {
        int i = t;
#pragma unroll
        for (int it = 0; it < 2; it++, i += 256)
            if (i < 2 * gb_chunks) *(uint4*)((i < gb_chunks ? s_g + i * VN : s_b + (i - gb_chunks) * VN)) = gb_raw[it];
    }
    int blo, bhi;
    block_range(rr.lo, rr.hi, sh, blo, bhi);           // (a barrier: gamma / beta are in LDS for everyone behind it)
    const T* Kb = K + b * a.k.sb + h * a.k.sh;
    const T* Vb = V + b * a.v.sb + h * a.v.sh;
    int staged_k0 = -1;
    if constexpr (FAST) {     // the first key tile starts travelling to LDS now and lands while the projection runs
        if (blo < bhi) {
            staged_k0 = (blo / kTile) * kTile;
            DmaStage64::issue(sK, Kb, a.k.sr, staged_k0, a.n_kv);
            DmaStage64::issue(sV, Vb, a.v.sr, staged_k0, a.n_kv);
        }
    }
    FF_XTL(1);

    // ---- LayerNorm statistics: sums shifted by the mean of the row's first piece (x0), so the one-pass variance has two-pass accuracy ----
    {
        float x0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int base = 0; base < nchunk; base += TPR * NB) {
            const int left = ppt - base / TPR;           // wave-uniform
            if (base > 0) {
#pragma unroll
                for (int u = 0; u < NB; u++)
                    if (u < left) raw[u] = *(const uint4*)(yr + (base + sub + u * TPR) * VN);
            } else {
                float first[VN];
                unpack16(raw[0], first, T());
                float fs = 0.f;                                 // the shift: mean of the row's first 16-byte piece (held by the row's first thread) -
#pragma unroll
                for (int e = 0; e < VN; e++) fs += first[e];    // a single element would lose the variance to cancellation when it is an outlier of its row
                x0 = __shfl(fs * (1.f / VN), (t & 63) - sub, 64);
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                if (u < left) {
                    float v[VN];
                    unpack16(raw[u], v, T());
#pragma unroll
                    for (int e = 0; e < VN; e++) {
                        const float dlt = v[e] - x0;
                        s1 += dlt;
                        s2 += dlt * dlt;
                    }
                }
            }
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        const float sh1 = s1 / (float)a.dim;
        const float mu_r = x0 + sh1;
        const float rs_r = rsqrtf(fmaxf(s2 / (float)a.dim - sh1 * sh1, 0.f) + a.eps);
        if (sub == 0) {
            s_mean[r] = rok ? mu_r : 0.f;
            s_rstd[r] = rok ? rs_r : 0.f;
            if (rok && h == 0) { mean[grow] = mu_r; rstd[grow] = rs_r; }
        }
        // The normalised rows are an operand of d to_q.weight: written from the registers that still hold the row, the pieces of a row
        // shared out over the heads' workgroups (gamma / beta from LDS: published by the barrier in block_range).
        if (rok && yn) {
            T* ynr = yn + grow * a.dim;
            for (int base = 0; base < nchunk; base += TPR * NB) {
                const int left = ppt - base / TPR;
                int owner = 0;                                               // u % heads without a division per piece
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const int ch = base + sub + u * TPR;
                    const bool mine = owner == h;
                    owner = owner + 1 == a.heads ? 0 : owner + 1;
                    if (u < left && mine) {                                  // this head's share of the row's pieces
                        float v[VN], gv[VN], bv[VN];
                        if (one_batch) unpack16(raw[u], v, T());
                        else Vec<T>::load(yr + ch * VN, v);
                        Vec<T>::load(s_g + ch * VN, gv);
                        Vec<T>::load(s_b + ch * VN, bv);
#pragma unroll
                        for (int e = 0; e < VN; e++) v[e] = (v[e] - mu_r) * rs_r * gv[e] + bv[e];
                        Vec<T>::store(ynr + ch * VN, v);
                    }
                }
            }
        }
    }
    FF_XTL(2);
    __syncthreads();      // statistics visible

    // ---- q[m][n] = sum_k LN(y)[m][k] Wq[h*DH + n][k] ----
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float mu = m_own < BM ? s_mean[m_own] : 0.f, rs = m_own < BM ? s_rstd[m_own] : 0.f;
    if constexpr (IsBf16<T>::value) {
        const LnPrologue ln{s_g, s_b, s_mean, s_rstd};
        project_bf16<DH, BM, NS, 0>((bf16*)smem, y + (long long)b * a.n_q * a.dim, a.dim, row0, a.n_q, Wq + (long long)h * DH * a.dim, a.dim, 0,
                                    DH, a.dim, w, l, acc, ln);
    } else {
        if (w * 16 < BM) {
            const float* yr = y + ((long long)b * a.n_q + row0 + (own_ok ? m_own : 0)) * a.dim;
            for (int k = 0; k < a.dim; k += 4) {
                const int kk = k + g;
                const bool kok = kk < a.dim;
                const float av = own_ok && kok ? (yr[kk] - mu) * rs * s_g[kk] + s_b[kk] : 0.f;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const float bv = kok ? Wq[((long long)h * DH + j * 16 + c) * a.dim + kk] : 0.f;
                    acc[j] = mfma_f32(bv, av, acc[j]);       // D[n][m]
                }
            }
        }
    }
    FF_XTL(3);
    __syncthreads();      // the operand ring is dead: its memory becomes the Q tile (and, on the synchronous path, the K / V tiles)
    if (w * 16 < BM) park_rows<T, DH, L>(sQ, acc, a.scale, w, c, g);
    __syncthreads();
    tile_to_global<T, DH, L>(sQ, Qs + ((long long)b * a.n_q + row0) * a.inner + h * DH, a.inner, n_rows);   // saved for backward

    FF_XTL(4);
    // ---- O = softmax_masked(q K^T) V for the wave's 16 own queries ----
    OwnFrag<T, DH> fq;
    fq.template load_tile<L>(sQ, m_own, g, own_ok);
    f32x4 o[NT];
#pragma unroll
    for (int dt = 0; dt < NT; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = kNegBig, lsum = 0.f;
    attn_fwd_loop<T, DH, St>(d, fq, rr, blo, bhi, Kb, Vb, sK, sV, o, m, lsum, staged_k0);
    FF_XTL(5);
    lsum = group_sum(lsum);
    if (own_ok) {
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        store_acc_row<T, DH>(O + ((long long)b * a.n_q + q) * a.inner + h * DH, o, inv, g);
        if (g == 0) lse[((long long)b * a.heads + h) * a.n_q + q] = lsum > 0.f ? m + __logf(lsum) : kPosBig;
    }
    FF_XTL(6);
}

// =====================================================================================================
// backward: d O projection (one head) -> attention backward
// =====================================================================================================
template <typename T, int DH, int BM, bool SINGLE>
__global__ __launch_bounds__(256) void xa_dattn_bwd_kernel(const XaFusedArgs a_in, const T* __restrict__ dy1, const T* __restrict__ Wo,
                                                           const T* __restrict__ gate, const T* __restrict__ Qs, const T* __restrict__ K,
                                                           const T* __restrict__ V, const int* __restrict__ tt, const T* __restrict__ O,
                                                           const float* __restrict__ lse, T* __restrict__ dO_out, T* __restrict__ dQ,
                                                           T* __restrict__ dK, T* __restrict__ dV, float* __restrict__ Dsum) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FF_XTL(0);
    const XaFusedArgs a = fetch_args(a_in);
    constexpr bool FAST = Fast<T, DH>::value;
    constexpr int NS = ring_stages<BM, DH>(), NT = DH / 16;
    typedef typename StageOf<T, DH, FAST>::type St;
    typedef typename St::L L;
    typedef Carve<T, DH, BM, NS, 1, 3, 4> CV;        // parks dO; Q, K, V tiles (+ O on the DMA path)
    // work list = (sample, query tile, head) with the head fastest: the workgroups that read the same rows of y / dy1 sit on one XCD
    const int n_qt = (a.n_q + BM - 1) / BM;
    const int lin = xcd_remap(blockIdx.x, n_qt * a.heads * a.batch);
    const int h = lin % a.heads, qt = (lin / a.heads) % n_qt, b = lin / (a.heads * n_qt);
    const int t = threadIdx.x, l = t & 63, c = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row0 = qt * BM;
    const int n_rows = min(BM, a.n_q - row0);
    T* sDO = (T*)smem;
    T* sQ = FAST ? (T*)(smem + CV::regionA) : sDO + L::tile_elems;
    T* sK = sQ + L::tile_elems;
    T* sV = sK + L::tile_elems;
    T* sO = sV + L::tile_elems;                          // DMA path only
    int* s_lo = (int*)(smem + CV::fixed);
    int* s_hi = s_lo + kTile;
    int* s_flag = s_hi + kTile;
    float* s_lse = (float*)(s_flag + kTile);
    float* s_D = s_lse + kTile;
    int* sh = (int*)(s_D + kTile);

    // ---- key ranges of the own queries; the tiles the attention will need start travelling to LDS right away ----
    const ff_attn_desc d = attn_view(a, DH);
    const int m_own = w * 16 + c;
    const bool own_ok = m_own < n_rows;
    const int q = row0 + m_own;
    const RowRange rr_full = row_range(d, tt, b, q);
    RowRange rr = rr_full;
    if (!own_ok || !rr.softmax) rr.lo = rr.hi = 0;      // zero / uniform rows: no gradient reaches the scores
    if (SINGLE && t < kTile) { s_lo[t] = 0; s_hi[t] = 0; s_flag[t] = 0; s_lse[t] = kPosBig; s_D[t] = 0.f; }
    int blo, bhi;
    block_range(rr.lo, rr.hi, sh, blo, bhi);
    const T* Kb = K + b * a.k.sb + h * a.k.sh;
    const T* Vb = V + b * a.v.sb + h * a.v.sh;
    const T* Qb = Qs + (long long)b * a.n_q * a.inner + h * DH;
    const int q_lim = SINGLE ? a.n_q : min(a.n_q, row0 + BM);      // rows past the tile / the sample read as zeros
    int staged_k0 = -1;
    if constexpr (FAST) {
        if (blo < bhi) {
            staged_k0 = (blo / kTile) * kTile;
            DmaStage64::issue(sK, Kb, a.k.sr, staged_k0, a.n_kv);
            DmaStage64::issue(sV, Vb, a.v.sr, staged_k0, a.n_kv);
        }
        DmaStage64::issue(sQ, Qb, a.inner, row0, q_lim);
        DmaStage64::issue(sO, O + (long long)b * a.n_q * a.inner + h * DH, a.inner, row0, q_lim);
    }
    const long long sidx = ((long long)b * a.heads + h) * a.n_q + q;
    const float Lq = own_ok ? lse[sidx] : kPosBig;

    FF_XTL(1);
    // ---- dO[m][n] = tanh(alpha) * sum_k dy1[m][k] Wo[k][h*DH + n] ----
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (IsBf16<T>::value) {
        project_bf16<DH, BM, NS, 1>((bf16*)smem, dy1 + (long long)b * a.n_q * a.dim, a.dim, row0, a.n_q, Wo, a.inner, h * DH, (h + 1) * DH,
                                    a.dim, w, l, acc, NoPrologue());
    } else {
        if (w * 16 < BM) {
            const float* dr = dy1 + ((long long)b * a.n_q + row0 + (own_ok ? m_own : 0)) * a.dim;
            for (int k = 0; k < a.dim; k += 4) {
                const int kk = k + g;
                const bool kok = kk < a.dim;
                const float av = own_ok && kok ? dr[kk] : 0.f;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const float bv = kok ? Wo[(long long)kk * a.inner + h * DH + j * 16 + c] : 0.f;
                    acc[j] = mfma_f32(bv, av, acc[j]);
                }
            }
        }
    }
    FF_XTL(2);
    __syncthreads();      // operand ring dead -> dO tile (and, on the synchronous path, the Q / K / V tiles)
    const float gt = tanhf(to_f32(gate[0]));
    if (w * 16 < BM) park_rows<T, DH, L>(sDO, acc, gt, w, c, g);
    else {                // rows BM .. 63 of the 64-row dO tile do not exist: zero them (they are "other" rows of the dK / dV products)
        typedef __attribute__((ext_vector_type(4))) T vec4;
        vec4 z;
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = from_f32<T>(0.f);
#pragma unroll
        for (int j = 0; j < NT; j++) *(vec4*)(sDO + L::off(w * 16 + c, j * 16 + g * 4)) = z;
    }
    if constexpr (!FAST) stage_tile<T, DH>(sQ, Qb, a.inner, row0, q_lim);
    __syncthreads();
    if (!SINGLE && dO_out) tile_to_global<T, DH, L>(sDO, dO_out + ((long long)b * a.n_q + row0) * a.inner + h * DH, a.inner, n_rows);

    FF_XTL(3);
    // ---- own rows = queries: D = sum_d dO * O, dQ ----
    OwnFrag<T, DH> fq, fdo, fo;
    fq.template load_tile<L>(sQ, m_own, g, own_ok);
    fdo.template load_tile<L>(sDO, m_own, g, own_ok);
    if constexpr (FAST) fo.template load_tile<L>(sO, m_own, g, own_ok);
    else fo.load(own_ok ? O + ((long long)b * a.n_q + q) * a.inner + h * DH : nullptr, g);
    const float Dq = group_sum(fdo.dot(fo));
    if (own_ok && g == 0 && Dsum) Dsum[sidx] = Dq;
    if (SINGLE && own_ok && g == 0) {
        s_lo[m_own] = rr_full.lo; s_hi[m_own] = rr_full.hi; s_flag[m_own] = rr_full.softmax | (rr_full.uniform << 1);
        s_lse[m_own] = Lq; s_D[m_own] = Dq;
    }
    f32x4 dq[NT];
#pragma unroll
    for (int dt = 0; dt < NT; dt++) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    attn_dq_loop<T, DH, St>(d, fq, fdo, rr, Lq, Dq, blo, bhi, Kb, Vb, sK, sV, dq, staged_k0);
    FF_XTL(4);
    if (own_ok) store_acc_row<T, DH>(dQ + ((long long)b * a.n_q + q) * a.inner + h * DH, dq, 1.f, g);
    FF_XTL(5);

    // ---- own rows = keys (the sample's queries are all in this tile): dK, dV ----
    if constexpr (SINGLE) {
        __syncthreads();
        for (int k0 = 0; k0 < a.n_kv; k0 += kTile) {
            const int key = k0 + m_own;
            const bool kok = key < a.n_kv;
            OwnFrag<T, DH> fk, fv;
            if (FAST && k0 == staged_k0) {          // the tile is still in LDS from the dQ pass (when that pass touched no other tile)
                const bool one_tile = bhi <= staged_k0 + kTile;
                if (one_tile) { fk.template load_tile<L>(sK, m_own, g, kok); fv.template load_tile<L>(sV, m_own, g, kok); }
                else { fk.load(kok ? Kb + (long long)key * a.k.sr : nullptr, g); fv.load(kok ? Vb + (long long)key * a.v.sr : nullptr, g); }
            } else {
                fk.load(kok ? Kb + (long long)key * a.k.sr : nullptr, g);
                fv.load(kok ? Vb + (long long)key * a.v.sr : nullptr, g);
            }
            f32x4 acc_k[NT], acc_v[NT];
#pragma unroll
            for (int dt = 0; dt < NT; dt++) { acc_k[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_v[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            attn_dkv_step<T, DH, L>(key, fk, fv, sQ, sDO, s_lo, s_hi, s_flag, s_lse, s_D, acc_k, acc_v);
            if (kok) {
                store_acc_row<T, DH>(dK + b * a.dk.sb + (long long)key * a.dk.sr + h * a.dk.sh, acc_k, 1.f, g);
                store_acc_row<T, DH>(dV + b * a.dv.sb + (long long)key * a.dv.sr + h * a.dv.sh, acc_v, 1.f, g);
            }
        }
    }
    FF_XTL(6);
}

// =====================================================================================================
// Resident-operand variants (bf16, 64-wide heads, a sample's text in one 32-row tile, its keys in one 64-row tile: the training shape of
// every published configuration - 32 tokens, one image).  One 8-wave workgroup per (sample, head).
//
// The kernels above read the 32 x dim activation rows twice (once from global memory for the LayerNorm statistics, once by DMA as the
// A operand) and stream A and B tiles through one ring with two issuing waves; their timeline is statistics 5.5 us + projection 11 us,
// both bound by how fast ONE CU can pull bytes (~12-20 B/clk), not by MFMA or HBM.  Here the activation rows are brought in ONCE, by DMA
// issued from all eight waves in the first microsecond, and STAY in LDS as the [dim / 64][32][64] K-major operand (80 KiB at dim 1280):
// the LayerNorm statistics are computed from LDS, the rows are normalised in place, and the projection streams only the weight slice
// (64 x dim, 8 KiB per k-step) through a ring that four DMA waves fill while four MFMA waves (2 row blocks x 2 column halves) consume -
// per workgroup 245 KiB of operand traffic instead of 325 KiB, and none of it behind a dependent global round trip.
// =====================================================================================================
namespace {

constexpr int kResBM = 32, kResDH = 64;
// LDS bytes: activation rows + weight ring + fixed tiles (+ LayerNorm vectors and statistics in the forward kernel)
// (gamma / beta are padded to whole wave-level DMA instructions - 64 pieces of 8 elements: a DMA writes all 64 lanes, out-of-range ones as zeros)
constexpr int res_vec_elems(int dim) { return (dim + 511) / 512 * 512; }
constexpr size_t res_lds_fwd(int dim, int nsb) {
    return (size_t)(dim / kBK) * kResBM * kBK * 2 + (size_t)nsb * kResDH * kBK * 2 + 2 * 8192 + (size_t)2 * res_vec_elems(dim) * 2 + 64;
}
constexpr size_t res_lds_bwd(int dim, int nsb) {
    return (size_t)(dim / kBK) * kResBM * kBK * 2 + (size_t)nsb * kResDH * kBK * 2 + 4 * 8192 + (size_t)5 * kTile * 4 + 64;
}

// workgroup barrier that only waits for this wave's LDS operations (a __syncthreads() would also drain the LDS-DMA still in flight)
FF_DEV void res_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// all eight waves: bring rows [0, n_rows) x [0, dim) of `rows` (leading dimension dim) into sA as [dim / 64][32][64] K-major swizzled tiles
FF_DEV void res_issue_rows(const bf16* rows, int dim, int n_rows, bf16* sA, int w, int l) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)rows, 0, 0x7fffffff, 0x00020000);
    const int p = w & 3, row = p * 8 + (l >> 3), cp = l & 7;            // a wave always carries the same 8 rows: one loop-invariant offset
    const unsigned voff = row < n_rows ? (unsigned)(row * dim + ((cp ^ (row & 7)) << 3)) * 2u : kOobOffset;
    const int nk = dim / kBK;
    for (int tile = w >> 2; tile < nk; tile += 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FF_LDS_PTR(void, sA + tile * (kResBM * kBK) + p * 8 * kBK), 16, voff, (unsigned)tile * (kBK * 2), 0, 0);
}

// The projection with a resident A operand: acc[j][r] = sum_k A[rb*16 + c][k] * B[ch*32 + j*16 + g*4 + r][k] for the four MFMA waves
// (cw = rb | ch << 1), the weight tiles streamed by the four DMA waves.  BL 0: B rows = output columns, k contiguous (Wq head slice);
// BL 1: B stored [k][ldb] with the output columns contiguous from n_base (Wo head slice).  `vb` / ring prologue: see res_ring_start.
template <int NSB, int BL>
FF_DEV void res_ring_start(__amdgpu_buffer_rsrc_t rb, bf16* sB, const RowMap& b_map, int n_base, int n_lim, int nk, int pw, int l, unsigned (&vb)[2]) {
    dma_prepare<kResDH, BL, 4>(b_map, n_base, n_lim, pw, l, vb);
    const unsigned b_step = BL == 0 ? 2u : (unsigned)b_map.ld * 2u;
#pragma unroll
    for (int s = 0; s < NSB - 1; s++)
        if (s < nk) dma_tile_fast<kResDH, BL, 4>(rb, sB + s * (kResDH * kBK), vb, (unsigned)(s * kBK) * b_step, pw);
}
template <int NSB, int BL>
FF_DEV void res_project(const bf16* sA, bf16* sB, __amdgpu_buffer_rsrc_t rb, const RowMap& b_map, const unsigned (&vb)[2], int nk, int w, f32x4 (&acc)[2]) {
    const unsigned b_step = BL == 0 ? 2u : (unsigned)b_map.ld * 2u;
    if (w >= 4) {           // DMA waves
        const int pw = w - 4;
        for (int kt = 0; kt < nk; kt++) {
            wait_tiles<2, NSB - 2>(min(nk - 1 - kt, NSB - 2));          // tile kt has landed; up to NSB - 2 younger ones stay in flight
            __builtin_amdgcn_s_barrier();                               // ... for everybody; the MFMA waves have left stage (kt - 1) % NSB
            if (kt + NSB - 1 < nk)
                dma_tile_fast<kResDH, BL, 4>(rb, sB + ((kt + NSB - 1) % NSB) * (kResDH * kBK), vb, (unsigned)((kt + NSB - 1) * kBK) * b_step, pw);
        }
    } else {                // MFMA waves
        const int rb16 = (w & 1) * 16, ch32 = (w >> 1) * 32;
        for (int kt = 0; kt < nk; kt++) {
            __builtin_amdgcn_s_barrier();
            const bf16* sAt = sA + kt * (kResBM * kBK);
            const bf16* sBt = sB + (kt % NSB) * (kResDH * kBK);
#pragma unroll
            for (int ks = 0; ks < kBK / 32; ks++) {
                const bf16x8 fa = frag_read2<kResBM, 0>(sAt, rb16, ks);
#pragma unroll
                for (int j = 0; j < 2; j++) acc[j] = mfma_bf16(frag_read2<kResDH, BL>(sBt, ch32 + j * 16, ks), fa, acc[j]);      // D[n][m]
            }
        }
    }
}
// the four MFMA waves park their accumulators (x scale) in a SwzLayout tile: row = rb*16 + c, columns ch*32 + j*16 + g*4 .. + 3
FF_DEV void res_park(bf16* tile, const f32x4 (&acc)[2], float scale, int w, int c, int g) {
    const int row = (w & 1) * 16 + c, ch32 = (w >> 1) * 32;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = (bf16)(acc[j][r] * scale);
        *(bf16x4*)(tile + SwzLayout::off(row, ch32 + j * 16 + g * 4)) = v;
    }
}


// Which (head, sample) a workgroup of the resident kernels works on.  Workgroup b runs on XCD b % 8 (speed only) and xcd_remap hands every XCD
// a contiguous chunk of the work list.  Head fastest (the default): an XCD holds all eight heads of batch / 8 samples - a sample's rows enter
// ONE private L2, every head's weight slices (W_q, and W_o / W_q again in phase 2) enter all eight.  xcd_split: an XCD holds FOUR heads of
// batch / 4 samples - the rows enter two L2s, the weight slices four: 1.3 MB x 8 -> x 4 per weight matrix against 2.6 MB x 1 -> x 2 of rows at
// the benchmark's shape (both served by the Infinity Cache after the first fetch).  Correctness does not depend on the placement: the
// exchange of phase 2 goes through write-through stores and sc1 loads.
FF_DEV void res_work_item(const XaFusedArgs& a, int& h, int& b) {
    const int total = a.heads * a.batch;
    const int lin = xcd_remap(blockIdx.x, total);
    if (a.xcd_split && a.heads == 8 && (a.batch & 3) == 0) {
        const int chunk = total >> 3;                   // workgroups per XCD (= batch)
        const int x = lin / chunk, j = lin - x * chunk;
        h = ((x & 1) << 2) | (j & 3);
        b = (x >> 1) * (a.batch >> 2) + (j >> 2);
    } else {
        h = lin % a.heads;
        b = lin / a.heads;
    }
}

// ---- phase 2 of the resident kernels (round 5): the product that runs over ALL heads of a sample, inside the same launch ----
// to_out (forward: attn_out = O . Wo^T, gated_cross_attention.py:124-126) and d LN(y) = scale * dQs . Wq (backward) contract over heads * 64,
// i.e. over what the eight (sample, head) workgroups of a sample produced.  They used to be launches of their own (64 x 64 tiles, 5-8 % of
// the MFMA peak: ~11 us of fixed cost for 1.3 GFLOP).  Here each workgroup publishes its 32 x 64 tile of O (d Q) with write-through
// stores, counts itself in on the sample's counter, and once all `heads` have arrived reads the whole 32 x 512 operand back (sc1 loads:
// MI355X guide, Guideline 16, the write-through form - no fences) and computes the output columns [h, h + 1) * dim / heads of all 32 rows:
// the weight slice (dim / heads rows x 512, or 512 k-rows x dim / heads columns) streams through a 4-deep ring that lives where the
// activation rows were, its first three tiles requested while the attention is still running.
// The counters are caller-owned (ff_xattn_desc.sync), zero when first handed over and never reset: every launch adds exactly `heads` to
// each sample's counter, a workgroup's ticket tells it which multiple of `heads` to wait for, and the comparison is wrap-safe.
// Co-residency.  A waiting workgroup holds its CU (one workgroup per CU: the LDS), so a sample's `heads` workgroups must all get a CU while
// the first of them waits.  Workgroup b is dispatched to XCD b % 8 and every XCD dispatches its own workgroups in blockIdx order; res_work_item
// (through xcd_remap) gives XCD x the work items [x * chunk, (x + 1) * chunk) in that order, head fastest - so an XCD works through its samples
// one after the other, the oldest unfinished sample's remaining heads are always the next workgroups that XCD dispatches, and the launch makes
// progress whenever EVERY XCD can hold `heads` (8) of these workgroups at a time.  (blockIdx-wise a sample's workgroups are b, b + 8, ...,
// b + 56: on a device whose dispatcher did NOT spread consecutive workgroups over 8 XCDs - a partitioned (CPX) device, one XCD - the first 32
// resident workgroups would be heads 0..3 of eight samples and nothing would ever complete.)  The host therefore takes this path only on a
// device that reports 8 XCDs with at least kMinCusPerXcd CUs each (xa_exchange_device_ok); what it cannot see - a CU mask on the stream, another
// process's persistent kernels holding most of an XCD - is caught by the bounded spin below: a violated assumption becomes an error word
// (sync[kSyncStatus], ff_xattn_sync_status) that the Python layer checks after warm-up steps, every N graph replays, at every eager optimizer
// step and at the end of bench.py, not a hung GPU and not silence.
constexpr int kSyncSlots = FF_XATTN_SYNC_SLOTS;         // samples per counter bank: forward bank, backward bank, status word
constexpr int kSyncStatus = 2 * kSyncSlots;
constexpr int kOutNS = 4;                               // ring depth of the phase-2 weight stream
constexpr int kOutMaxPer = 6;                           // 32-column groups per head slice: dim / heads <= 192
constexpr unsigned long long kSpinTicks = 5000000ull;     // 50 ms of the 100 MHz wall clock and ...
constexpr int kSpinPolls = 1 << 14;                      // ... this many polls of its own: the bound of an arrival wait (res_await)

// one accumulator row -> global, write-through (sc1): the line leaves this XCD's L2, every other CU's sc1 load sees it.
// 16-byte stores: lane (c, g) holds columns g*4 .. g*4+3 of every 16-column tile; the lanes g and g ^ 1 swap halves so that the even one
// writes columns g*4 .. g*4+7 of tiles 0 / 2 and the odd one columns (g-1)*4 .. +7 of tiles 1 / 3 (8-byte sc1 stores are one fabric write
// each and drained in 3 us where these take half of that: timeline r5s3 / r5s4)
FF_DEV void store_acc_row_wt(bf16* row, const f32x4 (&acc)[kResDH / 16], float scale, int g) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
    u32x2 v[kResDH / 16];
#pragma unroll
    for (int dt = 0; dt < kResDH / 16; dt++) {
        bf16x4 q;
#pragma unroll
        for (int e = 0; e < 4; e++) q[e] = (bf16)(acc[dt][e] * scale);
        v[dt] = __builtin_bit_cast(u32x2, q);
    }
    const bool odd = g & 1;
#pragma unroll
    for (int pr = 0; pr < kResDH / 32; pr++) {
        const u32x2 send = odd ? v[2 * pr] : v[2 * pr + 1];              // what the partner's store is missing
        u32x2 recv;
        recv[0] = __shfl_xor(send[0], 16, 64);
        recv[1] = __shfl_xor(send[1], 16, 64);
        const u32x4 out = odd ? u32x4{recv[0], recv[1], v[2 * pr + 1][0], v[2 * pr + 1][1]} : u32x4{v[2 * pr][0], v[2 * pr][1], recv[0], recv[1]};
        const int col = (odd ? 2 * pr + 1 : 2 * pr) * 16 + (g & ~1) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(out, r, (unsigned)col * 2u, 0, 16);
    }
}
// thread 0, after a workgroup barrier behind the drained payload stores: count this workgroup in; returns the count to wait for
FF_DEV unsigned res_arrive(unsigned* cnt, unsigned group) {
    const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (ticket / group + 1u) * group;
}
// The wait is bounded in polls AND in time.  A poll (s_sleep + one sc1 load) takes ~0.2 us when a launch is denied co-residency (r6s18: 233 000 -
// 266 000 polls in 50 ms), but its duration follows the memory system's load, so a poll count alone is an uncalibrated bound (rounds 5's 2^18
// polls were ~55 ms there); the constant 100 MHz clock alone is no bound either: it keeps running while a queue is preempted (two processes
// time-slicing one GPU), and a workgroup that comes back from a long slice must not give up on partners that were suspended with it.  So a wait
// gives up after 50 ms of wall time during which it also completed 2^14 polls of its own - three orders of magnitude beyond a legitimate wait (a
// sample's heads are dispatched back to back; the launch itself lasts ~30 us) - and a launch denied co-residency fails within a fraction of a second.
// (s_memrealtime, not wall_clock64(): the latter was hoisted out of the loop when it sat behind the poll-count test - r6s17 - and never fired.)
FF_DEV void res_await(unsigned* cnt, unsigned target, unsigned* status) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int polls = 0;
    while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > kSpinPolls) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) {
                __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
}
// res_issue_rows with agent-scope (sc1) loads: the rows were written by other CUs during this launch
FF_DEV void res_issue_rows_sc1(const bf16* rows, int dim, int n_rows, bf16* sA, int w, int l) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)rows, 0, 0x7fffffff, 0x00020000);
    const int p = w & 3, row = p * 8 + (l >> 3), cp = l & 7;
    const unsigned voff = row < n_rows ? (unsigned)(row * dim + ((cp ^ (row & 7)) << 3)) * 2u : kOobOffset;
    const int nk = dim / kBK;
    for (int tile = w >> 2; tile < nk; tile += 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FF_LDS_PTR(void, sA + tile * (kResBM * kBK) + p * 8 * kBK), 16, voff, (unsigned)tile * (kBK * 2), 0, 16);
}
// The weight slice of phase 2.  BL 0 (forward, Wo [dim][inner]): the slice's rows are the output columns n0 .. n0 + 32 PER - 1, k contiguous:
// one K-major tile [32 PER][64] per k-step.  BL 1 (backward, Wq [inner][dim]): output columns contiguous: PER M-major sub-tiles [64 k][32].
// Either way a DMA wave issues PER instructions per k-step.
template <int PER, int BL>
FF_DEV void res_out_prepare(const RowMap& b_map, int n0, int pw, int l, unsigned (&vb)[kOutMaxPer]) {
    if (BL == 0) dma_prepare<32 * PER, 0, 4>(b_map, n0, n0 + 32 * PER, pw, l, vb);
    else {
#pragma unroll
        for (int sidx = 0; sidx < PER; sidx++) dma_prepare<32, 1, 4>(b_map, n0 + 32 * sidx, n0 + 32 * PER, pw, l, vb + sidx);
    }
}
template <int PER, int BL>
FF_DEV void res_out_issue(__amdgpu_buffer_rsrc_t rb, bf16* ring, const unsigned (&vb)[kOutMaxPer], unsigned b_step, int tile, int pw) {
    bf16* st = ring + (tile % kOutNS) * (32 * PER * kBK);
    const unsigned soff = (unsigned)(tile * kBK) * b_step;
    if (BL == 0) dma_tile_fast<32 * PER, 0, 4>(rb, st, vb, soff, pw);
    else {
#pragma unroll
        for (int sidx = 0; sidx < PER; sidx++) dma_tile_fast<32, 1, 4>(rb, st + sidx * (32 * kBK), vb + sidx, soff, pw);
    }
}
// acc[j][r] = sum_k A2[rb*16 + c][k] * B[n0 + ch * 16 PER + j*16 + g*4 + r][k]   (cw = rb | ch << 1; the four DMA waves keep the ring full).
// On entry the first kOutNS - 1 tiles have been issued AND have landed (the caller waited for vmcnt(0) when it staged A2).
template <int PER, int BL>
FF_DEV void res_out_project(const bf16* sA2, bf16* ring, __amdgpu_buffer_rsrc_t rb, const unsigned (&vb)[kOutMaxPer], unsigned b_step, int nk2, int w,
                            f32x4 (&acc)[kOutMaxPer]) {
    constexpr int STAGE = 32 * PER * kBK;
    if (w >= 4) {
        const int pw = w - 4;
        for (int kt = 0; kt < nk2; kt++) {
            wait_tiles<PER, kOutNS - 2>(min(nk2 - 1 - kt, kOutNS - 2));
            __builtin_amdgcn_s_barrier();
            if (kt + kOutNS - 1 < nk2) res_out_issue<PER, BL>(rb, ring, vb, b_step, kt + kOutNS - 1, pw);
        }
    } else {
        const int rb16 = (w & 1) * 16, ch = w >> 1;
        for (int kt = 0; kt < nk2; kt++) {
            __builtin_amdgcn_s_barrier();
            const bf16* sAt = sA2 + kt * (kResBM * kBK);
            const bf16* sBt = ring + (kt % kOutNS) * STAGE;
#pragma unroll
            for (int ks = 0; ks < kBK / 32; ks++) {
                const bf16x8 fa = frag_read2<kResBM, 0>(sAt, rb16, ks);
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    bf16x8 fb;
                    if (BL == 0) fb = frag_read2<32 * PER, 0>(sBt, ch * (16 * PER) + j * 16, ks);
                    else {
                        const int jn = ch * PER + j;
                        fb = frag_read2<32, 1>(sBt + (jn >> 1) * (32 * kBK), (jn & 1) * 16, ks);
                    }
                    acc[j] = mfma_bf16(fb, fa, acc[j]);                  // D[n][m]
                }
            }
        }
    }
}
// the four MFMA waves park their fp32 tiles as rows of cs + 4 floats (16-byte pieces, conflict-free) for the row-contiguous epilogue
template <int PER> FF_DEV void res_out_park(float* tile, const f32x4 (&acc)[kOutMaxPer], int w, int c, int g) {
    constexpr int LD = 32 * PER + 4;
    const int row = (w & 1) * 16 + c, col0 = (w >> 1) * (16 * PER);
#pragma unroll
    for (int j = 0; j < PER; j++) *(f32x4*)(tile + row * LD + col0 + j * 16 + g * 4) = acc[j];
}
// DMA waves: byte offsets of this wave's pieces + the first kOutNS - 1 tiles of the slice
template <int PER, int BL>
FF_DEV void res_out_start(const RowMap& b_map, int n0, __amdgpu_buffer_rsrc_t rb, bf16* ring, unsigned (&vb)[kOutMaxPer], unsigned b_step, int nk2, int pw, int l) {
    res_out_prepare<PER, BL>(b_map, n0, pw, l, vb);
#pragma unroll
    for (int s2 = 0; s2 < kOutNS - 1; s2++)
        if (s2 < nk2) res_out_issue<PER, BL>(rb, ring, vb, b_step, s2, pw);
}
// all eight waves: the product, then the fp32 tile parked where the ring was (`sP`, rows of 32 PER + 4 floats)
template <int PER, int BL>
FF_DEV void res_out_phase(const bf16* sA2, bf16* ring, __amdgpu_buffer_rsrc_t rb, const unsigned (&vb)[kOutMaxPer], unsigned b_step, int nk2, int w, int c, int g,
                          float* sP) {
    f32x4 acc2[kOutMaxPer];
#pragma unroll
    for (int j = 0; j < kOutMaxPer; j++) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    res_out_project<PER, BL>(sA2, ring, rb, vb, b_step, nk2, w, acc2);
    __syncthreads();                                                   // the ring is dead: it takes the fp32 tile
    if (w < 4) res_out_park<PER>(sP, acc2, w, c, g);
}
#define FF_RES_PER(per, ...)                                         \
    switch (per) {                                                   \
        case 1: { constexpr int PER = 1; __VA_ARGS__; } break;       \
        case 2: { constexpr int PER = 2; __VA_ARGS__; } break;       \
        case 3: { constexpr int PER = 3; __VA_ARGS__; } break;       \
        case 4: { constexpr int PER = 4; __VA_ARGS__; } break;       \
        case 5: { constexpr int PER = 5; __VA_ARGS__; } break;       \
        default: { constexpr int PER = 6; __VA_ARGS__; } break;      \
    }

// ---- phase 3 of the resident kernels (round 5): the LayerNorm behind phase 2's output, inside the same launch ----
// After phase 2 a workgroup holds the columns [n0, n0 + dim / heads) of its sample's 32 rows; a LayerNorm (forward) or its backward needs two
// sums per ROW over all columns.  Each workgroup reduces its slice to a pair per row, publishes the 32 pairs (8-byte write-through stores),
// arrives on the sample's counter of a SECOND bank (the first bank's counter of the same launch is still being waited on by slower heads), and
// once all heads are in reads the heads x 32 pairs back with sc1 loads: 256 bytes out, 2 KiB in per workgroup.
constexpr int kLn3Bank = 2 * kSyncSlots + 64;           // forward bank of the phase-3 counters; the backward bank follows at + kSyncSlots
constexpr int kLn3Items = 32 * 32 * kOutMaxPer / 8;     // 16-byte pieces of a workgroup's output slice: 32 rows x dim / heads / 8 <= 768
struct Ln3Smem {
    float *item_a, *item_b, *stat, *fin, *cols;
};
FF_DEV Ln3Smem ln3_smem(float* base) { return Ln3Smem{base, base + kLn3Items, base + 2 * kLn3Items, base + 2 * kLn3Items + 512, base + 2 * kLn3Items + 576}; }
// thread r < n_rows holds row r's pair (va, vb); on return stat[(r * heads + hh) * 2 + {0, 1}] = head hh's pair of row r, for every row
FF_DEV void ln3_exchange(float* part, unsigned* cnt, unsigned* status, int b, int h, int heads, int n_rows, int t, float va, float vb, float* stat) {
    // one naturally aligned 8-byte granule per (sample, head, row), written and read with relaxed agent-scope atomics (MI355X guide, Guideline 16:
    // "8-byte agent atomics both sides" - they lower to sc1 accesses and a granule is never torn).  (The first version used the 8-byte raw
    // buffer store / load builtins with the sc1 bit: the second dword never arrived - r5s17: mean right, M2 zero.)
    unsigned long long* granule = (unsigned long long*)part;
    if (t < n_rows)
        __hip_atomic_store(granule + ((long long)(b * heads + h) * 32 + t),
                           ((unsigned long long)__builtin_bit_cast(unsigned, vb) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, va), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the pairs have left the CU
    res_barrier();
    if (t == 0) res_await(cnt, res_arrive(cnt, (unsigned)heads), status);
    res_barrier();
    if (t < n_rows * heads) {
        const int r = t / heads, hh = t - r * heads;
        const unsigned long long v = __hip_atomic_load(granule + ((long long)(b * heads + hh) * 32 + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stat[t * 2] = __builtin_bit_cast(float, (unsigned)(v & 0xffffffffull));
        stat[t * 2 + 1] = __builtin_bit_cast(float, (unsigned)(v >> 32));
    }
    __syncthreads();
}

}  // namespace

// OUTP: phase 2 - to_out + tanh gate + residual for this workgroup's column slice of all 32 rows (see the helpers above); `O` is then
// written through and read back by the sample's other workgroups, so it is NOT __restrict__ in that instantiation's eyes.
template <int NSB, bool OUTP>
__global__ __launch_bounds__(512) void xa_qattn_fwd_res_kernel(const XaFusedArgs a_in, const bf16* __restrict__ y, const bf16* __restrict__ gamma,
                                                               const bf16* __restrict__ beta, const bf16* __restrict__ Wq, const bf16* __restrict__ K,
                                                               const bf16* __restrict__ V, const int* __restrict__ tt, bf16* __restrict__ yn,
                                                               bf16* __restrict__ Qs, bf16* O, float* __restrict__ mean,
                                                               float* __restrict__ rstd, float* __restrict__ lse, const XaOutArgs o_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FF_XTL(0);
    const XaFusedArgs a = fetch_args(a_in);
    constexpr int DH = kResDH, BM = kResBM, NT = DH / 16;
    int h, b;
    res_work_item(a, h, b);
    const int t = threadIdx.x, l = t & 63, c = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nk = a.dim / kBK, n_rows = a.n_q;
    bf16* sA = (bf16*)smem;
    bf16* sB = sA + nk * (BM * kBK);
    bf16* sK = sB + NSB * (DH * kBK);
    bf16* sV = sK + SwzLayout::tile_elems;
    bf16* s_g = sV + SwzLayout::tile_elems;
    bf16* s_b = s_g + res_vec_elems(a.dim);                             // (each padded to whole DMA instructions, see res_vec_elems)
    bf16* sQ = sB;                                                      // the weight ring is dead when Q is parked

    // ---- every byte the workgroup will read is requested now: its text_time row, the activation rows, gamma / beta, K / V, the first weight tiles ----
    const int m_own = w * 16 + c;                                      // own query of an attention wave (waves 0, 1)
    const bool own_ok = w < 2 && m_own < n_rows;
    const ff_attn_desc d = attn_view(a, DH);
    RowRange rr = row_range(d, tt, b, own_ok ? m_own : a.n_q);
    const bf16* yb = y + (long long)b * a.n_q * a.dim;
    res_issue_rows(yb, a.dim, n_rows, sA, w, l);
    {   // gamma, beta: dim / 8 sixteen-byte pieces each, lane-linear in LDS
        const int nch = a.dim / 8;
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gamma, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rbt = __builtin_amdgcn_make_buffer_rsrc((void*)beta, 0, 0x7fffffff, 0x00020000);
        for (int i0 = w * 64; i0 < nch; i0 += 512) {
            const unsigned off = i0 + l < nch ? (unsigned)(i0 + l) * 16u : kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, FF_LDS_PTR(void, s_g + i0 * 8), 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rbt, FF_LDS_PTR(void, s_b + i0 * 8), 16, off, 0, 0, 0);
        }
    }
    const bf16* Kb = K + b * a.k.sb + h * a.k.sh;
    const bf16* Vb = V + b * a.v.sb + h * a.v.sh;
    {   // the (single) key tile: waves 0-3 carry K, waves 4-7 V
        const bf16* src = w < 4 ? Kb : Vb;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
        dma_tile<64, 0, 4>(rk, w < 4 ? sK : sV, RowMap{w < 4 ? a.k.sr : a.v.sr, 0, 0}, 0, a.n_kv, 0, kBK, w & 3, l);
    }
    const bf16* Wh = Wq + (long long)h * DH * a.dim;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)Wh, 0, 0x7fffffff, 0x00020000);
    const RowMap w_map{a.dim, 0, 0};
    unsigned vb[2] = {kOobOffset, kOobOffset};
    if (w >= 4) {
        res_ring_start<NSB, 0>(rw, sB, w_map, 0, DH, nk, w - 4, l, vb);
        wait_vmcnt<2 * (NSB - 1)>();                                   // everything but the weight tiles (issued last) has landed (nk >= NSB - 1)
    } else wait_vmcnt<0>();
    res_barrier();                                                     // (raw barrier: the weight tiles stay in flight across it)
    FF_XTL(1);

    // ---- LayerNorm of the rows, from LDS and in place: 16 threads per row ----
    {
        constexpr int MAXC = 12;                                       // 16-byte pieces per thread: dim <= 1536
        const int r = t >> 4, s16 = t & 15;
        const int nch = nk * 8;
        // statistics in ONE pass around the row's first element (sums of (x - c) and (x - c)^2), eight independent accumulators per thread:
        // round 3's two passes were two serial chains of up to 96 dependent additions per thread (round 4: ff_decode.hip's probe put the same
        // code at half of that kernel's time)
        uint4 raw[MAXC];
        // the shift: mean of the row's first 16 elements (chunks 0 and 1 sit in slots 0 ^ (r & 7) and 1 ^ (r & 7)).  One element as the shift
        // loses the variance to cancellation when that element is an outlier of its row (massive-activation channels of an LM's residual stream).
        float shift;
        {
            float h0[8], h1[8];
            unpack16(*(const uint4*)(sA + r * kBK + ((r & 7) << 3)), h0, bf16());
            unpack16(*(const uint4*)(sA + r * kBK + ((1 ^ (r & 7)) << 3)), h1, bf16());
            float hs = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) hs += h0[e] + h1[e];
            shift = hs * (1.f / 16.f);
        }
        float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < MAXC; u++) {
            const int ci = s16 + 16 * u;
            if (ci < nch) {
                raw[u] = *(const uint4*)(sA + (ci >> 3) * (BM * kBK) + r * kBK + (((ci & 7) ^ (r & 7)) << 3));
                float v[8];
                unpack16(raw[u], v, bf16());
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float d = v[e] - shift;
                    s1[e] += d;
                    s2[e] = fmaf(d, d, s2[e]);
                }
            }
        }
        float sum = ((s1[0] + s1[1]) + (s1[2] + s1[3])) + ((s1[4] + s1[5]) + (s1[6] + s1[7]));
        float sq = ((s2[0] + s2[1]) + (s2[2] + s2[3])) + ((s2[4] + s2[5]) + (s2[6] + s2[7]));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
        const float dm = sum / (float)a.dim;                            // mean - shift
        const float mu = shift + dm;
        const float rs = rsqrtf(fmaxf(sq / (float)a.dim - dm * dm, 0.f) + a.eps);
        const bool rok = r < n_rows;
        const long long grow = (long long)b * a.n_q + (rok ? r : 0);
        if (s16 == 0 && rok && h == 0) { mean[grow] = mu; rstd[grow] = rs; }
        bf16* ynr = yn ? yn + grow * a.dim : nullptr;
#pragma unroll
        for (int u = 0; u < MAXC; u++) {
            const int ci = s16 + 16 * u;
            if (ci < nch) {
                float v[8], gv[8], bv[8];
                unpack16(raw[u], v, bf16());
                Vec<bf16>::load(s_g + ci * 8, gv);
                Vec<bf16>::load(s_b + ci * 8, bv);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (v[e] - mu) * rs * gv[e] + bv[e];
                Vec<bf16>::store(sA + (ci >> 3) * (BM * kBK) + r * kBK + (((ci & 7) ^ (r & 7)) << 3), v);
                // the normalised rows are an operand of d to_q.weight: each head's workgroup writes its share of a row's pieces
                if (ynr && rok && (ci % a.heads) == h) Vec<bf16>::store(ynr + ci * 8, v);
            }
        }
    }
    res_barrier();
    FF_XTL(2);

    // ---- q[m][n] = sum_k LN(y)[m][k] Wq[h*DH + n][k] ----
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    res_project<NSB, 0>(sA, sB, rw, w_map, vb, nk, w, acc);
    __syncthreads();                                                   // the ring is dead: it becomes the Q tile
    FF_XTL(3);
    if (w < 4) res_park(sQ, acc, a.scale, w, c, g);
    res_barrier();                                                     // (raw: phase 2's weight tiles, requested next, stay in flight across later barriers)
    // ---- phase 2, requests first: the activation rows are dead too - their place takes the ring of the Wo slice (rows n0 .. n0 + cs - 1) ----
    const XaOutArgs oa = OUTP ? fetch_args(o_in) : XaOutArgs{};
    const int cs = a.dim / a.heads, per = cs / 32, n0 = h * cs, nk2 = a.inner / kBK;
    bf16* ring2 = sA;
    unsigned vo[kOutMaxPer] = {kOobOffset, kOobOffset, kOobOffset, kOobOffset, kOobOffset, kOobOffset};
    const __amdgpu_buffer_rsrc_t rwo = __builtin_amdgcn_make_buffer_rsrc((void*)(OUTP ? oa.W : Wq), 0, 0x7fffffff, 0x00020000);
    const RowMap wo_map{a.inner, 0, 0};
    if (OUTP && w >= 4) {
        FF_RES_PER(per, (res_out_start<PER, 0>(wo_map, n0, rwo, ring2, vo, 2u, nk2, w - 4, l)));
    }
    if (t < 256) {                                                     // saved for backward: 32 rows x 8 pieces
        const int r = t >> 3, ch = t & 7;
        if (r < n_rows) *(uint4*)(Qs + ((long long)b * a.n_q + r) * a.inner + h * DH + ch * 8) = *(const uint4*)(sQ + SwzLayout::off(r, ch * 8));
    }
    // ---- O = softmax_masked(q K^T) V: waves 0 and 1, 16 own queries each, the key tile already in LDS ----
    if (w < 2) {
        if (!own_ok) { rr.lo = rr.hi = 0; rr.softmax = 0; rr.uniform = 0; }
        OwnFrag<bf16, DH> fq;
        fq.template load_tile<SwzLayout>(sQ, m_own, g, own_ok);
        f32x4 o[NT];
#pragma unroll
        for (int dt = 0; dt < NT; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m = kNegBig, lsum = 0.f;
        const int bhi = __any(rr.hi > rr.lo) ? min(a.n_kv, kTile) : 0;
        attn_fwd_loop<bf16, DH, DmaStage64>(d, fq, rr, 0, bhi, Kb, Vb, sK, sV, o, m, lsum, 0);
        lsum = group_sum(lsum);
        if (own_ok) {
            const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
            bf16* orow = O + ((long long)b * a.n_q + m_own) * a.inner + h * DH;
            if (OUTP) store_acc_row_wt(orow, o, inv, g);
            else store_acc_row<bf16, DH>(orow, o, inv, g);
            if (g == 0) lse[((long long)b * a.heads + h) * a.n_q + m_own] = lsum > 0.f ? m + __logf(lsum) : kPosBig;
        }
        if (OUTP) wait_vmcnt<0>();                                     // this wave's share of O has left the CU
    }
    FF_XTL(4);
    if (!OUTP) return;

    // ---- phase 2: attn_out = O . Wo^T for the columns [n0, n0 + cs) of the sample's rows; y1 = y + tanh(alpha) * attn_out ----
    res_barrier();                                                     // the workgroup's tile of O is out
    unsigned* cnt = oa.sync + (b % kSyncSlots);
    if (t == 0) res_await(cnt, res_arrive(cnt, (unsigned)a.heads), oa.sync + kSyncStatus);
    res_barrier();                                                     // every head's tile of this sample is out
    FF_XTL(5);
    bf16* sA2 = sB;                                                    // [inner / 64][32][64]: the dead weight ring, Q tile and key tiles
    res_issue_rows_sc1(O + (long long)b * a.n_q * a.inner, a.inner, n_rows, sA2, w, l);
    wait_vmcnt<0>();                                                   // (also: the first three tiles of the Wo slice, requested long ago)
    res_barrier();
    FF_XTL(6);
    float* sP = (float*)ring2;
    // the residual rows of the epilogue are requested now (at most two 16-byte pieces per thread: 32 rows x cs / 8 <= 768 pieces) and
    // arrive under the product (timeline r5s3: the epilogue was 1.35 us with the loads inside it, 0.4 us in the backward kernel without any)
    const int cpr = cs / 8, ld = cs + 4, n_items = n_rows * cpr;
    const bool ln3 = oa.ln_out != nullptr;                             // phase 3: LN(y1) of the feed-forward in this launch too
    uint4 yreg[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}}, greg[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}}, breg[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}};
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = t + u * 512;
        if (i < n_items) {
            const int r = i / cpr, c8 = i - r * cpr;
            yreg[u] = *(const uint4*)(y + ((long long)b * a.n_q + r) * a.dim + n0 + c8 * 8);
            if (ln3) {
                greg[u] = *(const uint4*)(oa.ln_g + n0 + c8 * 8);
                breg[u] = *(const uint4*)(oa.ln_b + n0 + c8 * 8);
            }
        }
    }
    FF_RES_PER(per, (res_out_phase<PER, 0>(sA2, ring2, rwo, vo, 2u, nk2, w, c, g, sP)));
    __syncthreads();
    FF_XTL(7);
    float y1r[2][8];                                                   // the workgroup's slice of y1 as stored (bf16-rounded): what LN(y1) sees
    {
        const float gt = tanhf(to_f32(oa.gate[0]));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = t + u * 512;
            if (i < n_items) {
                const int r = i / cpr, c8 = i - r * cpr;
                const float* src = sP + r * ld + c8 * 8;
                const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
                const long long go = ((long long)b * a.n_q + r) * a.dim + n0 + c8 * 8;
                float yv[8], av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, ov[8];
                unpack16(yreg[u], yv, bf16());
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    ov[e] = av[e] * gt + yv[e];
                    y1r[u][e] = (float)(bf16)ov[e];
                }
                Vec<bf16>::store(oa.aux + go, av);                     // to_out(attention): an operand of d alpha_attn
                Vec<bf16>::store(oa.out + go, ov);
            }
        }
    }
    FF_XTL(8);
    if (!ln3) return;

    // ---- phase 3: xn = LN(y1) (utils.py:46) for the same slice.  Per row: mean and M2 of the slice's cs columns, exact two-pass; the heads'
    //      pairs are combined as equal-sized groups (Chan et al.): mean = avg(mean_h), M2 = sum(M2_h) + cs * sum((mean_h - mean)^2) ----
    const Ln3Smem L3 = ln3_smem((float*)sB);                           // the operand rows of phase 2 are dead (res_out_phase ended behind a barrier)
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = t + u * 512;
        if (i < n_items) {
            float sx = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) sx += y1r[u][e];
            L3.item_a[i] = sx;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = t + u * 512;
        if (i < n_items) {
            const int r = i / cpr;
            float sx = 0.f;
            for (int q = 0; q < cpr; q++) sx += L3.item_a[r * cpr + q];
            const float mw = sx / (float)cs;
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) m2 = fmaf(y1r[u][e] - mw, y1r[u][e] - mw, m2);
            L3.item_b[i] = m2;
        }
    }
    __syncthreads();
    float pa = 0.f, pb = 0.f;
    if (t < n_rows) {
        for (int q = 0; q < cpr; q++) { pa += L3.item_a[t * cpr + q]; pb += L3.item_b[t * cpr + q]; }
        pa /= (float)cs;
    }
    ln3_exchange(oa.ln_part, oa.sync + kLn3Bank + (b % kSyncSlots), oa.sync + kSyncStatus, b, h, a.heads, n_rows, t, pa, pb, L3.stat);
    if (t < n_rows) {
        float mean = 0.f, m2 = 0.f;
        for (int hh = 0; hh < a.heads; hh++) mean += L3.stat[(t * a.heads + hh) * 2];
        mean /= (float)a.heads;
        for (int hh = 0; hh < a.heads; hh++) {
            const float dm = L3.stat[(t * a.heads + hh) * 2] - mean;
            m2 += L3.stat[(t * a.heads + hh) * 2 + 1] + (float)cs * dm * dm;
        }
        const float rs = rsqrtf(m2 / (float)a.dim + a.eps);
        L3.fin[2 * t] = mean;
        L3.fin[2 * t + 1] = rs;
        if (h == 0) {
            oa.ln_mean[(long long)b * a.n_q + t] = mean;
            oa.ln_rstd[(long long)b * a.n_q + t] = rs;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = t + u * 512;
        if (i < n_items) {
            const int r = i / cpr, c8 = i - r * cpr;
            const float mean = L3.fin[2 * r], rs = L3.fin[2 * r + 1];
            float gv[8], bv[8], xn[8];
            unpack16(greg[u], gv, bf16());
            unpack16(breg[u], bv, bf16());
#pragma unroll
            for (int e = 0; e < 8; e++) xn[e] = (y1r[u][e] - mean) * rs * gv[e] + bv[e];
            Vec<bf16>::store(oa.ln_out + ((long long)b * a.n_q + r) * a.dim + n0 + c8 * 8, xn);
        }
    }
    FF_XTL(9);
}

// OUTP: phase 2 - d LN(y) = scale * dQs . Wq for this workgroup's column slice of all 32 rows (dQ is written through and read back).
template <int NSB, bool OUTP>
__global__ __launch_bounds__(512) void xa_dattn_bwd_res_kernel(const XaFusedArgs a_in, const bf16* __restrict__ dy1, const bf16* __restrict__ Wo,
                                                               const bf16* __restrict__ gate, const bf16* __restrict__ Qs, const bf16* __restrict__ K,
                                                               const bf16* __restrict__ V, const int* __restrict__ tt, const bf16* __restrict__ O,
                                                               const float* __restrict__ lse, bf16* dQ, bf16* __restrict__ dK,
                                                               bf16* __restrict__ dV, float* __restrict__ Dsum, const XaOutArgs o_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FF_XTL(0);
    const XaFusedArgs a = fetch_args(a_in);
    constexpr int DH = kResDH, BM = kResBM, NT = DH / 16;
    typedef SwzLayout L;
    int h, b;
    res_work_item(a, h, b);
    const int t = threadIdx.x, l = t & 63, c = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nk = a.dim / kBK, n_rows = a.n_q;
    bf16* sA = (bf16*)smem;
    bf16* sB = sA + nk * (BM * kBK);
    bf16* sQ = sB + NSB * (DH * kBK);
    bf16* sK = sQ + L::tile_elems;
    bf16* sV = sK + L::tile_elems;
    bf16* sO = sV + L::tile_elems;
    int* s_lo = (int*)(sO + L::tile_elems);
    int* s_hi = s_lo + kTile;
    int* s_flag = s_hi + kTile;
    float* s_lse = (float*)(s_flag + kTile);
    float* s_D = s_lse + kTile;
    bf16* sDO = sB;                                                     // the weight ring is dead when dO is parked

    // ---- requests first: text_time / lse of the own query, the d y1 rows, Q / K / V / O tiles, the first weight tiles ----
    const int m_own = (w & 3) * 16 + c;                                 // waves 0-3: own query (dQ pass) resp. own key (dK / dV pass)
    const bool att = w < 4;
    const bool own_ok = att && m_own < n_rows;
    const ff_attn_desc d = attn_view(a, DH);
    const RowRange rr_full = row_range(d, tt, b, own_ok ? m_own : a.n_q);
    RowRange rr = rr_full;
    if (!own_ok || !rr.softmax) rr.lo = rr.hi = 0;                      // zero / uniform rows: no gradient reaches the scores
    const long long sidx = ((long long)b * a.heads + h) * a.n_q + m_own;
    const float Lq = own_ok ? lse[sidx] : kPosBig;
    if (t < kTile) { s_lo[t] = 0; s_hi[t] = 0; s_flag[t] = 0; s_lse[t] = kPosBig; s_D[t] = 0.f; }
    res_issue_rows(dy1 + (long long)b * a.n_q * a.dim, a.dim, n_rows, sA, w, l);
    const bf16* Kb = K + b * a.k.sb + h * a.k.sh;
    const bf16* Vb = V + b * a.v.sb + h * a.v.sh;
    const bf16* Qb = Qs + (long long)b * a.n_q * a.inner + h * DH;
    const bf16* Ob = O + (long long)b * a.n_q * a.inner + h * DH;
    {   // four 64-row tiles, two waves each
        const int which = w >> 1;                                       // 0: Q, 1: K, 2: V, 3: O
        const bf16* src = which == 0 ? Qb : which == 1 ? Kb : which == 2 ? Vb : Ob;
        bf16* dst = which == 0 ? sQ : which == 1 ? sK : which == 2 ? sV : sO;
        const long long sr = which == 0 || which == 3 ? (long long)a.inner : which == 1 ? a.k.sr : a.v.sr;
        const int lim = which == 0 || which == 3 ? a.n_q : a.n_kv;
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
        dma_tile<64, 0, 2>(rt, dst, RowMap{sr, 0, 0}, 0, lim, 0, kBK, w & 1, l);
    }
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)Wo, 0, 0x7fffffff, 0x00020000);
    const RowMap w_map{a.inner, 0, 0};
    unsigned vb[2] = {kOobOffset, kOobOffset};
    if (w >= 4) {
        res_ring_start<NSB, 1>(rw, sB, w_map, h * DH, (h + 1) * DH, nk, w - 4, l, vb);
        wait_vmcnt<2 * (NSB - 1)>();
    } else wait_vmcnt<0>();
    res_barrier();
    FF_XTL(1);

    // ---- dO[m][n] = tanh(alpha) * sum_k dy1[m][k] Wo[k][h*DH + n] ----
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    res_project<NSB, 1>(sA, sB, rw, w_map, vb, nk, w, acc);
    __syncthreads();                                                    // ring dead -> dO tile
    FF_XTL(2);
    const float gt = tanhf(to_f32(gate[0]));
    if (w < 4) res_park(sDO, acc, gt, w, c, g);
    else {                // rows 32 .. 63 of the dO tile do not exist: zero them (they are "other" rows of the dK / dV products)
        const int row = 32 + (w - 4) * 8 + (l >> 3), ch = l & 7;
        *(uint4*)(sDO + row * 64 + ch * 8) = uint4{0, 0, 0, 0};
    }
    __syncthreads();

    // ---- own rows = queries: D = sum_d dO * O, dQ ----
    OwnFrag<bf16, DH> fq, fdo, fo;
    fq.template load_tile<L>(sQ, m_own, g, own_ok);
    fdo.template load_tile<L>(sDO, m_own, g, own_ok);
    fo.template load_tile<L>(sO, m_own, g, own_ok);
    const float Dq = group_sum(fdo.dot(fo));
    if (own_ok && g == 0) {
        if (Dsum) Dsum[sidx] = Dq;
        s_lo[m_own] = rr_full.lo; s_hi[m_own] = rr_full.hi; s_flag[m_own] = rr_full.softmax | (rr_full.uniform << 1);
        s_lse[m_own] = Lq; s_D[m_own] = Dq;
    }
    if (att) {
        f32x4 dq[NT];
#pragma unroll
        for (int dt = 0; dt < NT; dt++) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int bhi = __any(rr.hi > rr.lo) ? min(a.n_kv, kTile) : 0;
        attn_dq_loop<bf16, DH, DmaStage64>(d, fq, fdo, rr, Lq, Dq, 0, bhi, Kb, Vb, sK, sV, dq, 0);
        if (own_ok) {
            bf16* qrow = dQ + ((long long)b * a.n_q + m_own) * a.inner + h * DH;
            if (OUTP) store_acc_row_wt(qrow, dq, 1.f, g);
            else store_acc_row<bf16, DH>(qrow, dq, 1.f, g);
        }
        if (OUTP) wait_vmcnt<0>();                                      // this wave's share of dQ has left the CU
    }
    __syncthreads();                                                    // the per-query tables are complete (and, OUTP, the workgroup's tile of dQ is out)
    FF_XTL(3);
    // ---- phase 2, first half: count this workgroup in and request the first tiles of the Wq slice (columns n0 .. n0 + cs - 1) into the dead
    //      activation rows - the other heads' workgroups arrive while this one computes dK / dV ----
    const XaOutArgs oa = OUTP ? fetch_args(o_in) : XaOutArgs{};
    const int cs = a.dim / a.heads, per = cs / 32, n0 = h * cs, nk2 = a.inner / kBK;
    bf16* ring2 = sA;
    unsigned vo[kOutMaxPer] = {kOobOffset, kOobOffset, kOobOffset, kOobOffset, kOobOffset, kOobOffset};
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc((void*)(OUTP ? oa.W : Wo), 0, 0x7fffffff, 0x00020000);
    const RowMap wq_map{a.dim, 0, 0};
    const unsigned wq_step = (unsigned)a.dim * 2u;
    unsigned* cnt = OUTP ? oa.sync + kSyncSlots + (b % kSyncSlots) : nullptr;
    unsigned target = 0;
    if (OUTP) {
        if (t == 0) target = res_arrive(cnt, (unsigned)a.heads);
        if (w >= 4) FF_RES_PER(per, (res_out_start<PER, 1>(wq_map, n0, rwq, ring2, vo, wq_step, nk2, w - 4, l)));
    }

    // ---- own rows = keys (all of the sample's queries and keys are in this tile): dK, dV ----
    if (att) {
        const int key = m_own;
        const bool kok = key < a.n_kv;
        OwnFrag<bf16, DH> fk, fv;
        fk.template load_tile<L>(sK, key, g, kok);
        fv.template load_tile<L>(sV, key, g, kok);
        f32x4 acc_k[NT], acc_v[NT];
#pragma unroll
        for (int dt = 0; dt < NT; dt++) { acc_k[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_v[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        attn_dkv_step<bf16, DH, L>(key, fk, fv, sQ, sDO, s_lo, s_hi, s_flag, s_lse, s_D, acc_k, acc_v);
        if (kok) {
            store_acc_row<bf16, DH>(dK + b * a.dk.sb + (long long)key * a.dk.sr + h * a.dk.sh, acc_k, 1.f, g);
            store_acc_row<bf16, DH>(dV + b * a.dv.sb + (long long)key * a.dv.sr + h * a.dv.sh, acc_v, 1.f, g);
        }
    }
    FF_XTL(4);
    if (!OUTP) return;

    // ---- phase 2, second half: d LN(y)[m][n] = scale * sum_k dQs[m][k] Wq[k][n] for the columns [n0, n0 + cs) of the sample's rows ----
    res_barrier();                                                      // the tiles in LDS are dead
    if (t == 0) res_await(cnt, target, oa.sync + kSyncStatus);
    res_barrier();                                                      // every head's tile of dQ is out
    FF_XTL(5);
    bf16* sA2 = sB;                                                     // [inner / 64][32][64] over the dead ring / dO tile
    res_issue_rows_sc1(dQ + (long long)b * a.n_q * a.inner, a.inner, n_rows, sA2, w, l);
    wait_vmcnt<0>();
    res_barrier();
    FF_XTL(6);
    float* sP = (float*)ring2;
    const bool ln3 = oa.ln_out != nullptr;                              // phase 3: the backward of LN(y) in this launch too
    const int cpr = cs / 8, ld = cs + 4, n_items = n_rows * cpr;
    // phase 3's other operands are requested ahead of the product: the slice of y (the LayerNorm's input), of gamma, of d y1 (the residual the
    // result is added to) and the rows' saved statistics
    uint4 xreg[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}}, greg[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}}, rreg[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}};
    float mu_r[2] = {0.f, 0.f}, rs_r[2] = {0.f, 0.f};
    if (ln3) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = t + u * 512;
            if (i < n_items) {
                const int r = i / cpr, c8 = i - r * cpr;
                const long long go = ((long long)b * a.n_q + r) * a.dim + n0 + c8 * 8;
                xreg[u] = *(const uint4*)(oa.ln_x + go);
                rreg[u] = *(const uint4*)(oa.ln_res + go);
                greg[u] = *(const uint4*)(oa.ln_g + n0 + c8 * 8);
                mu_r[u] = oa.ln_mean[(long long)b * a.n_q + r];
                rs_r[u] = oa.ln_rstd[(long long)b * a.n_q + r];
            }
        }
    }
    FF_RES_PER(per, (res_out_phase<PER, 1>(sA2, ring2, rwq, vo, wq_step, nk2, w, c, g, sP)));
    __syncthreads();
    FF_XTL(7);
    float dyh[2][8], xh[2][8];                                          // phase 3: d LN(y) * gamma and x-hat of the workgroup's slice
    const Ln3Smem L3 = ln3_smem((float*)sB);                            // the operand rows of phase 2 and every tile of phase 1 are dead
    const int cld = cs + 1;                                             // row pitch of the two column images (d gamma, d beta terms)
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = t + u * 512;
        if (i < n_items) {
            const int r = i / cpr, c8 = i - r * cpr;
            const float* src = sP + r * ld + c8 * 8;
            const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
            float ov[8] = {a0[0] * a.scale, a0[1] * a.scale, a0[2] * a.scale, a0[3] * a.scale, a1[0] * a.scale, a1[1] * a.scale, a1[2] * a.scale, a1[3] * a.scale};
            if (oa.out) Vec<bf16>::store(oa.out + ((long long)b * a.n_q + r) * a.dim + n0 + c8 * 8, ov);
            if (ln3) {
                float xv[8], gv[8], s1 = 0.f, s2 = 0.f;
                unpack16(xreg[u], xv, bf16());
                unpack16(greg[u], gv, bf16());
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float dr = (float)(bf16)ov[e];                // d LN(y) as the separate launch stored it
                    xh[u][e] = (xv[e] - mu_r[u]) * rs_r[u];
                    dyh[u][e] = dr * gv[e];
                    s1 += dyh[u][e];
                    s2 = fmaf(dyh[u][e], xh[u][e], s2);
                    L3.cols[r * cld + c8 * 8 + e] = dr * xh[u][e];                       // d gamma term
                    L3.cols[(kResBM + r) * cld + c8 * 8 + e] = dr;                       // d beta term
                }
                L3.item_a[i] = s1;
                L3.item_b[i] = s2;
            }
        }
    }
    FF_XTL(8);
    if (!ln3) return;

    // ---- phase 3: d y = rstd * (dyh - mean(dyh) - xh * mean(dyh * xh)) + d y1 for the same slice; the row means need every head's sums ----
    __syncthreads();
    float pa = 0.f, pb = 0.f;
    if (t < n_rows)
        for (int q = 0; q < cpr; q++) { pa += L3.item_a[t * cpr + q]; pb += L3.item_b[t * cpr + q]; }
    if (t < cs) {                                                       // the sample's column sums of this slice: d gamma | d beta partials
        float sg = 0.f, sb2 = 0.f;
        for (int r = 0; r < n_rows; r++) { sg += L3.cols[r * cld + t]; sb2 += L3.cols[(kResBM + r) * cld + t]; }
        float* wp = oa.ln_wpart + (long long)b * (2 * a.dim + 2);
        wp[n0 + t] = sg;
        wp[a.dim + n0 + t] = sb2;
    }
    ln3_exchange(oa.ln_part, oa.sync + kLn3Bank + kSyncSlots + (b % kSyncSlots), oa.sync + kSyncStatus, b, h, a.heads, n_rows, t, pa, pb, L3.stat);
    if (t < n_rows) {
        float m1 = 0.f, m2 = 0.f;
        for (int hh = 0; hh < a.heads; hh++) { m1 += L3.stat[(t * a.heads + hh) * 2]; m2 += L3.stat[(t * a.heads + hh) * 2 + 1]; }
        L3.fin[2 * t] = m1 / (float)a.dim;
        L3.fin[2 * t + 1] = m2 / (float)a.dim;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = t + u * 512;
        if (i < n_items) {
            const int r = i / cpr, c8 = i - r * cpr;
            const float m1 = L3.fin[2 * r], m2 = L3.fin[2 * r + 1];
            float rv[8], dx[8];
            unpack16(rreg[u], rv, bf16());
#pragma unroll
            for (int e = 0; e < 8; e++) dx[e] = rs_r[u] * (dyh[u][e] - m1 - xh[u][e] * m2) + rv[e];
            Vec<bf16>::store(oa.ln_out + ((long long)b * a.n_q + r) * a.dim + n0 + c8 * 8, dx);
        }
    }
    FF_XTL(9);
}

// =====================================================================================================
// host side
// =====================================================================================================
bool xa_fused_supported(int dtype, int dim_head, int dim, int inner) {
    // the LayerNorm pass hands every thread of a row the same number of 16-byte pieces: dim % (8 threads x elements per piece) == 0
    if (dtype == FF_DTYPE_BF16) return (dim_head == 64 || dim_head == 128) && dim % 64 == 0 && inner % 8 == 0;
    if (dtype == FF_DTYPE_F32) return (dim_head == 16 || dim_head == 32 || dim_head == 64 || dim_head == 128) && dim % 32 == 0 && inner % 4 == 0;
    return false;
}

template <typename T, int DH, int BM> static size_t fwd_lds(int dim) {
    const int dimp = (dim + kBK - 1) / kBK * kBK;
    return Carve<T, DH, BM, ring_stages<BM, DH>(), 1, 2, 2>::fixed + (size_t)2 * dimp * sizeof(T) + (size_t)2 * BM * sizeof(float) + 8 * sizeof(int);
}
template <typename T, int DH, int BM> static size_t bwd_lds() {
    return Carve<T, DH, BM, ring_stages<BM, DH>(), 1, 3, 4>::fixed + (size_t)5 * kTile * 4 + 8 * sizeof(int);
}

template <typename KernelT> static int allow_lds(KernelT kernel, size_t lds, const char* what) {
    if (lds > 64 * 1024) {      // per kernel instantiation (this function template) and per device; the largest request so far is remembered
        static size_t allowed[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || allowed[dev] < lds) {
            hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "hipFuncSetAttribute(%s, lds=%zu): %s", what, lds, hipGetErrorString(e));
            if (dev >= 0 && dev < 64) allowed[dev] = lds;
        }
    }
    FF_CHECK(lds <= 160 * 1024, FF_ERR_UNSUPPORTED, "%s needs %zu bytes of LDS (dim too large for the fused kernel)", what, lds);
    return FF_OK;
}

#define FF_XA_DISPATCH(dtype, dim_head, ...)                                                                         \
    do {                                                                                                             \
        if ((dtype) == FF_DTYPE_BF16) {                                                                              \
            typedef bf16 T;                                                                                          \
            if ((dim_head) == 64) { constexpr int DH = 64; __VA_ARGS__; }                                                   \
            else { constexpr int DH = 128; __VA_ARGS__; }                                                                   \
        } else {                                                                                                     \
            typedef float T;                                                                                         \
            if ((dim_head) == 64) { constexpr int DH = 64; __VA_ARGS__; }                                                   \
            else if ((dim_head) == 16) { constexpr int DH = 16; __VA_ARGS__; }                                              \
            else if ((dim_head) == 32) { constexpr int DH = 32; __VA_ARGS__; }                                              \
            else { constexpr int DH = 128; __VA_ARGS__; }                                                                   \
        }                                                                                                            \
    } while (0)

template <typename T, int DH, int BM>
static int launch_fwd(const XaFusedArgs& a, const void* y, const void* gamma, const void* beta, const void* Wq, const void* K, const void* V,
                      const int* tt, void* yn, void* Qs, void* O, float* mean, float* rstd, float* lse, hipStream_t st) {
    const size_t lds = fwd_lds<T, DH, BM>(a.dim);
    auto kernel = xa_qattn_fwd_kernel<T, DH, BM>;
    FF_TRY(allow_lds(kernel, lds, "xa_qattn_fwd"));
    const dim3 grid(cdiv(a.n_q, BM) * a.heads * a.batch);
    kernel<<<grid, dim3(256), lds, st>>>(a, (const T*)y, (const T*)gamma, (const T*)beta, (const T*)Wq, (const T*)K, (const T*)V, tt, (T*)yn, (T*)Qs,
                                         (T*)O, mean, rstd, lse);
    return check_launch("xa_qattn_fwd");
}

// ring depth of the resident-operand kernels for this problem, 0 = not applicable (bf16, 64-wide heads, one 32-row text tile, one key tile,
// rows 16-byte aligned in every tensor, everything within the 160 KiB of LDS)
static int res_ring_depth(const XaFusedArgs& a, int dtype, int dim_head, bool bwd) {
    static const int on = dbg_switch("FF_XATTN_RES", 1);
    if (!on || dtype != FF_DTYPE_BF16 || dim_head != 64 || a.n_q > kResBM || a.n_kv > kTile || a.dim % kBK != 0 || a.dim > 1536 || a.inner % 8 != 0) return 0;
    if ((a.k.sr | a.k.sh | a.k.sb | a.v.sr | a.v.sh | a.v.sb | a.dk.sr | a.dk.sh | a.dk.sb | a.dv.sr | a.dv.sh | a.dv.sb) % 8 != 0) return 0;
    const int nk = a.dim / kBK;
    for (int nsb : {6, 4})
        if (nk >= nsb && (bwd ? res_lds_bwd(a.dim, nsb) : res_lds_fwd(a.dim, nsb)) <= 160 * 1024) return nsb;
    return 0;
}
template <int NSB, bool OUTP>
static int launch_fwd_res(const XaFusedArgs& a, const void* y, const void* gamma, const void* beta, const void* Wq, const void* K, const void* V,
                          const int* tt, void* yn, void* Qs, void* O, float* mean, float* rstd, float* lse, const XaOutArgs& out, hipStream_t st) {
    const size_t lds = res_lds_fwd(a.dim, NSB);
    auto kernel = xa_qattn_fwd_res_kernel<NSB, OUTP>;
    FF_TRY(allow_lds(kernel, lds, "xa_qattn_fwd_res"));
    kernel<<<dim3(a.heads * a.batch), dim3(512), lds, st>>>(a, (const bf16*)y, (const bf16*)gamma, (const bf16*)beta, (const bf16*)Wq, (const bf16*)K,
                                                             (const bf16*)V, tt, (bf16*)yn, (bf16*)Qs, (bf16*)O, mean, rstd, lse, out);
    return check_launch("xa_qattn_fwd_res");
}
template <int NSB, bool OUTP>
static int launch_bwd_res(const XaFusedArgs& a, const void* dy1, const void* Wo, const void* gate, const void* Qs, const void* K, const void* V,
                          const int* tt, const void* O, const float* lse, void* dQ, void* dK, void* dV, float* Dsum, const XaOutArgs& out, hipStream_t st) {
    const size_t lds = res_lds_bwd(a.dim, NSB);
    auto kernel = xa_dattn_bwd_res_kernel<NSB, OUTP>;
    FF_TRY(allow_lds(kernel, lds, "xa_dattn_bwd_res"));
    kernel<<<dim3(a.heads * a.batch), dim3(512), lds, st>>>(a, (const bf16*)dy1, (const bf16*)Wo, (const bf16*)gate, (const bf16*)Qs, (const bf16*)K,
                                                             (const bf16*)V, tt, (const bf16*)O, lse, (bf16*)dQ, (bf16*)dK, (bf16*)dV, Dsum, out);
    return check_launch("xa_dattn_bwd_res");
}

// Phase 2 needs the resident kernels in BOTH directions (a block that fused to_out forward must find d LN(y) fused backward), eight heads
// of 64 (the contraction is 8 k-steps, the operand of all heads 32 KiB), a head's column slice dim / 8 that is a whole number of 32-column
// groups, and one counter per sample.
size_t xa_ln3_part_bytes(int batch, int heads) { return (size_t)batch * heads * 32 * 2 * sizeof(float); }
// The in-launch exchange needs the device described at "Co-residency" above: 8 XCDs (xcd_remap's constant) with room for a sample's 8
// workgroups on each.  Asked once per device; a device that does not qualify keeps the separate launches.
constexpr int kMinCusPerXcd = 16;
static bool xa_exchange_device_ok() {
    static std::atomic<int> cache[64];          // 0 = not asked, 1 = ok, 2 = no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        int cus = 0, xccs = 0;
        const bool have = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                          hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess;
        c = (have && xccs == 8 && cus / 8 >= kMinCusPerXcd) ? 1 : 2;
        cache[dev].store(c, std::memory_order_relaxed);
    }
    return c == 1;
}
bool xa_out_fusable(const XaFusedArgs& a, int dtype, int dim_head) {
    static const int on = dbg_switch("FF_XATTN_OUTFUSE", 1);
    if (!on || !xa_exchange_device_ok()) return false;
    if (a.heads != 8 || a.inner != 8 * kResDH || a.dim % 256 != 0 || a.dim / 8 > 32 * kOutMaxPer || a.batch > kSyncSlots) return false;
    return res_ring_depth(a, dtype, dim_head, false) != 0 && res_ring_depth(a, dtype, dim_head, true) != 0;
}

int xa_qattn_fwd(const XaFusedArgs& a, int dtype, int dim_head, const void* y, const void* gamma, const void* beta, const void* Wq, const void* K,
                 const void* V, const int* tt, void* yn, void* Qs, void* O, float* mean, float* rstd, float* lse, hipStream_t st, const XaOutArgs* out) {
    FF_CHECK(xa_fused_supported(dtype, dim_head, a.dim, a.inner), FF_ERR_UNSUPPORTED, "xa_qattn_fwd: unsupported dtype / head size");
    FF_CHECK(y && gamma && beta && Wq && K && V && tt && Qs && O && mean && rstd && lse, FF_ERR_SHAPE, "xa_qattn_fwd: null argument");
    FF_CHECK(!out || (xa_out_fusable(a, dtype, dim_head) && out->W && out->gate && out->out && out->aux && out->sync), FF_ERR_SHAPE,
             "xa_qattn_fwd: phase 2 (to_out inside the launch) asked for a problem that does not take it, or with a null argument");
    FF_CHECK(!out || !out->ln_out || (out->ln_g && out->ln_b && out->ln_mean && out->ln_rstd && out->ln_part), FF_ERR_SHAPE,
             "xa_qattn_fwd: phase 3 (LayerNorm of the feed-forward inside the launch) with a null argument");
    const int pid = profile_begin(dtype, out ? (out->ln_out ? -8 : -6) : -4, a.heads, 0, a.n_q, a.n_kv, a.dim, a.batch * a.heads, dim_head, st);
    int rc;
    const int nsb = res_ring_depth(a, dtype, dim_head, false);
    const XaOutArgs none = {};
    if (nsb == 6 && out) rc = launch_fwd_res<6, true>(a, y, gamma, beta, Wq, K, V, tt, yn, Qs, O, mean, rstd, lse, *out, st);
    else if (nsb == 4 && out) rc = launch_fwd_res<4, true>(a, y, gamma, beta, Wq, K, V, tt, yn, Qs, O, mean, rstd, lse, *out, st);
    else if (nsb == 6) rc = launch_fwd_res<6, false>(a, y, gamma, beta, Wq, K, V, tt, yn, Qs, O, mean, rstd, lse, none, st);
    else if (nsb == 4) rc = launch_fwd_res<4, false>(a, y, gamma, beta, Wq, K, V, tt, yn, Qs, O, mean, rstd, lse, none, st);
    else if (a.n_q <= 32) FF_XA_DISPATCH(dtype, dim_head, rc = (launch_fwd<T, DH, 32>(a, y, gamma, beta, Wq, K, V, tt, yn, Qs, O, mean, rstd, lse, st)));
    else FF_XA_DISPATCH(dtype, dim_head, rc = (launch_fwd<T, DH, 64>(a, y, gamma, beta, Wq, K, V, tt, yn, Qs, O, mean, rstd, lse, st)));
    profile_end(pid, st);
    return rc;
}

template <typename T, int DH, int BM, bool SINGLE>
static int launch_bwd(const XaFusedArgs& a, const void* dy1, const void* Wo, const void* gate, const void* Qs, const void* K, const void* V,
                      const int* tt, const void* O, const float* lse, void* dO, void* dQ, void* dK, void* dV, float* Dsum, hipStream_t st) {
    const size_t lds = bwd_lds<T, DH, BM>();
    auto kernel = xa_dattn_bwd_kernel<T, DH, BM, SINGLE>;
    FF_TRY(allow_lds(kernel, lds, "xa_dattn_bwd"));
    const dim3 grid(cdiv(a.n_q, BM) * a.heads * a.batch);
    kernel<<<grid, dim3(256), lds, st>>>(a, (const T*)dy1, (const T*)Wo, (const T*)gate, (const T*)Qs, (const T*)K, (const T*)V, tt, (const T*)O, lse,
                                         (T*)dO, (T*)dQ, (T*)dK, (T*)dV, Dsum);
    return check_launch("xa_dattn_bwd");
}

// *single_tile = 1: d K / d V were produced too (every sample's queries fit one tile); 0: the caller runs the d K / d V kernel on dO / Dsum.
int xa_dattn_bwd(const XaFusedArgs& a, int dtype, int dim_head, const void* dy1, const void* Wo, const void* gate, const void* Qs, const void* K,
                 const void* V, const int* tt, const void* O, const float* lse, void* dO, void* dQ, void* dK, void* dV, float* Dsum,
                 int* single_tile, hipStream_t st, const XaOutArgs* out) {
    FF_CHECK(xa_fused_supported(dtype, dim_head, a.dim, a.inner), FF_ERR_UNSUPPORTED, "xa_dattn_bwd: unsupported dtype / head size");
    FF_CHECK(dy1 && Wo && gate && Qs && K && V && tt && O && lse && dQ && dK && dV && Dsum && single_tile, FF_ERR_SHAPE, "xa_dattn_bwd: null argument");
    FF_CHECK(!out || (xa_out_fusable(a, dtype, dim_head) && out->W && (out->out || out->ln_out) && out->sync), FF_ERR_SHAPE,
             "xa_dattn_bwd: phase 2 (d LN(y) inside the launch) asked for a problem that does not take it, or with a null argument");
    FF_CHECK(!out || !out->ln_out || (out->ln_g && out->ln_x && out->ln_res && out->ln_mean && out->ln_rstd && out->ln_part && out->ln_wpart), FF_ERR_SHAPE,
             "xa_dattn_bwd: phase 3 (LayerNorm backward inside the launch) with a null argument");
    const bool single = a.n_q <= 64;
    FF_CHECK(single || dO, FF_ERR_SHAPE, "xa_dattn_bwd: dO buffer needed when the queries span several tiles");
    *single_tile = single ? 1 : 0;
    const int pid = profile_begin(dtype, out ? (out->ln_out ? -9 : -7) : -5, a.heads, 0, a.n_q, a.n_kv, a.dim, a.batch * a.heads, dim_head, st);
    int rc;
    const int nsb = res_ring_depth(a, dtype, dim_head, true);
    const XaOutArgs none = {};
    if (nsb == 6 && out) rc = launch_bwd_res<6, true>(a, dy1, Wo, gate, Qs, K, V, tt, O, lse, dQ, dK, dV, Dsum, *out, st);
    else if (nsb == 4 && out) rc = launch_bwd_res<4, true>(a, dy1, Wo, gate, Qs, K, V, tt, O, lse, dQ, dK, dV, Dsum, *out, st);
    else if (nsb == 6) rc = launch_bwd_res<6, false>(a, dy1, Wo, gate, Qs, K, V, tt, O, lse, dQ, dK, dV, Dsum, none, st);
    else if (nsb == 4) rc = launch_bwd_res<4, false>(a, dy1, Wo, gate, Qs, K, V, tt, O, lse, dQ, dK, dV, Dsum, none, st);
    else if (a.n_q <= 32) FF_XA_DISPATCH(dtype, dim_head, rc = (launch_bwd<T, DH, 32, true>(a, dy1, Wo, gate, Qs, K, V, tt, O, lse, dO, dQ, dK, dV, Dsum, st)));
    else if (single) FF_XA_DISPATCH(dtype, dim_head, rc = (launch_bwd<T, DH, 64, true>(a, dy1, Wo, gate, Qs, K, V, tt, O, lse, dO, dQ, dK, dV, Dsum, st)));
    else FF_XA_DISPATCH(dtype, dim_head, rc = (launch_bwd<T, DH, 64, false>(a, dy1, Wo, gate, Qs, K, V, tt, O, lse, dO, dQ, dK, dV, Dsum, st)));
    profile_end(pid, st);
    return rc;
}

}  // namespace ff

#ifdef FF_XA_TIMELINE
extern "C" int ff_debug_xa_timeline_read(unsigned long long* out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff::g_xa_timeline), sizeof(unsigned long long) * 16 * n_blocks);
}
#endif
