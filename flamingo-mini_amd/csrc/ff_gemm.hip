// GEMM family of the fusion path: every nn.Linear of the resampler / gated-xattn blocks, forward (X W^T),
// data-gradient (dY W) and weight-gradient (dY^T X), with the elementwise neighbours fused into the epilogue
// (activation, activation-backward, residual, tanh(alpha) gate, branch-output store).
//
//   bf16 : v_mfma_f32_16x16x32_bf16, 128x128 / 64x128 / 64x64 block tiles x K 64, 4 waves (2x2), LDS ring fed by
//          LDS-DMA (buffer_load ... lds).  Operands whose contraction index is the slow (strided) one are
//          staged as-is and read with ds_read_b64_tr_b16.
//   fp32 : v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) — the verification precision.
//
// Accumulators are kept transposed (D[n][m] = mfma(Bfrag, Afrag)) so a lane owns 4 consecutive n of one row m
// and the epilogue issues 8/16-byte row-contiguous loads and stores.
#include <atomic>
#include <type_traits>
#include "ff_common.h"
#include <stdlib.h>
#include "ff_internal.h"
#include "ff_gemm_tiles.h"

namespace ff {

// ------------------------------------------------------------------------------------------------
// epilogue shared by the direct and the split-K paths: 4 consecutive columns n..n+3 of row m
// ------------------------------------------------------------------------------------------------
template <typename T>
FF_DEV void epilogue4(const GemmParams& P, const GemmProblem& pr, int m, int n, float (&v)[4]) {
    typedef __attribute__((ext_vector_type(4))) T vec4;
    const long long off = P.c_map.off(m) + n;
    const bool full = (n + 3 < P.N);
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] *= P.scale;
    auto load4 = [&](const void* base, long long at, float (&o)[4]) {
        const T* p = (const T*)base + at;
        if (full) {
            vec4 t = *(const vec4*)p;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = to_f32(t[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (n + r < P.N) ? to_f32(p[r]) : 0.f;
        }
    };
    auto store4 = [&](void* base, const float (&o)[4]) {
        T* p = (T*)base + off;
        if (full) {
            vec4 t;
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = from_f32<T>(o[r]);
            *(vec4*)p = t;
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (n + r < P.N) p[r] = from_f32<T>(o[r]);
        }
    };
    if (pr.aux_out) store4(pr.aux_out, v);
    if (pr.gate) {
        const float g = tanhf(to_f32(*(const T*)pr.gate));
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] *= g;
    }
    if (P.act >= 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = act_fwd_t<T>(v[r], P.act);
    }
    if (P.act_bwd >= 0) {
        float h[4];
        load4(pr.aux_in, off, h);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] *= act_grad_t<T>(h[r], P.act_bwd);
    }
    if (pr.residual) {
        float q[4];
        load4(pr.residual, P.r_map.off(m) + n, q);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] += q[r];
    }
    store4(pr.C, v);
}

// 8 consecutive columns [n, n + nv) of output row m: scale -> aux_out -> tanh-gate -> act -> act'(aux_in) -> + residual -> C.
// `vec`: all of C / aux / residual may be touched with 16-byte accesses (host flag c_vec8) and nv == 8.  `gate` = tanh(*gate) or 1.
// F < 0: which steps exist is decided at run time; F >= 0: bit mask of kEpi* (the branches fold away, see tile_epilogue_bf16).
constexpr int kEpiAux = 1, kEpiGate = 2, kEpiAct = 4, kEpiActBwd = 8, kEpiRes = 16;
template <typename T, int F = -1>
FF_DEV void epilogue8(const GemmParams& P, const GemmProblem& pr, int m, int n, int nv, bool vec, float gate, float (&v)[8]) {
    constexpr int VN = Vec<T>::N;   // 8 bf16 or 4 floats per 16 bytes
    const bool has_aux = F < 0 ? pr.aux_out != nullptr : (F & kEpiAux) != 0, has_gate = F < 0 || (F & kEpiGate);
    const bool has_act = F < 0 ? P.act >= 0 : (F & kEpiAct) != 0, has_act_bwd = F < 0 ? P.act_bwd >= 0 : (F & kEpiActBwd) != 0;
    const bool has_res = F < 0 ? pr.residual != nullptr : (F & kEpiRes) != 0;
    auto load8 = [&](const void* base, long long at, float (&o)[8]) {
        const T* p = (const T*)base + at;
        if (vec) {
            if constexpr (VN == 8) Vec<T>::load(p, o);
            else {
                float a[VN], b[VN];
                Vec<T>::load(p, a); Vec<T>::load(p + VN, b);
#pragma unroll
                for (int e = 0; e < VN; e++) { o[e] = a[e]; o[e + 4] = b[e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = e < nv ? to_f32(p[e]) : 0.f;
        }
    };
    auto store8 = [&](void* base, long long at, const float (&o)[8]) {
        T* p = (T*)base + at;
        if (vec) {
            if constexpr (VN == 8) Vec<T>::store(p, o);
            else {
                float a[VN], b[VN];
#pragma unroll
                for (int e = 0; e < VN; e++) { a[e] = o[e]; b[e] = o[e + 4]; }
                Vec<T>::store(p, a); Vec<T>::store(p + VN, b);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (e < nv) p[e] = from_f32<T>(o[e]);
        }
    };
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] *= P.scale;
    const long long off = P.c_map.off(m) + n;
    if (has_aux) store8(pr.aux_out, off, v);
    if (has_gate) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] *= gate;
    }
    if (has_act) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = act_fwd_t<T>(v[e], P.act);
    }
    if (has_act_bwd) {
        float h[8];
        load8(pr.aux_in, off, h);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] *= act_grad_t<T>(h[e], P.act_bwd);
    }
    if (has_res) {
        float q[8];
        load8(pr.residual, P.r_map.off(m) + n, q);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += q[e];
    }
    store8(pr.C, off, v);
}

// Epilogue of one bf16 output tile parked in LDS as fp32 (see gemm_bf16_dma_kernel): thread -> 8 consecutive columns of a row,
// rows strided over the 256 threads in a rolled loop, so the code exists once and every global access is a 16-byte piece of a row.
// Row swizzle of the parked fp32 tile: 16-byte chunk `ch` of row r sits at chunk ch ^ (r & kMask); the mask must keep the chunk inside
// the row (BN / 4 chunks): 15 for the power-of-two widths, 7 for BN = 160 (40 chunks = 5 groups of 8).
template <int BN> struct CtSwz { static constexpr int kMask = (BN / 4) % 16 == 0 ? 15 : 7; };
template <int BM, int BN, int NTHR = 256>
FF_DEV void tile_epilogue_bf16(const GemmParams& P, const GemmProblem& pr, const float* ct, int m_base, int n_base) {
    constexpr int TPR = BN / 8, RPP = NTHR / TPR;
    const int t = threadIdx.x, tr = t / TPR, col = (t % TPR) * 8, n = n_base + col;
    if (t >= TPR * RPP || n >= P.N) return;
    const int nv = min(8, P.N - n);
    const bool vec = P.c_vec8 && nv == 8;
    const float gate = pr.gate ? tanhf(to_f32(*(const bf16*)pr.gate)) : 1.f;
    auto rows = [&](auto feat) {   // branch-free row loop for one feature combination, 4 rows in flight
        constexpr int F = decltype(feat)::value;
        constexpr int UNROLL = (F < 0 || (F & (kEpiAct | kEpiActBwd))) ? 2 : 4;   // the erf-based activations are ~40 instructions per element
        const int nrows = min(BM, P.M - m_base);          // loop bound instead of a break: lets the rows' LDS reads go out together
#pragma unroll UNROLL
        for (int r = tr; r < nrows; r += RPP) {
            const int m = m_base + r;
            const int ch = col >> 2, sw = r & CtSwz<BN>::kMask;
            const f32x4 lo = *(const f32x4*)(ct + r * BN + ((ch ^ sw) << 2)), hi = *(const f32x4*)(ct + r * BN + (((ch + 1) ^ sw) << 2));
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            epilogue8<bf16, F>(P, pr, m, n, nv, vec, gate, v);
        }
    };
    const int feat = (pr.aux_out ? kEpiAux : 0) | (pr.gate ? kEpiGate : 0) | (P.act >= 0 ? kEpiAct : 0) | (P.act_bwd >= 0 ? kEpiActBwd : 0) |
                     (pr.residual ? kEpiRes : 0);
    if (feat == 0) rows(std::integral_constant<int, 0>());                                                      // plain projection / weight gradient
    else if (feat == kEpiRes) rows(std::integral_constant<int, kEpiRes>());                                         // attention / ff output + skip
    else if (feat == (kEpiRes | kEpiGate)) rows(std::integral_constant<int, kEpiRes | kEpiGate>());            // ... tanh-gated (xattn block)
    else if (feat == (kEpiAux | kEpiAct)) rows(std::integral_constant<int, kEpiAux | kEpiAct>());              // ff up-projection
    else if (feat == kEpiActBwd) rows(std::integral_constant<int, kEpiActBwd>());                              // its dgrad
    else if (feat == (kEpiActBwd | kEpiGate)) rows(std::integral_constant<int, kEpiActBwd | kEpiGate>());
    else rows(std::integral_constant<int, -1>());
}

// Kernel arguments live in HBM and a scalar load that misses costs ~0.7 us; left to itself the compiler fetches them lazily in
// dependent groups (five of them, 2.4 us, before the first operand load of the MFMA kernel, another one in its epilogue).
// FF_GEMM_ARGS fetches everything a kernel will ever need in one batch into Q / pr and pins it in SGPRs.
#define FF_PIN(x) asm volatile("" : "+s"(x))
#define FF_GEMM_ARGS(Q, pr, P)                                                                                                        \
    GemmParams Q;                                                                                                                     \
    Q.M = P.M; Q.N = P.N; Q.K = P.K; Q.split_k = P.split_k; Q.k_per_split = P.k_per_split; Q.nz = P.nz;                              \
    Q.xcd_ms = P.xcd_ms; Q.xcd_ns = P.xcd_ns; Q.a_map = P.a_map; Q.b_map = P.b_map; Q.c_map = P.c_map; Q.r_map = P.r_map;             \
    Q.scale = P.scale; Q.act = P.act; Q.act_bwd = P.act_bwd; Q.c_vec8 = P.c_vec8; Q.partial = P.partial;                              \
    GemmProblem pr = P.p[0];                                                                                                          \
    FF_PIN(Q.M); FF_PIN(Q.N); FF_PIN(Q.K); FF_PIN(Q.split_k); FF_PIN(Q.k_per_split); FF_PIN(Q.nz); FF_PIN(Q.xcd_ms); FF_PIN(Q.xcd_ns); \
    FF_PIN(Q.a_map.ld); FF_PIN(Q.a_map.seg_stride); FF_PIN(Q.a_map.rows_per_seg);                                                     \
    FF_PIN(Q.b_map.ld); FF_PIN(Q.b_map.seg_stride); FF_PIN(Q.b_map.rows_per_seg);                                                     \
    FF_PIN(Q.c_map.ld); FF_PIN(Q.c_map.seg_stride); FF_PIN(Q.c_map.rows_per_seg);                                                     \
    FF_PIN(Q.r_map.ld); FF_PIN(Q.r_map.seg_stride); FF_PIN(Q.r_map.rows_per_seg);                                                     \
    FF_PIN(Q.scale); FF_PIN(Q.act); FF_PIN(Q.act_bwd); FF_PIN(Q.c_vec8); FF_PIN(Q.partial);                                           \
    FF_PIN(pr.A); FF_PIN(pr.B); FF_PIN(pr.C); FF_PIN(pr.aux_out); FF_PIN(pr.aux_in); FF_PIN(pr.residual); FF_PIN(pr.gate)


struct TileCoord {
    int z, split, tm, tn;
};
// a / b for 0 <= a < 2^20, 0 < b: one v_rcp + fix-up instead of the ~35-instruction integer division sequence
FF_DEV int fast_div(int a, int b) {
    int q = (int)(__fdividef((float)a, (float)b));
    if (q * b > a) q--;
    else if ((q + 1) * b <= a) q++;
    return q;
}
// blockIdx -> (problem, K split, tile).  The 8 XCDs have private L2s and workgroup b runs on XCD b % 8 (observed, used for
// speed only), so the tile grid is cut into xcd_ms x xcd_ns sub-grids (powers of two, ms * ns = 8 or 1), one per XCD: an XCD
// then fetches only 1/ms of A's row panels and 1/ns of B's column panels through its L2 (rocprofv3 FETCH_SIZE for
// 1024x5120x1280: 103 MB with one row panel per XCD - every L2 pulled the whole weight matrix - vs 15.7 MB algorithmic).
// The map is a bijection for any grid size: positions are the concatenation of the sub-grids, XCD x takes a contiguous chunk
// of positions.  Everything here is on the critical path of a ~20 us kernel: shifts and one reciprocal, no integer division.
// Grouped launches (an XCD's chunk = one or two whole problems) keep this order too: walking a problem with the smaller operand's panels
// as the inner index (tools/experiments/r4_grouped_resident_panel_walk.patch) cut 2 x FETCH_SIZE of a 12-block weight-gradient launch
// 416 -> 368 MB but cost 9 % of its time (190 -> 208 us) with two workgroups per CU in flight - profiles/r04_wgrad_walk_ab.txt.
FF_DEV TileCoord tile_coord(int nz, int split_k, int ms, int ns, int tiles_m, int tiles_n) {
    const int per_z = tiles_m * tiles_n;
    int bid = xcd_remap(blockIdx.x, per_z * split_k * nz);   // == gridDim.x, without the hidden-argument load
    TileCoord c;
    c.z = 0; c.split = 0;
    if (nz > 1) { c.z = fast_div(bid, per_z * split_k); bid -= c.z * per_z * split_k; }
    if (split_k > 1) { c.split = fast_div(bid, per_z); bid -= c.split * per_z; }
    int p = bid;
    const int lm = __builtin_ctz(ms), ln = __builtin_ctz(ns);
    c.tm = 0; c.tn = 0;
    for (int sgrid = 0; sgrid < ms * ns; sgrid++) {
        const int sm = sgrid >> ln, sn = sgrid & (ns - 1);
        const int r0 = (sm * tiles_m) >> lm, r1 = ((sm + 1) * tiles_m) >> lm;
        const int c0 = (sn * tiles_n) >> ln, c1 = ((sn + 1) * tiles_n) >> ln;
        const int cw = c1 - c0, cnt = (r1 - r0) * cw;
        if (p < cnt) {
            const int q = fast_div(p, cw);
            c.tm = r0 + q;
            c.tn = c0 + p - q * cw;
            break;
        }
        p -= cnt;
    }
    return c;
}

#ifdef FF_GEMM_TIMELINE   // debug build: per-workgroup phase timestamps (100 MHz constant clock), read with ff_debug_timeline_read
__device__ unsigned long long g_timeline[16384 * 8];
#define FF_TL(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_timeline[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FF_TL(i) do { } while (0)
#endif


// The leading scalar arguments are the ones the path to the first operand load needs; they are declared as plain kernel
// parameters (not inside the by-value struct) so the hardware can preload them into SGPRs at wave launch
// (-mllvm -amdgpu-kernarg-preload-count, see build.py) instead of a ~0.7 us scalar-load round trip.  Everything else is fetched in
// one batch *after* the first operand tiles are in flight.   h_xcd = xcd_ms | xcd_ns << 8;  h_seg != 0: an operand row map is
// segmented (rare: the full maps are then read from P up front).
template <int BM, int BN, int AL, int BL, int NS>
__global__ __launch_bounds__(256) void gemm_bf16_dma_kernel(const void* hA, const void* hB, int hM, int hN, int hK, int h_split, int h_kps,
                                                            int h_nz, int h_xcd, int h_ald, int h_bld, int h_seg, const GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* smem = (bf16*)smem_raw;
    FF_TL(0);
    constexpr int A_ELEMS = BM * kBK, B_ELEMS = BN * kBK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16;
    constexpr int PER_TILE = BM / 32 + BN / 32;   // DMA instructions per wave per k-step
    static_assert(PER_TILE * (NS - 1) <= 63, "vmcnt overflow");

    RowMap a_map{h_ald, 0, 0}, b_map{h_bld, 0, 0};
    if (h_seg) { a_map = P.a_map; b_map = P.b_map; }
    const void* opA = hA;
    const void* opB = hB;
    const int tiles_m = (hM + BM - 1) / BM, tiles_n = (hN + BN - 1) / BN;
    const TileCoord tc = tile_coord(h_nz, h_split, h_xcd & 255, h_xcd >> 8, tiles_m, tiles_n);
    if (tc.z > 0) { opA = P.p[tc.z].A; opB = P.p[tc.z].B; }   // grouped launches only: one kernarg round trip
    const int m_base = tc.tm * BM, n_base = tc.tn * BN;
    const int k_begin = tc.split * h_kps;
    const int k_end = min(hK, k_begin + h_kps);
    const int t = threadIdx.x, l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 1, wn = w & 1;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)opA, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)opB, 0, 0x7fffffff, 0x00020000);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (k_end - k_begin + kBK - 1) / kBK;
    unsigned va[BM / 32], vb[BN / 32];
    dma_prepare<BM, AL>(a_map, m_base, hM, w, l, va);
    dma_prepare<BN, BL>(b_map, n_base, hN, w, l, vb);
    const bool a_plain = AL == 0 || a_map.rows_per_seg <= 0, b_plain = BL == 0 || b_map.rows_per_seg <= 0;
    const unsigned a_step = AL == 0 ? 2u : (unsigned)a_map.ld * 2u, b_step = BL == 0 ? 2u : (unsigned)b_map.ld * 2u;   // bytes per unit of k
    auto issue = [&](int tile) {
        bf16* st = smem + (tile % NS) * STAGE;
        const int k0 = k_begin + tile * kBK;
        const bool full = k0 + kBK <= k_end;       // wave-uniform
        if (full && a_plain) dma_tile_fast<BM, AL>(ra, st, va, (unsigned)k0 * a_step, w);
        else dma_tile<BM, AL>(ra, st, a_map, m_base, hM, k0, k_end, w, l);
        if (full && b_plain) dma_tile_fast<BN, BL>(rb, st + A_ELEMS, vb, (unsigned)k0 * b_step, w);
        else dma_tile<BN, BL>(rb, st + A_ELEMS, b_map, n_base, hN, k0, k_end, w, l);
    };
    FF_TL(1);
#pragma unroll
    for (int s = 0; s < NS - 1; s++)
        if (s < nk) issue(s);
    FF_GEMM_ARGS(Q, pr, P);             // epilogue arguments: the round trip overlaps the first operand tiles
    if (tc.z > 0) pr = P.p[tc.z];

    for (int kt = 0; kt < nk; kt++) {
        // tile kt must have landed; the up to NS-2 younger tiles may stay in flight across the barrier
        const int younger = min(nk - 1 - kt, NS - 2);
        if (NS >= 4 && younger == 2) wait_vmcnt<2 * PER_TILE>();
        else if (NS >= 3 && younger >= 1) wait_vmcnt<PER_TILE>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                       // everyone's pieces of tile kt are in LDS; stage (kt-1)%NS is free
#ifdef FF_GEMM_TIMELINE
        if (kt == 0) FF_TL(2);
#endif
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
        const bf16* sA = smem + (kt % NS) * STAGE;
        const bf16* sB = sA + A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < kBK / 32; ks++) {
            bf16x8 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; i++) fa[i] = frag_read2<BM, AL>(sA, wm * WM + i * 16, ks);
#pragma unroll
            for (int j = 0; j < NT; j++) fb[j] = frag_read2<BN, BL>(sB, wn * WN + j * 16, ks);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++) acc[i][j] = mfma_bf16(fb[j], fa[i], acc[i][j]);  // D[n][m]
        }
    }

    FF_TL(3);
    const int c = l & 15, g = l >> 4;
    if (Q.split_k > 1) {   // fp32 partial slab; gemm_splitk_epilogue_kernel reduces the slabs and applies the epilogue
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int m = m_base + wm * WM + i * 16 + c;
            if (m >= Q.M) continue;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int n = n_base + wn * WN + j * 16 + g * 4;
                if (n >= Q.N) continue;
                *(f32x4*)(Q.partial + ((long long)(tc.z * Q.split_k + tc.split) * Q.M + m) * Q.N + n) = acc[i][j];
            }
        }
        return;
    }
    // The operand ring is dead: park the fp32 tile in it (16-byte chunks XOR-swizzled by row) and let one rolled loop apply the
    // epilogue on row-contiguous 8-element pieces.  Applying it per accumulator fragment inlined the activation code 64 times
    // (15 k instructions, far beyond the instruction cache: ~8 us per workgroup) and stored 8-byte pieces.
    static_assert(BM * BN * 4 <= NS * STAGE * 2, "fp32 tile must fit the operand ring");
    __syncthreads();
    float* ct = (float*)smem_raw;
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const int ml = wm * WM + i * 16 + c;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int ch = ((wn * WN + j * 16) >> 2) + g;
            *(f32x4*)(ct + ml * BN + ((ch ^ (c & CtSwz<BN>::kMask)) << 2)) = acc[i][j];
        }
    }
    __syncthreads();
    FF_TL(4);
    tile_epilogue_bf16<BM, BN>(Q, pr, ct, m_base, n_base);
    FF_TL(5);
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer variant for shapes whose tile grid can be made to fit the chip exactly: 8 waves, waves 0-3 only run MFMA
// (2 x 2 over the BM x BN tile), waves 4-7 only issue the LDS-DMA of the operand ring.  In the kernel above every wave does both,
// and a DMA instruction parks its in-order wave for ~100+ cycles, which a second co-resident workgroup has to hide; that is why a
// workgroup alone on a CU reaches only half of the CU's operand rate there.  Here a single workgroup per CU reaches it, so a launch
// can use ONE tile per CU: 1024 x 5120 -> 8 x 32 tiles of 128 x 160 = 256 workgroups (the 128 x 128 grid has 320 tiles: 64 CUs get
// two), and 1024 x 1280 x 5120 -> 64 tiles x split-K 4.  K-major operands only (BN = 160 has no M-major staging).
// Measured (tools/pc_bench.py, cold operands): 32.3 -> 28.1 us and 31.4 -> 28.4 us incl. the split-K reduce; in-model 37.60 -> 37.41
// ms/step.  A k-step still takes ~1.05 us = 14 B/clk of operand bytes (the 4-wave kernel alone on a CU: ~10; two of them: ~21 together):
// eight DMA waves instead of four, or a 4-deep ring, change nothing by themselves - together (round 3) they are worth 0.3 % of the step and are
// what the 128 x 160 launches use now.  tools/experiments/fill_rate.hip: with nothing consuming, four DMA waves pull L2-resident tiles at
// 15-16 B/clk, eight at 19-22, everything at ~12 when the bytes come from HBM - the kernel sits where a mix of warm activations and cold
// weights puts it.
// ------------------------------------------------------------------------------------------------
// Staging of the B operand tile: one piece, or - BN = 160 with N the contiguous index (M-major) - a 128-column piece and a 32-column piece:
// a 160-element k-row is 320 bytes, which no whole number of 1-KiB wave-level DMA instructions covers, 256 + 64 bytes do (4 + 1
// instructions per wave and k-step, the same count as for the K-major tile).  Fragment reads pick the piece by their column.
// NW = number of issuing waves.  With eight of them the 160-wide tile is always staged in two pieces (160 rows / columns do not divide by
// 8 waves x 8 rows): the 128 piece by all eight waves, the 32 piece by the first four - the other four issue the same instruction with an
// out-of-range source into a scratch kilobyte each (`dummy`), so that every wave has the same number of DMA instructions per k-step and the
// counted vmcnt waits stay uniform.  For K-major tiles the two pieces are consecutive rows: the LDS image is that of the unsplit tile.
template <int BN, int BL, int NW = 4> struct BStage {
    static constexpr bool SPLIT = BN == 160 && (BL == 1 || NW == 8);
    static constexpr int N128 = 128 / (NW * 8);        // DMA instructions per wave for the 128 piece
    static constexpr int PER_WAVE = SPLIT ? N128 + 1 : BN / (NW * 8);
    static FF_DEV void prepare(const RowMap& map, int n_base, int n_lim, int w, int l, unsigned* v) {
        if constexpr (SPLIT) {
            dma_prepare<128, BL, NW>(map, n_base, n_lim, w, l, v);
            dma_prepare<32, BL, 4>(map, n_base + 128, n_lim, w & 3, l, v + N128);
            if (NW == 8 && w >= 4) v[N128] = kOobOffset;
        } else dma_prepare<BN, BL, NW>(map, n_base, n_lim, w, l, v);
    }
    static FF_DEV void fast(__amdgpu_buffer_rsrc_t r, bf16* st, const unsigned* v, unsigned soff, int w, bf16* dummy = nullptr) {
        if constexpr (SPLIT) {
            dma_tile_fast<128, BL, NW>(r, st, v, soff, w);
            if (NW == 8 && w >= 4) dma_tile_fast<32, BL, 4>(r, dummy, v + N128, soff, 0);
            else dma_tile_fast<32, BL, 4>(r, st + 128 * kBK, v + N128, soff, w & 3);
        } else dma_tile_fast<BN, BL, NW>(r, st, v, soff, w);
    }
    static FF_DEV void slow(__amdgpu_buffer_rsrc_t r, bf16* st, const RowMap& map, int n_base, int n_lim, int k0, int k_end, int w, int l, bf16* dummy = nullptr) {
        if constexpr (SPLIT) {
            dma_tile<128, BL, NW>(r, st, map, n_base, n_lim, k0, k_end, w, l);
            if (NW == 8 && w >= 4) dma_tile<32, BL, 4>(r, dummy, map, n_base + 128, 0, k0, k_end, 0, l);      // row limit 0: every lane out of range
            else dma_tile<32, BL, 4>(r, st + 128 * kBK, map, n_base + 128, n_lim, k0, k_end, w & 3, l);
        } else dma_tile<BN, BL, NW>(r, st, map, n_base, n_lim, k0, k_end, w, l);
    }
    static FF_DEV bf16x8 frag(const bf16* s, int col0, int ks) {      // col0: wave-uniform
        if constexpr (SPLIT && BL == 1) return col0 < 128 ? frag_read2<128, 1>(s, col0, ks) : frag_read2<32, 1>(s + 128 * kBK, col0 - 128, ks);
        else return frag_read2<BN, BL>(s, col0, ks);
    }
};

// WPC = workgroups per CU the register budget is sized for (2: <= 128 registers per lane, two 8-wave workgroups share a CU).  The second
//       __launch_bounds__ argument of HIP is WAVES PER SIMD, not workgroups: WPC * waves / 4.  (Until round 4 it said WPC, the <128,128,1,1>
//       instantiation happened to need 124 registers, and an experiment in tile_coord moved it to 140: one workgroup per CU, the 12-block
//       weight-gradient launches 189 -> 289 us.  tests/test_kernel_resources.py now reads the compiler's resource remarks.)
// NPW = producer (DMA) waves: 4, or 8 (a 12-wave workgroup: one MFMA wave and two DMA waves per SIMD)
// NCW = consumer (MFMA) waves: 4 (2 x 2 over the tile) or 8 (4 x 2: two MFMA waves per SIMD, each on a 32-row slice - one wave's fragment-read
//       latency is covered by the other's MFMAs instead of being exposed twice per k-step; round 4, the 128 x 160 launches)
template <int BM, int BN, int AL, int BL, int NS, int WPC, int NPW = 4, int NCW = 4>
__global__ __launch_bounds__((NCW + NPW) * 64, WPC * (NCW + NPW) / 4) void gemm_bf16_pc_kernel(const void* hA, const void* hB, int hM, int hN, int hK, int h_split, int h_kps,
                                                           int h_nz, int h_xcd, int h_ald, int h_bld, int h_seg, const GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* smem = (bf16*)smem_raw;
    constexpr int A_ELEMS = BM * kBK, B_ELEMS = BN * kBK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int CM = NCW / 2;                                 // consumer waves along M (x 2 along N)
    constexpr int WM = BM / CM, WN = BN / 2, MT = WM / 16, NT = WN / 16;
    static_assert((NCW == 4 || NCW == 8) && WM % 16 == 0, "consumer layout");
    typedef BStage<BN, BL, NPW> BS;
    constexpr int PER_TILE = BM / (NPW * 8) + BS::PER_WAVE;   // DMA instructions per producer wave per k-step
    static_assert(WN % 16 == 0 && BM % 32 == 0 && BN % 32 == 0, "tile shape");
    static_assert((AL == 0 || BM == 256 || BM == 128 || BM == 64) && (BL == 0 || BN == 128 || BN == 64 || BN == 160), "M-major staging exists for 64 / 128 / 256-wide A tiles and 64 / 128 / 160-wide B tiles only");
    static_assert(PER_TILE * (NS - 1) <= 63, "vmcnt overflow");

    RowMap a_map{h_ald, 0, 0}, b_map{h_bld, 0, 0};
    if (h_seg) { a_map = P.a_map; b_map = P.b_map; }
    const void* opA = hA;
    const void* opB = hB;
    const int tiles_m = (hM + BM - 1) / BM, tiles_n = (hN + BN - 1) / BN;
    const TileCoord tc = tile_coord(h_nz, h_split, h_xcd & 255, h_xcd >> 8, tiles_m, tiles_n);
    if (tc.z > 0) { opA = P.p[tc.z].A; opB = P.p[tc.z].B; }
    const int m_base = tc.tm * BM, n_base = tc.tn * BN;
    const int k_begin = tc.split * h_kps;
    const int k_end = min(hK, k_begin + h_kps);
    const int t = threadIdx.x, l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool producer = w >= NCW;
    const int pw = producer ? w - NCW : w;         // index among the producers / the consumers
    const int wm = (w & (NCW - 1)) >> 1, wn = w & 1;
    bf16* dummy = smem + NS * STAGE + (pw & 3) * 512;   // NPW = 8, 160-wide tile: scratch kilobyte of producers 4..7 (see BStage)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)opA, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)opB, 0, 0x7fffffff, 0x00020000);
    const int nk = (k_end - k_begin + kBK - 1) / kBK;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned va[BM / (NPW * 8)], vb[BS::PER_WAVE];
    const bool a_plain = AL == 0 || a_map.rows_per_seg <= 0, b_plain = BL == 0 || b_map.rows_per_seg <= 0;
    const unsigned a_step = AL == 0 ? 2u : (unsigned)a_map.ld * 2u, b_step = BL == 0 ? 2u : (unsigned)b_map.ld * 2u;   // bytes per unit of k
    auto issue = [&](int tile) {
        bf16* st = smem + (tile % NS) * STAGE;
        const int k0 = k_begin + tile * kBK;
        const bool full = k0 + kBK <= k_end;       // wave-uniform
        if (full && a_plain) dma_tile_fast<BM, AL, NPW>(ra, st, va, (unsigned)k0 * a_step, pw);
        else dma_tile<BM, AL, NPW>(ra, st, a_map, m_base, hM, k0, k_end, pw, l);
        if (full && b_plain) BS::fast(rb, st + A_ELEMS, vb, (unsigned)k0 * b_step, pw, dummy);
        else BS::slow(rb, st + A_ELEMS, b_map, n_base, hN, k0, k_end, pw, l, dummy);
    };
    if (producer) {
        dma_prepare<BM, AL, NPW>(a_map, m_base, hM, pw, l, va);
        BS::prepare(b_map, n_base, hN, pw, l, vb);
#pragma unroll
        for (int s = 0; s < NS - 1; s++)
            if (s < nk) issue(s);
    }
    FF_GEMM_ARGS(Q, pr, P);             // epilogue arguments: the round trip overlaps the first operand tiles
    if (tc.z > 0) pr = P.p[tc.z];
#ifdef FF_GEMM_PCMODE    // timing builds (tools/build_pcmodes.sh: one library per mode, -DFF_GEMM_PCMODE=n; results are WRONG, timing only) -
    constexpr int pcmode = FF_GEMM_PCMODE;      // 1: no fragment reads / MFMA (the DMA side alone), 2: no DMA (the consumers alone),
#else                                           // 3: fragment reads only, 4: MFMA only.  A compile-time constant: a run-time switch cost the
    constexpr int pcmode = 0;                   // kernel its register budget (r6s7: 427 us instead of 63)
#endif
    if (producer) {
        for (int kt = 0; kt < nk; kt++) {
            if (pcmode == 2) { __builtin_amdgcn_s_barrier(); continue; }
            const int younger = min(nk - 1 - kt, NS - 2);
            if (NS >= 4 && younger == 2) wait_vmcnt<2 * PER_TILE>();
            else if (NS >= 3 && younger >= 1) wait_vmcnt<PER_TILE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();           // tile kt is in LDS for everybody; the consumers have left stage (kt - 1) % NS
            if (kt + NS - 1 < nk) issue(kt + NS - 1);
        }
    } else {
        for (int kt = 0; kt < nk; kt++) {
            __builtin_amdgcn_s_barrier();
            if (pcmode == 1) continue;
            const bf16* sA = smem + (kt % NS) * STAGE;
            const bf16* sB = sA + A_ELEMS;
#pragma unroll
            for (int ks = 0; ks < kBK / 32; ks++) {
                bf16x8 fa[MT], fb[NT];
                if (pcmode != 4) {
#pragma unroll
                    for (int i = 0; i < MT; i++) fa[i] = frag_read2<BM, AL>(sA, wm * WM + i * 16, ks);
#pragma unroll
                    for (int j = 0; j < NT; j++) fb[j] = BS::frag(sB, wn * WN + j * 16, ks);
                } else {
                    const bf16x8 fixed = __builtin_bit_cast(bf16x8, f32x4{(float)l, 1.f, 2.f, (float)kt});
#pragma unroll
                    for (int i = 0; i < MT; i++) fa[i] = fixed;
#pragma unroll
                    for (int j = 0; j < NT; j++) fb[j] = fixed;
                }
                if (pcmode == 3) {      // keep the reads alive without the MFMAs
#pragma unroll
                    for (int i = 0; i < MT; i++) acc[i][0] += __builtin_bit_cast(f32x4, fa[i]);
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[0][j] += __builtin_bit_cast(f32x4, fb[j]);
                    continue;
                }
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = mfma_bf16(fb[j], fa[i], acc[i][j]);  // D[n][m]
            }
        }
    }
    const int c = l & 15, g = l >> 4;
    if (Q.split_k > 1) {   // fp32 partial slab; gemm_splitk_epilogue_kernel reduces the slabs and applies the epilogue
        if (producer) return;
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int m = m_base + wm * WM + i * 16 + c;
            if (m >= Q.M) continue;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int n = n_base + wn * WN + j * 16 + g * 4;
                if (n >= Q.N) continue;
                *(f32x4*)(Q.partial + ((long long)(tc.z * Q.split_k + tc.split) * Q.M + m) * Q.N + n) = acc[i][j];
            }
        }
        return;
    }
    static_assert(BM * BN * 4 <= NS * STAGE * 2, "fp32 tile must fit the operand ring");
    __syncthreads();                                // the consumers' last reads of the ring are done
    float* ct = (float*)smem_raw;
    if (!producer) {
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int ml = wm * WM + i * 16 + c;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int ch = ((wn * WN + j * 16) >> 2) + g;
                *(f32x4*)(ct + ml * BN + ((ch ^ (c & CtSwz<BN>::kMask)) << 2)) = acc[i][j];
            }
        }
    }
    __syncthreads();
    tile_epilogue_bf16<BM, BN, (NCW + NPW) * 64>(Q, pr, ct, m_base, n_base);      // all waves share the row loop
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tiles on ONE 16-wave workgroup per CU (round 5): the tile for products big enough to give every CU several of them.
// Operand tiles enter a CU at ~21 B/clk whatever the kernel does (section 5 of DESIGN.md: the fill microbenchmark, this family's k-step
// anatomy, and hipBLASLt's MT256x160 kernel all sit at that rate), so a launch's ceiling is its tile's FLOP per operand byte: 64 for 128 x 128
// (~940 TFLOP/s), 85 for 256 x 128, 128 for 256 x 256.  A 256 x 256 fp32 accumulator is 64 registers per lane only when sixteen waves share it
// (4 x 4, each on a 64 x 64 slice like every other kernel of the family), and sixteen waves are a whole workgroup: there is no room for
// separate DMA waves, so every wave issues its four 1-KiB pieces of the next k-step (two of A, two of B) right after the barrier and then runs
// its 32 MFMAs - four waves per SIMD cover each other's DMA-issue stalls, which is what the producer / consumer split buys the smaller tiles.
// The LDS holds two and a half k-steps: five 32-KiB units, each one operand's tile of a k-step (see the ring in the kernel); the fp32 tile
// leaves through the dead ring in two 128-row halves.
// Instantiated for K-major operands only (the forward projections): with transposing fragment reads on either operand the first versions lost
// to the planned tiles (profiles/r05_gemm_u16_ab.txt), and those instantiations no longer exist.
// ------------------------------------------------------------------------------------------------
template <int AL, int BL>
__global__ __launch_bounds__(1024, 4) void gemm_bf16_u16_kernel(const void* hA, const void* hB, int hM, int hN, int hK, int h_split, int h_kps,
                                                                int h_nz, int h_xcd, int h_ald, int h_bld, int h_seg, const GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* smem = (bf16*)smem_raw;
    constexpr int BM = 256, BN = 256, NW = 16;
    constexpr int UNIT = BM * kBK, NU = 5;                      // one operand's k-step tile: 32 KiB; five of them are all of the LDS
    static_assert(BM == BN, "the ring's units hold either operand");
    constexpr int WM = 64, WN = 64, MT = WM / 16, NT = WN / 16;
    constexpr int PA = BM / (NW * 8), PB = BN / (NW * 8);       // DMA instructions per wave and k-step: 2 + 2

    RowMap a_map{h_ald, 0, 0}, b_map{h_bld, 0, 0};
    if (h_seg) { a_map = P.a_map; b_map = P.b_map; }
    const void* opA = hA;
    const void* opB = hB;
    const int tiles_m = (hM + BM - 1) / BM, tiles_n = (hN + BN - 1) / BN;
    const TileCoord tc = tile_coord(h_nz, h_split, h_xcd & 255, h_xcd >> 8, tiles_m, tiles_n);
    if (tc.z > 0) { opA = P.p[tc.z].A; opB = P.p[tc.z].B; }
    const int m_base = tc.tm * BM, n_base = tc.tn * BN;
    const int k_begin = tc.split * h_kps;
    const int k_end = min(hK, k_begin + h_kps);
    const int t = threadIdx.x, l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 2, wn = w & 3;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)opA, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)opB, 0, 0x7fffffff, 0x00020000);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (k_end - k_begin + kBK - 1) / kBK;
    unsigned va[PA], vb[PB];
    dma_prepare<BM, AL, NW>(a_map, m_base, hM, w, l, va);
    dma_prepare<BN, BL, NW>(b_map, n_base, hN, w, l, vb);
    const bool a_plain = AL == 0 || a_map.rows_per_seg <= 0, b_plain = BL == 0 || b_map.rows_per_seg <= 0;
    const unsigned a_step = AL == 0 ? 2u : (unsigned)a_map.ld * 2u, b_step = BL == 0 ? 2u : (unsigned)b_map.ld * 2u;   // bytes per unit of k
    // The ring: unit (2 kt + op) % 5 holds operand op (0 = A, 1 = B) of k-step kt.  Issue order A0 B0 A1 | B1 A2 | B2 A3 | ...: after the barrier
    // of k-step kt the two units of k-step kt - 1 are free and take B(kt + 1) and A(kt + 2), while A(kt + 1) - requested a k-step earlier - is
    // still streaming in: one and a half k-steps are always in flight, so the CU's load path never waits for a request to be issued (with two
    // whole stages - the first version - a k-step cost the latency of its tile PLUS its transfer: 4600 clocks instead of ~3100).
    auto unit = [&](int kt, int op) { return smem + ((2 * kt + op) % NU) * UNIT; };
    auto issue_a = [&](int tile) {
        const int k0 = k_begin + tile * kBK;
        if (k0 + kBK <= k_end && a_plain) dma_tile_fast<BM, AL, NW>(ra, unit(tile, 0), va, (unsigned)k0 * a_step, w);
        else dma_tile<BM, AL, NW>(ra, unit(tile, 0), a_map, m_base, hM, k0, k_end, w, l);
    };
    auto issue_b = [&](int tile) {
        const int k0 = k_begin + tile * kBK;
        if (k0 + kBK <= k_end && b_plain) dma_tile_fast<BN, BL, NW>(rb, unit(tile, 1), vb, (unsigned)k0 * b_step, w);
        else dma_tile<BN, BL, NW>(rb, unit(tile, 1), b_map, n_base, hN, k0, k_end, w, l);
    };
    if (nk > 0) { issue_a(0); issue_b(0); }
    if (nk > 1) issue_a(1);
    FF_GEMM_ARGS(Q, pr, P);             // epilogue arguments: the round trip overlaps the first operand tiles
    if (tc.z > 0) pr = P.p[tc.z];

    // (Measured and dropped, r5s10 / r5s12: a k-step's four pieces spread over its MFMAs instead of requested together - 1153 vs 1153 TFLOP/s;
    // two groups of waves half a k-step apart, one multiplying while the other reads its fragments - 1135 vs 1150.  The SQ counters of r5s13 say
    // why neither matters: the matrix pipe is busy 56 % of the launch at the ~2.0 GHz the chip holds under this load, LDS stalls are 3.6 % of the
    // wave-cycles, no bank conflicts - the waves wait at the barrier for operand pieces, i.e. for the CU's ~47 clocks per 1-KiB request.)
    auto sync = [&](int kt) {           // k-step kt may be read: this wave's pieces of A(kt), B(kt) have landed (A(kt + 1) may stay in flight), and so have
        if (kt + 1 < nk) wait_vmcnt<PA>();      // everybody's; everybody has left the units of k-step kt - 1, which take B(kt + 1) and A(kt + 2)
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue_b(kt + 1);
        if (kt + 2 < nk) issue_a(kt + 2);
    };
    bf16x8 fa[MT], fb[NT];
    auto read = [&](int kt, int ks) {
        const bf16* sA = unit(kt, 0);
        const bf16* sB = unit(kt, 1);
#pragma unroll
        for (int i = 0; i < MT; i++) fa[i] = frag_read2<BM, AL>(sA, wm * WM + i * 16, ks);
#pragma unroll
        for (int j = 0; j < NT; j++) fb[j] = frag_read2<BN, BL>(sB, wn * WN + j * 16, ks);
    };
    auto mma = [&]() {
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc[i][j] = mfma_bf16(fb[j], fa[i], acc[i][j]);  // D[n][m]
    };
    for (int kt = 0; kt < nk; kt++) {
        sync(kt);
        read(kt, 0);
        mma();
        read(kt, 1);
        mma();
    }

    const int c = l & 15, g = l >> 4;
    if (Q.split_k > 1) {   // fp32 partial slab; gemm_splitk_epilogue_kernel reduces the slabs and applies the epilogue
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int m = m_base + wm * WM + i * 16 + c;
            if (m >= Q.M) continue;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int n = n_base + wn * WN + j * 16 + g * 4;
                if (n >= Q.N) continue;
                *(f32x4*)(Q.partial + ((long long)(tc.z * Q.split_k + tc.split) * Q.M + m) * Q.N + n) = acc[i][j];
            }
        }
        return;
    }
    // the fp32 tile is 256 KiB, the dead ring 128: rows 0..127 (wave rows 0, 1) go through it first, then rows 128..255
    static_assert((BM / 2) * BN * 4 <= NU * UNIT * 2, "half of the fp32 tile must fit the operand ring");
    float* ct = (float*)smem_raw;
    __syncthreads();                                        // the last fragment reads of the ring are done
#pragma unroll                                              // (two copies of the epilogue's rolled row loops: kept in a rolled loop the accumulators spill)
    for (int half = 0; half < 2; half++) {
        if ((wm >> 1) == half) {
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const int ml = (wm & 1) * WM + i * 16 + c;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const int ch = ((wn * WN + j * 16) >> 2) + g;
                    *(f32x4*)(ct + ml * BN + ((ch ^ (c & CtSwz<BN>::kMask)) << 2)) = acc[i][j];
                }
            }
        }
        __syncthreads();
        tile_epilogue_bf16<BM / 2, BN, NW * 64>(Q, pr, ct, m_base + half * (BM / 2), n_base);      // (no rows left: the loop bound is <= 0)
        if (half == 0) __syncthreads();
    }
}
template <int AL, int BL> static int launch_bf16_u16(const GemmParams& P, hipStream_t st) {
    constexpr size_t lds = (size_t)5 * 256 * kBK * sizeof(bf16);       // 160 KiB: the whole LDS of a CU
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_u16_kernel<AL, BL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "hipFuncSetAttribute(gemm u16 lds=%zu): %s", lds, hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int grid = cdiv(P.M, 256) * cdiv(P.N, 256) * P.split_k * P.nz;
    const int seg = P.a_map.rows_per_seg > 0 || P.b_map.rows_per_seg > 0 || P.a_map.ld >= (1LL << 31) || P.b_map.ld >= (1LL << 31);
    gemm_bf16_u16_kernel<AL, BL><<<dim3(grid), dim3(1024), lds, st>>>(P.p[0].A, P.p[0].B, P.M, P.N, P.K, P.split_k, P.k_per_split, P.nz,
                                                                      P.xcd_ms | (P.xcd_ns << 8), (int)P.a_map.ld, (int)P.b_map.ld, seg, P);
    return check_launch("gemm_bf16_u16");
}

// ------------------------------------------------------------------------------------------------
// Weight-streaming kernel for decode-shaped products: M <= 32 activation rows (one token per sequence), both operands K-major.
// Nothing is staged in LDS: a workgroup owns 16 output columns, its NW waves split K between them, every lane loads the 16 bytes of weight
// row (n0 + c), k-chunk g that ARE its B fragment of v_mfma_f32_16x16x32_bf16, plus the two matching A fragments of the 32 activation rows,
// keeps two batches of four k-steps in flight, and the partial 32 x 16 tiles of the waves meet in LDS (NW x 2 KiB) for the usual epilogue.
// Measured in isolation (tools/decode_gemm_bench.py: cold weights, graph replay, us per launch incl. the gap) against the 32 x 64 tiles:
// to_out 1280 x 512: 3.7 vs 5.0; to_q 512 x 1280: 5.8-6.7 vs 7.9; FFW up 5120 x 1280: 12.2-13.0 vs 9.6; FFW down 1280 x 5120: 20-23 vs 10.7
// (incl. its split-K reduce).  The long products lose because every workgroup re-reads ALL activation rows as fragments - 2 bytes of
// activations per byte of weights, 26 MB of 64-byte gathers on the same few hundred L2 lines; with those loads removed the same kernel
// needs 5.9 / 7.4 us, of which 2.9 us is the fixed cost of a launch.  The planner therefore uses it for short-K products only (K <= 1024).
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_rows32_kernel(const GemmParams P) {
    __shared__ f32x4 red[NW][2][64];
    const int t = threadIdx.x, l = t & 63, c = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n0 = blockIdx.x * 16;
    const GemmProblem& pr = P.p[0];
    const int steps = P.K / 32, spw = (steps + NW - 1) / NW;
    const int s_begin = w * spw, s_end = min(steps, s_begin + spw);
    // buffer loads: a lane without a row (n0 + c >= N, c >= M) and a k-step beyond the wave's share read zeros through an out-of-range
    // offset - no branches around the loads
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)pr.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)pr.B, 0, 0x7fffffff, 0x00020000);
    const int gk = g * 8;                           // k-chunk of lane group g in k-step s: elements s * 32 + 8 g .. + 7
    const unsigned b_off = n0 + c < P.N ? (unsigned)(P.b_map.off(n0 + c) + gk) * 2u : kOobOffset;
    const unsigned a0_off = c < P.M ? (unsigned)(P.a_map.off(c) + gk) * 2u : kOobOffset;
    const unsigned a1_off = 16 + c < P.M ? (unsigned)(P.a_map.off(16 + c) + gk) * 2u : kOobOffset;
    constexpr int SB = 4;                           // k-steps per batch; two batches (2 x 12 loads of 16 bytes per lane) in flight
    bf16x8 fb0[SB], fa00[SB], fa10[SB], fb1[SB], fa01[SB], fa11[SB];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto ld = [&](__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff, bool in) {      // (nontemporal weight loads measured 3-10 % slower here)
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, in ? off : kOobOffset, soff, 0));
    };
    auto load = [&](bf16x8* fb, bf16x8* fa0, bf16x8* fa1, int s0) {
#pragma unroll
        for (int i = 0; i < SB; i++) {
            const bool in = s0 + i < s_end;         // wave-uniform
            const int st = s0 + i;
            const unsigned soff = in ? (unsigned)st * 64u : 0u;
            fb[i] = ld(rb, b_off, soff, in);
            fa0[i] = ld(ra, a0_off, soff, in);
            fa1[i] = ld(ra, a1_off, soff, in);
        }
    };
    auto mma = [&](const bf16x8* fb, const bf16x8* fa0, const bf16x8* fa1, int s0) {
#pragma unroll
        for (int i = 0; i < SB; i++)
            if (s0 + i < s_end) {
                acc0 = mfma_bf16(fb[i], fa0[i], acc0);      // D[n][m]: lane (c, g) holds row m = c (+16), columns n0 + 4 g .. + 3
                acc1 = mfma_bf16(fb[i], fa1[i], acc1);
            }
    };
    load(fb0, fa00, fa10, s_begin);
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * SB) {
        load(fb1, fa01, fa11, s0 + SB);
        mma(fb0, fa00, fa10, s0);
        load(fb0, fa00, fa10, s0 + 2 * SB);
        mma(fb1, fa01, fa11, s0 + SB);
    }
    red[w][0][l] = acc0;
    red[w][1][l] = acc1;
    __syncthreads();
    if (t < 128) {                                  // wave 0: rows 0..15, wave 1: rows 16..31
        const int i = t >> 6;
        f32x4 sum = red[0][i][l];
#pragma unroll
        for (int ww = 1; ww < NW; ww++) sum += red[ww][i][l];
        const int m = i * 16 + c, n = n0 + g * 4;
        if (m < P.M && n < P.N) {
            float v[4] = {sum[0], sum[1], sum[2], sum[3]};
            epilogue4<bf16>(P, pr, m, n, v);
        }
    }
}
static int launch_bf16_rows32(const GemmParams& P, hipStream_t st) {
    const int steps = P.K / 32, grid = cdiv(P.N, 16);
    if (steps <= 40) hipLaunchKernelGGL(gemm_bf16_rows32_kernel<4>, dim3(grid), dim3(256), 0, st, P);      // up to 10 k-steps per wave
    else if (steps <= 80) hipLaunchKernelGGL(gemm_bf16_rows32_kernel<8>, dim3(grid), dim3(512), 0, st, P);
    else hipLaunchKernelGGL(gemm_bf16_rows32_kernel<16>, dim3(grid), dim3(1024), 0, st, P);
    return check_launch("gemm_bf16_rows32");
}

// ------------------------------------------------------------------------------------------------
// fp32 kernel (exact): 64x64x16 tiles, LDS tiles always stored [k][row]
// ------------------------------------------------------------------------------------------------
constexpr int kFBM = 64, kFBK = 16, kFLd = 80;

template <int LAYOUT>
FF_DEV void f32_tile_load(const float* __restrict__ base, const RowMap& map, int row_base, int row_lim, int k0, int k_end,
                          bool vec_ok, float (&reg)[4]) {
    const int t = threadIdx.x;
    if (LAYOUT == 0) {  // rows = tile rows, K contiguous: thread -> (row = t>>2, 4 k's at (t&3)*4)
        const int row = row_base + (t >> 2), k = k0 + (t & 3) * 4;
        reg[0] = reg[1] = reg[2] = reg[3] = 0.f;
        if (row < row_lim) {
            const float* p = base + map.off(row) + k;
            if (vec_ok && k + 3 < k_end) {
                f32x4 v = *(const f32x4*)p;
                reg[0] = v[0]; reg[1] = v[1]; reg[2] = v[2]; reg[3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (k + j < k_end) reg[j] = p[j];
            }
        }
    } else {  // rows = K, tile rows contiguous: thread -> (k = t>>4, 4 rows at (t&15)*4)
        const int k = k0 + (t >> 4), row = row_base + (t & 15) * 4;
        reg[0] = reg[1] = reg[2] = reg[3] = 0.f;
        if (k < k_end) {
            const float* p = base + map.off(k) + row;
            if (vec_ok && row + 3 < row_lim) {
                f32x4 v = *(const f32x4*)p;
                reg[0] = v[0]; reg[1] = v[1]; reg[2] = v[2]; reg[3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (row + j < row_lim) reg[j] = p[j];
            }
        }
    }
}
template <int LAYOUT> FF_DEV void f32_tile_store(float* s, const float (&reg)[4]) {
    const int t = threadIdx.x;
    if (LAYOUT == 0) {
        const int row = t >> 2, k = (t & 3) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) s[(k + j) * kFLd + row] = reg[j];
    } else {
        const int k = t >> 4, row = (t & 15) * 4;
        *(f32x4*)(s + k * kFLd + row) = f32x4{reg[0], reg[1], reg[2], reg[3]};
    }
}

template <int AL, int BL> __global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams P) {
    __shared__ __attribute__((aligned(16))) float sA[kFBK * kFLd];
    __shared__ __attribute__((aligned(16))) float sB[kFBK * kFLd];
    const int tiles_m = (P.M + kFBM - 1) / kFBM, tiles_n = (P.N + kFBM - 1) / kFBM;
    const TileCoord tc = tile_coord(P.nz, P.split_k, P.xcd_ms, P.xcd_ns, tiles_m, tiles_n);
    const GemmProblem& pr = P.p[tc.z];
    const int m_base = tc.tm * kFBM, n_base = tc.tn * kFBM;
    const int k_begin = tc.split * P.k_per_split;
    const int k_end = min(P.K, k_begin + P.k_per_split);
    const float* A = (const float*)pr.A;
    const float* B = (const float*)pr.B;
    const int t = threadIdx.x, w = t >> 6, l = t & 63, c = l & 15, g = l >> 4;
    const int wm = w >> 1, wn = w & 1;
    const bool a_vec = P.a_vec_ok, b_vec = P.b_vec_ok;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = k_begin; k0 < k_end; k0 += kFBK) {
        float ra[4], rb[4];
        f32_tile_load<AL>(A, P.a_map, m_base, P.M, k0, k_end, a_vec, ra);
        f32_tile_load<BL>(B, P.b_map, n_base, P.N, k0, k_end, b_vec, rb);
        __syncthreads();
        f32_tile_store<AL>(sA, ra);
        f32_tile_store<BL>(sB, rb);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kFBK / 4; ks++) {
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; i++) fa[i] = sA[(ks * 4 + g) * kFLd + wm * 32 + i * 16 + c];
#pragma unroll
            for (int j = 0; j < 2; j++) fb[j] = sB[(ks * 4 + g) * kFLd + wn * 32 + j * 16 + c];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = mfma_f32(fb[j], fa[i], acc[i][j]);  // D[n][m]
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int m = m_base + wm * 32 + i * 16 + c;
        if (m >= P.M) continue;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = n_base + wn * 32 + j * 16 + g * 4;
            if (n >= P.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (P.split_k > 1) {
                float* dst = P.partial + ((long long)(tc.z * P.split_k + tc.split) * P.M + m) * P.N + n;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (n + r < P.N) dst[r] = v[r];
            } else {
                epilogue4<float>(P, pr, m, n, v);
            }
        }
    }
}

// split-K: sum the fp32 partial slabs, then the same epilogue.  One thread = 8 consecutive columns of one row (one item per
// thread, no grid-stride loop); the slabs are read as 16-byte pieces, two splits in flight.
template <typename T> __global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(const GemmParams P) {
    FF_GEMM_ARGS(Q, pr, P);
    const unsigned n8 = (unsigned)(Q.N + 7) / 8u;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned row = idx / n8, nq = idx - row * n8;        // row = z * M + m
    if (row >= (unsigned)(Q.nz * Q.M)) return;
    const int z = Q.nz > 1 ? (int)(row / (unsigned)Q.M) : 0, m = (int)row - z * Q.M, n = (int)nq * 8;
    if (z > 0) pr = P.p[z];
    const int nv = min(8, Q.N - n);                             // 4 or 8: the host requires N % 4 == 0 for split-K
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = Q.partial + ((long long)z * Q.split_k * Q.M + m) * Q.N + n;
    const long long slab = (long long)Q.M * Q.N;
    int sp = 0;
    for (; sp + 3 < Q.split_k; sp += 4) {      // four slabs per pass: eight 16-byte loads in flight before the first add (the plans split 3 .. 8 ways)
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            lo[u] = *(const f32x4*)(src + (sp + u) * slab);
            hi[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (nv == 8) hi[u] = *(const f32x4*)(src + (sp + u) * slab + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] += lo[u][e]; v[e + 4] += hi[u][e]; }
    }
    for (; sp + 1 < Q.split_k; sp += 2) {
        const f32x4 a0 = *(const f32x4*)(src + sp * slab), b0 = *(const f32x4*)(src + (sp + 1) * slab);
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, b1 = a1;
        if (nv == 8) { a1 = *(const f32x4*)(src + sp * slab + 4); b1 = *(const f32x4*)(src + (sp + 1) * slab + 4); }
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] += a0[e]; v[e + 4] += a1[e]; }
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[e + 4] += b1[e]; }
    }
    if (sp < Q.split_k) {
        const f32x4 a0 = *(const f32x4*)(src + sp * slab);
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
        if (nv == 8) a1 = *(const f32x4*)(src + sp * slab + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] += a0[e]; v[e + 4] += a1[e]; }
    }
    const float gate = pr.gate ? tanhf(to_f32(*(const T*)pr.gate)) : 1.f;
    epilogue8<T>(Q, pr, m, n, nv, Q.c_vec8 && nv == 8, gate, v);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int dispatch_f32(const GemmParams& P, hipStream_t st) {
    const int grid = cdiv(P.M, kFBM) * cdiv(P.N, kFBM) * P.split_k * P.nz;
#define FF_F32_LAUNCH(AL, BL) hipLaunchKernelGGL((gemm_f32_kernel<AL, BL>), dim3(grid), dim3(256), 0, st, P)
    if (P.a_layout == 0 && P.b_layout == 0) FF_F32_LAUNCH(0, 0);
    else if (P.a_layout == 0 && P.b_layout == 1) FF_F32_LAUNCH(0, 1);
    else if (P.a_layout == 1 && P.b_layout == 0) FF_F32_LAUNCH(1, 0);
    else FF_F32_LAUNCH(1, 1);
#undef FF_F32_LAUNCH
    return check_launch("gemm_f32");
}


// ------------------------------------------------------------------------------------------------
// optional per-launch timing of the GEMM kernels with HIP events on the launch stream (bench.py's roofline leg)
// ------------------------------------------------------------------------------------------------
namespace {
// The opt-in launch log is the one piece of process-wide state in the library, and it has to be: PyTorch issues the backward pass from its
// autograd worker thread, so a per-thread log would only ever see forward launches.  Recording is thread-safe (slots are claimed with an
// atomic counter); enabling / reading / disabling is done by one thread while no launch is in flight.  Disabled (the default), the only
// cost is one relaxed atomic load per GEMM launch.
struct ProfState {
    std::atomic<bool> on{false};
    int cap = 0;
    std::atomic<int> n{0};
    hipEvent_t* ev = nullptr;     // 2 events per record
    ff_gemm_profile_record* rec = nullptr;
} g_prof;
}
int profile_begin(int dtype, int tile, int a_layout, int b_layout, int M, int N, int K, int nz, int split_k, hipStream_t st) {
    if (!g_prof.on.load(std::memory_order_relaxed)) return -1;
    const int i = g_prof.n.fetch_add(1);
    if (i >= g_prof.cap) return -1;
    ff_gemm_profile_record& r = g_prof.rec[i];
    r.dtype = dtype; r.tile = tile; r.a_layout = a_layout; r.b_layout = b_layout;
    r.M = M; r.N = N; r.K = K; r.nz = nz; r.split_k = split_k; r.ms = 0.f;
    hipEventRecord(g_prof.ev[2 * i], st);
    return i;
}
void profile_end(int i, hipStream_t st) {
    if (i >= 0) hipEventRecord(g_prof.ev[2 * i + 1], st);
}
static int prof_begin(const GemmParams& P, int dtype, int bm, hipStream_t st) {
    return profile_begin(dtype, bm, P.a_layout, P.b_layout, P.M, P.N, P.K, P.nz, P.split_k, st);
}
static void prof_end(int i, hipStream_t st) { profile_end(i, st); }

static int env_int(const char* name, int dflt) { return dbg_switch(name, dflt); }      // development builds only (ff_common.h)

// Block tile and split-K factor for a bf16 problem.  Measured on MI355X (tools/gemm_bench.py --sweep): a k-step's cost
// is the L2 -> LDS traffic of its operand tiles, so 128x128 (64 FLOP/B) beats 64x64 (32 FLOP/B) whenever it can put
// >= ~200 workgroups on the chip; long-K problems with few output tiles get there through split-K (fp32 partial slabs
// + a reduce/epilogue kernel), short-K ones use 64x64 tiles.  Re-checked in-model after the fixed-cost work (same box, graph replay,
// 43.74 ms/step): no split for the small 64-tile GEMMs +0.5 ms, one-workgroup-per-CU split counts +0.7 ms, no 64x128 rule +0.5 ms -
// in-model the operands arrive cold from HBM and more workgroups in flight hide that better than isolated timings suggest.
struct TilePlan { int tile, split; };
static TilePlan plan_bf16(int M, int N, int K, int nz, int want_split, int a_layout = 0, int b_layout = 0, int ft = 0) {
    const long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128) * nz, t64 = (long long)cdiv(M, 64) * cdiv(N, 64) * nz;
    TilePlan p;
    static const int env_tile = env_int("FF_GEMM_TILE", 0);
    if (ft == 0) ft = env_tile;
    // the 128 x 160 producer / consumer kernel needs a K-major A operand; B may be K-major or (split staging) N-contiguous
    static const int pc_bl1 = env_int("FF_GEMM_PC_BL1", 1);
    const bool pc_ok = a_layout == 0 && (b_layout == 0 || (pc_bl1 && N % 8 == 0));
    const bool skinny_ok = a_layout == 0 && b_layout == 0;      // the 32 x 64 tile stages K-major operands only
    const bool rows_ok = skinny_ok && M <= 32 && nz == 1 && K % 32 == 0;      // the weight-streaming kernel (tile code 3216)
    if (ft == 3216 && rows_ok) return TilePlan{3216, 1};
    if (ft == 128168) ft = 128160;      // the 8-consumer-wave launch of the same tile (run_bf16_dma looks at force_tile)
    const long long t256 = (long long)cdiv(M, 256) * cdiv(N, 128) * nz;
    if (ft == 128 || ft == 64 || ft == 64002 || ft == 6412 || ft == 128002 || (ft == 128160 && pc_ok) || (ft == 3264 && skinny_ok) || ft == 256128 || (ft == 256256 && skinny_ok)) {
        p.tile = ft;
        if (want_split > 0) { p.split = want_split; return p; }
        const long long t = ft == 128 || ft == 128002 ? t128 : ft == 64 || ft == 64002 ? t64 : ft == 128160 ? (long long)cdiv(M, 128) * cdiv(N, 160) * nz
                            : ft == 3264 ? (long long)cdiv(M, 32) * cdiv(N, 64) * nz : ft == 256128 ? t256 : ft == 256256 ? (long long)cdiv(M, 256) * cdiv(N, 256) * nz
                            : (long long)cdiv(M, 64) * cdiv(N, 128) * nz;
        p.split = (t >= 128 || K < 1024) ? 1 : std::max(1, std::min(std::min((int)(256 / t), K / 512), 16));
        return p;
    }
    // 200..450 tiles of 128x128 leave most CUs with a single, latency-bound workgroup; 64x128 tiles (1.5x the operand traffic
    // but 2-3 workgroups per CU) measured 10-25 % faster there when A is row-major (sweep in tools/gemm_bench.py)
    // one 128 x 160 tile per CU (producer / consumer kernel) when that grid, times a small split-K, lands on 224..256 workgroups
    static const int pc_on = env_int("FF_GEMM_PC", 1);
    // Decode (one token per sequence: M = batch <= 32 rows): 32 x 64 tiles instead of 64 x 64 ones with half their rows empty; the grid is
    // N / 64 workgroups, so short-K products need no split-K (and no reduce launch) at all and long-K ones split just enough to occupy
    // ~160 CUs (measured in the caption leg: 8 -> 5 library launches per gated block and token)
    static const int skinny_on = env_int("FF_GEMM_SKINNY", 1);
    // ... and, round 3, no tiles at all for short-K products (the attention output projection): the weight-streaming kernel, never split
    static const int rows_on = env_int("FF_GEMM_ROWS32", 1);
    if (rows_on && rows_ok && K <= 1024 && want_split <= 1) return TilePlan{3216, 1};
    if (skinny_on && skinny_ok && M <= 32 && nz == 1) {
        const int t = cdiv(N, 64);
        int sp = 1;
        if (K >= 2048) {
            sp = std::min(8, std::max(1, 160 / t));
            while (sp > 1 && K / (64 * sp) < 8) sp--;
        }
        p = TilePlan{3264, want_split > 0 ? want_split : sp};
        return p;
    }
    // 256 x 256 tiles on 16-wave workgroups (gemm_bf16_u16_kernel) for products with >= 4096 rows, a K-major weight and at least one tile per CU
    // (config E's feed-forward up- and down-projections).  Measured against the 256 x 128 tile (r5s9 / r5s10, tools/gemm_graph_bench.py, cold weights):
    // 4096 x 16384 x 4096 1079-1086 -> 1153-1156 TFLOP/s (with the GELU epilogue 1005 -> 1070), 4096 x 4096 x 16384 1157-1164 -> 1197-1206 (gated residual
    // 1139 -> 1180), 8192^3 1185 -> 1273.  With an N-contiguous weight (data gradients) and for the weight gradients it LOSES to the planned tiles
    // (999 -> 962, 1066 -> 943; 1014 -> 916) and at K = 1024 it is no better in the model's 12-block launches (config B: 41.27 vs 41.29 ms per step, r5s8),
    // so those keep their tiles; spreading a k-step's four pieces over its MFMAs instead of requesting them together changed nothing (1153 vs 1153).
    static const int u16_on = env_int("FF_GEMM_U16", 1);
    const long long t256sq = (long long)cdiv(M, 256) * cdiv(N, 256) * nz;
    if (u16_on && a_layout == 0 && b_layout == 0 && M >= 4096 && K >= 2048 && t256sq >= 256 && want_split <= 1) return TilePlan{256256, 1};
    // Round 5: products with >= 4096 rows (config E: 4 x 1024 tokens, d = 4096, 16384 hidden - 97 % of that configuration's FLOPs) get 256 x 128
    // tiles on a 16-wave workgroup (eight MFMA waves 4 x 2, each on the same 64 x 64 slice as in the 128 x 128 kernel, + eight DMA waves), one
    // workgroup per CU: 85 FLOP per operand byte instead of 64, at the fill rate eight DMA waves reach alone on a CU.
    static const int t256_on = env_int("FF_GEMM_T256", 1);
    // Measured (tools/gemm_graph_bench.py, cold weights, graph replay, r5s5): 4096 x 16384 x 4096 1013 -> 1090 TFLOP/s (N-contiguous weight 1016 ->
    // 1070), 4096 x 4096 x 16384 1057 -> 1170 (1066 -> 1118), with the GELU / GELU' / gated-residual epilogues 956 -> 1007, 939 -> 988, 1033 -> 1160;
    // 4096 x 2048 x 8192 (256 tiles: one per CU) 1057 -> 1098; four DMA waves instead of eight: the same within 1 %.
    // With an M-major A operand (the weight gradients; mswz<256>, two 512-byte k-rows per DMA instruction, eight MFMA + four DMA waves) the tile
    // is selectable (ff_gemm_desc.tile = 256128) but not planned: r5s7, same method - 4096 x 16384 x 4096 994 -> 980 TFLOP/s, 16384 x 4096 x 4096
    // 1038 -> 1054, M-major A with a K-major B 1001 -> 1042; below 4096 contraction rows it loses (4096 x 1024 x 2048: 669 -> 543; 1280 x 5120 x 1024:
    // 591 -> 578).  Both operands are read with the half-width transposing fragment reads, so the tile's extra FLOP per DMA byte buy nothing.
    if (t256_on && a_layout == 0 && M >= 4096 && N >= 1024 && K >= 1024 && t256 >= 256 && N % 8 == 0 && want_split <= 1) return TilePlan{256128, 1};
    const long long t160 = (long long)cdiv(M, 128) * cdiv(N, 160) * nz;
    int pc_split = 0;
    if (pc_on && pc_ok && N % 160 == 0 && K % 64 == 0 && t160 <= 256)
        for (int sp = 1; sp <= 8 && !pc_split; sp++)
            if (t160 * sp >= 224 && t160 * sp <= 256 && K / sp >= 1024 && K % (64 * sp) == 0) pc_split = sp;
    if (pc_split) p = TilePlan{128160, pc_split};
    else if (t128 >= 200 && t128 < 450 && a_layout == 0 && (long long)cdiv(M, 64) * cdiv(N, 128) * nz >= 400) p = TilePlan{6412, 1};
    else if (t128 >= 200) p = TilePlan{128, 1};
    else if (K >= 2048) {
        int s = (int)std::min<long long>(8, std::max<long long>(2, cdiv(384, t128)));
        while (s > 1 && K / (64 * s) < 8) s--;
        p = TilePlan{128, s};
    } else if (t64 >= 200) p = TilePlan{64, 1};
    else {
        int s = (int)std::min<long long>(4, cdiv(256, t64));
        while (s > 1 && K / (64 * s) < 4) s--;
        p = TilePlan{64, s};
    }
    if (want_split > 0) p.split = want_split;
    static const int pc128 = env_int("FF_GEMM_PC128", 2);      // the producer / consumer kernel for every 128 x 128 launch (0: the 4-wave kernel)
    if (pc128 && p.tile == 128) p.tile = 128002;
    if (pc128 >= 2 && p.tile == 6412) p.tile = 128002;         // ... and instead of the 64 x 128 tiles (1: keep those)
    // ... and for the 64 x 64 tiles (short-K projections, small weight gradients): 35.32 -> 35.10 ms/step in a same-box A/B (round 3)
    static const int pc64 = env_int("FF_GEMM_PC64", 1);
    if (pc64 && p.tile == 64) p.tile = 64002;
    return p;
}

template <int BM, int BN, int AL, int BL, int NS> static int launch_bf16_dma(const GemmParams& P, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * (BM + BN) * kBK * sizeof(bf16);
    if (lds > 64 * 1024) {      // the attribute is per device: remember it per device (benign if two threads both set it)
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_dma_kernel<BM, BN, AL, BL, NS>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "hipFuncSetAttribute(gemm dma lds=%zu): %s", lds, hipGetErrorString(e));
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    const int grid = cdiv(P.M, BM) * cdiv(P.N, BN) * P.split_k * P.nz;
    const int seg = P.a_map.rows_per_seg > 0 || P.b_map.rows_per_seg > 0 || P.a_map.ld >= (1LL << 31) || P.b_map.ld >= (1LL << 31);
    gemm_bf16_dma_kernel<BM, BN, AL, BL, NS><<<dim3(grid), dim3(256), lds, st>>>(P.p[0].A, P.p[0].B, P.M, P.N, P.K, P.split_k, P.k_per_split, P.nz,
                                                                           P.xcd_ms | (P.xcd_ns << 8), (int)P.a_map.ld, (int)P.b_map.ld, seg, P);
    return check_launch("gemm_bf16_dma");
}
template <int BM, int BN, int NS> static int dispatch_bf16_dma(const GemmParams& P, hipStream_t st) {
    if (P.a_layout == 0 && P.b_layout == 0) return launch_bf16_dma<BM, BN, 0, 0, NS>(P, st);
    if (P.a_layout == 0 && P.b_layout == 1) return launch_bf16_dma<BM, BN, 0, 1, NS>(P, st);
    if (P.a_layout == 1 && P.b_layout == 0) return launch_bf16_dma<BM, BN, 1, 0, NS>(P, st);
    return launch_bf16_dma<BM, BN, 1, 1, NS>(P, st);
}
template <int BM, int BN> static int run_bf16_dma_tile(const GemmParams& P, int ns, hipStream_t st) {
    return ns == 2 ? dispatch_bf16_dma<BM, BN, 2>(P, st) : ns == 4 ? dispatch_bf16_dma<BM, BN, 4>(P, st) : dispatch_bf16_dma<BM, BN, 3>(P, st);
}
template <int BM, int BN, int AL, int BL, int NS, int WPC, int NPW = 4, int NCW = 4> static int launch_bf16_pc(const GemmParams& P, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * (BM + BN) * kBK * sizeof(bf16) + (NPW == 8 && BN == 160 ? 4096 : 0);
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_pc_kernel<BM, BN, AL, BL, NS, WPC, NPW, NCW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "hipFuncSetAttribute(gemm pc lds=%zu): %s", lds, hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int grid = cdiv(P.M, BM) * cdiv(P.N, BN) * P.split_k * P.nz;
    const int seg = P.a_map.rows_per_seg > 0 || P.b_map.rows_per_seg > 0 || P.a_map.ld >= (1LL << 31) || P.b_map.ld >= (1LL << 31);
    gemm_bf16_pc_kernel<BM, BN, AL, BL, NS, WPC, NPW, NCW><<<dim3(grid), dim3((NCW + NPW) * 64), lds, st>>>(P.p[0].A, P.p[0].B, P.M, P.N, P.K, P.split_k, P.k_per_split,
                                                                                 P.nz, P.xcd_ms | (P.xcd_ns << 8), (int)P.a_map.ld,
                                                                                 (int)P.b_map.ld, seg, P);
    return check_launch("gemm_bf16_pc");
}
// 128 x 128 tiles, two 8-wave workgroups per CU (experiment: FF_GEMM_TILE=128002 / FF_GEMM_PC128=1)
template <int NS> static int dispatch_bf16_pc128(const GemmParams& P, hipStream_t st) {
    if (P.a_layout == 0 && P.b_layout == 0) return launch_bf16_pc<128, 128, 0, 0, NS, 2>(P, st);
    if (P.a_layout == 0 && P.b_layout == 1) return launch_bf16_pc<128, 128, 0, 1, NS, 2>(P, st);
    if (P.a_layout == 1 && P.b_layout == 0) return launch_bf16_pc<128, 128, 1, 0, NS, 2>(P, st);
    return launch_bf16_pc<128, 128, 1, 1, NS, 2>(P, st);
}
static int run_bf16_dma(const GemmParams& P, hipStream_t st) {
    static const int ns_env = env_int("FF_GEMM_STAGES", 0);
    // default 2 stages: 64 KiB (128x128) / 16 KiB (64x64) per workgroup, so several workgroups per CU overlap each other's
    // DMA-issue and barrier stalls - measured faster than deeper rings at lower occupancy (tools/gemm_bench.py --sweep)
    const int ns = P.force_stages > 0 ? P.force_stages : ns_env > 0 ? ns_env : 2;
    if (P.tile == 128160) {      // one workgroup per CU, so the ring has to cover the DMA latency by itself (2 stages would not hold the parked fp32 tile)
        // Round 3: eight DMA waves (12-wave workgroups) AND a 4-deep ring (148 KiB) - each alone changes nothing (round 2), together 35.46 ->
        // 35.36 ms/step in a same-box A/B (gpurun_out/r3s22); FF_GEMM_NPW=4 / FF_GEMM_STAGES=3 in the development build restore the old launch.
        static const int npw_full = env_int("FF_GEMM_NPW", 8), npw_split = env_int("FF_GEMM_NPW_SPLIT", 4);
        // (split-K launches - fp32 partial slabs, no epilogue for the extra waves to share - were ~1 us slower with the 12-wave workgroup: 24.5 -> 25.9 us)
        const int npw = P.split_k > 1 ? npw_split : npw_full;
        const int pns = P.force_stages > 0 ? P.force_stages : ns_env > 0 ? ns_env : npw == 8 ? 4 : 3;
        // Round 4: eight MFMA waves (4 x 2) + four DMA waves, 4-deep ring - tile code 128168 or FF_GEMM_NCW=8 in the development build
        static const int ncw_env = env_int("FF_GEMM_NCW", 4), ncw_split_env = env_int("FF_GEMM_NCW_SPLIT", 4);
        const int ncw = P.force_tile == 128168 ? 8 : (P.split_k > 1 ? ncw_split_env : ncw_env);
        if (ncw == 8) {
            if (P.b_layout == 0) return pns == 3 ? launch_bf16_pc<128, 160, 0, 0, 3, 1, 4, 8>(P, st) : launch_bf16_pc<128, 160, 0, 0, 4, 1, 4, 8>(P, st);
            return pns == 3 ? launch_bf16_pc<128, 160, 0, 1, 3, 1, 4, 8>(P, st) : launch_bf16_pc<128, 160, 0, 1, 4, 1, 4, 8>(P, st);
        }
        if (npw == 8) {
            if (P.b_layout == 0) return pns == 4 ? launch_bf16_pc<128, 160, 0, 0, 4, 1, 8>(P, st) : launch_bf16_pc<128, 160, 0, 0, 3, 1, 8>(P, st);
            return pns == 4 ? launch_bf16_pc<128, 160, 0, 1, 4, 1, 8>(P, st) : launch_bf16_pc<128, 160, 0, 1, 3, 1, 8>(P, st);
        }
        if (P.b_layout == 0) return pns == 4 ? launch_bf16_pc<128, 160, 0, 0, 4, 1>(P, st) : launch_bf16_pc<128, 160, 0, 0, 3, 1>(P, st);
        return pns == 4 ? launch_bf16_pc<128, 160, 0, 1, 4, 1>(P, st) : launch_bf16_pc<128, 160, 0, 1, 3, 1>(P, st);
    }
    if (P.tile == 128002) return dispatch_bf16_pc128<2>(P, st);
    if (P.tile == 256128) {      // 3-deep ring of 48 KiB stages (the parked 256 x 128 fp32 tile needs 128 KiB of it): one workgroup per CU
        static const int npw256 = env_int("FF_GEMM_NPW256", 8);
        if (P.a_layout == 1)     // M-major A (the weight gradients d Y^T . X): 512-byte k-rows, two per wave-level DMA instruction, ds_read_b64_tr_b16 fragments;
                                 // four DMA waves (12-wave workgroup, 168 registers per lane): with eight the transposing reads' addressing spills at the 128-register cap
            return P.b_layout == 0 ? launch_bf16_pc<256, 128, 1, 0, 3, 1, 4, 8>(P, st) : launch_bf16_pc<256, 128, 1, 1, 3, 1, 4, 8>(P, st);
        if (npw256 == 8) return P.b_layout == 0 ? launch_bf16_pc<256, 128, 0, 0, 3, 1, 8, 8>(P, st) : launch_bf16_pc<256, 128, 0, 1, 3, 1, 8, 8>(P, st);
        return P.b_layout == 0 ? launch_bf16_pc<256, 128, 0, 0, 3, 1, 4, 8>(P, st) : launch_bf16_pc<256, 128, 0, 1, 3, 1, 4, 8>(P, st);
    }
    if (P.tile == 256256) return launch_bf16_u16<0, 0>(P, st);
    if (P.tile == 3216) return launch_bf16_rows32(P, st);
    if (P.tile == 3264) return launch_bf16_pc<32, 64, 0, 0, 4, 2>(P, st);
    if (P.tile == 128) return run_bf16_dma_tile<128, 128>(P, ns, st);
    if (P.tile == 6412) return run_bf16_dma_tile<64, 128>(P, ns, st);
    if (P.tile == 64002) {       // 64 x 64 tiles on the 8-wave producer / consumer kernel, 3-stage ring
        if (P.a_layout == 0 && P.b_layout == 0) return launch_bf16_pc<64, 64, 0, 0, 3, 2>(P, st);
        if (P.a_layout == 0 && P.b_layout == 1) return launch_bf16_pc<64, 64, 0, 1, 3, 2>(P, st);
        if (P.a_layout == 1 && P.b_layout == 0) return launch_bf16_pc<64, 64, 1, 0, 3, 2>(P, st);
        return launch_bf16_pc<64, 64, 1, 1, 3, 2>(P, st);
    }
    // 64 x 64 tiles (short-K projections: 8 k-steps of 16 KiB): 3 stages - several workgroups still share a CU at 24 KiB each, and with so
    // few k-steps the extra tile in flight is worth more than the occupancy (same-box A/B of the step: 35.44 -> 35.32 ms, twice)
    return run_bf16_dma_tile<64, 64>(P, P.force_stages > 0 ? P.force_stages : ns_env > 0 ? ns_env : 3, st);
}

int gemm_pick_split(int dtype, int M, int N, int K, int nz) {
    if (dtype == FF_DTYPE_BF16)     // workspace sizing does not know the operand layouts: the larger of the two plans they can select
        return std::max(plan_bf16(M, N, K, nz, 0, 0, 0).split, plan_bf16(M, N, K, nz, 0, 0, 1).split);
    const long long tiles = (long long)cdiv(M, kFBM) * cdiv(N, kFBM) * nz;
    if (tiles >= 128 || K < 1024) return 1;
    int s = (int)(256 / tiles);
    s = std::min(s, K / 512);
    s = std::min(s, 16);
    return std::max(s, 1);
}

size_t gemm_workspace_bytes(int dtype, int M, int N, int K, int nz, int split_k) {
    if (split_k <= 0) split_k = gemm_pick_split(dtype, M, N, K, nz);
    return split_k > 1 ? (size_t)split_k * nz * M * N * sizeof(float) : 0;
}

int gemm_launch(GemmParams P, int dtype, void* workspace, size_t ws_bytes, hipStream_t st, int* leave_partial) {
    if (leave_partial) *leave_partial = 1;
    FF_CHECK(P.M > 0 && P.N > 0 && P.K > 0 && P.nz >= 1 && P.nz <= kGemmMaxZ, FF_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d nz=%d",
             P.M, P.N, P.K, P.nz);
    P.tile = kFBM;
    if (dtype == FF_DTYPE_BF16) {
        const TilePlan plan = plan_bf16(P.M, P.N, P.K, P.nz, P.split_k, P.a_layout, P.b_layout, P.force_tile);
        P.tile = plan.tile;
        P.split_k = plan.split;
    } else if (P.split_k <= 0) P.split_k = gemm_pick_split(dtype, P.M, P.N, P.K, P.nz);
    const int kq = dtype == FF_DTYPE_BF16 ? kBK : kFBK;
    P.k_per_split = cdiv(cdiv(P.K, P.split_k), kq) * kq;
    P.split_k = cdiv(P.K, P.k_per_split);
    P.partial = nullptr;
    if (P.split_k > 1) {
        const size_t need = gemm_workspace_bytes(dtype, P.M, P.N, P.K, P.nz, P.split_k);
        FF_CHECK(workspace && ws_bytes >= need, FF_ERR_WORKSPACE, "gemm split-K workspace: need %zu have %zu", need, ws_bytes);
        FF_CHECK(P.N % 4 == 0, FF_ERR_UNSUPPORTED, "gemm split-K needs N %% 4 == 0 (N=%d)", P.N);
        P.partial = (float*)workspace;
    }
    {   // XCD partition of the tile grid: minimise (A bytes)/ms + (B bytes)/ns over ms * ns = 8
        const int tm_edge = dtype == FF_DTYPE_BF16 ? (P.tile == 256128 || P.tile == 256256 ? 256 : P.tile == 128 || P.tile == 128160 || P.tile == 128002 ? 128 : P.tile == 3264 || P.tile == 3216 ? 32 : 64) : kFBM;
        const int tn_edge = dtype == FF_DTYPE_BF16 ? (P.tile == 3216 ? 16 : P.tile == 64 || P.tile == 64002 || P.tile == 3264 ? 64 : P.tile == 128160 ? 160 : P.tile == 256256 ? 256 : 128) : kFBM;
        const int tiles_m = cdiv(P.M, tm_edge), tiles_n = cdiv(P.N, tn_edge);
        double best = 1e300;
        P.xcd_ms = 1; P.xcd_ns = 1;
        for (int ms = 1; ms <= 8; ms *= 2) {
            const int ns = 8 / ms;
            if (ms > tiles_m || ns > tiles_n) continue;
            const double cost = (double)P.M / ms + (double)P.N / ns;
            if (cost < best) { best = cost; P.xcd_ms = ms; P.xcd_ns = ns; }
        }
        static const int disable = env_int("FF_GEMM_XCD2D", 1) == 0;
        if (disable) { P.xcd_ms = 1; P.xcd_ns = 1; }
    }
    const int vec = dtype == FF_DTYPE_BF16 ? 8 : 4;
    auto map_ok = [&](const RowMap& m) { return m.ld % vec == 0 && (m.rows_per_seg <= 0 || m.seg_stride % vec == 0); };
    const int a_contig = P.a_layout == 0 ? P.K : P.M, b_contig = P.b_layout == 0 ? P.K : P.N;
    P.a_vec_ok = map_ok(P.a_map) && a_contig % vec == 0;
    P.b_vec_ok = map_ok(P.b_map) && b_contig % vec == 0;
    for (int z = 0; z < P.nz; z++) {
        P.a_vec_ok = P.a_vec_ok && ((uintptr_t)P.p[z].A % 16 == 0);
        P.b_vec_ok = P.b_vec_ok && ((uintptr_t)P.p[z].B % 16 == 0);
    }
    P.c_vec8 = P.N % vec == 0 && map_ok(P.c_map) && map_ok(P.r_map);   // 8-column pieces of C / aux / residual as 16-byte accesses
    for (int z = 0; z < P.nz; z++)
        P.c_vec8 = P.c_vec8 && ((uintptr_t)P.p[z].C | (uintptr_t)P.p[z].aux_out | (uintptr_t)P.p[z].aux_in | (uintptr_t)P.p[z].residual) % 16 == 0;
    int rc;
    const int prof_id = prof_begin(P, dtype, P.tile, st);
    if (dtype == FF_DTYPE_BF16) {
        FF_CHECK(P.a_vec_ok && P.b_vec_ok, FF_ERR_UNSUPPORTED,
                 "bf16 gemm needs 16-byte aligned operands with contiguous dims %% 8 == 0 (M=%d N=%d K=%d)", P.M, P.N, P.K);
        FF_CHECK(P.N % 4 == 0 && P.c_map.ld % 4 == 0, FF_ERR_UNSUPPORTED, "bf16 gemm needs N %% 4 == 0 (N=%d)", P.N);
        // operands are addressed through 32-bit buffer offsets
        auto span_ok = [&](const RowMap& m, int rows, int contig) {
            const long long last = (m.rows_per_seg > 0 ? (long long)((rows - 1) / m.rows_per_seg) * m.seg_stride + (long long)((rows - 1) % m.rows_per_seg) * m.ld
                                                      : (long long)(rows - 1) * m.ld) + contig;
            return last < (1LL << 30);
        };
        FF_CHECK(span_ok(P.a_map, P.a_layout == 0 ? P.M : P.K, a_contig) && span_ok(P.b_map, P.b_layout == 0 ? P.N : P.K, b_contig),
                 FF_ERR_UNSUPPORTED, "bf16 gemm operands must span < 2^30 elements (M=%d N=%d K=%d)", P.M, P.N, P.K);
        rc = run_bf16_dma(P, st);
    } else {
        rc = dispatch_f32(P, st);
    }
    prof_end(prof_id, st);   // the record covers the MFMA main kernel only (what rocprofv3 lists under the same name)
    FF_TRY(rc);
    if (P.split_k > 1 && leave_partial && P.nz == 1 && P.scale == 1.f && P.act < 0 && P.act_bwd < 0 && !P.p[0].aux_out && !P.p[0].gate &&
        !P.p[0].residual) {
        *leave_partial = P.split_k;         // the consumer sums the slabs (see ff_internal.h)
        return FF_OK;
    }
    if (P.split_k > 1) {
        const long long total = (long long)P.nz * P.M * ((P.N + 7) / 8);
        FF_CHECK(total < (1LL << 31), FF_ERR_UNSUPPORTED, "gemm split-K output too large (M=%d N=%d nz=%d)", P.M, P.N, P.nz);
        const int grid = (int)((total + 255) / 256);
        if (dtype == FF_DTYPE_BF16) hipLaunchKernelGGL(gemm_splitk_epilogue_kernel<bf16>, dim3(grid), dim3(256), 0, st, P);
        else hipLaunchKernelGGL(gemm_splitk_epilogue_kernel<float>, dim3(grid), dim3(256), 0, st, P);
        FF_TRY(check_launch("gemm_splitk_epilogue"));
    }
    return FF_OK;
}

}  // namespace ff

#ifdef FF_GEMM_TIMELINE
extern "C" int ff_debug_timeline_read(unsigned long long* out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff::g_timeline), sizeof(unsigned long long) * 8 * n_blocks);
}
#endif

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" size_t ff_gemm_workspace_bytes(const ff_gemm_desc* d) {
    int split = d->split_k;
    if (split <= 0 && d->tile > 0 && d->dtype == FF_DTYPE_BF16) split = ff::plan_bf16(d->M, d->N, d->K, 1, 0, d->a_layout, d->b_layout, d->tile).split;
    return ff::gemm_workspace_bytes(d->dtype, d->M, d->N, d->K, 1, split);
}

extern "C" int ff_gemm(const ff_gemm_desc* d, const void* A, const void* B, void* C, void* aux_out, const void* aux_in,
                       const void* residual, const void* gate, void* workspace, size_t workspace_bytes, ff_stream_t stream) {
    using namespace ff;
    FF_CHECK(d && A && B && C, FF_ERR_SHAPE, "ff_gemm: null argument");
    FF_CHECK(d->dtype == FF_DTYPE_F32 || d->dtype == FF_DTYPE_BF16, FF_ERR_UNSUPPORTED, "ff_gemm: dtype %d", d->dtype);
    FF_CHECK(d->act_bwd < 0 || aux_in, FF_ERR_SHAPE, "ff_gemm: act_bwd needs aux_in");
    GemmParams P = {};
    P.M = d->M; P.N = d->N; P.K = d->K;
    P.a_layout = d->a_layout; P.b_layout = d->b_layout;
    P.a_map = make_rowmap(d->a_map); P.b_map = make_rowmap(d->b_map); P.c_map = make_rowmap(d->c_map); P.r_map = P.c_map;
    P.scale = d->scale; P.act = d->act; P.act_bwd = d->act_bwd; P.split_k = d->split_k;
    P.force_tile = d->tile; P.force_stages = d->stages;
    P.nz = 1;
    P.p[0] = GemmProblem{A, B, C, aux_out, aux_in, residual, gate};
    return gemm_launch(P, d->dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ff_gemm_profile_enable(int max_records) {
    using namespace ff;
    if (max_records <= 0) { g_prof.on = false; return FF_OK; }
    g_prof.on = false;
    if (max_records > g_prof.cap) {
        hipEvent_t* ev = (hipEvent_t*)realloc(g_prof.ev, sizeof(hipEvent_t) * 2 * max_records);
        ff_gemm_profile_record* rec = (ff_gemm_profile_record*)realloc(g_prof.rec, sizeof(ff_gemm_profile_record) * max_records);
        FF_CHECK(ev && rec, FF_ERR_WORKSPACE, "gemm profile: out of host memory");
        g_prof.ev = ev; g_prof.rec = rec;
        for (int i = 2 * g_prof.cap; i < 2 * max_records; i++) {
            hipError_t e = hipEventCreate(&g_prof.ev[i]);
            FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "hipEventCreate: %s", hipGetErrorString(e));
        }
        g_prof.cap = max_records;
    }
    g_prof.n = 0;
    g_prof.on = true;
    return FF_OK;
}
extern "C" int ff_gemm_profile_read(ff_gemm_profile_record* out, int max_records) {
    using namespace ff;
    const int n = std::min(std::min(g_prof.n.load(), g_prof.cap), max_records);
    for (int i = 0; i < n; i++) {
        hipEventSynchronize(g_prof.ev[2 * i + 1]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        g_prof.rec[i].ms = ms;
        out[i] = g_prof.rec[i];
    }
    g_prof.n = 0;
    return n;
}

extern "C" int ff_gemm_plan(const ff_gemm_desc* d, int* bm, int* bn, int* split_k) {   /* introspection: the tile / split-K the launcher would use */
    using namespace ff;
    FF_CHECK(d && bm && bn && split_k, FF_ERR_SHAPE, "ff_gemm_plan: null argument");
    if (d->dtype == FF_DTYPE_BF16) {
        const TilePlan p = plan_bf16(d->M, d->N, d->K, 1, d->split_k, d->a_layout, d->b_layout, d->tile);
        *bm = p.tile == 256128 || p.tile == 256256 ? 256 : p.tile == 128 || p.tile == 128160 || p.tile == 128002 ? 128 : p.tile == 3264 || p.tile == 3216 ? 32 : 64;
        *bn = p.tile == 3216 ? 16 : p.tile == 64 || p.tile == 64002 || p.tile == 3264 ? 64 : p.tile == 128160 ? 160 : p.tile == 256256 ? 256 : 128;
        *split_k = p.split;
    } else {
        *bm = *bn = kFBM;
        *split_k = d->split_k > 0 ? d->split_k : gemm_pick_split(d->dtype, d->M, d->N, d->K, 1);
    }
    return FF_OK;
}
