// GEMM family of the fusion path: every nn.Linear of the resampler / gated-xattn blocks, forward (X W^T),
// data-gradient (dY W) and weight-gradient (dY^T X), with the elementwise neighbours fused into the epilogue
// (activation, activation-backward, residual, tanh(alpha) gate, branch-output store).
//
//   bf16 : v_mfma_f32_16x16x32_bf16, 128x128x64 or 64x64x64 block tiles, 4 waves (2x2), LDS double buffer fed
//          through registers (global loads of tile t+1 fly under the MFMAs of tile t).  Operands whose
//          contraction index is the slow (strided) one are staged as-is and read with ds_read_b64_tr_b16.
//   fp32 : v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) — the verification precision.
//
// Accumulators are kept transposed (D[n][m] = mfma(Bfrag, Afrag)) so a lane owns 4 consecutive n of one row m
// and the epilogue issues 8/16-byte row-contiguous loads and stores.
#include "ff_common.h"
#include "ff_internal.h"

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
        for (int r = 0; r < 4; r++) v[r] = act_fwd(v[r], P.act);
    }
    if (P.act_bwd >= 0) {
        float h[4];
        load4(pr.aux_in, off, h);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] *= act_grad(h[r], P.act_bwd);
    }
    if (pr.residual) {
        float q[4];
        load4(pr.residual, P.r_map.off(m) + n, q);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] += q[r];
    }
    store4(pr.C, v);
}

// XCD-aware tile order: consecutive logical tiles (same A row panel) land on the same XCD / L2.
FF_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct TileCoord {
    int z, split, tm, tn;
};
FF_DEV TileCoord tile_coord(const GemmParams& P, int tiles_m, int tiles_n) {
    const int per_z = tiles_m * tiles_n;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    TileCoord c;
    c.z = bid / (per_z * P.split_k);
    bid -= c.z * per_z * P.split_k;
    c.split = bid / per_z;
    bid -= c.split * per_z;
    c.tm = bid / tiles_n;
    c.tn = bid - c.tm * tiles_n;
    return c;
}

// ------------------------------------------------------------------------------------------------
// bf16 kernel
// ------------------------------------------------------------------------------------------------
constexpr int kBK = 64;    // bf16 K tile
constexpr int kMPad = 16;  // row padding (elements) of M-major LDS tiles

template <int BR, int LAYOUT> struct TileGeom {  // one operand tile: BR rows (M or N) x kBK
    static constexpr int elems = LAYOUT == 0 ? BR * kBK : kBK * (BR + kMPad);
    static constexpr int nreg = BR / 32;  // 16-byte registers per thread per tile
};

// global -> registers.  LAYOUT 0: source rows are the BR tile rows, K contiguous.  LAYOUT 1: source rows are K.
template <int BR, int LAYOUT>
FF_DEV void tile_load(const bf16* __restrict__ base, const RowMap& map, int row_base, int row_lim, int k0, int k_end,
                      const long long* row_off, uint4 (&reg)[BR / 32]) {
    const int t = threadIdx.x;
    if (LAYOUT == 0) {
        const int chunk = t & 7;
        const int k = k0 + chunk * 8;
#pragma unroll
        for (int p = 0; p < BR / 32; p++) {
            const int row = p * 32 + (t >> 3);
            uint4 v = {0, 0, 0, 0};
            if (row_base + row < row_lim && k < k_end) v = *(const uint4*)(base + row_off[p] + k);
            reg[p] = v;
        }
    } else {
        constexpr int CPR = BR / 8;  // 16-byte chunks per k row
        constexpr int RPP = 256 / CPR;
        const int mc = t % CPR;
        const int col = row_base + mc * 8;
#pragma unroll
        for (int p = 0; p < BR / 32; p++) {
            const int kr = k0 + p * RPP + t / CPR;
            uint4 v = {0, 0, 0, 0};
            if (kr < k_end && col < row_lim) v = *(const uint4*)(base + map.off(kr) + col);
            reg[p] = v;
        }
    }
}

template <int BR, int LAYOUT> FF_DEV void tile_store(bf16* s, const uint4 (&reg)[BR / 32]) {
    const int t = threadIdx.x;
    if (LAYOUT == 0) {
        const int chunk = t & 7;
#pragma unroll
        for (int p = 0; p < BR / 32; p++) {
            const int row = p * 32 + (t >> 3);
            *(uint4*)(s + row * kBK + ((chunk ^ (row & 7)) << 3)) = reg[p];
        }
    } else {
        constexpr int CPR = BR / 8;
        constexpr int RPP = 256 / CPR;
        const int mc = t % CPR;
#pragma unroll
        for (int p = 0; p < BR / 32; p++) {
            const int kr = p * RPP + t / CPR;
            *(uint4*)(s + kr * (BR + kMPad) + mc * 8) = reg[p];
        }
    }
}

// LDS -> MFMA fragment of 16 tile rows starting at r0, k-step ks (32 wide)
template <int BR, int LAYOUT> FF_DEV bf16x8 frag_read(const bf16* s, int r0, int ks) {
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
    if (LAYOUT == 0) {
        const int row = r0 + c;
        const int chunk = ks * 4 + g;
        return *(const bf16x8*)(s + row * kBK + ((chunk ^ (row & 7)) << 3));
    } else {
        const int k = ks * 32 + g * 8 + (c >> 2);
        const bf16* p = s + k * (BR + kMPad) + r0 + (c & 3) * 4;
        return cat4(lds_read_tr16(p), lds_read_tr16(p + 4 * (BR + kMPad)));
    }
}

template <int BM, int BN, int AL, int BL>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef TileGeom<BM, AL> GA;
    typedef TileGeom<BN, BL> GB;
    constexpr int STAGE = GA::elems + GB::elems;
    bf16* smem = (bf16*)smem_raw;
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16;

    const int tiles_m = (P.M + BM - 1) / BM, tiles_n = (P.N + BN - 1) / BN;
    const TileCoord tc = tile_coord(P, tiles_m, tiles_n);
    const GemmProblem& pr = P.p[tc.z];
    const int m_base = tc.tm * BM, n_base = tc.tn * BN;
    const int k_begin = tc.split * P.k_per_split;
    const int k_end = min(P.K, k_begin + P.k_per_split);
    const bf16* A = (const bf16*)pr.A;
    const bf16* B = (const bf16*)pr.B;

    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int wm = w >> 1, wn = w & 1;

    long long a_off[GA::nreg], b_off[GB::nreg];
    if (AL == 0) {
#pragma unroll
        for (int p = 0; p < GA::nreg; p++) a_off[p] = P.a_map.off(min(m_base + p * 32 + (t >> 3), P.M - 1));
    }
    if (BL == 0) {
#pragma unroll
        for (int p = 0; p < GB::nreg; p++) b_off[p] = P.b_map.off(min(n_base + p * 32 + (t >> 3), P.N - 1));
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[GA::nreg], rb[GB::nreg];
    const int nk = (k_end - k_begin + kBK - 1) / kBK;
    if (nk > 0) {
        tile_load<BM, AL>(A, P.a_map, m_base, P.M, k_begin, k_end, a_off, ra);
        tile_load<BN, BL>(B, P.b_map, n_base, P.N, k_begin, k_end, b_off, rb);
        tile_store<BM, AL>(smem, ra);
        tile_store<BN, BL>(smem + GA::elems, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const bf16* sA = smem + (kt & 1) * STAGE;
        const bf16* sB = sA + GA::elems;
        const bool more = kt + 1 < nk;
        if (more) {
            const int k0 = k_begin + (kt + 1) * kBK;
            tile_load<BM, AL>(A, P.a_map, m_base, P.M, k0, k_end, a_off, ra);
            tile_load<BN, BL>(B, P.b_map, n_base, P.N, k0, k_end, b_off, rb);
        }
#pragma unroll
        for (int ks = 0; ks < kBK / 32; ks++) {
            bf16x8 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; i++) fa[i] = frag_read<BM, AL>(sA, wm * WM + i * 16, ks);
#pragma unroll
            for (int j = 0; j < NT; j++) fb[j] = frag_read<BN, BL>(sB, wn * WN + j * 16, ks);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++) acc[i][j] = mfma_bf16(fb[j], fa[i], acc[i][j]);  // D[n][m]
        }
        if (more) {
            bf16* dA = smem + ((kt + 1) & 1) * STAGE;
            tile_store<BM, AL>(dA, ra);
            tile_store<BN, BL>(dA + GA::elems, rb);
        }
        __syncthreads();
    }

    // epilogue: lane owns row m = ..+(l&15), columns n = ..+(l>>4)*4 .. +3
    const int c = l & 15, g = l >> 4;
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const int m = m_base + wm * WM + i * 16 + c;
        if (m >= P.M) continue;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int n = n_base + wn * WN + j * 16 + g * 4;
            if (n >= P.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (P.split_k > 1) {
                float* dst = P.partial + ((long long)(tc.z * P.split_k + tc.split) * P.M + m) * P.N + n;
                *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};  // N % 4 == 0 checked on the host
            } else {
                epilogue4<bf16>(P, pr, m, n, v);
            }
        }
    }
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
    const TileCoord tc = tile_coord(P, tiles_m, tiles_n);
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

// split-K: sum the fp32 partial slabs, then the same epilogue
template <typename T> __global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(const GemmParams P) {
    const int n4 = (P.N + 3) / 4;
    const long long total = (long long)P.nz * P.M * n4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int nq = (int)(idx % n4);
        const long long rest = idx / n4;
        const int m = (int)(rest % P.M), z = (int)(rest / P.M);
        const int n = nq * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < P.split_k; s++) {
            const float* src = P.partial + ((long long)(z * P.split_k + s) * P.M + m) * P.N + n;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (n + r < P.N) v[r] += src[r];
        }
        epilogue4<T>(P, P.p[z], m, n, v);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int AL, int BL> static int launch_bf16(const GemmParams& P, hipStream_t st) {
    constexpr size_t lds = 2 * (TileGeom<BM, AL>::elems + TileGeom<BN, BL>::elems) * sizeof(bf16);
    static bool attr_done = false;
    if (!attr_done) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, AL, BL>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "hipFuncSetAttribute(gemm lds=%zu): %s", lds, hipGetErrorString(e));
        }
        attr_done = true;
    }
    const int tiles = cdiv(P.M, BM) * cdiv(P.N, BN);
    const int grid = tiles * P.split_k * P.nz;
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AL, BL>), dim3(grid), dim3(256), lds, st, P);
    return check_launch("gemm_bf16");
}
template <int BM, int BN> static int dispatch_bf16(const GemmParams& P, hipStream_t st) {
    if (P.a_layout == 0 && P.b_layout == 0) return launch_bf16<BM, BN, 0, 0>(P, st);
    if (P.a_layout == 0 && P.b_layout == 1) return launch_bf16<BM, BN, 0, 1>(P, st);
    if (P.a_layout == 1 && P.b_layout == 0) return launch_bf16<BM, BN, 1, 0>(P, st);
    return launch_bf16<BM, BN, 1, 1>(P, st);
}
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
struct ProfState {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;     // 2 events per record
    ff_gemm_profile_record* rec = nullptr;
} g_prof;
}
static int prof_begin(const GemmParams& P, int dtype, int bm, hipStream_t st) {
    if (!g_prof.on || g_prof.n >= g_prof.cap) return -1;
    const int i = g_prof.n++;
    ff_gemm_profile_record& r = g_prof.rec[i];
    r.dtype = dtype; r.tile = bm; r.a_layout = P.a_layout; r.b_layout = P.b_layout;
    r.M = P.M; r.N = P.N; r.K = P.K; r.nz = P.nz; r.split_k = P.split_k; r.ms = 0.f;
    hipEventRecord(g_prof.ev[2 * i], st);
    return i;
}
static void prof_end(int i, hipStream_t st) {
    if (i >= 0) hipEventRecord(g_prof.ev[2 * i + 1], st);
}

static bool big_tile(const GemmParams& P) { return (long long)cdiv(P.M, 128) * cdiv(P.N, 128) * P.nz >= 160; }

int gemm_pick_split(int dtype, int M, int N, int K, int nz) {
    const int bm = dtype == FF_DTYPE_BF16 ? ((long long)cdiv(M, 128) * cdiv(N, 128) * nz >= 160 ? 128 : 64) : kFBM;
    const long long tiles = (long long)cdiv(M, bm) * cdiv(N, bm) * nz;
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

int gemm_launch(GemmParams P, int dtype, void* workspace, size_t ws_bytes, hipStream_t st) {
    FF_CHECK(P.M > 0 && P.N > 0 && P.K > 0 && P.nz >= 1 && P.nz <= kGemmMaxZ, FF_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d nz=%d",
             P.M, P.N, P.K, P.nz);
    if (P.split_k <= 0) P.split_k = gemm_pick_split(dtype, P.M, P.N, P.K, P.nz);
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
    const int vec = dtype == FF_DTYPE_BF16 ? 8 : 4;
    auto map_ok = [&](const RowMap& m) { return m.ld % vec == 0 && (m.rows_per_seg <= 0 || m.seg_stride % vec == 0); };
    const int a_contig = P.a_layout == 0 ? P.K : P.M, b_contig = P.b_layout == 0 ? P.K : P.N;
    P.a_vec_ok = map_ok(P.a_map) && a_contig % vec == 0;
    P.b_vec_ok = map_ok(P.b_map) && b_contig % vec == 0;
    for (int z = 0; z < P.nz; z++) {
        P.a_vec_ok = P.a_vec_ok && ((uintptr_t)P.p[z].A % 16 == 0);
        P.b_vec_ok = P.b_vec_ok && ((uintptr_t)P.p[z].B % 16 == 0);
    }
    int rc;
    const int prof_id = prof_begin(P, dtype, dtype == FF_DTYPE_BF16 ? (big_tile(P) ? 128 : 64) : kFBM, st);
    if (dtype == FF_DTYPE_BF16) {
        FF_CHECK(P.a_vec_ok && P.b_vec_ok, FF_ERR_UNSUPPORTED,
                 "bf16 gemm needs 16-byte aligned operands with contiguous dims %% 8 == 0 (M=%d N=%d K=%d)", P.M, P.N, P.K);
        FF_CHECK(P.N % 4 == 0 && P.c_map.ld % 4 == 0, FF_ERR_UNSUPPORTED, "bf16 gemm needs N %% 4 == 0 (N=%d)", P.N);
        rc = big_tile(P) ? dispatch_bf16<128, 128>(P, st) : dispatch_bf16<64, 64>(P, st);
    } else {
        rc = dispatch_f32(P, st);
    }
    prof_end(prof_id, st);   // main kernel only: the split-K epilogue is a separate (HBM-bound) kernel
    FF_TRY(rc);
    if (P.split_k > 1) {
        const long long total = (long long)P.nz * P.M * ((P.N + 3) / 4);
        const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
        if (dtype == FF_DTYPE_BF16) hipLaunchKernelGGL(gemm_splitk_epilogue_kernel<bf16>, dim3(grid), dim3(256), 0, st, P);
        else hipLaunchKernelGGL(gemm_splitk_epilogue_kernel<float>, dim3(grid), dim3(256), 0, st, P);
        FF_TRY(check_launch("gemm_splitk_epilogue"));
    }
    return FF_OK;
}

}  // namespace ff

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" size_t ff_gemm_workspace_bytes(const ff_gemm_desc* d) {
    return ff::gemm_workspace_bytes(d->dtype, d->M, d->N, d->K, 1, d->split_k);
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
    P.nz = 1;
    P.p[0] = GemmProblem{A, B, C, aux_out, aux_in, residual, gate};
    return gemm_launch(P, d->dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ff_gemm_profile_enable(int max_records) {
    using namespace ff;
    if (max_records <= 0) { g_prof.on = false; return FF_OK; }
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
    const int n = g_prof.n < max_records ? g_prof.n : max_records;
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
