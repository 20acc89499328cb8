// Weight-streaming products for at most 32 activation rows - the cached decode step (one token per sequence: M = batch <= 32), which is
// pure weight traffic: every weight of a gated block is read once per token for a few MFMAs (gated_cross_attention.py:88-92,102-104,128-131
// through FeedForward, utils.py:45-50).
//
// The kernels of ff_gemm.hip serve these shapes badly: 32 x 64 tiles move the 13 MB of an FFW matrix at 1.2-1.4 TB/s (DMA ring, one tile's
// latency chain per workgroup, split-K + reduce launch for the long product), and the LDS-free streaming kernel (tile 3216) re-gathers all 32
// activation rows as MFMA fragments for every 16 output columns - 2 bytes of activations per byte of weights.  Here the roles are split the
// way the data wants it:
//   * the ACTIVATIONS (32 x K bf16, at most 128 KiB) come in ONCE per workgroup by LDS-DMA, in full 128-byte lines, and stay in LDS as the
//     K-major A operand of every MFMA (XOR-swizzled 16-byte chunks, ds_read_b128 fragments);
//   * the WEIGHTS never touch LDS: every lane loads the 16 bytes of weight row (n0 + c), k-chunk g that ARE its B fragment of
//     v_mfma_f32_16x16x32_bf16, and a wave issues ALL loads of its share of K before it waits for the first one - the whole weight slab of
//     a workgroup (~50 KiB) is in flight from the first microsecond, which is what a CU needs outstanding to pull its share of HBM bandwidth;
//   * one workgroup per CU, exactly: nb output columns per workgroup (a multiple of 4, <= 32) with N / nb (x K slices) = ~256 workgroups;
//   * the LayerNorm in front of the FFW up-projection (utils.py:45) is applied to the resident rows in LDS (two-pass statistics like torch),
//     so the separate LayerNorm launch and the normalised copy's round trip disappear (its by-products - mean, rstd, the normalised rows a
//     later weight gradient wants - are still written, spread over the workgroups);
//   * the long product (FFW down, K = 4 dim) splits K over workgroups and combines the fp32 partial tiles INSIDE the launch: each slice
//     publishes its 32 x nb tile (plain stores -> agent-scope release -> ticket), the last arriver of a tile acquires, sums the slabs and runs
//     the epilogue (tanh gate, residual).  The tickets are zeroed by the up-projection launch that always precedes it on the same stream.
#include "ff_common.h"
#include "ff_internal.h"
#include "ff_gemm_tiles.h"

namespace ff {

namespace {

constexpr int kDecRows = 32;          // activation rows held in LDS (two MFMA row groups)
constexpr int kDecWaves = 8;
constexpr int kDecMaxSteps = 8;       // 32-element k-steps per wave: K slice <= 8 waves x 8 steps x 32 = 2048

struct DecodeArgs {
    int M, N, kslice, kslices, nb, ln, act, has_next_tickets;
    float eps, scale;
    long long lda, ldb, ldc, ldr, ldxn;
    const bf16* A;            // [M][lda]: activation rows (raw y1 when ln != 0)
    const bf16* B;            // [N][ldb]: nn.Linear weight (out, in)
    const bf16* gamma;
    const bf16* beta;
    bf16* C;
    bf16* aux_out;
    const bf16* residual;
    const bf16* gate;
    bf16* xn_out;             // ln: the normalised rows (operand of a later weight gradient)
    float* mean;
    float* rstd;
    float* partial;           // kslices > 1: [kslices][kDecRows][N] fp32
    unsigned* tickets;        // kslices > 1: one per column group, zero at entry
    unsigned* zero_tickets;   // tickets of the NEXT launch on this stream, zeroed here (n_zero of them)
    int n_zero;
};

#ifdef FF_DEC_TIMELINE   // development probe (tools/experiments/decode_probe.hip): per-workgroup phase timestamps, 100 MHz constant clock
__device__ unsigned long long g_dec_timeline[1024 * 8];
#define FF_DTL(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_dec_timeline[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FF_DTL(i) do { } while (0)
#endif

FF_DEV void dec_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// 4 consecutive output columns n .. n + 3 of row m: scale -> aux_out -> tanh gate -> activation -> + residual -> C (ff_gemm.hip: epilogue4)
FF_DEV void dec_epilogue4(const DecodeArgs& a, int m, int n, float (&v)[4], float gate) {
    const bool full = n + 3 < a.N;
    auto store4 = [&](bf16* base, long long ld) {
        bf16* p = base + (long long)m * ld + n;
        if (full) {
            bf16x4 t;
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = (bf16)v[r];
            *(bf16x4*)p = t;
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (n + r < a.N) p[r] = (bf16)v[r];
        }
    };
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] *= a.scale;
    if (a.aux_out) store4(a.aux_out, a.ldc);
    if (a.gate) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] *= gate;
    }
    if (a.act >= 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = act_fwd_t<bf16>(v[r], a.act);
    }
    if (a.residual) {
        const bf16* q = a.residual + (long long)m * a.ldr + n;
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (n + r < a.N) v[r] += (float)q[r];
    }
    store4(a.C, a.ldc);
}

// NT: nontemporal weight loads (each weight byte is read once, by one CU).  LN: LayerNorm prologue on the resident rows (two instantiations
// so that a kernel trace tells the up-projection launches from the down-projection ones).
template <bool NT, bool LN>
__global__ __launch_bounds__(kDecWaves * 64) void decode_rows32_kernel(const bf16* hA, const bf16* hB, const bf16* hG, const bf16* hBt, int hM, int hN,
                                                                       int h_kslice, int h_kslices, int h_nb, int h_ld, const DecodeArgs a_in) {
    // The leading scalars are everything the path to the last load instruction needs: plain kernel parameters, which the hardware preloads
    // into SGPRs at wave launch (-amdgpu-kernarg-preload-count, build.py) - the struct behind them lives in HBM and costs a ~1 us scalar
    // round trip, which now overlaps the loads in flight (probe: "loads issued" at 2.5 us after launch with everything in the struct).
    // 14 dwords are preloaded (16 user SGPRs minus the kernarg pointer): four pointers and six integers; h_ld = leading dimension of BOTH
    // operands (K-major rows of the full contraction length).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FF_DTL(0);
    const int t = threadIdx.x, l = t & 63, c = l & 15, g = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n_groups = (hN + h_nb - 1) / h_nb;
    const int lin = xcd_remap(blockIdx.x, n_groups * h_kslices);      // the K slices of a column group are neighbours: same XCD (speed only)
    const int cg = lin / h_kslices, ks = lin - cg * h_kslices;
    const int n0 = cg * h_nb;
    const int k0 = ks * h_kslice;
    const int nk = h_kslice / kBK;                                      // 64-element LDS tiles of the slice
    bf16* sA = (bf16*)smem;                                             // [nk][32][64], 16-byte chunks XOR-swizzled by row
    const int vec_elems = (h_kslice + 511) / 512 * 512;                 // gamma / beta padded to whole DMA instructions (out-of-range lanes write zeros)
    bf16* s_g = sA + nk * (kDecRows * kBK);
    bf16* s_b = s_g + vec_elems;
    unsigned* s_flag = (unsigned*)(s_b + vec_elems);

    // ---- the activation rows of this K slice: LDS-DMA, 8 rows x 128 bytes per wave instruction, four instructions per 64-element tile ----
    {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)hA, 0, 0x7fffffff, 0x00020000);
        const int p = w & 3, row = p * 8 + (l >> 3), cp = l & 7;
        const unsigned voff = row < hM ? (unsigned)((long long)row * h_ld + k0 + ((cp ^ (row & 7)) << 3)) * 2u : kOobOffset;
        for (int tile = w >> 2; tile < nk; tile += 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, FF_LDS_PTR(void, sA + tile * (kDecRows * kBK) + p * 8 * kBK), 16, voff, (unsigned)tile * (kBK * 2), 0, 0);
        if (LN) {       // gamma / beta of the slice, lane-linear
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)hG, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rbt = __builtin_amdgcn_make_buffer_rsrc((void*)hBt, 0, 0x7fffffff, 0x00020000);
            const int nch = h_kslice / 8;
            for (int i0 = w * 64; i0 < nch; i0 += kDecWaves * 64) {
                const unsigned off = i0 + l < nch ? (unsigned)(k0 / 8 + i0 + l) * 16u : kOobOffset;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, FF_LDS_PTR(void, s_g + i0 * 8), 16, off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rbt, FF_LDS_PTR(void, s_b + i0 * 8), 16, off, 0, 0, 0);
            }
        }
    }
    asm volatile("" ::: "memory");                                      // (the weight loads below stay behind the DMA pieces: the counted wait relies on it)
    // ---- ALL weight fragments of this wave: k-steps [s_begin, s_end) of the slice, rows n0 .. n0 + nb - 1 in two 16-row groups ----
    const int steps = h_kslice / 32, spw = (steps + kDecWaves - 1) / kDecWaves;
    const int s_begin = w * spw, s_end = min(steps, s_begin + spw);
    bf16x8 fb0[kDecMaxSteps], fb1[kDecMaxSteps];
    {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)hB, 0, 0x7fffffff, 0x00020000);
        const unsigned b0_off = (c < h_nb && n0 + c < hN) ? (unsigned)((long long)(n0 + c) * h_ld + k0 + g * 8) * 2u : kOobOffset;
        const unsigned b1_off = (16 + c < h_nb && n0 + 16 + c < hN) ? (unsigned)((long long)(n0 + 16 + c) * h_ld + k0 + g * 8) * 2u : kOobOffset;
#pragma unroll
        for (int i = 0; i < kDecMaxSteps; i++) {
            const bool in = s_begin + i < s_end;                        // wave-uniform
            const unsigned soff = in ? (unsigned)(s_begin + i) * 64u : 0u;
            fb0[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, in ? b0_off : kOobOffset, soff, NT ? 2 : 0));
            fb1[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, in ? b1_off : kOobOffset, soff, NT ? 2 : 0));
        }
    }
    FF_DTL(1);
    const DecodeArgs a = fetch_args(a_in);                              // epilogue arguments: the scalar round trip overlaps the loads in flight
    if (a.zero_tickets && blockIdx.x == 0)                              // the next launch's tickets (visible at the kernel boundary)
        for (int i = t; i < a.n_zero; i += kDecWaves * 64) a.zero_tickets[i] = 0u;
    // the DMA pieces were issued before the 2 x kDecMaxSteps weight loads: they have landed once at most that many loads are outstanding
    // (the ticket stores above are younger still: vmcnt counts them too, so the bound only gets safer)
    wait_vmcnt<2 * kDecMaxSteps>();
    dec_barrier();
    FF_DTL(2);

    // ---- LayerNorm of the resident rows, in place: 16 threads per row.  One pass for the statistics - sums of (x - c) and (x - c)^2 around
    //      the row's first element c, eight independent accumulators per thread (the first version's two serial passes of 80 dependent adds
    //      per thread cost 6 us of a 12 us launch) - and one to normalise; every pass re-reads LDS, registers belong to the weights in flight ----
    if (LN) {
        const int r = t >> 4, s16 = t & 15;
        const int nch = nk * 8;
        auto piece = [&](int ci) { return sA + (ci >> 3) * (kDecRows * kBK) + r * kBK + (((ci & 7) ^ (r & 7)) << 3); };
        // (the slice is a whole number of 128-element blocks - decode_ffw_supported - so every thread owns nch / 16 pieces; two pieces per
        // iteration, and when that count is odd (d an odd multiple of 128) the last iteration's second piece does not exist: `two`)
        // shift of the one-pass sums: the mean of the row's first 16 elements (two 16-byte pieces, read by every thread of the row).  A single
        // element as the shift loses the variance to cancellation when that element is an outlier of its row (massive-activation channels).
        float shift;
        {
            const bf16x8 h0 = *(const bf16x8*)piece(0), h1 = *(const bf16x8*)piece(1);
            float hs = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) hs += (float)h0[e] + (float)h1[e];
            shift = hs * (1.f / 16.f);
        }
        float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c0 = s16; c0 < nch; c0 += 32) {
            const bool two = c0 + 16 < nch;
            const bf16x8 x0 = *(const bf16x8*)piece(c0), x1 = *(const bf16x8*)piece(two ? c0 + 16 : c0);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float d0 = (float)x0[e] - shift, d1 = two ? (float)x1[e] - shift : 0.f;
                s1[e] += d0 + d1;
                s2[e] = fmaf(d1, d1, fmaf(d0, d0, s2[e]));
            }
        }
        float sum = ((s1[0] + s1[1]) + (s1[2] + s1[3])) + ((s1[4] + s1[5]) + (s1[6] + s1[7]));
        float sq = ((s2[0] + s2[1]) + (s2[2] + s2[3])) + ((s2[4] + s2[5]) + (s2[6] + s2[7]));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
        const float inv_n = 1.f / (float)h_kslice;
        const float dm = sum * inv_n;                                   // mean - shift
        const float mu = shift + dm;
        const float rs = rsqrtf(fmaxf(sq * inv_n - dm * dm, 0.f) + a.eps);
        const bool rok = r < hM;
        // by-products for backward, written once: row r by workgroup r % gridDim (every workgroup holds every row)
        const bool writer = rok && (int)(r % gridDim.x) == (int)blockIdx.x;
        if (writer && s16 == 0) { a.mean[r] = mu; a.rstd[r] = rs; }
        const float nmr = -mu * rs;
        for (int c0 = s16; c0 < nch; c0 += 32) {
            const int np = c0 + 16 < nch ? 2 : 1;
            bf16x8 x[2], gq[2], bq[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int ci = u < np ? c0 + 16 * u : c0;
                x[u] = *(const bf16x8*)piece(ci);
                gq[u] = *(const bf16x8*)(s_g + ci * 8);
                bq[u] = *(const bf16x8*)(s_b + ci * 8);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (u >= np) break;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = fmaf(fmaf((float)x[u][e], rs, nmr), (float)gq[u][e], (float)bq[u][e]);
                Vec<bf16>::store(piece(c0 + 16 * u), v);
                if (writer && a.xn_out) Vec<bf16>::store(a.xn_out + (long long)r * a.ldxn + (c0 + 16 * u) * 8, v);
            }
        }
        dec_barrier();
    }

    FF_DTL(3);
    // ---- D[n][m] += W[n][k] . X[m][k] over this wave's k-steps: 2 row groups x 2 column groups ----
    f32x4 acc[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
    const bool two_cols = a.nb > 16;
#pragma unroll
    for (int i = 0; i < kDecMaxSteps; i++) {
        const int s = s_begin + i;
        if (s < s_end) {
            const bf16* tile = sA + (s >> 1) * (kDecRows * kBK);
            const int chunk = (s & 1) * 4 + g;
            const bf16x8 fa0 = *(const bf16x8*)(tile + c * kBK + ((chunk ^ (c & 7)) << 3));
            const bf16x8 fa1 = *(const bf16x8*)(tile + (16 + c) * kBK + ((chunk ^ (c & 7)) << 3));       // (16 + c) & 7 == c & 7
            acc[0][0] = mfma_bf16(fb0[i], fa0, acc[0][0]);              // D[n][m]: lane (c, g) holds row m = c, columns n0 + 4 g .. + 3
            acc[1][0] = mfma_bf16(fb0[i], fa1, acc[1][0]);
            if (two_cols) {
                acc[0][1] = mfma_bf16(fb1[i], fa0, acc[0][1]);
                acc[1][1] = mfma_bf16(fb1[i], fa1, acc[1][1]);
            }
        }
    }
    // ---- the eight waves' partial tiles meet in LDS (the activation rows are dead) ----
    __syncthreads();
    FF_DTL(4);
    f32x4* red = (f32x4*)smem;                                          // [wave][row group][column group][64 lanes]
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) red[((w * 2 + i) * 2 + j) * 64 + l] = acc[i][j];
    __syncthreads();
    const int n_items = 2 * (two_cols ? 2 : 1) * 64;                    // (row group, column group, lane) = 4 columns of one row
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int m = 0, n = 0;
    bool mine = false;
    if (t < n_items) {
        const int i = t >> (two_cols ? 7 : 6), j = two_cols ? (t >> 6) & 1 : 0, ll = t & 63;
        f32x4 sum = red[((0 * 2 + i) * 2 + j) * 64 + ll];
#pragma unroll
        for (int ww = 1; ww < kDecWaves; ww++) sum += red[((ww * 2 + i) * 2 + j) * 64 + ll];
        m = i * 16 + (ll & 15);
        const int col = j * 16 + (ll >> 4) * 4;
        n = n0 + col;
        mine = m < a.M && col < a.nb && n < a.N;
        v[0] = sum[0]; v[1] = sum[1]; v[2] = sum[2]; v[3] = sum[3];
    }
    const float gate = a.gate ? tanhf((float)*a.gate) : 1.f;
    if (a.kslices == 1) {
        if (mine) dec_epilogue4(a, m, n, v, gate);
        FF_DTL(5);
        return;
    }
    // ---- K split over workgroups: publish the partial tile, the last arriver of the column group combines.  The slab is 2.5 KB per workgroup:
    //      written THROUGH (sc1 stores), drained, then one relaxed agent-scope ticket; the last arriver reads all slabs with sc1 loads.  No
    //      release / acquire fences (MI355X guide, Guideline 16, the write-through form): the first version's plain stores + agent release
    //      fence cost 4.6 us per launch - buffer_wbl2 writes back whatever is dirty in the XCD's L2, not just this tile. ----
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.partial, 0, 0x7fffffff, 0x00020000);
    if (mine) {
        const unsigned off = (unsigned)(((long long)ks * kDecRows + m) * a.N + n) * 4u;      // (N % 4 == 0: 16-byte aligned)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, f32x4{v[0], v[1], v[2], v[3]}), rp, off, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) *s_flag = __hip_atomic_fetch_add(a.tickets + cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    FF_DTL(5);
    if (*s_flag != (unsigned)(a.kslices - 1)) return;
    if (mine) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < a.kslices; s++) {
            const unsigned off = (unsigned)(((long long)s * kDecRows + m) * a.N + n) * 4u;
            sum += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, 16));
        }
        float o[4] = {sum[0], sum[1], sum[2], sum[3]};
        dec_epilogue4(a, m, n, o, gate);
    }
    FF_DTL(6);
}

int launch_decode(const DecodeArgs& a, hipStream_t st) {
    // rows + gamma / beta + flag; at least the 32 KiB the eight waves' partial tiles need when they meet in the (then dead) row buffer
    const size_t lds = std::max((size_t)kDecRows * a.kslice * 2 + (size_t)2 * ((a.kslice + 511) / 512 * 512) * 2 + 64, (size_t)kDecWaves * 4 * 64 * sizeof(f32x4) + 64);
    static const int nt = dbg_switch("FF_DECODE_NT", 1);
    const int grid = cdiv(a.N, a.nb) * a.kslices;
    static bool attr_done[64][4] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto set_attr = [&](const void* fn, int which) -> int {
        if (dev < 0 || dev >= 64 || !attr_done[dev][which]) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            FF_CHECK(e == hipSuccess, FF_ERR_LAUNCH, "hipFuncSetAttribute(decode_rows32): %s", hipGetErrorString(e));
            if (dev >= 0 && dev < 64) attr_done[dev][which] = true;
        }
        return FF_OK;
    };
    FF_CHECK(lds <= 160 * 1024, FF_ERR_UNSUPPORTED, "decode kernel: K slice %d does not fit LDS", a.kslice);
    FF_CHECK(a.lda == a.ldb && a.lda == (long long)a.kslice * a.kslices, FF_ERR_SHAPE, "decode kernel: both operands K-major with leading dimension K");
#define FF_DEC_LAUNCH(NT_, LN_)                                                                       \
    do {                                                                                              \
        FF_TRY(set_attr((const void*)decode_rows32_kernel<NT_, LN_>, (NT_ ? 2 : 0) + (LN_ ? 1 : 0))); \
        decode_rows32_kernel<NT_, LN_><<<dim3(grid), dim3(kDecWaves * 64), lds, st>>>(a.A, a.B, a.gamma, a.beta, a.M, a.N, a.kslice, a.kslices, a.nb, \
                                                                                      (int)a.lda, a);                                               \
    } while (0)
    if (nt && a.ln) FF_DEC_LAUNCH(true, true);
    else if (nt) FF_DEC_LAUNCH(true, false);
    else if (a.ln) FF_DEC_LAUNCH(false, true);
    else FF_DEC_LAUNCH(false, false);
#undef FF_DEC_LAUNCH
    return check_launch("decode_rows32");
}

// columns per workgroup so that (N / nb) * kslices lands on ~one workgroup per CU
int pick_nb(int N, int kslices) {
    int nb = (int)((((long long)N * kslices + 255) / 256 + 3) / 4 * 4);
    return std::max(4, std::min(32, nb));
}
int pick_kslices(int K) {       // fewest slices whose rows fit LDS (32 x 2048 bf16 = 128 KiB)
    for (int s = 1; s <= 8; s++)
        if (K % s == 0 && (K / s) % kBK == 0 && K / s <= kDecMaxSteps * kDecWaves * 32) return s;
    return 0;
}

}  // namespace

bool decode_ffw_supported(int dtype, int M, int d, int ffi) {
    static const int on = dbg_switch("FF_DECODE_FFW", 1);
    if (!on || dtype != FF_DTYPE_BF16 || M > kDecRows || M < 1) return false;
    if (d % 128 != 0 || ffi % kBK != 0 || d > kDecMaxSteps * kDecWaves * 32) return false;      // (d % 128: the LayerNorm pass, two 64-element tiles per step)
    return pick_kslices(ffi) > 0;
}
size_t decode_ffw_workspace_bytes(int d, int ffi) {
    const int ks = pick_kslices(ffi);
    if (ks <= 1) return 256;
    return align_up((size_t)ks * kDecRows * d * sizeof(float)) + align_up((size_t)cdiv(d, pick_nb(d, ks)) * sizeof(unsigned));
}

// y_out = y1 + tanh(alpha) * W3 act(W1 LN(y1))   (utils.py:45-50 inside gated_cross_attention.py:182) for M <= 32 rows, two launches.
// Saves what the general path saves: mean / rstd / the normalised rows, the pre-activation H, the activation A, the branch output.
int decode_ffw(int M, int d, int ffi, int act, float eps, const void* y1, const void* gamma, const void* beta, const void* W1, const void* W3,
               const void* alpha, void* xn, float* mean, float* rstd, void* Hpre, void* Aact, void* ffw_out, void* y_out, void* ws, size_t ws_bytes,
               hipStream_t st) {
    const int ks = pick_kslices(ffi);
    FF_CHECK(ks > 0, FF_ERR_UNSUPPORTED, "decode_ffw: unsupported width %d", ffi);
    FF_CHECK(ws_bytes >= decode_ffw_workspace_bytes(d, ffi) && ws, FF_ERR_WORKSPACE, "decode_ffw: workspace too small");
    const int nb_down = pick_nb(d, ks);
    float* partial = (float*)ws;
    unsigned* tickets = (unsigned*)((char*)ws + align_up((size_t)ks * kDecRows * d * sizeof(float)));
    DecodeArgs up = {};
    up.M = M; up.N = ffi; up.kslice = d; up.kslices = 1; up.nb = pick_nb(ffi, 1); up.ln = 1; up.act = act; up.eps = eps; up.scale = 1.f;
    up.lda = d; up.ldb = d; up.ldc = ffi; up.ldr = 0; up.ldxn = d;
    up.A = (const bf16*)y1; up.B = (const bf16*)W1; up.gamma = (const bf16*)gamma; up.beta = (const bf16*)beta;
    up.C = (bf16*)Aact; up.aux_out = (bf16*)Hpre; up.xn_out = (bf16*)xn; up.mean = mean; up.rstd = rstd;
    if (ks > 1) { up.zero_tickets = tickets; up.n_zero = cdiv(d, nb_down); }
    FF_TRY(launch_decode(up, st));
    DecodeArgs dn = {};
    dn.M = M; dn.N = d; dn.kslice = ffi / ks; dn.kslices = ks; dn.nb = nb_down; dn.ln = 0; dn.act = FF_ACT_NONE; dn.eps = 0.f; dn.scale = 1.f;
    dn.lda = ffi; dn.ldb = ffi; dn.ldc = d; dn.ldr = d;
    dn.A = (const bf16*)Aact; dn.B = (const bf16*)W3; dn.C = (bf16*)y_out; dn.aux_out = (bf16*)ffw_out; dn.residual = (const bf16*)y1;
    dn.gate = (const bf16*)alpha; dn.partial = partial; dn.tickets = tickets;
    return launch_decode(dn, st);
}

}  // namespace ff
