// Operand-fill microbenchmark for the producer side of gemm_bf16_pc_kernel (round 3): how fast can ONE CU pull the operand stages of a
// 128 x 160 GEMM tile (K-major A and B, 64-element k-steps = 288 rows x 128 B = 36 KiB per stage) when nothing consumes them?
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), the library's way          mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR, folded into a checksum (no LDS write)
// swept over the number of issuing waves (1, 2, 4, 8), the ring depth (stages in flight) and the sharing pattern of the sources
// (`gemm`: tiles of a 1024 x 5120 x 1280 product laid over the chip like the library does, operands shared through the L2s;
//  `private`: every CU streams its own bytes).  Prints GB/s per CU and bytes per clock at 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/fill_rate.hip -o /tmp/fill_rate && /tmp/fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int kRows = 288, kRowBytes = 128, kStageBytes = kRows * kRowBytes;     // 36 KiB
constexpr int kInstrPerStage = kStageBytes / 1024;                               // 36 wave instructions of 1 KiB
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const char* a; const char* b; long long lda_bytes, ldb_bytes; int ksteps, repeats, nw, depth, tiles_n, priv; unsigned* sink;
};

template <int MODE, int CNT = 5, int SETS = 2>
__global__ __launch_bounds__(512) void fill_kernel(Args g) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (w >= g.nw) return;
    // tile of this workgroup: XCD x = id % 8 owns tile columns [4x, 4x+4) and all 8 tile rows (32 tiles per XCD)
    const int id = blockIdx.x, xcd = id & 7, loc = id >> 3;
    int tm, tn;
    if (g.priv) { tm = id; tn = id; } else { tm = loc & 7; tn = xcd * 4 + (loc >> 3); }
    const char* abase = g.a + (long long)tm * 128 * g.lda_bytes;
    const char* bbase = g.b + (long long)tn * 160 * g.ldb_bytes;
    // wave instruction i (0..35) of a stage covers rows 8 i .. 8 i + 7; rows 0..127 come from A, 128..287 from B
    const int ipw = kInstrPerStage / g.nw;          // 36, 18, 9 (nw = 8 -> 4 per wave, waves 0..3 one more)
    const int extra = kInstrPerStage - ipw * g.nw;
    const int first = w * ipw + (w < extra ? w : extra), count = ipw + (w < extra ? 1 : 0);
    unsigned acc = 0;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, 0x7fffffff, 0x00020000);
    const int total = g.ksteps * g.repeats;
    // ring of `depth` stages in flight per wave: before issuing step s wait until step s - depth has landed
    for (int s = 0; s < total; s++) {
        const int kstep = s % g.ksteps, slot = s % (g.depth + 1);
        char* stage = lds + slot * kStageBytes;
        if (MODE == 0) {
            for (int i = 0; i < count; i++) {
                const int ins = first + i, row = ins * 8 + (l >> 3);
                const bool isa = row < 128;
                const long long off = (isa ? (long long)row * g.lda_bytes : (long long)(row - 128) * g.ldb_bytes) + kstep * kRowBytes + (((l & 7) ^ (row & 7)) << 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isa ? ra : rb, LDS_PTR(void, stage + ins * 1024), 16, (unsigned)off, 0, 0, 0);
            }
            // counted wait: allow `depth` steps of this wave's instructions in flight (vmcnt saturates at 63)
            const int allow = g.depth * count;
            if (allow >= 54) asm volatile("s_waitcnt vmcnt(54)" ::: "memory");
            else if (allow >= 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
            else if (allow >= 27) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
            else if (allow >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else if (allow >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (allow >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (allow >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (allow >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else break;
    }
    if (MODE != 0) {
        // register staging (8 waves, <= 5 loads of 16 B per lane and step): two named register sets = two steps in flight
        u32x4 r0[CNT], r1[CNT], r2[SETS == 3 ? CNT : 1];
        auto load = [&](u32x4* r, int s) {
            const int kstep = s % g.ksteps;
#pragma unroll
            for (int i = 0; i < CNT; i++) if (i < count) {
                const int ins = first + i, row = ins * 8 + (l >> 3);
                const bool isa = row < 128;
                const long long off = (isa ? (long long)row * g.lda_bytes : (long long)(row - 128) * g.ldb_bytes) + kstep * kRowBytes + (((l & 7) ^ (row & 7)) << 4);
                r[i] = *(const u32x4*)((isa ? abase : bbase) + off);
            }
        };
        auto drain = [&](u32x4* r, int s) {
#pragma unroll
            for (int i = 0; i < CNT; i++) if (i < count) {
                if (MODE == 1) *(u32x4*)(lds + (s % (g.depth + 1)) * kStageBytes + (first + i) * 1024 + l * 16) = r[i];
                else acc ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
            }
        };
        if (SETS == 2) {
            load(r0, 0); load(r1, 1);
            for (int s = 2; s + 1 < total; s += 2) {
                drain(r0, s - 2); load(r0, s);
                drain(r1, s - 1); load(r1, s + 1);
            }
            drain(r0, total - 2); drain(r1, total - 1);
        } else {
            load(r0, 0); load(r1, 1); load(r2, 2);
            int s = 3;
            for (; s + 2 < total; s += 3) {
                drain(r0, s - 3); load(r0, s);
                drain(r1, s - 2); load(r1, s + 1);
                drain(r2, s - 1); load(r2, s + 2);
            }
            drain(r0, s - 3); drain(r1, s - 2); drain(r2, s - 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1) acc ^= *(unsigned*)(lds + l * 4);
    if (acc == 0x12345678u) g.sink[0] = acc;
}

int main() {
    const int M = 1024, N = 5120, K = 1280;
    const size_t priv_rows = 256 * 160;
    char *a, *b; unsigned* sink;
    CK(hipMalloc(&a, (size_t)priv_rows * K * 2)); CK(hipMalloc(&b, (size_t)priv_rows * K * 2)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, (size_t)priv_rows * K * 2)); CK(hipMemset(b, 2, (size_t)priv_rows * K * 2));
    (void)M; (void)N;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto big = [](const void* f) { CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); };
    big((const void*)fill_kernel<0>); big((const void*)fill_kernel<1, 5, 2>); big((const void*)fill_kernel<2, 5, 2>); big((const void*)fill_kernel<1, 5, 3>);
    big((const void*)fill_kernel<1, 9, 2>); big((const void*)fill_kernel<2, 9, 2>); big((const void*)fill_kernel<1, 9, 3>); big((const void*)fill_kernel<1, 18, 2>);
    printf("mode sharing waves depth/sets  us/launch  GB/s/CU  B/clk@2.4GHz  chip TB/s\n");
    struct Case { int mode, nw, depth, sets; };
    std::vector<Case> cases;
    for (int nw : {1, 2, 4, 8}) for (int depth : {1, 3}) cases.push_back({0, nw, depth, 0});
    cases.push_back({1, 8, 1, 2}); cases.push_back({2, 8, 1, 2}); cases.push_back({1, 8, 3, 3});
    cases.push_back({1, 4, 1, 2}); cases.push_back({2, 4, 1, 2}); cases.push_back({1, 4, 3, 3}); cases.push_back({1, 2, 1, 2});
    for (int priv = 0; priv < 2; priv++)
        for (const Case& c : cases) {
            Args g{a, b, (long long)K * 2, (long long)K * 2, K / 64, 8, c.nw, c.depth, 32, priv, sink};
            const size_t lds_bytes = (size_t)(c.depth + 1) * kStageBytes;      // one workgroup per CU from depth 2 on (>= 108 KiB)
            auto launch = [&] {
                if (c.mode == 0) fill_kernel<0><<<256, 512, lds_bytes>>>(g);
                else if (c.nw == 8 && c.mode == 1 && c.sets == 2) fill_kernel<1, 5, 2><<<256, 512, lds_bytes>>>(g);
                else if (c.nw == 8 && c.mode == 2) fill_kernel<2, 5, 2><<<256, 512, lds_bytes>>>(g);
                else if (c.nw == 8 && c.mode == 1 && c.sets == 3) fill_kernel<1, 5, 3><<<256, 512, lds_bytes>>>(g);
                else if (c.nw == 4 && c.mode == 1 && c.sets == 2) fill_kernel<1, 9, 2><<<256, 512, lds_bytes>>>(g);
                else if (c.nw == 4 && c.mode == 2) fill_kernel<2, 9, 2><<<256, 512, lds_bytes>>>(g);
                else if (c.nw == 4 && c.mode == 1 && c.sets == 3) fill_kernel<1, 9, 3><<<256, 512, lds_bytes>>>(g);
                else if (c.nw == 2) fill_kernel<1, 18, 2><<<256, 512, lds_bytes>>>(g);
            };
            for (int i = 0; i < 3; i++) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int reps = 20;
            for (int i = 0; i < reps; i++) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / reps, bytes = (double)kStageBytes * g.ksteps * g.repeats;
            const double gbs = bytes / us * 1e-3;
            printf("%4d %7s %5d %5d/%d  %9.1f  %7.1f  %12.1f  %9.2f\n", c.mode, priv ? "private" : "gemm", c.nw, c.depth, c.sets, us, gbs, gbs / 2.4, gbs * 256 * 1e-3);
            fflush(stdout);
        }
    return 0;
}
