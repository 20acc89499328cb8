// Round 6 microbenchmark: the producer side of the 128 x 160 GEMM workgroup (one workgroup per CU, 4 DMA waves, 3-stage ring of 36 KiB
// stages, counted vmcnt waits, one barrier per k-step) streaming COLD operands laid over the chip exactly like the library lays a
// 1024 x 5120 x K product (XCD x owns tile columns [4x, 4x + 4) and all 8 tile rows: a weight tile is requested by 8 CUs of an XCD at the
// same time, an activation tile by 4), with and without an L2 PRE-TOUCH: four more waves (standing in for the MFMA waves, which have nothing
// else to issue on the vector-memory path) request ONE dword of every 128-byte line of k-step kt + D, each workgroup only ITS share of the
// lines it has in common with its neighbours (A lines: 1 of 4, B lines: 1 of 8 -> 52 lines per k-step per workgroup).
//
// Why: l1_share.hip (same round) shows a CU taking ~50 B/clk of LDS-DMA bytes when the lines are PRESENT in its XCD's L2, while every GEMM of
// the library sees 14-24 B/clk on operands that are requested for the first time by several CUs at once.  If what bounds a k-step is
// (outstanding requests per CU) x (latency of a first-touch line), then touching lines a few k-steps ahead - shared out, so that a CU pays the
// long latency for 1/4 .. 1/8 of its lines only - turns the rest into true L2 hits.  Round 3 built this into the GEMM kernel
// (l2_pretouch.patch: 28.3 -> 29.3 us) and concluded "request rate, not latency"; this isolates the question from the kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/pretouch_stream.hip -o /tmp/pretouch_stream && /tmp/pretouch_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int kStageBytes = 288 * 128, kNS = 3;             // 128 A rows + 160 B rows of 64 bf16
struct Args {
    const char* a; const char* b; long long ld_bytes; int ksteps, dist, share, touch_waves; unsigned* sink;
};

__global__ __launch_bounds__(512) void stream_kernel(Args g) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int id = blockIdx.x, xcd = id & 7, loc = id >> 3;
    const int tm = loc & 7, tn = xcd * 4 + (loc >> 3);
    const char* abase = g.a + (long long)tm * 128 * g.ld_bytes;
    const char* bbase = g.b + (long long)tn * 160 * g.ld_bytes;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, 0x7fffffff, 0x00020000);
    if (w < 4) {            // ---- DMA waves: 9 wave instructions of 1 KiB per k-step each (rows 8 i .. 8 i + 7 of the 288-row stage)
        auto issue = [&](int s) {
            char* stage = lds + (s % kNS) * kStageBytes;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int ins = w * 9 + i, row = ins * 8 + (l >> 3);
                const bool isa = row < 128;
                const long long off = (isa ? (long long)row : (long long)(row - 128)) * g.ld_bytes + (long long)s * 128 + (((l & 7) ^ (row & 7)) << 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isa ? ra : rb, LDS_PTR(void, stage + ins * 1024), 16, (unsigned)off, 0, 0, 0);
            }
        };
        issue(0); issue(1);
        for (int s = 0; s < g.ksteps; s++) {
            const int younger = min(g.ksteps - 1 - s, kNS - 2);
            if (younger >= 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + kNS - 1 < g.ksteps) issue(s + kNS - 1);
        }
    } else {                // ---- the other four waves: barriers, and (dist > 0) the pre-touch of k-step s + dist
        unsigned* scratch = (unsigned*)(lds + kNS * kStageBytes) + (w - 4) * 64;
        // wave 4: A lines (128 rows, 1 of share_a), wave 5: B lines (160 rows, 1 of share_b; lanes 0..19 when shared), waves 6, 7: the rest when unshared
        const int share_a = g.share ? 4 : 1, share_b = g.share ? 8 : 1;
        const int mine_a = (loc >> 3) & 3, mine_b = tm;                 // position among the 4 (8) workgroups of this XCD that read the same panel
        long long toff = -1; bool is_a = true;
        const int tw = w - 4;
        if (g.dist > 0) {
            if (g.share) {
                if (tw == 0) { const int line = l * share_a + mine_a; if (l < 128 / share_a) toff = (long long)line * g.ld_bytes; }
                else if (tw == 1) { const int line = l * share_b + mine_b; is_a = false; if (l < 160 / share_b) toff = (long long)line * g.ld_bytes; }
            } else {            // every workgroup touches all 288 lines of its stage: 64 + 64 (A), 64 + 64 + 32 (B) over waves 4..7 (two rounds for B)
                if (tw < 2) toff = (long long)(tw * 64 + l) * g.ld_bytes;
                else { is_a = false; toff = (long long)((tw - 2) * 64 + l) * g.ld_bytes; }
            }
        }
        auto touch = [&](int s) {
            if (toff >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? ra : rb, LDS_PTR(void, scratch), 4, (unsigned)(toff + (long long)s * 128), 0, 0, 0);
            if (!g.share && tw == 3 && l < 32 && g.dist > 0)      // B rows 128..159
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, scratch), 4, (unsigned)((long long)(128 + l) * g.ld_bytes + (long long)s * 128), 0, 0, 0);
        };
        if (g.dist > 0)
            for (int s = kNS - 1; s < min(g.dist, g.ksteps); s++) touch(s);
        for (int s = 0; s < g.ksteps; s++) {
            __builtin_amdgcn_s_barrier();
            if (g.dist > 0 && s + g.dist < g.ksteps) touch(s + g.dist);
            if ((s & 7) == 7) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // keep the wave's own queue of touches bounded
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    unsigned acc = *(unsigned*)(lds + l * 4);
    if (acc == 0x12345678u) g.sink[0] = acc;
}

int main() {
    const int K = 1280 * 4, ksteps = K / 64;                     // 80 k-steps per launch: a long steady state
    const size_t ld = (size_t)K * 2, a_bytes = 1024 * ld, b_bytes = 5120 * ld;        // 10.5 MB + 52 MB per set
    const int sets = 6;                                         // 375 MB in rotation: beyond the 256 MB Infinity Cache
    std::vector<char*> A(sets), B(sets);
    unsigned* sink;
    for (int i = 0; i < sets; i++) { CK(hipMalloc(&A[i], a_bytes)); CK(hipMalloc(&B[i], b_bytes)); CK(hipMemset(A[i], 1, a_bytes)); CK(hipMemset(B[i], 2, b_bytes)); }
    CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute((const void*)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kNS * kStageBytes + 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("1024 x 5120 x %d operand stream, 256 workgroups (1 per CU), %d k-steps of 36 KiB per workgroup; cold operands (%d sets in rotation)\n", K, ksteps, sets);
    printf("pre-touch   distance  shared   us/launch   us/k-step   B/clk per CU @2.4GHz\n");
    struct Case { int dist, share; };
    const Case cases[] = {{0, 0}, {2, 1}, {4, 1}, {6, 1}, {8, 1}, {12, 1}, {16, 1}, {4, 0}, {8, 0}, {0, 0}, {8, 1}};
    int rot = 0;
    for (const Case& c : cases) {
        float best = 1e9f;
        for (int it = 0; it < 7; it++) {
            Args g{A[rot % sets], B[rot % sets], (long long)ld, ksteps, c.dist, c.share, 4, sink};
            rot++;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            stream_kernel<<<256, 512, kNS * kStageBytes + 1024>>>(g);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        const double bytes = (double)ksteps * kStageBytes;
        printf("%9s  %9d  %6s  %10.1f  %10.3f  %12.2f\n", c.dist ? "on" : "off", c.dist, c.share ? "yes" : "no", best * 1e3, best * 1e3 / ksteps, bytes / (best * 1e-3) / 2.4e9);
    }
    return 0;
}
