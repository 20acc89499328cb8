// Round 6 microbenchmark: (1) where do the workgroups of a 2-per-CU launch land, and (2) does a CU fill its LDS faster when its two
// co-resident workgroups request the SAME operand tile at about the same time (the second request served by the CU's vector L1 instead of
// taking a slot on the L2 request path)?
//
// Background ("the operand-fill wall"): a CU accepts LDS-DMA operand bytes at ~14 B/clk with one workgroup and ~21 with
// two, whatever the ring depth or the number of issuing waves; every GEMM of the library is bound by that rate times its tile's FLOP per byte.
// If L1 hits do not count against that rate, two co-resident 128 x 128 workgroups working on N-adjacent tiles (same A row panel) would move
// 3/4 of the bytes through the miss path - the one lever left that changes bytes per FLOP without a larger tile.
//
// The kernel is the producer side of the 128 x 128, two-stage GEMM workgroup: 4 waves, per k-step a 16 KiB A tile and a 16 KiB B tile by
// buffer_load_dwordx4 ... lds, counted vmcnt waits, one barrier per k-step, nothing consuming.  64 KiB of LDS per workgroup: two per CU.
// Per XCD the sources are the panels of an 8 x 8 tile sub-grid (8 A panels, 8 B panels of K = 1024: 4 MiB, L2-resident after first touch).
//   assign 0: (i, j) from the XCD-local launch index           - co-resident workgroups share a panel only by accident
//   assign 1: (i, j) from the PHYSICAL CU (HW_ID) + arrival slot - the two workgroups of a CU read the same A panel, different B panels
//   assign 2: as 1, but both read the same A AND B (upper bound: every second request is an L1 hit)
//   assign 3: as 0 with ONE workgroup per CU's worth of work missing (one workgroup per CU, 128 KiB LDS): the 14 B/clk reference
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/l1_share.hip -o /tmp/l1_share && /tmp/l1_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int kTileBytes = 128 * 128;          // 128 rows x 64 bf16
constexpr int kK = 1024, kSteps = kK / 64;     // 16 k-steps per pass over a panel
constexpr size_t kPanelBytes = (size_t)128 * kK * 2;

struct Args {
    const char* a; const char* b; unsigned* cu_slots; unsigned* where; unsigned* sink; int assign, repeats;
};

__device__ inline unsigned hw_id() { return __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4); }      // HW_REG_HW_ID
__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20); }      // HW_REG_XCC_ID

__global__ __launch_bounds__(256) void fill2(Args g) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    __shared__ int s_ij[2];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int id = blockIdx.x, xcd = id & 7, loc = id >> 3;
    if (threadIdx.x == 0) {
        const unsigned hw = hw_id(), xc = xcc_id() & 15;
        const unsigned cu_key = (xc << 8) | ((hw >> 8) & 0xff);            // XCC | SE / SH / CU fields of HW_ID
        int i, j;
        if (g.assign == 0 || g.assign == 3) { i = loc & 7; j = (loc >> 3) & 7; }
        else {
            const unsigned slot = atomicAdd(&g.cu_slots[cu_key], 1u) & 1u;
            const unsigned cu_in_x = (hw >> 8) & 0xff;                       // not dense: fold it
            const unsigned f = (cu_in_x ^ (cu_in_x >> 3)) & 31;
            i = f & 7;
            j = g.assign == 2 ? (f >> 3) * 2 : (f >> 3) * 2 + slot;
        }
        s_ij[0] = i; s_ij[1] = j;
        g.where[id * 4 + 0] = hw; g.where[id * 4 + 1] = xc; g.where[id * 4 + 2] = (unsigned)wall_clock64(); g.where[id * 4 + 3] = (unsigned)(i | (j << 8));
    }
    __syncthreads();
    const char* abase = g.a + ((size_t)xcd * 8 + s_ij[0]) * kPanelBytes;
    const char* bbase = g.b + ((size_t)xcd * 8 + s_ij[1]) * kPanelBytes;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, 0x7fffffff, 0x00020000);
    // wave w carries rows [32 w, 32 w + 32) of both tiles: 4 + 4 wave instructions of 1 KiB per k-step
    const int total = kSteps * g.repeats;
    auto issue = [&](int s) {
        const int kstep = s % kSteps;
        char* stage = lds + (s & 1) * (2 * kTileBytes);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = w * 32 + q * 8 + (l >> 3);
            const unsigned off = (unsigned)row * (kK * 2) + kstep * 128 + (((l & 7) ^ (row & 7)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, stage + (w * 4 + q) * 1024), 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, stage + kTileBytes + (w * 4 + q) * 1024), 16, off, 0, 0, 0);
        }
    };
    issue(0);
    for (int s = 0; s < total; s++) {
        if (s + 1 < total) { issue(s + 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    unsigned acc = *(unsigned*)(lds + l * 4);
    if (acc == 0x12345678u) g.sink[0] = acc;
}

int main() {
    char *a, *b; unsigned *slots, *where, *sink;
    const size_t bytes = (size_t)64 * kPanelBytes;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&slots, 65536 * 4)); CK(hipMalloc(&where, 1024 * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    CK(hipFuncSetAttribute((const void*)fill2, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int repeats = 8;
    printf("assign  workgroups  LDS/WG   us/launch   GB/s per CU   B/clk @2.4GHz   (bytes DELIVERED to LDS; %d passes over K = %d)\n", repeats, kK);
    for (int assign : {0, 1, 2, 3, 0, 1, 2, 3}) {
        const int grid = assign == 3 ? 256 : 512;
        const size_t lds = assign == 3 ? 128 * 1024 : 64 * 1024;
        Args g{a, b, slots, where, sink, assign, repeats};
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CK(hipMemset(slots, 0, 65536 * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            fill2<<<grid, 256, lds>>>(g);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        const double per_wg = (double)kSteps * repeats * 2 * kTileBytes, per_cu = per_wg * grid / 256.0;
        printf("%6d  %10d  %4zu K  %9.1f  %12.1f  %13.2f\n", assign, grid, lds / 1024, best * 1e3, per_cu / (best * 1e-3) / 1e9, per_cu / (best * 1e-3) / 2.4e9);
        if (assign <= 1) {          // placement of the last launch: which launch indices shared a CU, and did the pairs get the same A panel
            std::vector<unsigned> h(grid * 4);
            CK(hipMemcpy(h.data(), where, grid * 16, hipMemcpyDeviceToHost));
            std::map<unsigned, std::vector<int>> by_cu;
            for (int i = 0; i < grid; i++) by_cu[(h[i * 4 + 1] << 8) | ((h[i * 4] >> 8) & 0xff)].push_back(i);
            int pairs = 0, same_a = 0, d8 = 0, d256 = 0, other = 0; std::map<int, int> sizes;
            for (auto& kv : by_cu) {
                sizes[(int)kv.second.size()]++;
                if (kv.second.size() == 2) {
                    pairs++;
                    const int x = kv.second[0], y = kv.second[1];
                    if ((h[x * 4 + 3] & 0xff) == (h[y * 4 + 3] & 0xff)) same_a++;
                    const int d = abs(x - y);
                    if (d == 8) d8++; else if (d == 256) d256++; else other++;
                }
            }
            printf("        placement: %zu distinct CUs;", by_cu.size());
            for (auto& kv : sizes) printf(" %d CUs hold %d workgroups;", kv.second, kv.first);
            printf(" of %d pairs: %d share their A panel; launch-index distance 8: %d, 256: %d, other: %d\n", pairs, same_a, d8, d256, other);
            if (assign == 0) {
                printf("        first 16 CUs (xcc | hw_id bits 15:8 -> launch indices):");
                int n = 0;
                for (auto& kv : by_cu) { if (n++ >= 16) break; printf(" %03x:", kv.first); for (int v : kv.second) printf("%d,", v); }
                printf("\n");
            }
        }
    }
    return 0;
}
