// LDS-DMA operand tiles of the bf16 MFMA kernels (shared by the GEMM family in ff_gemm.hip and the fused
// projection + attention kernels in ff_xattn_fused.hip): global -> LDS staging, swizzles, fragment reads.
#pragma once
#include "ff_common.h"

namespace ff {

// XCD-aware work order: workgroup b runs on XCD b % 8 (observed, used for speed only); the remap hands every XCD a contiguous chunk
// of the logical work list (bijective for any size), so neighbours in the list share an L2.
FF_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------------
// bf16 kernel
// ------------------------------------------------------------------------------------------------
constexpr int kBK = 64;    // bf16 K tile

// ------------------------------------------------------------------------------------------------
// bf16 kernel: operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds, no VGPR staging, no ds_write)
// into an NS-deep ring; tiles t+1 .. t+NS-2 stay in flight across the single s_barrier of a k-step (counted vmcnt).
// The LDS-DMA destination is lane-linear (wave base + lane*16 B), so the bank-conflict swizzles live on the SOURCE
// address: LDS 16-byte slot (row, c') holds global chunk c' ^ swz(row); readers apply the same XOR.
//   K-major tile  [BR rows][64 k]  : swz = row & 7                       (ds_read_b128 fragment reads)
//   M-major tile  [64 k][BR rows]  : swz = f(k) spreading the 4 k-rows of a tr-read block (and the two 16-lane
//                                    groups of a half-wave) over distinct bank ranges   (ds_read_b64_tr_b16 reads)
// Rows / k beyond the matrix are fetched with an out-of-range buffer offset, which the hardware returns as 0.
// ------------------------------------------------------------------------------------------------
constexpr unsigned kOobOffset = 0x80000000u;   // >= num_records of every descriptor we build

template <int BR> FF_DEV int mswz(int k) {      // chunk XOR of k-row `k` in an M-major tile
    if (BR == 128 || BR == 256) return ((k & 3) << 1) | (((k >> 3) & 1) << 3);      // k-rows of 256 / 512 bytes: whole bank rows, the XOR stays inside the low 16 chunks
    if (BR == 32) return ((k >> 3) & 1) << 1;       // 64-byte k-rows (4 chunks): the two 16-lane groups of a half-wave (k, k + 8) land on disjoint banks
    return (((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2);
}

// NW = number of waves that share the issue of a tile (w = 0 .. NW-1 among them); 4 = the whole workgroup.
template <int BR, int LAYOUT, int NW = 4>
FF_DEV void dma_tile(__amdgpu_buffer_rsrc_t rsrc, bf16* stage, const RowMap& map, int row_base, int row_lim, int k0, int k_end, int w, int l) {
    if (LAYOUT == 0) {   // wave instruction = 8 rows x 128 B
        const int cp = l & 7;
#pragma unroll
        for (int p = 0; p < BR / (NW * 8); p++) {
            const int row = p * (NW * 8) + w * 8 + (l >> 3);
            const int k = k0 + ((cp ^ (row & 7)) << 3);
            unsigned off = kOobOffset;
            if (row_base + row < row_lim && k < k_end) off = (unsigned)(map.off(row_base + row) + k) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, FF_LDS_PTR(void, stage + (p * (NW * 8) + w * 8) * kBK), 16, off, 0, 0, 0);
        }
    } else {             // wave instruction = 1 KiB of consecutive k-rows (BR*2 bytes each)
        constexpr int CPR = BR / 8;            // 16-byte chunks per k-row
        constexpr int RPI = 64 / CPR;          // k-rows per wave instruction
        const int cp = l % CPR;
#pragma unroll
        for (int p = 0; p < kBK / (NW * RPI); p++) {
            const int kr = p * NW * RPI + w * RPI + l / CPR;
            const int col = row_base + ((cp ^ mswz<BR>(kr)) << 3);
            unsigned off = (unsigned)(map.off(min(k0 + kr, k_end - 1)) + col) * 2u;
            if (k0 + kr >= k_end || col >= row_lim) off = kOobOffset;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, FF_LDS_PTR(void, stage + (p * NW * RPI + w * RPI) * BR), 16, off, 0, 0, 0);
        }
    }
}


// Fast path of dma_tile for k-steps that lie fully inside [k_begin, k_end) (and plain row maps for M-major operands):
// the per-lane byte offsets are loop invariants computed once; a k-step only advances the scalar soffset.
template <int BR, int LAYOUT, int NW = 4>
FF_DEV void dma_prepare(const RowMap& map, int row_base, int row_lim, int w, int l, unsigned* voff) {
    if (LAYOUT == 0) {
        const int cp = l & 7;
#pragma unroll
        for (int p = 0; p < BR / (NW * 8); p++) {
            const int row = p * (NW * 8) + w * 8 + (l >> 3);
            voff[p] = row_base + row < row_lim ? (unsigned)map.off(row_base + row) * 2u + (unsigned)((cp ^ (row & 7)) << 4) : kOobOffset;
        }
    } else {
        constexpr int CPR = BR / 8, RPI = 64 / CPR;
        const int cp = l % CPR;
#pragma unroll
        for (int p = 0; p < BR / (NW * 8); p++) {
            const int kr = p * NW * RPI + w * RPI + l / CPR;
            const int col = row_base + ((cp ^ mswz<BR>(kr)) << 3);
            voff[p] = col < row_lim ? (unsigned)((long long)kr * map.ld + col) * 2u : kOobOffset;
        }
    }
}
template <int BR, int LAYOUT, int NW = 4>
FF_DEV void dma_tile_fast(__amdgpu_buffer_rsrc_t rsrc, bf16* stage, const unsigned* voff, unsigned soff, int w) {
    constexpr int ROWS_PER_PASS_ELEMS = LAYOUT == 0 ? (NW * 8) * kBK : (NW * (64 / (BR / 8))) * BR;   // LDS elements covered by one pass of the NW waves
    constexpr int WAVE_ELEMS = ROWS_PER_PASS_ELEMS / NW;
#pragma unroll
    for (int p = 0; p < BR / (NW * 8); p++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, FF_LDS_PTR(void, stage + p * ROWS_PER_PASS_ELEMS + w * WAVE_ELEMS), 16, voff[p], soff, 0, 0);
}

template <int BR, int LAYOUT> FF_DEV bf16x8 frag_read2(const bf16* s, int r0, int ks) {
    const int l = threadIdx.x & 63, c = l & 15, g = l >> 4;
    if (LAYOUT == 0) {
        const int row = r0 + c;
        const int chunk = ks * 4 + g;
        return *(const bf16x8*)(s + row * kBK + ((chunk ^ (row & 7)) << 3));
    } else {
        const int k = ks * 32 + g * 8 + (c >> 2);
        const int col = r0 + (c & 3) * 4;
        const bf16* p = s + k * BR + (((col >> 3) ^ mswz<BR>(k)) << 3) + (col & 7);
        return cat4(lds_read_tr16(p), lds_read_tr16(p + 4 * BR));   // mswz(k + 4) == mswz(k)
    }
}

template <int N> FF_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace ff
