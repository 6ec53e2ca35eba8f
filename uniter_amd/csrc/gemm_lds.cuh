// gemm_lds.cuh — LDS layout, fragment reads and LDS-DMA staging of K-contiguous operand tiles, shared by the GEMM
// family (gemm.hip) and the persistent per-XCD forward (xcd_forward.hip).
#pragma once
#include "common.cuh"

namespace {

// ---- LDS layouts -------------------------------------------------------------------------------
// K-contiguous tile: [rows][64] bf16, 8 chunks of 16 B per row, chunk c stored at c ^ ((row>>1)&7).
__device__ __forceinline__ int kc_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}
__device__ __forceinline__ bf16x8 lds_read_b128(const bf16_t* p) {
    return *reinterpret_cast<const bf16x8*>(p);
}

// fragment (8 k-values for row/col `i` of a 16-wide sub-tile) from a K-contiguous tile
__device__ __forceinline__ bf16x8 frag_kc(const bf16_t* tile, int row, int ks, int g) {
    return lds_read_b128(tile + kc_off(row, ks * 4 + g));
}

// ---- K-strided tiles ---------------------------------------------------------------------------
// K-strided tile: [64][W] bf16 (W = 64 or 128).  8-byte chunk ch of row r stored at ch ^ (h(r) << 2).
// A 192-wide K-strided tile is a [64][128] sub-tile followed by a [64][64] sub-tile (columns 128..191), each in its
// own swizzle; only the N-side operand may be 192 wide (dgrad with 192x192 / 128x192 / 96x192 tiles).
template <int W>
__device__ __forceinline__ int ks_swz(int r) {
    if (W == 128) return (r & 3) | (((r >> 3) & 1) << 2);
    return ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
}
template <int W>
__device__ __forceinline__ int ks_off8(int r, int ch8) {   // element offset of 8-byte chunk ch8 of row r
    return r * W + ((ch8 ^ (ks_swz<W>(r) << 2)) << 2);
}

__device__ __forceinline__ s16x4 lds_read_tr(const bf16_t* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

// same from a K-strided tile: lane (g, i = 4j+q) supplies row ks*32+8g+j (+4), cols cb+4q..
template <int W>
__device__ __forceinline__ bf16x8 frag_ks(const bf16_t* tile, int cb, int ks, int g, int i) {
    if constexpr (W == 192) {
        if (cb < 128) return frag_ks<128>(tile, cb, ks, g, i);
        return frag_ks<64>(tile + 64 * 128, cb - 128, ks, g, i);
    }
    const int j = i >> 2, q = i & 3;
    const int r0 = ks * 32 + 8 * g + j;
    const int ch = (cb >> 2) + q;
    const s16x4 lo = lds_read_tr(tile + ks_off8<W>(r0, ch));
    const s16x4 hi = lds_read_tr(tile + ks_off8<W>(r0 + 4, ch));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// ---- direct global -> LDS staging (global_load_lds_dwordx4) ---------------------------------------
// One wave instruction moves 64 lanes x 16 B = 1 KiB to LDS base + lane*16 (the destination is lane-linear by
// hardware), so the XOR swizzle is applied on the SOURCE address: lane l, which lands in physical 16-byte
// chunk c' of row r, fetches the logical chunk c = c' ^ swz(r).  Lanes of one row still read one contiguous
// 128/256-byte row segment, so HBM/L2 coalescing is unchanged.  Rows beyond the matrix are clamped to the last
// valid row (their products land in output rows the epilogue never stores); partial K tiles do not use this
// path (they need zero fill).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

// AUX: cache-policy bits of the load (0 = default; 16 = sc1: the source may have been written by another CU of this XCD
// inside the same launch and must not come out of this CU's L1 — xcd_forward.hip)
template <int AUX = 0>
__device__ __forceinline__ void glds16(const bf16_t* src, bf16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((global_void_t*)src, (lds_void_t*)lds_wave_base, 16, 0, AUX);
}

// K-contiguous tile [ROWS][64]: instruction j covers rows 8j..8j+7.
template <int ROWS, int AUX = 0>
__device__ __forceinline__ void glds_kc(bf16_t* tile, const bf16_t* base, int64_t ld, int row0, int rows_total,
                                        int k0, int wid, int lane) {
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it) {
        const int j = it * 4 + wid;
        const int r = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;
        glds16<AUX>(base + (int64_t)gr * ld + k0 + c * 8, tile + j * 512);
    }
}

// Wait until this wave's DMA share of the current tile has landed while the younger tiles of the ring (at most
// NSTAGE-2 of them, `later` = tiles still to come after this one) stay in flight: vmcnt counts outstanding VMEM
// instructions in issue order and every tile is G instructions per wave.
template <int NSTAGE, int G>
__device__ __forceinline__ void wait_tile(int later) {
    if (NSTAGE >= 4 && later >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
    else if (NSTAGE >= 3 && later >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


}  // namespace
