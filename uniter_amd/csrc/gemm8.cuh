// gemm8.cuh — the deep-pipelined bf16 GEMM tile for gfx950: 256 x 256 x 64, eight waves, eight phases per two K tiles.
//
// Why a second tile family.  The 64...192-wide tiles of gemm.hip run one or two barriers per K step with every wave
// doing the same thing at the same time; that structure tops out near 36 % of the MFMA peak however its ring is tuned
// (EXPERIMENTS.md §7/§8).  This one follows the schedule cdna_hip_programming.md §5 "256² 8-phase template" describes:
//
//   * tile 256 x 256, K step 64, 8 waves as 2 (M) x 4 (N): a wave owns four 64 x 32 quadrants (mq, nq) of the output,
//     rows mq*128 + wr*64 + [0,64), columns nq*128 + wc*32 + [0,32) — 128 accumulator registers;
//   * a K tile is four HALF tiles of 16 KiB (A0, A1 = rows 0-127 / 128-255 of the M-side operand, B0, B1 = columns
//     0-127 / 128-255 of the N-side operand), LDS = 2 buffers x 4 half tiles = 128 KiB, filled by LDS-DMA
//     (global_load_lds_dwordx4, two instructions per wave per half tile, XOR swizzle applied on the source address);
//   * four phases per K tile, one quadrant (16 MFMAs) each:   P1: read B0, A0 -> (0,0)   P2: read B1 -> (0,1)
//     P3: read A1 -> (1,1)   P4: no read -> (1,0) (B0 is still in registers).  Every phase also issues the DMA of ONE half
//     tile, seven half tiles ahead of the one being consumed;
//   * a phase is  {ds_reads, DMA issue} s_barrier {s_waitcnt lgkmcnt(0), 16 MFMAs under s_setprio 1} s_barrier, and the
//     two wave groups (waves 0-3 / 4-7: one wave of each on every SIMD) run ONE barrier apart, so that one wave of a SIMD
//     reads LDS and issues DMA while the other owns the matrix pipe;
//   * s_waitcnt vmcnt(6) once per K tile (P4), never 0 inside the loop: three half tiles stay in flight across the
//     barriers.
//
// Hazards, by construction (phase numbers count from the consumer's point of view; S_j = j-th half tile in consumption
// order B0 A0 B1 A1, issued in phase j-7, the first seven in the prologue):
//   RAW  a half tile is waited for (vmcnt) before the FIRST barrier of P4 of the previous K tile by every wave, and first
//        read in P1 of its own K tile: the lagging group's wait and the leading group's read are two barriers apart.
//   WAR  S_j overwrites S_{j-8}.  B0 is last read in P1 and re-filled in P2: its reads are retired (lgkmcnt) before P1's
//        first barrier, so both groups have them back before either group's P2 issue.  A0 (read P1, re-filled P3), B1
//        (P2 -> P4) and A1 (P3 -> next P1) have two phases between their last read and the re-fill.
#pragma once
#include "gemm_args.cuh"
#include "gemm_lds.cuh"

namespace {

constexpr int G8_HALF = 128 * 64;                    // elements of one half tile (16 KiB)
constexpr int G8_BUF = 4 * G8_HALF;                  // one K tile in LDS: A0 A1 B0 B1
constexpr int G8_LDS_BYTES = 2 * G8_BUF * 2;         // 131072
constexpr int G8_THREADS = 512;

template <int V> struct G8C { static constexpr int value = V; };

// Byte offset (from the operand's origin of the current K tile) of the 16 bytes this lane fetches with LDS-DMA
// instruction j (0..15) of half tile h.  K-contiguous operand [extent][ld]: instruction j covers tile rows 8j..8j+7, the
// lane lands in physical chunk lane&7 and fetches logical chunk (lane&7) ^ swz(row) (kc_off's swizzle, gemm_lds.cuh);
// rows beyond the matrix are clamped to the last one (their products land in rows the epilogue never stores).
// K-strided operand [K][ld], `origin` = first tile column: instruction j covers k rows 4j..4j+3 (ks_off8<128>'s swizzle).
template <bool TR>
__device__ __forceinline__ uint32_t g8_src_off(int j, int lane, int h, int64_t ld, int origin, int extent) {
    if constexpr (!TR) {
        const int r = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = origin + h * 128 + r;
        gr = gr < extent ? gr : extent - 1;
        return (uint32_t)(((int64_t)gr * ld + c * 8) * 2);
    } else {
        const int r = 4 * j + (lane >> 4);
        const int c = (lane & 15) ^ (ks_swz<128>(r) << 1);
        return (uint32_t)(((int64_t)r * ld + origin + h * 128 + c * 8) * 2);
    }
}

__device__ __forceinline__ void g8_sc1_store16(void* p, const f32x4 v) {
    // write-through store: the bytes leave this XCD's L2 while the kernel runs (the partner slice may sit on another XCD)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((__attribute__((address_space(1))) f32x4*)p), "v"(v) : "memory");
}

// LDS-DMA of 64 lanes x 16 bytes: global address = uniform 64-bit origin (scalar registers) + per-lane 32-bit offset, LDS
// destination = lds_addr (uniform, through M0) + lane * 16.  Inline asm: the builtin takes a flat 64-bit per-lane pointer,
// which costs a 64-bit VALU add and a register pair per instruction in a kernel that has neither to spare; M0 is declared
// clobbered, so the compiler keeps its own uses of it (movrel, the LDS-DMA builtin) apart.  The loads are invisible to the compiler's vmcnt bookkeeping — every wait on them is written by hand.
__device__ __forceinline__ void g8_glds16(uint32_t voff, const void* origin, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(origin), "s"(lds_addr) : "memory", "m0");
}

// ---- epilogue pieces shared by the tile shapes ------------------------------------------------------------------------------
// Two accumulator fragments of one row block (columns 0-15 / 16-31 of a 32-column group): lane (g, i) holds columns 4g..4g+3
// of row i in each.  v_permlane16_swap exchanges the odd lane rows of the first with the even lane rows of the second, after
// which a lane owns 8 consecutive columns — (g & 1) * 16 + (g >> 1) * 8 ... + 7 — of row i: 16-byte stores and loads.
__device__ __forceinline__ void g8_swap8(const f32x4& lo16, const f32x4& hi16, float (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo16[r]), __float_as_uint(hi16[r]), false, false);
        v[r] = __uint_as_float(sw[0]);
        v[4 + r] = __uint_as_float(sw[1]);
    }
}

// The element-wise tail of a GEMM for one piece of W (4 or 8) consecutive output columns of row m starting at column n:
// the arithmetic (and its roundings) of gemm.hip's epilogue.
template <int EPI>
struct G8Epi {
    static constexpr bool HAS_BIAS = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_DROP_RES);
    static constexpr bool HAS_AUX = (EPI == EPI_BIAS_DROP_RES || EPI == EPI_RES || EPI == EPI_GELU_BWD || EPI == EPI_WGRAD);
    bool to_partial;          // fp32 partials of a K slice (reduced by another kernel)
    const bf16_t* abase;      // residual / pre-activation / the gradient a weight gradient accumulates into
    int64_t ald;
    __device__ __forceinline__ explicit G8Epi(const GemmArgs& p) {
        to_partial = (EPI == EPI_WGRAD || EPI == EPI_RES) && p.partial != nullptr && (EPI != EPI_WGRAD || p.pair == nullptr);
        abase = nullptr;
        ald = 0;
        if constexpr (HAS_AUX) {
            if (EPI == EPI_WGRAD) { abase = (p.accumulate && !to_partial) ? p.C : nullptr; ald = p.ldc; }
            else { abase = to_partial ? nullptr : p.aux; ald = p.ldaux; }
        }
    }
    template <int W>
    __device__ __forceinline__ void bias_words(const GemmArgs& p, const int n, uint32_t (&w)[W / 2]) const {
#pragma unroll
        for (int e = 0; e < W / 2; ++e) w[e] = 0u;
        if constexpr (HAS_BIAS) {
            if (p.bias != nullptr) {
                const uint32_t* bp = reinterpret_cast<const uint32_t*>(p.bias + n);
#pragma unroll
                for (int e = 0; e < W / 2; ++e) w[e] = bp[e];
            }
        }
    }
    template <int W>
    __device__ __forceinline__ void aux_words(const GemmArgs& p, const int m, const int n, uint32_t (&w)[W / 2]) const {
#pragma unroll
        for (int e = 0; e < W / 2; ++e) w[e] = 0u;
        if constexpr (HAS_AUX) {
            if (abase != nullptr && m < p.M) {
                const bf16_t* ap = abase + (int64_t)m * ald + n;
                if constexpr (W == 8) {
                    const u32x4 q = *reinterpret_cast<const u32x4*>(ap);
                    w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
                } else {
                    const u32x2 q = *reinterpret_cast<const u32x2*>(ap);
                    w[0] = q[0]; w[1] = q[1];
                }
            }
        }
    }
    // Output stores are write-through (sc1), not the `nt` of the smaller tiles: a lane row of these tiles writes 32- and
    // 64-byte pieces of a 128-byte line that a neighbouring wave completes, and `nt` partial lines measured 1.8x slower on
    // the two-output GELU epilogue (49.6 -> 27.3 us for 3072 x 3072 x 768; default-policy stores 29.6 us).
    // INVARIANT (the bucketed deferred launch depends on it, g8_bucket_arrive below): EVERY output store of a tile epilogue is
    // write-through (sc1).  A bucket's flag is raised for another stream (the RCCL stream's hipStreamWaitValue32) after a
    // RELAXED agent-scope count of work items whose stores were only drained with s_waitcnt vmcnt(0); that publishes sc1
    // stores and nothing else.  An epilogue that adds a plain (write-back) store must pass plain_stores = true for its work
    // item, which makes the arrival a RELEASE (one buffer_wbl2 of this XCD's L2).
    template <int W>
    __device__ __forceinline__ static void store(bf16_t* dst, const float (&x)[W]) {
        if constexpr (W == 8) {
            const u32x4 v = pack8(x);
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((gmem_u32x4*)dst), "v"(v) : "memory");
        } else {
            const u32x2 v = pack4(x);
            asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"((gmem_u32x2*)dst), "v"(v) : "memory");
        }
    }
    template <int W>
    __device__ __forceinline__ void emit(const GemmArgs& p, const int by, const int m, const int n, float (&v)[W],
                                         const uint32_t (&auxw)[W / 2], const uint32_t (&biasw)[W / 2]) const {
#pragma clang fp contract(off)                              // bias, dropout, residual: three roundings, the same in every tile family
        if (m >= p.M) return;
        if (to_partial) {
            float* dst = p.partial + ((int64_t)by * p.M + m) * p.N + n;
#pragma unroll
            for (int e = 0; e < W; e += 4) *reinterpret_cast<f32x4*>(dst + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
            return;
        }
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int e = 0; e < W / 2; ++e) { v[2 * e] += bits2f_lo(biasw[e]); v[2 * e + 1] += bits2f_hi(biasw[e]); }
        }
        bf16_t* cptr = p.C + (int64_t)m * p.ldc + n;
        if constexpr (EPI == EPI_BIAS_GELU) {
            // the activation is applied to the bf16-rounded u so that backward (which only has u) is consistent
            float uq[W], gq[W];
#pragma unroll
            for (int e = 0; e < W; ++e) uq[e] = bf2f(f2bf(v[e]));
            if (p.relu & UH_ACT_SAVE_GRAD) {               // C <- act'(u) instead of u (common.cuh, act_fwd_grad2)
                float dq[W];
#pragma unroll
                for (int e = 0; e < W; e += 2) {
                    f32x2_t gp, dp;
                    act_fwd_grad2(p.relu & UH_ACT_MASK, f32x2_t{uq[e], uq[e + 1]}, gp, dp);
                    gq[e] = gp.x; gq[e + 1] = gp.y; dq[e] = dp.x; dq[e + 1] = dp.y;
                }
                store<W>(cptr, dq);
                store<W>(p.C2 + (int64_t)m * p.ldc + n, gq);
                return;
            }
            store<W>(cptr, v);
#pragma unroll
            for (int e = 0; e < W; e += 2) {
                const f32x2_t gp = act_fwd2(p.relu, f32x2_t{uq[e], uq[e + 1]});
                gq[e] = gp.x; gq[e + 1] = gp.y;
            }
            store<W>(p.C2 + (int64_t)m * p.ldc + n, gq);
            return;
        }
        if constexpr (EPI == EPI_BIAS_DROP_RES) {
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < W; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (p.drop.p > 0.f) {
                float mv[W];
                if constexpr (W == 8) dropout_mult8(p.drop, ((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 3, mv);
                else                  dropout_mult4(p.drop, ((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 2, mv);
#pragma unroll
                for (int e = 0; e < W; ++e) v[e] *= mv[e];
            }
        }
        if constexpr (HAS_AUX) {
            if (abase != nullptr) {
#pragma unroll
                for (int e = 0; e < W / 2; ++e) {
                    const float lo = bits2f_lo(auxw[e]), hi = bits2f_hi(auxw[e]);
                    if constexpr (EPI == EPI_GELU_BWD) {
                        if (p.relu & UH_ACT_SAVE_GRAD) { v[2 * e] *= lo; v[2 * e + 1] *= hi; }     // aux holds act'(u) already
                        else {
                            const f32x2_t gp = act_grad2(p.relu, f32x2_t{lo, hi});
                            v[2 * e] *= gp.x; v[2 * e + 1] *= gp.y;
                        }
                    }
                    else { v[2 * e] += lo; v[2 * e + 1] += hi; }
                }
            }
        }
        store<W>(cptr, v);
    }
};

// One 256 x 256 output tile.  bx = tile slot, by = K slice.
// COLSUM (weight gradients): the tiles of the first tile column also produce p.C2[m] (+)= sum_k R[k][m] — the bias gradient
// that belongs to the weight gradient — from the M-side fragments they hold in registers anyway (v_dot2c with a pair of ones
// per register, waves of the first wave column only), instead of a second pass over dy by column-strip workgroups: in the
// twelve-layer launch those strips were a quarter of the launch's HBM reads (510 MB of 2 053, PMC).
template <bool TRA, bool TRB, int EPI, bool COLSUM = false>
__device__ __forceinline__ void gemm8_tile(const GemmArgs& p, const int bx, const int by, char* smem_raw) {
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int g = lane >> 4, i = lane & 15;

    const int tiles_n = p.N >> 8;
    const int tiles_m = (p.M + 255) >> 8;
    int tm, tn;
    tile_of_block(p.xr, bx, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;

    const int k_begin = by * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int nk = (k_end - k_begin) >> 6;                  // whole K tiles (the launcher guarantees it)
    if constexpr (!TRA) chain_wait(p.chain, m0, min(256, p.M - m0));      // overlapped chain (common.cuh)

    // ---- LDS-DMA sources and destinations -------------------------------------------------------------------------------
    uint32_t offA[2][2], offB[2][2];                        // [half][instruction of this wave]: byte offsets from the K tile's origin
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            offA[h][jj] = g8_src_off<TRA>(wid + 8 * jj, lane, h, p.ldr, m0, p.M);
            offB[h][jj] = g8_src_off<TRB>(wid + 8 * jj, lane, h, p.ldcc, n0, p.N);
        }
    const char* gA = reinterpret_cast<const char*>(p.R + (TRA ? (int64_t)k_begin * p.ldr : (int64_t)k_begin));
    const char* gB = reinterpret_cast<const char*>(p.Cc + (TRB ? (int64_t)k_begin * p.ldcc : (int64_t)k_begin));
    const int64_t stepA = (TRA ? 64 * p.ldr : (int64_t)64) * 2, stepB = (TRB ? 64 * p.ldcc : (int64_t)64) * 2;   // bytes per K tile
    typedef __attribute__((address_space(3))) char lds_char_t;
    const uint32_t lds0 = (uint32_t)(size_t)(lds_char_t*)smem_raw + (uint32_t)wid * 1024u;   // this wave's first piece of buffer 0, slot 0

    // half tile X of K tile kt into buffer kt & 1: X = 0: B0, 1: A0, 2: B1, 3: A1 (consumption order); LDS slots: A0 A1 B0 B1
    auto stage = [&](auto xc, const int kt) {
        constexpr int X = decltype(xc)::value;
        constexpr bool isA = (X & 1) != 0;
        constexpr int h = X >> 1;
        const uint32_t dst = lds0 + (uint32_t)(kt & 1) * (G8_BUF * 2) + (uint32_t)((isA ? h : 2 + h) * G8_HALF * 2);
        const char* origin = isA ? gA + (int64_t)kt * stepA : gB + (int64_t)kt * stepB;
        g8_glds16(isA ? offA[h][0] : offB[h][0], origin, dst);
        g8_glds16(isA ? offA[h][1] : offB[h][1], origin, dst + 8 * 1024);
    };

    // ---- fragment addresses (elements inside the current buffer; toggled between the buffers by XOR after every K tile) ---
    // K-contiguous half tile [128][64]: row r, 16-byte chunk c at kc_off(r, c); the swizzle depends on (r >> 1) & 7 only, i.e.
    // on the lane, so one address per k half serves all 16-row blocks through the instruction's offset field.
    // K-strided half tile [64][128]: ks_off8<128>(k row, 8-byte chunk); the swizzle mixes into the column block, one address
    // per 16-column block.
    uint32_t aoff[TRA ? 4 : 2], boff[2];
    if constexpr (TRA) {
#pragma unroll
        for (int b = 0; b < 4; ++b) aoff[b] = (uint32_t)ks_off8<128>(8 * g + (i >> 2), ((wr * 64 + b * 16) >> 2) + (i & 3));
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aoff[ks] = (uint32_t)kc_off(wr * 64 + i, ks * 4 + g);
    }
    if constexpr (TRB) {
#pragma unroll
        for (int a = 0; a < 2; ++a) boff[a] = (uint32_t)ks_off8<128>(8 * g + (i >> 2), ((wc * 32 + a * 16) >> 2) + (i & 3)) + 2 * G8_HALF;
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) boff[ks] = (uint32_t)kc_off(wc * 32 + i, ks * 4 + g) + 2 * G8_HALF;
    }
    auto tr_pair = [&](uint32_t e) {                        // rows r0 and r0 + 4 of a K-strided tile -> one MFMA operand
        const s16x4 lo = lds_read_tr(smem + e);
        const s16x4 hi = lds_read_tr(smem + e + 4 * 128);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8, v);
    };

    bf16x8 fA[2][4];                 // [k half of the tile][16-row block of the quadrant]
    bf16x8 fB0[2][2], fB1[2][2];     // [k half][16-column block]
    auto read_a = [&](const int h, const int ks) {          // half tile A_h, k half ks
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (TRA) fA[ks][b] = tr_pair(aoff[b] + h * G8_HALF + ks * 32 * 128);
            else               fA[ks][b] = lds_read_b128(smem + aoff[ks] + h * G8_HALF + b * 16 * 64);
        }
    };
    auto read_b = [&](bf16x8 (&f)[2][2], const int h) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if constexpr (TRB) f[ks][a] = tr_pair(boff[a] + h * G8_HALF + ks * 32 * 128);
                else               f[ks][a] = lds_read_b128(smem + boff[ks] + h * G8_HALF + a * 16 * 64);
            }
    };
    constexpr int A_HALF_READS = TRA ? 8 : 4;   // DS instructions of one read_a(., ks)

    f32x4 acc[2][2][2][4];           // [mq][nq][a][b]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[q >> 1][q & 1][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    float csum[2][4];                // COLSUM: partial sums of this lane's k values, [row half][16-row block]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < 4; ++b) csum[h][b] = 0.f;
    const bool do_colsum = COLSUM && p.C2 != nullptr && tn == 0 && wc == 0;
    auto colsum = [&](const int h) {                        // the fragments of row half h are in fA
        if constexpr (COLSUM) {
            if (do_colsum) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const u32x4 w = __builtin_bit_cast(u32x4, fA[ks][b]);
                        // acc += lo(w) + hi(w): v_dot2c with the bf16 pair (1, 1).  (Inline asm: with the builtin,
                        // __builtin_amdgcn_fdot2_f32_bf16, this LLVM read the fragment's FIRST register four times.)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            asm("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(csum[h][b]) : "v"(w[e]));
                    }
            }
        }
    };

    auto mma = [&](f32x4 (&c)[2][4], const bf16x8 (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    c[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][a], fA[ks][b], c[a][b], 0, 0, 0);
        // MFMAs are pure register operations: without a user on the side-effect chain LLVM sinks half of a cluster below the
        // phase's closing barrier (seen in the optimised IR).  The empty statements emit nothing and pin the cluster here.
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(c[a][b]));
        __builtin_amdgcn_s_setprio(0);
    };
    // first barrier of a phase (the reads and the DMA issue are behind us), then the reads' data
    auto to_mma = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_phase = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                      // (s_barrier alone does not order the compiler's LDS accesses)
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nk > 0) {                                           // (an empty K slice contributes zeros)
    // ---- prologue: seven half tiles in flight, K tile 0 landed -------------------------------------------------------------
    stage(G8C<0>{}, 0);
    stage(G8C<1>{}, 0);
    stage(G8C<2>{}, 0);
    stage(G8C<3>{}, 0);
    if (nk > 1) {
        stage(G8C<0>{}, 1);
        stage(G8C<1>{}, 1);
        stage(G8C<2>{}, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();              // the second wave group runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk; ++kt) {
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        // ---- P1: B0, A0 -> quadrant (0,0); DMA: A1 of K tile kt+1 (the other buffer)
        read_b(fB0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(0, 0);
        if (more1) stage(G8C<3>{}, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(A_HALF_READS) : "memory");      // B0's reads are back: B0 may be re-filled in P2
        __builtin_amdgcn_sched_barrier(0);
        read_a(0, 1);
        to_mma();
        mma(acc[0][0], fB0);
        colsum(0);
        end_phase();
        // ---- P2: B1 -> (0,1); DMA: B0 of K tile kt+2
        read_b(fB1, 1);
        if (more2) stage(G8C<0>{}, kt + 2);
        to_mma();
        mma(acc[0][1], fB1);
        end_phase();
        // ---- P3: A1 -> (1,1); DMA: A0 of K tile kt+2
        read_a(1, 0);
        read_a(1, 1);
        if (more2) stage(G8C<1>{}, kt + 2);
        to_mma();
        mma(acc[1][1], fB1);
        colsum(1);
        end_phase();
        // ---- P4: (1,0) from registers; DMA: B1 of K tile kt+2; K tile kt+1 has landed when three half tiles remain in flight
        if (more2) {
            stage(G8C<2>{}, kt + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        to_mma();
        mma(acc[1][0], fB0);
        end_phase();
        // the other buffer
#pragma unroll
        for (int k = 0; k < (TRA ? 4 : 2); ++k) aoff[k] ^= (uint32_t)G8_BUF;
#pragma unroll
        for (int k = 0; k < 2; ++k) boff[k] ^= (uint32_t)G8_BUF;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();              // the first group waits for the second group's last phase
    __builtin_amdgcn_sched_barrier(0);
    }

    // ---- two K slices combined inside the launch ------------------------------------------------------------------------
    // Whichever slice of a tile finishes first parks its accumulators in the tile's fp32 slab (thread-linear, write-through)
    // and raises the tile's counter; the other adds them to its own and runs the epilogue.  fp32 addition commutes, so the
    // result does not depend on the order of arrival.  Counter: 0 -> 1 (first ticket) -> 2 (slab written) -> 3 (second ticket);
    // the combiner leaves it at 0 for the next launch.
    if constexpr (EPI == EPI_WGRAD) {
        if (p.pair != nullptr) {
            unsigned* cnt = p.pair + (tm * tiles_n + tn);
            float* slab = p.partial + (size_t)(tm * tiles_n + tn) * (256 * 256);
            unsigned* lds_flag = reinterpret_cast<unsigned*>(smem_raw);
            if (t == 0) *lds_flag = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned ticket = *lds_flag;
            if (ticket == 0u) {                              // first to finish: park and leave
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            g8_sc1_store16(slab + ((size_t)((q * 2 + a) * 4 + b) * G8_THREADS + t) * 4, acc[q >> 1][q & 1][a][b]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (t == 0) {
                // ticket 1: the slab is complete at 3 (our own increment included); ticket 2 (the partner had already
                // finished writing when we drew): also 3
                // (bounded: a counter left dirty by an aborted launch must not hang the device; ~1 s)
                for (int spin = 0; spin < (1 << 21) && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 3u; ++spin)
                    __builtin_amdgcn_s_sleep(8);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {                    // a quadrant's eight loads in flight at a time (register budget)
                typedef __attribute__((address_space(1))) f32x4 gmem_f32x4;
                f32x4 o[2][4];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        o[a][b] = __builtin_nontemporal_load((const gmem_f32x4*)(slab + ((size_t)((q * 2 + a) * 4 + b) * G8_THREADS + t) * 4));
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[q >> 1][q & 1][a][b] += o[a][b];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if constexpr (COLSUM) {
        if (do_colsum) {                                     // lanes g = 0..3 of a row hold the four k quarters
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float v = csum[h][b];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    const int m = m0 + h * 128 + wr * 64 + b * 16 + i;
                    if (g == 0 && m < p.M) {
                        bf16_t* dst = p.C2 + m;
                        if (p.accumulate) v += bf2f(*dst);
                        *dst = f2bf(v);
                    }
                }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    // A lane holds C[m][n..n+3] for m = m0 + mq*128 + wr*64 + b*16 + i, n = n0 + nq*128 + wc*32 + a*16 + 4g.  The a = 0 / a = 1
    // pieces of lane rows g, g^1 are swapped (v_permlane16_swap) so that the lane owns 8 consecutive columns of fragment (g & 1).
    const G8Epi<EPI> ep(p);
    const int ncol = n0 + wc * 32 + (g & 1) * 16 + (g >> 1) * 8;           // + nq * 128
    uint32_t biasw[2][4];
    float sq = 0.f;                                          // EPI_WGRAD with sq_out: sum of squares of what this thread stores
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) ep.template bias_words<8>(p, ncol + nq * 128, biasw[nq]);
#pragma unroll
    for (int mq = 0; mq < 2; ++mq) {
        uint32_t auxw[2][4][4];
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)                       // all of this row half's operand loads at once
#pragma unroll
            for (int b = 0; b < 4; ++b) ep.template aux_words<8>(p, m0 + mq * 128 + wr * 64 + b * 16 + i, ncol + nq * 128, auxw[nq][b]);
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float v[8];
                g8_swap8(acc[mq][nq][0][b], acc[mq][nq][1][b], v);
                ep.template emit<8>(p, by, m0 + mq * 128 + wr * 64 + b * 16 + i, ncol + nq * 128, v, auxw[nq][b], biasw[nq]);
                if constexpr (EPI == EPI_WGRAD) {
                    if (p.sq_out != nullptr && m0 + mq * 128 + wr * 64 + b * 16 + i < p.M) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float r = bf2f(f2bf(v[e])); sq += r * r; }     // (v holds the final value: emit added the old gradient)
                    }
                }
            }
    }
    if constexpr (EPI == EPI_WGRAD) {
        // one float per tile, summed in a fixed order (lanes by butterfly, waves 0..7 by thread 0): the gradient norm is as
        // deterministic as the separate reduction kernel it replaces for these tensors (adamw.hip, uniter_adamw_grad_norm_ex)
        if (p.sq_out != nullptr && !ep.to_partial) {
            sq = wave_sum(sq);
            __syncthreads();                                 // (every wave is past the K loop and the slab exchange: LDS is free)
            float* red = reinterpret_cast<float*>(smem_raw);
            if (lane == 0) red[wid] = sq;
            __syncthreads();
            if (t == 0) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < G8_THREADS / 64; ++w) tot += red[w];
                *p.sq_out = tot;
            }
        }
    }
    if constexpr (!TRA) chain_signal(p.chain, m0, min(256, p.M - m0));    // (the epilogue's stores are write-through already)
}

// ---- the 192 x 192 x 64 tile: three phases per K tile, three LDS buffers ----------------------------------------------------
// The same machinery for outputs the 256-wide tile leaves half the chip idle on: a 3072 x 3072 output is exactly 256 tiles
// of 192 x 192 (144 of 256 x 256).  8 waves as 2 (M) x 4 (N), a wave owns 96 rows x 48 columns = 6 x 3 accumulator fragments;
// phase j of a K tile multiplies row third j (rows wr*96 + j*32 + [0,32)) with all 48 columns: 12 MFMAs.  The N-side
// fragments (6 reads) are read in phase 0 and stay in registers, every phase reads its 4 M-side fragments.  A K tile is six
// 8 KiB DMA units — M0 M1 M2 (the row thirds of both wave rows: 64 rows) and N0 N1 N2 (64 columns each) — one instruction
// per wave per unit, two units per phase; with THREE buffers of 48 KiB the units of K tile t+2 are issued during K tile t
// (N0 N1 | N2 M0 | M1 M2) into the buffer K tile t-1 has left, so no unit is re-filled sooner than three phases after its
// last read, and `s_waitcnt vmcnt(6)` in the last phase of K tile t leaves exactly those six in flight: K tile t+1 has landed.
constexpr int G6_UNIT = 64 * 64;                     // elements of one DMA unit (8 KiB)
constexpr int G6_BUF = 6 * G6_UNIT;                  // one K tile: M0 M1 M2 N0 N1 N2
constexpr int G6_LDS_BYTES = 3 * G6_BUF * 2;         // 147456

// byte offset of this lane's 16 bytes of unit `u` (instruction = wave index w) from the operand's origin of the K tile
template <bool TR, bool MSIDE>
__device__ __forceinline__ uint32_t g6_src_off(int w, int lane, int u, int64_t ld, int origin, int extent) {
    if constexpr (!TR) {                                    // K-contiguous unit [64 rows][64 k]
        const int lr = 8 * w + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        int row = MSIDE ? (lr >> 5) * 96 + u * 32 + (lr & 31) : 64 * u + lr;
        row += origin;
        row = row < extent ? row : extent - 1;
        return (uint32_t)(((int64_t)row * ld + c * 8) * 2);
    } else {                                                // K-strided unit [64 k][64 columns]
        const int r = 8 * w + (lane >> 3);
        const int c = (lane & 7) ^ (ks_swz<64>(r) << 1);
        const int cc = c * 8;
        const int col = MSIDE ? (cc >> 5) * 96 + u * 32 + (cc & 31) : 64 * u + cc;
        return (uint32_t)(((int64_t)r * ld + origin + col) * 2);
    }
}

template <bool TRA, bool TRB, int EPI>
__device__ __forceinline__ void gemm6_tile(const GemmArgs& p, const int bx, const int by, char* smem_raw) {
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int g = lane >> 4, i = lane & 15;

    const int tiles_n = p.N / 192;
    const int tiles_m = (p.M + 191) / 192;
    int tm, tn;
    tile_of_block(p.xr, bx, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 192, n0 = tn * 192;

    const int k_begin = by * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int nk = (k_end - k_begin) >> 6;
    if constexpr (!TRA) chain_wait(p.chain, m0, min(192, p.M - m0));      // overlapped chain (common.cuh)

    uint32_t offM[3], offN[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        offM[u] = g6_src_off<TRA, true>(wid, lane, u, p.ldr, m0, p.M);
        offN[u] = g6_src_off<TRB, false>(wid, lane, u, p.ldcc, n0, p.N);
    }
    const char* gA = reinterpret_cast<const char*>(p.R + (TRA ? (int64_t)k_begin * p.ldr : (int64_t)k_begin));
    const char* gB = reinterpret_cast<const char*>(p.Cc + (TRB ? (int64_t)k_begin * p.ldcc : (int64_t)k_begin));
    const int64_t stepA = (TRA ? 64 * p.ldr : (int64_t)64) * 2, stepB = (TRB ? 64 * p.ldcc : (int64_t)64) * 2;
    typedef __attribute__((address_space(3))) char lds_char_t;
    const uint32_t lds0 = (uint32_t)(size_t)(lds_char_t*)smem_raw + (uint32_t)wid * 1024u;

    // unit X of K tile kt into buffer `buf` (= kt % 3): X = 0..2: N0 N1 N2, 3..5: M0 M1 M2; LDS slots: M0 M1 M2 N0 N1 N2
    auto stage = [&](auto xc, const int kt, const int buf) {
        constexpr int X = decltype(xc)::value;
        constexpr bool isM = X >= 3;
        constexpr int u = isM ? X - 3 : X;
        const uint32_t dst = lds0 + (uint32_t)buf * (G6_BUF * 2) + (uint32_t)((isM ? u : 3 + u) * G6_UNIT * 2);
        const char* origin = isM ? gA + (int64_t)kt * stepA : gB + (int64_t)kt * stepB;
        g8_glds16(isM ? offM[u] : offN[u], origin, dst);
    };

    // fragment addresses inside a buffer (elements).  K-contiguous: the M units are [3][64 local rows][64 k] and the N units
    // [192 columns][64 k], both plain kc_off arrays; K-strided: units [64 k][64], ks_off8<64>.
    uint32_t aoff[2], boff[TRB ? 3 : 2];
    if constexpr (TRA) {
#pragma unroll
        for (int b = 0; b < 2; ++b) aoff[b] = (uint32_t)ks_off8<64>(8 * g + (i >> 2), ((wr * 32 + b * 16) >> 2) + (i & 3));
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aoff[ks] = (uint32_t)kc_off(wr * 32 + i, ks * 4 + g);
    }
    if constexpr (TRB) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int col = wc * 48 + a * 16;
            boff[a] = (uint32_t)((col >> 6) * G6_UNIT + ks_off8<64>(8 * g + (i >> 2), ((col & 63) >> 2) + (i & 3))) + 3 * G6_UNIT;
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) boff[ks] = (uint32_t)kc_off(wc * 48 + i, ks * 4 + g) + 3 * G6_UNIT;
    }
    auto tr_pair = [&](uint32_t e) {
        const s16x4 lo = lds_read_tr(smem + e);
        const s16x4 hi = lds_read_tr(smem + e + 4 * 64);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8, v);
    };

    bf16x8 fM[2][2];                 // [k half][16-row block of the row third]
    bf16x8 fN[2][3];                 // [k half][16-column block]
    auto read_m = [&](const uint32_t cur, const int j) {    // row third j of the buffer at element offset cur
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if constexpr (TRA) fM[ks][b] = tr_pair(cur + aoff[b] + j * G6_UNIT + ks * 32 * 64);
                else               fM[ks][b] = lds_read_b128(smem + cur + aoff[ks] + j * G6_UNIT + b * 16 * 64);
            }
    };
    auto read_n = [&](const uint32_t cur) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if constexpr (TRB) fN[ks][a] = tr_pair(cur + boff[a] + ks * 32 * 64);
                else               fN[ks][a] = lds_read_b128(smem + cur + boff[ks] + a * 16 * 64);
            }
    };

    f32x4 acc[3][3][2];              // [row third][a][b]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[j][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](f32x4 (&c)[3][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    c[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fN[ks][a], fM[ks][b], c[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(c[a][b]));   // pins the cluster above the closing barrier (see gemm8_tile)
        __builtin_amdgcn_s_setprio(0);
    };
    auto to_mma = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto end_phase = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nk > 0) {
    // prologue: K tiles 0 and 1 issued, K tile 0 landed
    stage(G8C<0>{}, 0, 0); stage(G8C<1>{}, 0, 0); stage(G8C<2>{}, 0, 0);
    stage(G8C<3>{}, 0, 0); stage(G8C<4>{}, 0, 0); stage(G8C<5>{}, 0, 0);
    if (nk > 1) {
        stage(G8C<0>{}, 1, 1); stage(G8C<1>{}, 1, 1); stage(G8C<2>{}, 1, 1);
        stage(G8C<3>{}, 1, 1); stage(G8C<4>{}, 1, 1); stage(G8C<5>{}, 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();              // the second wave group runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);

    int cur_buf = 0, st_buf = 2;                            // kt % 3, (kt + 2) % 3
    for (int kt = 0; kt < nk; ++kt) {
        const bool more2 = kt + 2 < nk;
        const uint32_t cur = (uint32_t)cur_buf * G6_BUF;
        // ---- phase 0: all N fragments, row third 0; DMA: N0 N1 of K tile kt+2
        read_n(cur);
        read_m(cur, 0);
        if (more2) { stage(G8C<0>{}, kt + 2, st_buf); stage(G8C<1>{}, kt + 2, st_buf); }
        to_mma();
        mma(acc[0]);
        end_phase();
        // ---- phase 1: row third 1; DMA: N2 M0
        read_m(cur, 1);
        if (more2) { stage(G8C<2>{}, kt + 2, st_buf); stage(G8C<3>{}, kt + 2, st_buf); }
        to_mma();
        mma(acc[1]);
        end_phase();
        // ---- phase 2: row third 2; DMA: M1 M2; K tile kt+1 has landed when only K tile kt+2's six units remain in flight
        read_m(cur, 2);
        if (more2) {
            stage(G8C<4>{}, kt + 2, st_buf); stage(G8C<5>{}, kt + 2, st_buf);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        to_mma();
        mma(acc[2]);
        end_phase();
        cur_buf = cur_buf == 2 ? 0 : cur_buf + 1;
        st_buf = st_buf == 2 ? 0 : st_buf + 1;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: per row block an 8-column piece (fragments a = 0, 1 exchanged between lane rows) and a 4-column piece (a = 2)
    const G8Epi<EPI> ep(p);
    const int n8 = n0 + wc * 48 + (g & 1) * 16 + (g >> 1) * 8;
    const int n4 = n0 + wc * 48 + 32 + 4 * g;
    uint32_t bias8[4], bias4[2];
    ep.template bias_words<8>(p, n8, bias8);
    ep.template bias_words<4>(p, n4, bias4);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        uint32_t aux8[2][4], aux4[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wr * 96 + j * 32 + b * 16 + i;
            ep.template aux_words<8>(p, m, n8, aux8[b]);
            ep.template aux_words<4>(p, m, n4, aux4[b]);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wr * 96 + j * 32 + b * 16 + i;
            float v8[8], v4[4];
            g8_swap8(acc[j][0][b], acc[j][1][b], v8);
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = acc[j][2][b][r];
            ep.template emit<8>(p, by, m, n8, v8, aux8[b], bias8);
            ep.template emit<4>(p, by, m, n4, v4, aux4[b], bias4);
        }
    }
    if constexpr (!TRA) chain_signal(p.chain, m0, min(192, p.M - m0));    // (the epilogue's stores are write-through already)
}

template <bool TRA, bool TRB, int EPI>
__global__ __launch_bounds__(G8_THREADS, 2) void gemm6_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    gemm6_tile<TRA, TRB, EPI>(p, (int)blockIdx.x, (int)blockIdx.y, smem_raw);
}

// ---- grouped weight gradients on the eight-phase tile ------------------------------------------------------------------
// Up to four weight gradients dw_q[M_q][N_q] (+)= R_q^T Cc_q over the same contraction (the tokens) in ONE launch, optionally
// as two K slices per tile combined inside the launch, plus — on workgroups appended to the grid — the bias gradients: column
// sums of the M-side operand (dy) in strips of 256 columns.  Tile order: slice-major, then problem after problem, each walked
// with its longer tile dimension outermost; XCD x (hardware blocks b with b % 8 == x) runs the contiguous segment
// [x*per, (x+1)*per) of that order, so an XCD's L2 holds the K slabs of one compact rectangle of tiles (as gemm_group_kernel).
struct G8GroupArgs {
    GemmArgs g[4];
    int tile_start[5];     // cumulative tile counts (one K slice)
    int strip_start[5];    // cumulative 256-column strip counts of the bias gradients
    int n;
    int splits;            // 1 or 2
    int per;               // tiles per XCD segment
    int gemm_blocks;       // 8 * per; blocks beyond it are column-sum strips
};

__device__ __forceinline__ void g8_colsum_strip(const GemmArgs& p, const int strip, char* smem_raw) {
    // db[c] (+)= sum over the contraction of R[k][c], c in [256*strip, +256): thread = (row lane, 8-column chunk)
    const int t = threadIdx.x;
    const int cc = t & 31, rl = t >> 5;
    const bf16_t* src = p.R + (int64_t)strip * 256 + cc * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = (strip * 256 + cc * 8 < p.M) ? rl : p.K;        // (the last strip may be narrower than 256 columns; p.M % 8 == 0)
    for (; r + 48 < p.K; r += 64) {                         // four rows in flight per thread
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(src + (int64_t)(r + 16 * u) * p.ldr);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float f[8];
            unpack8(v[u], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += f[e];
        }
    }
    for (; r < p.K; r += 16) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(src + (int64_t)r * p.ldr), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
    float* red = reinterpret_cast<float*>(smem_raw);        // [16][256]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl * 256 + cc * 8 + e] = s[e];
    __syncthreads();
    if (t < 256 && strip * 256 + t < p.M) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k * 256 + t];
        bf16_t* dst = p.C2 + strip * 256 + t;
        if (p.accumulate) tot += bf2f(*dst);
        *dst = f2bf(tot);
    }
}

__global__ __launch_bounds__(G8_THREADS, 2) void gemm8_group_kernel(const G8GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int b = (int)blockIdx.x;
    if (b >= ga.gemm_blocks) {
        const int s = b - ga.gemm_blocks;
        int q = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (k < ga.n && s >= ga.strip_start[k]) q = k;
        g8_colsum_strip(ga.g[q], s - ga.strip_start[q], smem_raw);
        return;
    }
    const int total = ga.tile_start[ga.n];
    const int pos = (b & 7) * ga.per + (b >> 3);            // position in the linear order
    if (pos >= total * ga.splits) return;                   // the last segment may be short
    const int slice = pos >= total ? 1 : 0;
    const int lin = pos - slice * total;
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < ga.n && lin >= ga.tile_start[k]) q = k;
    const GemmArgs& p = ga.g[q];
    const int bx = lin - ga.tile_start[q];
    const int tiles_m = p.M >> 8, tiles_n = p.N >> 8;
    const int tm = tiles_n >= tiles_m ? bx % tiles_m : bx / tiles_n;      // longer tile dimension outermost
    const int tn = tiles_n >= tiles_m ? bx / tiles_m : bx % tiles_n;
    gemm8_tile<true, true, EPI_WGRAD>(p, tm * tiles_n + tn, slice, smem_raw);
}

// ---- the weight gradients of SEVERAL layers in one launch ----------------------------------------------------------------
// One layer's four weight gradients are 108 tiles of 256 x 256 — less than half the chip for 48 K tiles.  Nothing downstream
// needs a weight gradient before the optimizer (or the bucket's allreduce), so the backward pass can keep every layer's dy
// operands and hand ALL of them over at once: 12 layers = 1 296 tiles, five full rounds of 256 CUs on the tile whose K loop
// runs at 80 % of the matrix pipe, instead of twelve half-filled launches that each fight the data-gradient chain for CUs.
// The problems (up to 4 per layer) come from a table in device memory; tile order and XCD mapping as gemm8_group_kernel;
// the bias gradients come from the tiles of each problem's first tile column (gemm8_tile COLSUM; `bias_strips` = 0 — the
// column-strip form of gemm8_group_kernel is still understood); blocks beyond them are the LayerNorm strips.
// meta = tile_start[n+1] followed by strip_start[n+1].
// A LayerNorm's parameter gradients as one more kind of column strip of the multi launch: dgamma[c] (+)= sum_t dy[t][c] * xhat[t][c],
// dbeta[c] (+)= sum_t dy[t][c] with xhat = (z - mean[t]) * rstd[t] (model/layer.py:108,149: BertLayerNorm backward, parameter half;
// the row half stays in layernorm.hip on the critical path).
struct G8LnJob {
    const bf16_t* dy;
    const bf16_t* z;
    const float* mean;
    const float* rstd;
    bf16_t* dgamma;
    bf16_t* dbeta;
    int rows, H;
    int accumulate, pad;
};
// Strips of 64 columns: thread = (one of 64 row lanes, 8-column chunk), four rows in flight per thread — 12 trips over 3 072
// rows.  (256-column strips with 16 row lanes took 96 dependent trips of ~1 us each: longer than a GEMM tile of the same
// launch, and since the strips are dispatched last they WERE the launch's tail.)
constexpr int G8_LN_STRIP = 64;
__device__ __forceinline__ void g8_ln_cols_strip(const G8LnJob& j, const int strip, char* smem_raw) {
    const int t = threadIdx.x;
    const int cc = t & 7, rl = t >> 3;
    const int col = strip * G8_LN_STRIP + cc * 8;
    float ag[8], ab[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[e] = 0.f; ab[e] = 0.f; }
    if (col < j.H) {
        int r = rl;
        for (; r + 192 < j.rows; r += 256) {
            u32x4 d[4], z[4];
            float m[4], sd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = *reinterpret_cast<const u32x4*>(j.dy + (int64_t)(r + 64 * u) * j.H + col);
                z[u] = *reinterpret_cast<const u32x4*>(j.z + (int64_t)(r + 64 * u) * j.H + col);
                m[u] = j.mean[r + 64 * u];
                sd[u] = j.rstd[r + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float dv[8], zv[8];
                unpack8(d[u], dv); unpack8(z[u], zv);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ag[e] += dv[e] * ((zv[e] - m[u]) * sd[u]); ab[e] += dv[e]; }
            }
        }
        for (; r < j.rows; r += 64) {
            float dv[8], zv[8];
            unpack8(*reinterpret_cast<const u32x4*>(j.dy + (int64_t)r * j.H + col), dv);
            unpack8(*reinterpret_cast<const u32x4*>(j.z + (int64_t)r * j.H + col), zv);
            const float m0 = j.mean[r], s0 = j.rstd[r];
#pragma unroll
            for (int e = 0; e < 8; ++e) { ag[e] += dv[e] * ((zv[e] - m0) * s0); ab[e] += dv[e]; }
        }
    }
    float* red = reinterpret_cast<float*>(smem_raw);        // [2][64 row lanes][64 columns]
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[rl * 64 + cc * 8 + e] = ag[e]; red[64 * 64 + rl * 64 + cc * 8 + e] = ab[e]; }
    __syncthreads();
    if (t < 128) {
        const int k = t >> 6, c = t & 63;                   // threads 0-63: dgamma, 64-127: dbeta
        if (strip * G8_LN_STRIP + c < j.H) {
            float tot = 0.f;
            for (int w = 0; w < 64; ++w) tot += red[k * 64 * 64 + w * 64 + c];
            bf16_t* base = k == 0 ? j.dgamma : j.dbeta;
            if (base != nullptr) {
                bf16_t* dst = base + strip * G8_LN_STRIP + c;
                if (j.accumulate) tot += bf2f(*dst);
                *dst = f2bf(tot);
            }
        }
    }
}

// Gradient buckets of a data-parallel step (round 4): the problems of a multi launch are grouped into up to 24 buckets (the layers
// of one allreduce); every XCD walks bucket 0's share of tiles first, then bucket 1's, ..., so the buckets complete one after the
// other INSIDE the launch, and the last work item of a bucket raises that bucket's flag — a word of signal memory a communication
// stream waits on with hipStreamWaitValue32 before the bucket's allreduce (utils/distributed.py: no per-bucket launches, no host
// hand-back).  nb == 0: the plain single-segment order.
constexpr int G8_MAX_BUCKETS = 24;
struct G8Buckets {
    int nb;
    int tile_start[G8_MAX_BUCKETS + 1];      // first tile (linear order) of every bucket
    int slot_start[G8_MAX_BUCKETS + 1];      // first slot of every bucket inside an XCD's walk (cumulative per-XCD shares)
    unsigned total[G8_MAX_BUCKETS];          // work items (tiles + LayerNorm strips) of every bucket
    int strip_start[G8_MAX_BUCKETS + 1];     // first LayerNorm strip (job order) of every bucket: a bucket's strips run AHEAD of its tiles
    unsigned* count;                         // [nb] arrivals, left at zero by the last arrival
    unsigned* flag[G8_MAX_BUCKETS];          // signal memory, one word per bucket
    unsigned epoch;                          // value written to a completed bucket's flag
    const int* prob_bucket;                  // [n] bucket of every problem
    const int* ln_bucket;                    // [LayerNorm jobs]
};
// plain_stores: this work item also wrote results with ordinary (write-back) stores — a bias gradient, LayerNorm parameter
// gradients — which have to be written back from this XCD's L2 before the item is counted; tile outputs are write-through
// (sc1) and acknowledged by the s_waitcnt, so most items need no cache operation at all (a release fence per tile cost the
// bucketed launch ~100 us: buffer_wbl2 walks the L2).
__device__ __forceinline__ void g8_bucket_arrive(const G8Buckets& bk, const int k, const bool plain_stores) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned seen;
        if (plain_stores) seen = __hip_atomic_fetch_add(bk.count + k, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else seen = __hip_atomic_fetch_add(bk.count + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen + 1u == bk.total[k]) {
            // every other item of the bucket had its results in memory before it was counted
            __hip_atomic_store(bk.count + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bk.flag[k], bk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(G8_THREADS, 2) void gemm8_multi_kernel(const GemmArgs* __restrict__ tbl, const int* __restrict__ meta,
                                                                    const int n, const int per, const int full, const int gemm_blocks,
                                                                    const G8LnJob* __restrict__ ln_jobs, const int ln_strips_per_job,
                                                                    const int bias_strips, unsigned* __restrict__ tail_pairs,
                                                                    float* __restrict__ tail_slabs, unsigned long long* __restrict__ stamps,
                                                                    const int lead_strips, const G8Buckets bk, const int lead_tiles,
                                                                    const int lead_strips2, float* __restrict__ sq) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // Block order (all four counts multiples of 8, so a tile keeps the XCD its index implies): `lead_tiles` tiles, `lead_strips`
    // LayerNorm strips, `lead_strips2` more strips, the remaining tiles, the remaining strips.  The CUs that start with one 28 us
    // strip (or two: the second group lands on the CUs whose first strip ends first) run their 83 us tiles 28 / 56 us behind the
    // CUs that started with a tile, for the whole launch — so CUs come free three times per tile time instead of once, which is
    // when the small kernels of the caller's stream (the embedding backward runs beside this launch) get to run at all: nothing
    // can share a CU with a tile, and a kernel launched beside this one waits for the next turnover.
    int b = (int)blockIdx.x;
    if (bk.nb > 0) {
        // Bucketed launch: hardware order = for every bucket its LayerNorm strips (padded to a multiple of 8 blocks, so that a
        // tile keeps the XCD its index implies), then its tiles — a bucket's flag can only rise when ALL its items are done, and
        // strips left to the end of the launch (the plain order below) would hold every bucket but the first until then.  The
        // strips ahead of a bucket's tiles also stagger the CUs, as the leading strips of the plain order do.
        int base = 0, mapped = -1;
        for (int k = 0; k < bk.nb && mapped < 0; ++k) {
            const int sk = bk.strip_start[k + 1] - bk.strip_start[k], skp = (sk + 7) & ~7;
            const int tk = 8 * (bk.slot_start[k + 1] - bk.slot_start[k]);
            if (b < base + skp) {
                if (b - base >= sk) return;                  // padding block
                mapped = gemm_blocks + bias_strips + bk.strip_start[k] + (b - base);
            } else if (b < base + skp + tk) {
                mapped = 8 * bk.slot_start[k] + (b - base - skp);
            }
            base += skp + tk;
        }
        if (mapped < 0) return;
        b = mapped;
    } else {
        const int work = gemm_blocks + bias_strips;          // tiles (+ bias strips) in the logical order; LayerNorm strips follow
        const int s12 = lead_strips + lead_strips2;
        if (b < lead_tiles) { /* a leading tile: already at its index */ }
        else if (b < lead_tiles + s12) b = work + (b - lead_tiles);                 // a leading LayerNorm strip
        else if (b < s12 + work) b -= s12;                                          // the other tiles / bias strips
        // (else: one of the remaining LayerNorm strips, already at its index)
    }
    // profiling (UNITER_AMD_MULTI_STAMPS, harness only): start / end of every workgroup on the chip-wide 100 MHz clock
    struct Stamp {
        unsigned long long* p;
        __device__ explicit Stamp(unsigned long long* q) : p(q) { if (p && threadIdx.x == 0) p[0] = __builtin_amdgcn_s_memrealtime(); }
        __device__ ~Stamp() { if (p && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p[1] = __builtin_amdgcn_s_memrealtime(); } }
    } stamp(stamps ? stamps + 2 * (size_t)b : nullptr);
    const int* tile_start = meta;
    const int* strip_start = meta + n + 1;
    auto find = [&](const int* starts, int v) {             // largest q with starts[q] <= v (starts ascend, starts[0] == 0)
        int lo = 0, hi = n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (starts[mid] <= v) lo = mid; else hi = mid;
        }
        return lo;
    };
    if (b >= gemm_blocks + bias_strips) {                   // LayerNorm parameter gradients
        const int s = b - gemm_blocks - bias_strips;
        const G8LnJob j = ln_jobs[s / ln_strips_per_job];
        g8_ln_cols_strip(j, s % ln_strips_per_job, smem_raw);
        if (bk.nb > 0) g8_bucket_arrive(bk, bk.ln_bucket[s / ln_strips_per_job], true);
        return;
    }
    if (b >= gemm_blocks) {
        const int s = b - gemm_blocks;
        const int q = find(strip_start, s);
        const GemmArgs p = tbl[q];
        g8_colsum_strip(p, s - strip_start[q], smem_raw);
        return;
    }
    // XCD x (hardware blocks with b % 8 == x) walks the contiguous segment [x*per, (x+1)*per) of the tile order.  Its first
    // `full` tiles fill whole rounds of the XCD's 32 CUs; the few left over (full < per only when they are at most a quarter of a
    // round) would be a round of their own at one tile per handful of CUs — they run as TWO K slices combined in the launch
    // (gemm8_tile's pair path: half the K tiles each, one fp32 slab per tile), twice as many workgroups for half as long.
    const int xcd = b & 7, loc = b >> 3;
    if (bk.nb > 0) {
        // bucketed walk: slot `loc` of this XCD lies in bucket k = the last one with slot_start[k] <= loc
        int k = 0;
        for (int c = 1; c < bk.nb; ++c) k = (bk.slot_start[c] <= loc) ? c : k;
        const int share = bk.slot_start[k + 1] - bk.slot_start[k];
        const int bpos = bk.tile_start[k] + xcd * share + (loc - bk.slot_start[k]);
        if (bpos >= bk.tile_start[k + 1]) return;           // (an XCD's share of a bucket rounds up: a few idle workgroups)
        const int q = find(tile_start, bpos);
        const GemmArgs p = tbl[q];
        const int bx = bpos - tile_start[q];
        const int tiles_m = p.M >> 8, tiles_n = p.N >> 8;
        const int tm = tiles_n >= tiles_m ? bx % tiles_m : bx / tiles_n;
        const int tn = tiles_n >= tiles_m ? bx / tiles_m : bx % tiles_n;
        gemm8_tile<true, true, EPI_WGRAD, true>(p, tm * tiles_n + tn, 0, smem_raw);
        g8_bucket_arrive(bk, k, tn == 0 && p.C2 != nullptr);         // (the first tile column also stored the bias gradient)
        return;
    }
    int pos = xcd * per + loc, slice = 0, slot = -1;
    if (loc >= full) {
        const int t = loc - full;
        pos = xcd * per + full + (t >> 1);
        slice = t & 1;
        slot = xcd * (per - full) + (t >> 1);
    }
    if (pos >= tile_start[n]) return;
    const int q = find(tile_start, pos);
    GemmArgs p = tbl[q];
    const int bx = pos - tile_start[q];
    const int tiles_m = p.M >> 8, tiles_n = p.N >> 8;
    const int tm = tiles_n >= tiles_m ? bx % tiles_m : bx / tiles_n;      // longer tile dimension outermost
    const int tn = tiles_n >= tiles_m ? bx / tiles_m : bx % tiles_n;
    const int tile = tm * tiles_n + tn;
    p.sq_out = sq != nullptr ? sq + pos : nullptr;           // (plain order only: a data-parallel step norms the REDUCED gradients)
    if (slot >= 0) {
        if (tn == 0 && p.C2 != nullptr) {                   // this tile also sums the bias gradient over the WHOLE contraction
            if (slice == 1) return;
        } else {
            p.k_per_split = (p.K >> 7) << 6;                // two slices of whole K tiles (the second takes an odd one)
            if (p.K - p.k_per_split > p.k_per_split) p.k_per_split += 64;
            p.pair = tail_pairs + slot - tile;              // gemm8_tile indexes both by the tile number
            p.partial = tail_slabs + ((int64_t)slot - tile) * (256 * 256);
        }
        gemm8_tile<true, true, EPI_WGRAD, true>(p, tile, slice, smem_raw);
        return;
    }
    gemm8_tile<true, true, EPI_WGRAD, true>(p, tile, 0, smem_raw);
}

__global__ __launch_bounds__(G8_THREADS, 2) void gemm6_group_kernel(const G8GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int b = (int)blockIdx.x;
    if (b >= ga.gemm_blocks) {
        const int s = b - ga.gemm_blocks;
        int q = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (k < ga.n && s >= ga.strip_start[k]) q = k;
        g8_colsum_strip(ga.g[q], s - ga.strip_start[q], smem_raw);
        return;
    }
    const int pos = (b & 7) * ga.per + (b >> 3);
    if (pos >= ga.tile_start[ga.n]) return;
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < ga.n && pos >= ga.tile_start[k]) q = k;
    const GemmArgs& p = ga.g[q];
    const int bx = pos - ga.tile_start[q];
    const int tiles_m = p.M / 192, tiles_n = p.N / 192;
    const int tm = tiles_n >= tiles_m ? bx % tiles_m : bx / tiles_n;
    const int tn = tiles_n >= tiles_m ? bx / tiles_m : bx % tiles_n;
    gemm6_tile<true, true, EPI_WGRAD>(p, tm * tiles_n + tn, 0, smem_raw);
}

template <bool TRA, bool TRB, int EPI>
__global__ __launch_bounds__(G8_THREADS, 2) void gemm8_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    gemm8_tile<TRA, TRB, EPI>(p, (int)blockIdx.x, (int)blockIdx.y, smem_raw);
}

}  // namespace
