// gemm_args.cuh — argument block and epilogue kinds shared by the GEMM tile families (gemm.hip, gemm8.cuh).
#pragma once
#include "common.cuh"

namespace {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_DROP_RES = 2, EPI_RES = 3, EPI_GELU_BWD = 4, EPI_WGRAD = 5,
       // fused Q / K / V projection + self-attention of one (example, head) unit per 96 x 192 tile (gemm.hip, round 6): the tile's
       // columns are the head's query | key | value columns of the fused [3H, H] weight, its rows one example's 96 tokens; the
       // epilogue stores qkv (+ bias) as EPI_BIAS does AND keeps the three [96 x 64] blocks in LDS, where the attention forward of
       // the unit runs at once (C2 = ctx, attn_mask, attn_lse, drop = the attention dropout)
       EPI_QKV_ATTN = 6 };

//   C[M,N] = sum_k R(m,k) * Cc(n,k)
//     fwd   : R = x  [M][K]   (k contiguous)          Cc = w  [N][K]   (k contiguous)
//     dgrad : R = dy [M][K]   (k contiguous)          Cc = w  [K][N]   (n contiguous)  -> TRB
//     wgrad : R = dy [K][M]   (m contiguous) -> TRA   Cc = x  [K][N]   (n contiguous)  -> TRB
struct GemmArgs {
    const bf16_t* R;      // M-side operand
    const bf16_t* Cc;     // N-side operand
    int64_t ldr, ldcc;    // leading dimensions (elements)
    bf16_t* C;            // output [M][N]
    bf16_t* C2;           // second output (EPI_BIAS_GELU: g; EPI_WGRAD: db[M] = sums of the M-side operand over the
                          // contraction, i.e. the bias gradient that belongs to this weight gradient; nullptr = none)
    int64_t ldc;
    const bf16_t* bias;   // [N] or nullptr
    const bf16_t* aux;    // residual [M][N] / pre-activation u [M][N] / nullptr
    int64_t ldaux;
    float* partial;       // split-K fp32 partials [splits][M][N] (nullptr when splits == 1)
    int M, N, K;          // K = contraction length
    int k_per_split;      // multiple of 64
    int accumulate;       // EPI_WGRAD, splits == 1: C += result
    int xr;               // 2-D XCD blocking: rows of the XCD grid (0 = 1-D contiguous ranges)
    int relu;             // EPI_BIAS_DROP_RES: 1 = clamp at zero after the bias, before the dropout (Linear + ReLU + Dropout heads);
                          // EPI_BIAS_GELU / EPI_GELU_BWD: the activation (UH_ACT_*: 0 = erf GELU, 1 = ReLU, 2 = swish)
    unsigned* pair;       // gemm8, two K slices combined inside the launch: one counter per output tile (zero between launches)
    float* sq_out;        // gemm8 multi launch, EPI_WGRAD: this tile's sum of squares of the STORED (bf16-rounded) gradient values goes
                          // here (one float per tile; the gradient norm adds them instead of re-reading the weights' gradients)
#ifdef UNITER_GEMM_PROBE
    unsigned long long* probe;   // cycle stamps of wave 0 of every workgroup: [block][kt][5] (profiling builds only)
#endif
    DropoutCfg drop;
    ChainLink chain;      // overlapped kernel chain (common.cuh): wait for the M-side rows' producer, signal the output rows
    const float* attn_mask;   // EPI_QKV_ATTN: additive key mask [B, L] fp32 (model/model.py:342-345)
    float* attn_lse;          // EPI_QKV_ATTN: log-sum-exp of the score rows [B * heads, L] fp32 (saved for the backward)
};

// blockIdx -> (tile row, tile column) for a tiles_m x tiles_n grid: the caller's choice (xr < 0), 2-D XCD blocking
// (xr > 0: hardware block b runs on XCD b % 8; XCD (xi, xj) of an xr x xc grid owns a (tiles_m/xr) x (tiles_n/xc)
// sub-block of tiles, so its private L2 holds only that sub-block's operand rows) or contiguous 1-D ranges per XCD.
__device__ __forceinline__ void tile_of_block(int xr, int bx, int tiles_m, int tiles_n, int& tm, int& tn) {
    if (xr < 0) {
        tm = bx / tiles_n;
        tn = bx % tiles_n;
    } else if (xr > 0) {
        const int xcd = bx & 7, loc = bx >> 3;
        const int xc = 8 / xr;
        const int sub_m = tiles_m / xr, sub_n = tiles_n / xc;
        const int xi = xcd / xc, xj = xcd % xc;
        tm = xi * sub_m + loc / sub_n;
        tn = xj * sub_n + loc % sub_n;
    } else {
        const int tile = xcd_remap(bx, tiles_m * tiles_n);
        tm = tile / tiles_n;
        tn = tile % tiles_n;
    }
}

}  // namespace
