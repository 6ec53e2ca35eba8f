// layernorm.hip — LayerNorm forward / backward and column-sum reductions (HBM-bound row kernels).
//
// Reference: apex FusedLayerNorm(H, eps=1e-12) as used at model/layer.py:108,149 and
// model/model.py:229,252,253,258 — biased variance, eps inside the sqrt, affine, fp32 statistics.
// One wave64 owns a row; every lane keeps its slice of the row in registers (8-byte bf16x4 loads,
// row fully coalesced), statistics by wave-level reductions, no LDS on the forward path.
#include <cstdlib>
#include "common.cuh"
#include "kernels.h"
#include "layernorm_fwd.cuh"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // forward: 4 waves, one row each
constexpr int BWD_WAVES = 16;       // backward / column sums: 16 waves per block (fewer partial rows to finalize)

template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ gamma,
                                                     const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, int H, float eps, const DropoutCfg drop, const ChainLink chain) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int blk = (int)blockIdx.x;
    const int row = blk * ROWS_PER_BLOCK + wid;
    chain_wait(chain, blk * ROWS_PER_BLOCK, ROWS_PER_BLOCK);      // overlapped chain (common.cuh): z rows of this block
    if (row < rows) ln_fwd_row<NC, false>(z, gamma, beta, y, mean_out, rstd_out, row, H, eps, drop, lane, chain.signal != nullptr);
    chain_signal(chain, blk * ROWS_PER_BLOCK, ROWS_PER_BLOCK);
}

// Backward.  partial layout: [gridDim.x][3][H] fp32 = per-block column sums of (dgamma, dbeta, dbias).
template <int NC, int NW>
struct LnBwd {
    static __device__ void run(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy_extra,
                               const bf16_t* __restrict__ z, const float* __restrict__ mean_in,
                               const float* __restrict__ rstd_in, const bf16_t* __restrict__ gamma,
                               bf16_t* __restrict__ dz, bf16_t* __restrict__ dd, float* __restrict__ partial,
                               int rows, int H, int want_dbias, int post_drop, const DropoutCfg& drop, float* red) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const int nch = H >> 2;
        float gv[NC][4];
        float ag[NC][4], ab[NC][4], ad[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) unpack4(*reinterpret_cast<const u32x2*>(gamma + ch * 4), gv[c]);
            else gv[c][0] = gv[c][1] = gv[c][2] = gv[c][3] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; ad[c][e] = 0.f; }
        }
        const bool use_drop = drop.p > 0.f && !post_drop;   // dropout sat on the dense branch feeding z
        const bool use_post = drop.p > 0.f && post_drop;    // dropout sat on the LN output (embedding blocks)
        for (int row = blockIdx.x * NW + wid; row < rows; row += gridDim.x * NW) {
            const float mean = mean_in[row], rstd = rstd_in[row];
            const int64_t ro = (int64_t)row * H;
            float xh[NC][4], gy[NC][4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nch) {
                    float zv[4], dv[4];
                    unpack4(*reinterpret_cast<const u32x2*>(z + ro + ch * 4), zv);
                    unpack4(*reinterpret_cast<const u32x2*>(dy + ro + ch * 4), dv);
                    if (dy_extra != nullptr) {
                        float ev[4];
                        unpack4(*reinterpret_cast<const u32x2*>(dy_extra + ro + ch * 4), ev);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dv[e] += ev[e];
                    }
                    if (use_post) {
                        float mult[4];
                        dropout_mult4(drop, ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 4) >> 2, mult);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dv[e] *= mult[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xh[c][e] = (zv[e] - mean) * rstd;
                        gy[c][e] = dv[e] * gv[c][e];
                        s1 += gy[c][e];
                        s2 += gy[c][e] * xh[c][e];
                        ag[c][e] += dv[e] * xh[c][e];
                        ab[c][e] += dv[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { xh[c][e] = 0.f; gy[c][e] = 0.f; }
                }
            }
            const float c1 = wave_sum(s1) / (float)H;
            const float c2 = wave_sum(s2) / (float)H;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nch) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rstd * (gy[c][e] - c1 - xh[c][e] * c2);
                    const u32x2 packed = pack4(o);
                    *reinterpret_cast<u32x2*>(dz + ro + ch * 4) = packed;
                    if (use_drop) {
                        float oq[4], mult[4];
                        unpack4(packed, oq);
                        dropout_mult4(drop, ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 4) >> 2, mult);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { oq[e] *= mult[e]; ad[c][e] += oq[e]; }
                        if (dd != nullptr) *reinterpret_cast<u32x2*>(dd + ro + ch * 4) = pack4(oq);
                    } else {
                        if (dd != nullptr) *reinterpret_cast<u32x2*>(dd + ro + ch * 4) = packed;   // no dropout: dd = dz
                        if (want_dbias) {
                            float oq[4];
                            unpack4(packed, oq);
#pragma unroll
                            for (int e = 0; e < 4; ++e) ad[c][e] += oq[e];
                        }
                    }
                }
            }
        }
        // block reduction over the NW waves, one (quantity, 256-column chunk) at a time: red[NW][256]
        float* pout = partial + (int64_t)blockIdx.x * 3 * H;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    red[wid * 256 + lane * 4 + e] = (k == 0) ? ag[c][e] : ((k == 1) ? ab[c][e] : ad[c][e]);
                __syncthreads();
                if (threadIdx.x < 256) {
                    const int col = c * 256 + threadIdx.x;
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) s += red[w * 256 + threadIdx.x];
                    if (col < H) pout[k * H + col] = s;
                }
                __syncthreads();
            }
        }
    }
};

template <int NC, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel2(const bf16_t* dy, const bf16_t* dy_extra, const bf16_t* z,
                                                      const float* mean_in, const float* rstd_in, const bf16_t* gamma,
                                                      bf16_t* dz, bf16_t* dd, float* partial, int rows, int H,
                                                      int want_dbias, int post_drop, const DropoutCfg drop) {
    __shared__ float red[NW * 256];
    LnBwd<NC, NW>::run(dy, dy_extra, z, mean_in, rstd_in, gamma, dz, dd, partial, rows, H, want_dbias, post_drop, drop, red);
}

// ---- split backward (H % 8 == 0): a row kernel on the critical path + a column-sum kernel that can run elsewhere ----
// Row part: dz (and the dropout-masked dd) of every row; one wave per row, no cross-row traffic at all.
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy_extra,
                                                          const bf16_t* __restrict__ z, const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, const bf16_t* __restrict__ gamma,
                                                          bf16_t* __restrict__ dz, bf16_t* __restrict__ dd, int rows, int H,
                                                          int post_drop, const DropoutCfg drop, const ChainLink chain) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nch = H >> 2;
    // overlapped chain (common.cuh): such launches cover all rows in one pass of the grid, so a block owns rows 4b .. 4b+3
    const int blk = (int)blockIdx.x;
    chain_wait(chain, blk * 4, 4);
    const bool wt = chain.signal != nullptr;
    float gv[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) unpack4(*reinterpret_cast<const u32x2*>(gamma + ch * 4), gv[c]);
        else gv[c][0] = gv[c][1] = gv[c][2] = gv[c][3] = 0.f;
    }
    const bool use_drop = drop.p > 0.f && !post_drop;
    const bool use_post = drop.p > 0.f && post_drop;
    for (int row = blk * 4 + wid; row < rows; row += gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        const int64_t ro = (int64_t)row * H;
        float xh[NC][4], gy[NC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float zv[4], dv[4];
                unpack4(*reinterpret_cast<const u32x2*>(z + ro + ch * 4), zv);
                unpack4(*reinterpret_cast<const u32x2*>(dy + ro + ch * 4), dv);
                if (dy_extra != nullptr) {
                    float ev[4];
                    unpack4(*reinterpret_cast<const u32x2*>(dy_extra + ro + ch * 4), ev);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] += ev[e];
                }
                if (use_post) {
                    float mult[4];
                    dropout_mult4(drop, ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 4) >> 2, mult);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] *= mult[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[c][e] = (zv[e] - mean) * rstd;
                    gy[c][e] = dv[e] * gv[c][e];
                    s1 += gy[c][e];
                    s2 += gy[c][e] * xh[c][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { xh[c][e] = 0.f; gy[c][e] = 0.f; }
            }
        }
        const float c1 = wave_sum(s1) / (float)H;
        const float c2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (gy[c][e] - c1 - xh[c][e] * c2);
                const u32x2 packed = pack4(o);
                out_store8c(dz + ro + ch * 4, packed, wt);
                if (dd != nullptr) {            // dd = dropout-masked dz (a plain copy when there is no dropout)
                    u32x2 dpk = packed;
                    if (use_drop) {
                        float oq[4], mult[4];
                        unpack4(packed, oq);
                        dropout_mult4(drop, ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 4) >> 2, mult);
#pragma unroll
                        for (int e = 0; e < 4; ++e) oq[e] *= mult[e];
                        dpk = pack4(oq);
                    }
                    out_store8c(dd + ro + ch * 4, dpk, wt);
                }
            }
        }
    }
    chain_signal(chain, blk * 4, 4);
}

// ---- 16-byte forms (H % 8 == 0, H <= 2048): a lane owns 8-column chunks lane, lane + 64, ... of its wave's row --------------------
// The 8-byte forms above move one row with three 512-byte wave instructions per tensor; 8-byte accesses run at 0.54-0.70 of the
// 16-byte rate (MI355X_MICROARCH.md) and these kernels are latency chains of loads -> two reductions -> stores.  Same arithmetic per
// element, same dropout element groups (4 columns per Philox field group); only the partial sums a lane forms before the wave
// reduction cover 8 columns instead of 4.
template <int NC8>
__global__ __launch_bounds__(256) void ln_fwd_kernel8(const bf16_t* __restrict__ z, const bf16_t* __restrict__ gamma,
                                                      const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      int rows, int H, float eps, const DropoutCfg drop, const ChainLink chain) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int blk = (int)blockIdx.x;
    const int row = blk * ROWS_PER_BLOCK + wid;
    chain_wait(chain, blk * ROWS_PER_BLOCK, ROWS_PER_BLOCK);
    const bool wt = chain.signal != nullptr;
    if (row < rows) {
        const int nch = H >> 3;
        const bf16_t* zr = z + (int64_t)row * H;
        float x[NC8][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC8; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                unpack8(*reinterpret_cast<const u32x4*>(zr + ch * 8), x[c]);
                s += ((x[c][0] + x[c][1]) + (x[c][2] + x[c][3])) + ((x[c][4] + x[c][5]) + (x[c][6] + x[c][7]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[c][e] = 0.f;
            }
        }
        const float mean = wave_sum(s) / (float)H;
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < NC8; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = x[c][e] - mean; v += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        bf16_t* yr = y + (int64_t)row * H;
#pragma unroll
        for (int c = 0; c < NC8; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float gv[8], bv[8], o[8];
                unpack8(*reinterpret_cast<const u32x4*>(gamma + ch * 8), gv);
                unpack8(*reinterpret_cast<const u32x4*>(beta + ch * 8), bv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (x[c][e] - mean) * rstd * gv[e] + bv[e];
                if (drop.p > 0.f) {
                    // the dropped value is the bf16-rounded LN output (what a separate dropout kernel would see)
                    float oq[8], m0[4], m1[4];
                    unpack8(pack8(o), oq);
                    const uint64_t grp = ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 8) >> 2;
                    dropout_mult4(drop, grp, m0);
                    dropout_mult4(drop, grp + 1, m1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = oq[e] * m0[e]; o[4 + e] = oq[4 + e] * m1[e]; }
                }
                out_store16c(yr + ch * 8, pack8(o), wt);
            }
        }
    }
    chain_signal(chain, blk * ROWS_PER_BLOCK, ROWS_PER_BLOCK);
}

template <int NC8>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel8(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy_extra,
                                                           const bf16_t* __restrict__ z, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, const bf16_t* __restrict__ gamma,
                                                           bf16_t* __restrict__ dz, bf16_t* __restrict__ dd, int rows, int H,
                                                           int post_drop, const DropoutCfg drop, const ChainLink chain) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nch = H >> 3;
    const int blk = (int)blockIdx.x;
    chain_wait(chain, blk * 4, 4);
    const bool wt = chain.signal != nullptr;
    float gv[NC8][8];
#pragma unroll
    for (int c = 0; c < NC8; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) unpack8(*reinterpret_cast<const u32x4*>(gamma + ch * 8), gv[c]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[c][e] = 0.f;
        }
    }
    const bool use_drop = drop.p > 0.f && !post_drop;
    const bool use_post = drop.p > 0.f && post_drop;
    for (int row = blk * 4 + wid; row < rows; row += gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        const int64_t ro = (int64_t)row * H;
        float xh[NC8][8], gy[NC8][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC8; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float zv[8], dv[8];
                unpack8(*reinterpret_cast<const u32x4*>(z + ro + ch * 8), zv);
                unpack8(*reinterpret_cast<const u32x4*>(dy + ro + ch * 8), dv);
                if (dy_extra != nullptr) {
                    float ev[8];
                    unpack8(*reinterpret_cast<const u32x4*>(dy_extra + ro + ch * 8), ev);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dv[e] += ev[e];
                }
                if (use_post) {
                    float m0[4], m1[4];
                    const uint64_t grp = ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 8) >> 2;
                    dropout_mult4(drop, grp, m0);
                    dropout_mult4(drop, grp + 1, m1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dv[e] *= m0[e]; dv[4 + e] *= m1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[c][e] = (zv[e] - mean) * rstd;
                    gy[c][e] = dv[e] * gv[c][e];
                    s1 += gy[c][e];
                    s2 += gy[c][e] * xh[c][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[c][e] = 0.f; gy[c][e] = 0.f; }
            }
        }
        const float c1 = wave_sum(s1) / (float)H;
        const float c2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int c = 0; c < NC8; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (gy[c][e] - c1 - xh[c][e] * c2);
                const u32x4 packed = pack8(o);
                out_store16c(dz + ro + ch * 8, packed, wt);
                if (dd != nullptr) {            // dd = dropout-masked dz (a plain copy when there is no dropout)
                    u32x4 dpk = packed;
                    if (use_drop) {
                        float oq[8], m0[4], m1[4];
                        unpack8(packed, oq);
                        const uint64_t grp = ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 8) >> 2;
                        dropout_mult4(drop, grp, m0);
                        dropout_mult4(drop, grp + 1, m1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { oq[e] *= m0[e]; oq[4 + e] *= m1[e]; }
                        dpk = pack8(oq);
                    }
                    out_store16c(dd + ro + ch * 8, dpk, wt);
                }
            }
        }
    }
    chain_signal(chain, blk * 4, 4);
}

// the 16-byte forms run wherever their shape conditions hold (A/B against the 8-byte forms: profiles/r06_layernorm_wide_ab.txt;
// the 8-byte kernels remain for the other widths)
static bool ln_wide() { return true; }

// Column part: per-block partial sums over rows of (dy*xhat, dy, d) with d = `dsrc` (the bf16 dd / dz the row kernel
// wrote; masked on the fly when `mask_dsrc`).  Grid (strips of 512 columns, row blocks); partial [gridDim.y][3][H].
__global__ __launch_bounds__(1024) void ln_bwd_cols_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy_extra,
                                                           const bf16_t* __restrict__ z, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, const bf16_t* __restrict__ dsrc,
                                                           float* __restrict__ partial, int rows, int H, int post_drop,
                                                           int mask_dsrc, const DropoutCfg drop) {
    __shared__ float red[BWD_WAVES][512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    float ag[8], ab[8], ad[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[e] = 0.f; ab[e] = 0.f; ad[e] = 0.f; }
    const bool use_post = drop.p > 0.f && post_drop;
    const bool use_mask = drop.p > 0.f && !post_drop && mask_dsrc;
    if (col < H) {
        for (int row = blockIdx.y * BWD_WAVES + wid; row < rows; row += gridDim.y * BWD_WAVES) {
            const float mean = mean_in[row], rstd = rstd_in[row];
            const int64_t o = (int64_t)row * H + col;
            float zv[8], dv[8];
            unpack8(*reinterpret_cast<const u32x4*>(z + o), zv);
            unpack8(*reinterpret_cast<const u32x4*>(dy + o), dv);
            if (dy_extra != nullptr) {
                float ev[8];
                unpack8(*reinterpret_cast<const u32x4*>(dy_extra + o), ev);
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[e] += ev[e];
            }
            float mult[8];
            if (use_post || use_mask) {
                dropout_mult8(drop, (uint64_t)o >> 3, mult);       // o % 8 == 0 (H % 8 == 0, 8 columns per lane)
            }
            if (use_post) {
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[e] *= mult[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ag[e] += dv[e] * ((zv[e] - mean) * rstd);
                ab[e] += dv[e];
            }
            if (dsrc != nullptr) {
                float sv[8];
                unpack8(*reinterpret_cast<const u32x4*>(dsrc + o), sv);
                if (use_mask) {
                    // same rounding as the row kernel's dd: mask the bf16 dz, round the product to bf16
                    float q0[4], q1[4], r0[4], r1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { q0[e] = sv[e] * mult[e]; q1[e] = sv[4 + e] * mult[4 + e]; }
                    unpack4(pack4(q0), r0);
                    unpack4(pack4(q1), r1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sv[e] = r0[e]; sv[4 + e] = r1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ad[e] += sv[e];
            }
        }
    }
    float* pout = partial + (int64_t)blockIdx.y * 3 * H;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wid][lane * 8 + e] = (k == 0) ? ag[e] : ((k == 1) ? ab[e] : ad[e]);
        __syncthreads();
        if (threadIdx.x < 512) {
            const int gc = blockIdx.x * 512 + threadIdx.x;
            if (gc < H) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < BWD_WAVES; ++w) s += red[w][threadIdx.x];
                pout[k * H + gc] = s;
            }
        }
        __syncthreads();
    }
}

// out_k[col] (+)= sum_b partial[b][k][col]   for k < nk (nk <= 3), partial [nb][nk][H].
// One block = 64 columns x 16 groups of partial rows (coalesced 256-byte reads), LDS tree over the groups.
__global__ __launch_bounds__(1024) void finalize_cols_kernel(const float* __restrict__ partial, int nb, int nk, int H,
                                                             bf16_t* o0, bf16_t* o1, bf16_t* o2, int accumulate) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cx;
    const int total = nk * H;
    float s = 0.f;
    if (idx < total) {
        for (int b = grp; b < nb; b += 16) s += partial[(int64_t)b * total + idx];
    }
    red[grp][cx] = s;
    __syncthreads();
    if (grp == 0 && idx < total) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][cx];
        const int k = idx / H, col = idx % H;
        bf16_t* out = k == 0 ? o0 : (k == 1 ? o1 : o2);
        if (out != nullptr) {
            if (accumulate) t += bf2f(out[col]);
            out[col] = f2bf(t);
        }
    }
}

// per-block column sums of a[rows][N]: grid (strips of 512 cols, row blocks); partial [gridDim.y][N]
__global__ __launch_bounds__(1024) void colsum_kernel(const bf16_t* __restrict__ a, float* __restrict__ partial,
                                                      int rows, int N) {
    __shared__ float red[BWD_WAVES][512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (col < N) {
        for (int row = blockIdx.y * BWD_WAVES + wid; row < rows; row += gridDim.y * BWD_WAVES) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(a + (int64_t)row * N + col), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wid][lane * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 512) {
        const int gc = blockIdx.x * 512 + threadIdx.x;
        if (gc < N) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < BWD_WAVES; ++w) s += red[w][threadIdx.x];
            partial[(int64_t)blockIdx.y * N + gc] = s;
        }
    }
}

int ln_bwd_waves(int nc) { return nc <= 3 ? 16 : (nc == 4 ? 8 : 4); }   // register budget: 128 / 256 / 512 per lane
int ln_bwd_blocks(int64_t rows, int nw) {
    int64_t nb = (rows + nw - 1) / nw;     // one row per wave until the chip is ~3x covered
    const int64_t cap = 4096 / nw;
    if (nb > cap) nb = cap;
    return (int)nb;
}
int ln_cols_blocks(int64_t rows, int64_t H) {
    const int64_t strips = (H + 511) / 512;
    int64_t nb = 256 / strips;                           // ~256 blocks x 16 waves
    const int64_t maxb = (rows + BWD_WAVES - 1) / BWD_WAVES;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    return (int)nb;
}
int colsum_blocks(int64_t rows, int64_t N) {
    const int64_t strips = (N + 511) / 512;
    int64_t nb = 256 / strips;                           // ~256 blocks x 16 waves in flight
    if (nb < 1) nb = 1;
    const int64_t maxb = (rows + BWD_WAVES * 2 - 1) / (BWD_WAVES * 2);   // >= 2 rows per wave
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

namespace uh {

int layernorm_fwd(const void* z, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                  int64_t rows, int64_t H, float eps, const DropoutCfg& drop, hipStream_t st, ChainStep* chain) {
    if (rows <= 0 || H <= 0 || H % 4 != 0 || H > 4096) { uh_set_error("layernorm_fwd: need H %% 4 == 0 and H <= 4096 (H=%lld)", (long long)H); return -1; }
    const int nc = (int)((H / 4 + 63) / 64);
    LaunchTimer lt(TIME_LN_FWD, rows, H, 0, st);
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
    // a chain step needs whole 32-row units (each receives 32 / ROWS_PER_BLOCK contributions)
    ChainLink link{nullptr, nullptr, nullptr, 0, 0};
    if (chain != nullptr) {
        if (rows % 32 == 0) { link = chain->link; chain->produced = 32 / ROWS_PER_BLOCK; }
        else { chain->anyorder = 0; chain->produced = 0; }
    }
#define LN_FWD(NCV)                                                                                         \
    chain_launch(chain, ln_fwd_kernel<NCV>, grid, block, 0, st, (const bf16_t*)z, (const bf16_t*)gamma,     \
                 (const bf16_t*)beta, (bf16_t*)y, mean, rstd, (int)rows, (int)H, eps, drop, link)
    if (ln_wide() && H % 8 == 0 && H <= 2048) {
        const int nc8 = (int)((H / 8 + 63) / 64);
#define LN_FWD8(NCV)                                                                                        \
    chain_launch(chain, ln_fwd_kernel8<NCV>, grid, block, 0, st, (const bf16_t*)z, (const bf16_t*)gamma,    \
                 (const bf16_t*)beta, (bf16_t*)y, mean, rstd, (int)rows, (int)H, eps, drop, link)
        if (nc8 <= 1) LN_FWD8(1);
        else if (nc8 == 2) LN_FWD8(2);
        else if (nc8 == 3) LN_FWD8(3);
        else LN_FWD8(4);
#undef LN_FWD8
        UH_LAUNCH_CHECK();
        return 0;
    }
    if (nc <= 1) LN_FWD(1);
    else if (nc == 2) LN_FWD(2);
    else if (nc == 3) LN_FWD(3);
    else if (nc == 4) LN_FWD(4);
    else if (nc <= 8) LN_FWD(8);
    else LN_FWD(16);
#undef LN_FWD
    UH_LAUNCH_CHECK();
    return 0;
}

size_t layernorm_bwd_workspace_bytes(int64_t rows, int64_t H) {
    const int nc = (int)((H / 4 + 63) / 64);
    const size_t fused = (size_t)ln_bwd_blocks(rows, ln_bwd_waves(nc)) * 3 * (size_t)H * sizeof(float);
    const size_t split = (size_t)ln_cols_blocks(rows, H) * 3 * (size_t)H * sizeof(float);
    return fused > split ? fused : split;
}

static int ln_bwd_check(int64_t rows, int64_t H) {
    if (rows <= 0 || H <= 0 || H % 4 != 0 || H > 2048) { uh_set_error("layernorm_bwd: need H %% 4 == 0 and H <= 2048 (H=%lld)", (long long)H); return -1; }
    return 0;
}

// Row half of the split backward (H % 8 == 0): dz, and dd = dropout-masked dz when the dropout sat on the dense branch.
int layernorm_bwd_rows(const void* dy, const void* dy_extra, const void* z, const float* mean, const float* rstd,
                       const void* gamma, void* dz, void* dd, int64_t rows, int64_t H, const DropoutCfg& drop,
                       int post_drop, hipStream_t st, ChainStep* chain) {
    if (ln_bwd_check(rows, H)) return -1;
    if (H % 8 != 0) { uh_set_error("layernorm_bwd_rows: need H %% 8 == 0"); return -1; }
    LaunchTimer lt(TIME_LN_BWD, rows, H, 0, st);
    const int nc = (int)((H / 4 + 63) / 64);
    int64_t nb = (rows + 3) / 4;
    // a chain step needs one pass of the grid over whole 32-row units (each receives 8 contributions)
    ChainLink link{nullptr, nullptr, nullptr, 0, 0};
    if (chain != nullptr) {
        if (rows % 32 == 0 && nb <= 65536) { link = chain->link; chain->produced = 8; }
        else { chain->anyorder = 0; chain->produced = 0; }
    }
    if (link.signal == nullptr && link.wait == nullptr && nb > 4096) nb = 4096;
#define LN_ROWS(NCV)                                                                                                   \
    chain_launch(chain, ln_bwd_rows_kernel<NCV>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)dy,             \
                 (const bf16_t*)dy_extra, (const bf16_t*)z, mean, rstd, (const bf16_t*)gamma, (bf16_t*)dz,             \
                 (bf16_t*)dd, (int)rows, (int)H, post_drop, drop, link)
    if (ln_wide()) {                                           // (H % 8 == 0 and H <= 2048 were checked above)
        const int nc8 = (int)((H / 8 + 63) / 64);
#define LN_ROWS8(NCV)                                                                                                  \
    chain_launch(chain, ln_bwd_rows_kernel8<NCV>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)dy,            \
                 (const bf16_t*)dy_extra, (const bf16_t*)z, mean, rstd, (const bf16_t*)gamma, (bf16_t*)dz,             \
                 (bf16_t*)dd, (int)rows, (int)H, post_drop, drop, link)
        if (nc8 <= 1) LN_ROWS8(1);
        else if (nc8 == 2) LN_ROWS8(2);
        else if (nc8 == 3) LN_ROWS8(3);
        else LN_ROWS8(4);
#undef LN_ROWS8
        UH_LAUNCH_CHECK();
        return 0;
    }
    if (nc <= 1) LN_ROWS(1);
    else if (nc == 2) LN_ROWS(2);
    else if (nc == 3) LN_ROWS(3);
    else if (nc == 4) LN_ROWS(4);
    else LN_ROWS(8);
#undef LN_ROWS
    UH_LAUNCH_CHECK();
    return 0;
}

// Column half: dgamma / dbeta / dbias (+)= column sums.  `dz` and `dd` are the row half's outputs (dd may be null).
int layernorm_bwd_cols(const void* dy, const void* dy_extra, const void* z, const float* mean, const float* rstd,
                       const void* dz, const void* dd, void* dgamma, void* dbeta, void* dbias,
                       int64_t rows, int64_t H, int accumulate, const DropoutCfg& drop, int post_drop,
                       void* workspace, size_t ws_bytes, hipStream_t st) {
    if (ln_bwd_check(rows, H)) return -1;
    if (H % 8 != 0) { uh_set_error("layernorm_bwd_cols: need H %% 8 == 0"); return -1; }
    const int nb = ln_cols_blocks(rows, H);
    if (ws_bytes < (size_t)nb * 3 * (size_t)H * sizeof(float)) { uh_set_error("layernorm_bwd: workspace too small"); return -1; }
    LaunchTimer lt(TIME_LN_BWD_COLS, rows, H, 0, st);
    const bool masked = drop.p > 0.f && !post_drop;
    const void* dsrc = nullptr;
    int mask_dsrc = 0;
    if (dbias != nullptr) {
        if (dd != nullptr) dsrc = dd;                        // the row half's (masked) copy: never the reusable dz buffer
        else { dsrc = dz; mask_dsrc = masked ? 1 : 0; }
    }
    dim3 grid((unsigned)((H + 511) / 512), nb);
    hipLaunchKernelGGL(ln_bwd_cols_kernel, grid, dim3(64 * BWD_WAVES), 0, st, (const bf16_t*)dy, (const bf16_t*)dy_extra,
                       (const bf16_t*)z, mean, rstd, (const bf16_t*)dsrc, (float*)workspace, (int)rows, (int)H, post_drop,
                       mask_dsrc, drop);
    UH_LAUNCH_CHECK();
    hipLaunchKernelGGL(finalize_cols_kernel, dim3((unsigned)((3 * H + 63) / 64)), dim3(1024), 0, st,
                       (const float*)workspace, nb, 3, (int)H, (bf16_t*)dgamma, (bf16_t*)dbeta, (bf16_t*)dbias, accumulate);
    UH_LAUNCH_CHECK();
    return 0;
}

int layernorm_bwd(const void* dy, const void* dy_extra, const void* z, const float* mean, const float* rstd,
                  const void* gamma, void* dz, void* dd, void* dgamma, void* dbeta, void* dbias,
                  int64_t rows, int64_t H, int accumulate, const DropoutCfg& drop, int post_drop,
                  void* workspace, size_t ws_bytes, hipStream_t st) {
    if (ln_bwd_check(rows, H)) return -1;
    if (ws_bytes < layernorm_bwd_workspace_bytes(rows, H)) { uh_set_error("layernorm_bwd: workspace too small"); return -1; }
    // Large row counts (the encoder's LayerNorms go through layernorm_bwd_rows / _cols directly): row kernel + column kernel.
    // The embedding LayerNorms (a few hundred to ~2000 rows, all on one stream) take the one-pass kernel below: two launches
    // instead of three, and dy / z are read once.
    if (H % 8 == 0 && rows > 2048) {
        int rc = layernorm_bwd_rows(dy, dy_extra, z, mean, rstd, gamma, dz, dd, rows, H, drop, post_drop, st);
        if (rc) return rc;
        return layernorm_bwd_cols(dy, dy_extra, z, mean, rstd, dz, dd, dgamma, dbeta, dbias, rows, H, accumulate, drop,
                                  post_drop, workspace, ws_bytes, st);
    }
    LaunchTimer lt(TIME_LN_BWD, rows, H, 0, st);
    const int nc = (int)((H / 4 + 63) / 64);
    const int nw = ln_bwd_waves(nc);
    const int nb = ln_bwd_blocks(rows, nw);
    float* partial = (float*)workspace;
    const int want_dbias = dbias != nullptr;
#define LN_BWD(NCV, NWV)                                                                                              \
    hipLaunchKernelGGL((ln_bwd_kernel2<NCV, NWV>), dim3(nb), dim3(64 * NWV), 0, st, (const bf16_t*)dy, (const bf16_t*)dy_extra, \
                       (const bf16_t*)z, mean, rstd, (const bf16_t*)gamma, (bf16_t*)dz, (bf16_t*)dd, partial,     \
                       (int)rows, (int)H, want_dbias, post_drop, drop)
    if (nc <= 1) LN_BWD(1, 16);
    else if (nc == 2) LN_BWD(2, 16);
    else if (nc == 3) LN_BWD(3, 16);
    else if (nc == 4) LN_BWD(4, 8);
    else LN_BWD(8, 4);
#undef LN_BWD
    UH_LAUNCH_CHECK();
    hipLaunchKernelGGL(finalize_cols_kernel, dim3((unsigned)((3 * H + 63) / 64)), dim3(1024), 0, st,
                       (const float*)partial, nb, 3, (int)H, (bf16_t*)dgamma, (bf16_t*)dbeta, (bf16_t*)dbias, accumulate);
    UH_LAUNCH_CHECK();
    return 0;
}

size_t colsum_workspace_bytes(int64_t rows, int64_t N) {
    return (size_t)colsum_blocks(rows, N) * (size_t)N * sizeof(float);
}

int colsum(const void* a, void* out, int64_t rows, int64_t N, int accumulate,
           void* workspace, size_t ws_bytes, hipStream_t st) {
    if (rows <= 0 || N <= 0 || N % 8 != 0) { uh_set_error("colsum: need N %% 8 == 0"); return -1; }
    if (ws_bytes < colsum_workspace_bytes(rows, N)) { uh_set_error("colsum: workspace too small"); return -1; }
    LaunchTimer lt(TIME_COLSUM, rows, N, 0, st);
    const int nb = colsum_blocks(rows, N);
    dim3 grid((unsigned)((N + 511) / 512), nb);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(64 * BWD_WAVES), 0, st, (const bf16_t*)a, (float*)workspace, (int)rows, (int)N);
    UH_LAUNCH_CHECK();
    hipLaunchKernelGGL(finalize_cols_kernel, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, st,
                       (const float*)workspace, nb, 1, (int)N, (bf16_t*)out, (bf16_t*)nullptr, (bf16_t*)nullptr, accumulate);
    UH_LAUNCH_CHECK();
    return 0;
}

}  // namespace uh
