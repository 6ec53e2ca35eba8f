// layernorm_fwd.cuh — LayerNorm row arithmetic (one wave per row) shared by layernorm.hip and the GEMM row tails (row_tail.cuh).  Reference: model/layer.py:108-115,149-156 (BertLayerNorm, eps inside the sqrt).
#pragma once
#include "common.cuh"

namespace {

// COH: z was written by another CU of this XCD inside the same launch (see ldg8 in common.cuh); y is then stored with
// the default cache policy so that its consumer finds it in the XCD's L2.
template <int NC, bool COH>
__device__ __forceinline__ void ln_fwd_row(const bf16_t* __restrict__ z, const bf16_t* __restrict__ gamma,
                                           const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                           const int row, const int H, const float eps, const DropoutCfg& drop, const int lane,
                                           const bool wt = false) {
#pragma clang fp contract(off)          // the same bits from every kernel this is inlined into
    const int nch = H >> 2;
    const bf16_t* zr = z + (int64_t)row * H;
    float x[NC][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            unpack4(ldg8<COH>(zr + ch * 4), x[c]);
            s += (x[c][0] + x[c][1]) + (x[c][2] + x[c][3]);
        } else {
            x[c][0] = x[c][1] = x[c][2] = x[c][3] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = x[c][e] - mean; v += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    bf16_t* yr = y + (int64_t)row * H;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float gv[4], bv[4], o[4];
            unpack4(ldg8<false>(gamma + ch * 4), gv);
            unpack4(ldg8<false>(beta + ch * 4), bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (x[c][e] - mean) * rstd * gv[e] + bv[e];
            if (drop.p > 0.f) {
                // the dropped value is the bf16-rounded LN output (what a separate dropout kernel would see)
                float mult[4], oq[4];
                unpack4(pack4(o), oq);
                dropout_mult4(drop, ((uint64_t)row * (uint64_t)H + (uint64_t)ch * 4) >> 2, mult);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = oq[e] * mult[e];
            }
            if constexpr (COH) stg8<true>(yr + ch * 4, pack4(o));
            else out_store8c(yr + ch * 4, pack4(o), wt);       // (wt: write-through for a consumer inside an overlapped chain)
        }
    }
}

// ---- batched row forms (row_tail.cuh: LayerNorm as the tail of the GEMM that produces its input) -------------------------------
// RB rows per wave with the loads of all of them in flight before the first use — a wave that walks its rows one by one
// pays one memory round trip per row.  The arithmetic per row is that of ln_fwd_row / ln_bwd_rows_kernel, expression by
// expression, so the fused and the separate launches give the same bits.
//   rows handled: row0 + j * step for j < cnt (cnt <= RB, wave-uniform)
// COH: z was written inside this launch (loads that this CU's L1 cannot serve)
template <int NC, int RB, bool COH>
__device__ __forceinline__ void ln_fwd_rows_batch(const bf16_t* __restrict__ z, const bf16_t* __restrict__ gamma,
                                                  const bf16_t* __restrict__ beta, bf16_t* __restrict__ y,
                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                  const int row0, const int step, const int cnt, const int H, const float eps,
                                                  const int lane) {
#pragma clang fp contract(off)
    const int nch = H >> 2;
    u32x2 raw[RB][NC], graw[NC], braw[NC];
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            raw[j][c] = u32x2{0u, 0u};
            if (j < cnt && ch < nch) raw[j][c] = ldg8<COH>(z + (int64_t)(row0 + j * step) * H + ch * 4);
        }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = lane + 64 * c;
        graw[c] = braw[c] = u32x2{0u, 0u};
        if (ch < nch) { graw[c] = ldg8<false>(gamma + ch * 4); braw[c] = ldg8<false>(beta + ch * 4); }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        if (j >= cnt) break;
        const int row = row0 + j * step;
        float x[NC][4];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                unpack4(raw[j][c], x[c]);
                s += (x[c][0] + x[c][1]) + (x[c][2] + x[c][3]);
            } else {
                x[c][0] = x[c][1] = x[c][2] = x[c][3] = 0.f;
            }
        }
        const float mean = wave_sum(s) / (float)H;
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = x[c][e] - mean; v += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        bf16_t* yr = y + (int64_t)row * H;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float gv[4], bv[4], o[4];
                unpack4(graw[c], gv);
                unpack4(braw[c], bv);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (x[c][e] - mean) * rstd * gv[e] + bv[e];
                out_store8(yr + ch * 4, pack4(o));
            }
        }
    }
}

// Row half of the LayerNorm backward (layernorm.hip: ln_bwd_rows_kernel is this with RB = 1): dz = rstd * (g*dy - mean(g*dy)
// - xhat * mean(g*dy*xhat)), dd = dz with the dense branch's dropout mask (a copy without dropout; nullptr = not wanted).
// gv: gamma of this lane's chunks as floats (zero beyond H).
template <int NC, int RB, bool COH = false>
__device__ __forceinline__ void ln_bwd_rows_batch(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy_extra,
                                                  const bf16_t* __restrict__ z, const float* __restrict__ mean_in,
                                                  const float* __restrict__ rstd_in, const float (&gv)[NC][4],
                                                  bf16_t* __restrict__ dz, bf16_t* __restrict__ dd, const int row0, const int step,
                                                  const int cnt, const int H, const bool use_drop, const bool use_post,
                                                  const DropoutCfg& drop, const int lane, const bool wt) {
    const int nch = H >> 2;
    u32x2 zraw[RB][NC], draw[RB][NC], eraw[RB][NC];
    float mean[RB], rstd[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int row = row0 + j * step;
        const int64_t ro = (int64_t)row * H;
        mean[j] = 0.f; rstd[j] = 0.f;
        if (j < cnt) { mean[j] = mean_in[row]; rstd[j] = rstd_in[row]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            zraw[j][c] = draw[j][c] = eraw[j][c] = u32x2{0u, 0u};
            if (j < cnt && ch < nch) {
                zraw[j][c] = *reinterpret_cast<const u32x2*>(z + ro + ch * 4);
                draw[j][c] = ldg8<COH>(dy + ro + ch * 4);
                if (dy_extra != nullptr) eraw[j][c] = *reinterpret_cast<const u32x2*>(dy_extra + ro + ch * 4);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        if (j >= cnt) break;
        const int row = row0 + j * step;
        const int64_t ro = (int64_t)row * H;
        float xh[NC][4], gy[NC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float zv[4], dv[4];
                unpack4(zraw[j][c], zv);
                unpack4(draw[j][c], dv);
                if (dy_extra != nullptr) {
                    float ev[4];
                    unpack4(eraw[j][c], ev);
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
                    xh[c][e] = (zv[e] - mean[j]) * rstd[j];
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
                for (int e = 0; e < 4; ++e) o[e] = rstd[j] * (gy[c][e] - c1 - xh[c][e] * c2);
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
}

}  // namespace
