// layernorm_fwd.cuh — LayerNorm forward of one row by one wave, shared by layernorm.hip (ln_fwd_kernel) and the
// persistent per-XCD forward (xcd_forward.hip).  Reference: model/layer.py:108-115,149-156 (BertLayerNorm, eps inside the sqrt).
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

}  // namespace
