// head.hip — the small dense pieces of the NLVR2 paired-attention head (model/nlvr2.py:150-204) that PyTorch would run
// as ~40 tiny launches per step: the backward mask of Linear + ReLU + Dropout, and the 2-way classifier with its cross
// entropy (Linear(2H, 2) + F.cross_entropy(reduction='none'), n = pairs per batch).
#include <algorithm>
#include "common.cuh"
#include "kernels.h"
#include "../../include/uniter_hip.h"

namespace {

// dpre = dy * scale where the (ReLU + dropout) output is positive, else 0: out = relu(pre) * keep * scale is positive
// exactly where the unit is active AND kept, so the saved output is the whole mask (no Philox replay needed).
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ out,
                                                               bf16_t* __restrict__ dpre, int64_t n8, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float g[8], o[8], r[8];
        unpack8(*reinterpret_cast<const u32x4*>(dy + i * 8), g);
        unpack8(*reinterpret_cast<const u32x4*>(out + i * 8), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = o[e] > 0.f ? g[e] * scale : 0.f;
        *reinterpret_cast<u32x4*>(dpre + i * 8) = pack8(r);
    }
}

// The masks of the paired head in one pass over attn_masks [2n rows in (pair, side) order][L] (model/nlvr2.py:172-176,183-186):
// rows regrouped as [left block; right block] (row o = side * n + pair), pad[o][t] = (mask == 0) for the attention pool, and
// partner_bias[o][t] = the additive key mask ((1 - m) * -10000, model/model.py:342-345 form) of the sequence row o attends TO,
// i.e. of the other image of its pair.  (PyTorch: view / transpose / == 0 / flip / reshape + the mask kernel = six launches.)
__global__ __launch_bounds__(256) void pair_masks_kernel(const int64_t* __restrict__ m, uint8_t* __restrict__ pad, float* __restrict__ partner_bias,
                                                         int n, int L) {
    const int64_t total = (int64_t)2 * n * L;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int t = (int)(idx % L);
        const int o = (int)(idx / L);
        const int side = o / n, pair = o - side * n;
        const int64_t own = m[((int64_t)2 * pair + side) * L + t];
        const int64_t other = m[((int64_t)2 * pair + (1 - side)) * L + t];
        pad[idx] = own == 0 ? 1 : 0;
        partner_bias[idx] = (1.0f - (float)other) * -10000.0f;
    }
}

constexpr int CLS_MAX = 8;      // classes

// one workgroup per row: logits[c] = x . W[c] + b[c]; loss = logsumexp(logits) - logits[target]; probs saved for backward
__global__ __launch_bounds__(256) void cls_ce_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         const bf16_t* __restrict__ b, const int64_t* __restrict__ target,
                                                         float* __restrict__ loss, float* __restrict__ probs, float* __restrict__ logits_out,
                                                         int D, int C) {
    __shared__ float red[4][CLS_MAX];
    const int row = blockIdx.x;
    const bf16_t* xr = x + (int64_t)row * D;
    float acc[CLS_MAX];
#pragma unroll
    for (int c = 0; c < CLS_MAX; ++c) acc[c] = 0.f;
    for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
        float xv[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + d), xv);
#pragma unroll
        for (int c = 0; c < CLS_MAX; ++c) {
            if (c < C) {
                float wv[8];
                unpack8(*reinterpret_cast<const u32x4*>(w + (int64_t)c * D + d), wv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[c] += xv[e] * wv[e];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CLS_MAX; ++c) acc[c] = wave_sum(acc[c]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < CLS_MAX; ++c) red[threadIdx.x >> 6][c] = acc[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lg[CLS_MAX], mx = -3.0e38f;
        for (int c = 0; c < C; ++c) {
            // the module computes the logits in bf16 (nn.Linear on bf16 tensors) and the loss on their fp32 copy
            const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]) + (b != nullptr ? bf2f(b[c]) : 0.f);
            lg[c] = bf2f(f2bf(v));
            mx = fmaxf(mx, lg[c]);
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(lg[c] - mx);
        const float lse = mx + __logf(se);
        const int64_t t = target[row];
        loss[row] = (t >= 0 && t < C) ? lse - lg[t] : 0.f;
        for (int c = 0; c < C; ++c) {
            probs[(int64_t)row * C + c] = __expf(lg[c] - lse);
            if (logits_out != nullptr) logits_out[(int64_t)row * C + c] = lg[c];
        }
    }
}

// dlogit[i][c] = (p[i][c] - [c == t_i]) * gloss[i]; thread d: dx[i][d] = sum_c dlogit W[c][d]; dW[c][d] += sum_i dlogit x[i][d]
__global__ __launch_bounds__(256) void cls_ce_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         const float* __restrict__ probs, const int64_t* __restrict__ target,
                                                         const float* __restrict__ gloss, bf16_t* __restrict__ dx,
                                                         bf16_t* __restrict__ gw, bf16_t* __restrict__ gb, int n, int D, int C) {
    extern __shared__ float dl[];            // [n][C]
    for (int idx = threadIdx.x; idx < n * C; idx += 256) {
        const int i = idx / C, c = idx % C;
        const int64_t t = target[i];
        float v = 0.f;
        if (t >= 0 && t < C) v = (probs[idx] - (c == (int)t ? 1.f : 0.f)) * gloss[i];
        dl[idx] = bf2f(f2bf(v));             // the gradient of the bf16 logits tensor is bf16
    }
    __syncthreads();
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d < D) {
        float wv[CLS_MAX], gwv[CLS_MAX];
#pragma unroll
        for (int c = 0; c < CLS_MAX; ++c) { wv[c] = c < C ? bf2f(w[(int64_t)c * D + d]) : 0.f; gwv[c] = 0.f; }
        for (int i = 0; i < n; ++i) {
            const float xv = bf2f(x[(int64_t)i * D + d]);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CLS_MAX; ++c) {
                if (c < C) { const float g = dl[i * C + c]; s += g * wv[c]; gwv[c] += g * xv; }
            }
            if (dx != nullptr) dx[(int64_t)i * D + d] = f2bf(s);
        }
        if (gw != nullptr) {
#pragma unroll
            for (int c = 0; c < CLS_MAX; ++c)
                if (c < C) gw[(int64_t)c * D + d] = f2bf(bf2f(gw[(int64_t)c * D + d]) + gwv[c]);
        }
    }
    if (gb != nullptr && blockIdx.x == 0 && threadIdx.x < C) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += dl[i * C + threadIdx.x];
        gb[threadIdx.x] = f2bf(bf2f(gb[threadIdx.x]) + s);
    }
}

}  // namespace

extern "C" {

int uniter_gemm_bias_relu_dropout_fwd(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                                      float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(x && w && y, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::gemm_fwd(uh::GEMM_EPI_BIAS_DROP_RES, x, w, bias, nullptr, y, nullptr, M, N, K, make_dropout(p_drop, seed, offset),
                        (hipStream_t)stream, 0, 0, 1);
}

int uniter_relu_dropout_bwd(const void* dy, const void* out, void* dpre, int64_t numel, float p_drop, void* stream) {
    UH_CHECK_ARG(dy && out && dpre, "null pointer");
    UH_CHECK_ARG(numel > 0 && numel % 8 == 0, "element count must be a positive multiple of 8");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    const float scale = make_dropout(p_drop, 0, 0).scale;
    const int64_t n8 = numel / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)out, (bf16_t*)dpre, n8, scale);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_cls_ce_fwd(const void* x, const void* w, const void* b, const int64_t* target, float* loss, float* probs, float* logits,
                      int64_t n, int64_t D, int64_t C, void* stream) {
    UH_CHECK_ARG(x && w && target && loss && probs, "null pointer");
    UH_CHECK_ARG(n > 0 && D > 0 && D % 8 == 0 && C >= 1 && C <= CLS_MAX, "need D %% 8 == 0 and 1..8 classes");
    hipLaunchKernelGGL(cls_ce_fwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w,
                       (const bf16_t*)b, target, loss, probs, logits, (int)D, (int)C);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_cls_ce_bwd(const void* x, const void* w, const float* probs, const int64_t* target, const float* gloss, void* dx,
                      void* gw, void* gb, int64_t n, int64_t D, int64_t C, void* stream) {
    UH_CHECK_ARG(x && w && probs && target && gloss, "null pointer");
    UH_CHECK_ARG(n > 0 && n <= 4096 && D > 0 && C >= 1 && C <= CLS_MAX, "need 1..4096 rows and 1..8 classes");
    UH_CHECK_ARG(n * C * (int64_t)sizeof(float) <= 64 * 1024, "rows x classes must fit the 64 KiB of LDS the kernel stages the probabilities in");
    hipLaunchKernelGGL(cls_ce_bwd_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), (size_t)(n * C) * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, probs, target, gloss, (bf16_t*)dx, (bf16_t*)gw,
                       (bf16_t*)gb, (int)n, (int)D, (int)C);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_nlvr2_pair_masks(const int64_t* attn_masks, uint8_t* pad, float* partner_bias, int64_t n_pairs, int64_t L, void* stream) {
    UH_CHECK_ARG(attn_masks && pad && partner_bias, "null pointer");
    UH_CHECK_ARG(n_pairs > 0 && L > 0, "bad shape");
    const int64_t total = 2 * n_pairs * L;
    hipLaunchKernelGGL(pair_masks_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                       attn_masks, pad, partner_bias, (int)n_pairs, (int)L);
    UH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
