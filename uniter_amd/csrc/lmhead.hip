// lmhead.hip — the HBM-bound pieces of the pre-training output heads:
//   * cross entropy over a materialised bf16 logits matrix [n, V] (V = 28996 for the tied MLM decoder,
//     model/layer.py:205-222 + model/pretrain.py:129-133; 1601 for region classification, model/pretrain.py:36-47),
//     forward (loss, log-sum-exp) and backward (softmax - one-hot, scaled by the incoming gradient, written over the
//     logits) — PyTorch runs the same thing as a cast to fp32 plus log_softmax / nll_loss kernels over a 4-byte copy;
//   * KL divergence against soft labels for MRC-KL (model/pretrain.py:206-229): same two sweeps with a target row;
//   * the element-wise GELU backward of the head transform dense -> GELU -> LayerNorm (model/layer.py:188-203).
// The GEMMs of the heads are the library's ordinary GEMM entry points (strided variants over the padded logits
// buffer); nothing here is MFMA work.  One workgroup per row; bf16 in, fp32 arithmetic.
#include "common.cuh"
#include "kernels.h"
#include "../../include/uniter_hip.h"
#include <algorithm>

namespace {

constexpr int CT = 256;

__device__ __forceinline__ float blk_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float blk_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// log-sum-exp of one row of bf16 logits (online max / sum per thread, then a block combine)
__device__ __forceinline__ float row_lse(const bf16_t* x, int V, float* red) {
    float m = -INFINITY, s = 0.f;
    const int nvec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? (V >> 3) : 0;
    for (int c = threadIdx.x; c < nvec; c += CT) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + c * 8), v);
        float mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
        if (mx > m) { s *= __expf(m - mx); m = mx; }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf(v[e] - m);
    }
    for (int c = nvec * 8 + threadIdx.x; c < V; c += CT) {
        const float v = bf2f(x[c]);
        if (v > m) { s *= __expf(m - v); m = v; }
        s += __expf(v - m);
    }
    const float M = blk_max(m, red);
    const float part = (m == -INFINITY) ? 0.f : s * __expf(m - M);
    const float S = blk_sum(part, red);
    return M + __logf(S);
}

// loss[row] = lse - logit[label] (0 for ignored rows: label < 0, F.cross_entropy(ignore_index=-1, reduction='none'))
__global__ __launch_bounds__(CT) void ce_fwd_kernel(const bf16_t* __restrict__ logits, int64_t ld,
                                                    const int64_t* __restrict__ labels, float* __restrict__ loss,
                                                    float* __restrict__ lse_out, int V) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const bf16_t* x = logits + (int64_t)row * ld;
    const float lse = row_lse(x, V, red);
    if (threadIdx.x == 0) {
        const int64_t lab = labels[row];
        lse_out[row] = lse;
        loss[row] = (lab >= 0 && lab < V) ? lse - bf2f(x[lab]) : 0.f;
    }
}

// in place: logits[row][v] <- (softmax(row)[v] - [v == label]) * gout[row]   (all zeros for ignored rows)
__global__ __launch_bounds__(CT) void ce_bwd_kernel(bf16_t* __restrict__ logits, int64_t ld,
                                                    const int64_t* __restrict__ labels, const float* __restrict__ lse_in,
                                                    const float* __restrict__ gout, int V) {
    const int row = blockIdx.x;
    bf16_t* x = logits + (int64_t)row * ld;
    const int64_t lab = labels[row];
    const bool live = lab >= 0 && lab < V;
    const float g = live ? gout[row] : 0.f;
    const float lse = lse_in[row];
    const int nvec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? (V >> 3) : 0;
    for (int c = threadIdx.x; c < nvec; c += CT) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + c * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__expf(v[e] - lse) - ((int64_t)(c * 8 + e) == lab ? 1.f : 0.f)) * g;
        *reinterpret_cast<u32x4*>(x + c * 8) = pack8(v);
    }
    for (int c = nvec * 8 + threadIdx.x; c < V; c += CT)
        x[c] = f2bf((__expf(bf2f(x[c]) - lse) - ((int64_t)c == lab ? 1.f : 0.f)) * g);
}

// F.kl_div(log_softmax(logits), target, reduction='none') (model/pretrain.py:217-221), element-wise [n, V]:
// loss[row][v] = t_v * (log t_v - (x_v - lse))   (0 where t_v == 0)
__global__ __launch_bounds__(CT) void kl_fwd_kernel(const bf16_t* __restrict__ logits, int64_t ld,
                                                    const float* __restrict__ target, float* __restrict__ loss,
                                                    float* __restrict__ lse_out, int V) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const bf16_t* x = logits + (int64_t)row * ld;
    const float* t = target + (int64_t)row * V;
    float* lo = loss + (int64_t)row * V;
    const float lse = row_lse(x, V, red);
    for (int c = threadIdx.x; c < V; c += CT) {
        const float tv = t[c];
        lo[c] = tv > 0.f ? tv * (__logf(tv) - (bf2f(x[c]) - lse)) : 0.f;
    }
    if (threadIdx.x == 0) lse_out[row] = lse;
}
// in place: logits[row][v] <- softmax[v] * sum_u(gout_u t_u) - gout_v t_v        (gout element-wise [n, V])
__global__ __launch_bounds__(CT) void kl_bwd_kernel(bf16_t* __restrict__ logits, int64_t ld, const float* __restrict__ target,
                                                    const float* __restrict__ lse_in, const float* __restrict__ gout, int V) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    bf16_t* x = logits + (int64_t)row * ld;
    const float* t = target + (int64_t)row * V;
    const float* g = gout + (int64_t)row * V;
    float a = 0.f;
    for (int c = threadIdx.x; c < V; c += CT) a += g[c] * t[c];
    a = blk_sum(a, red);
    const float lse = lse_in[row];
    for (int c = threadIdx.x; c < V; c += CT)
        x[c] = f2bf(__expf(bf2f(x[c]) - lse) * a - g[c] * t[c]);
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u,
                                                       bf16_t* __restrict__ dx, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const u32x4*>(dy + i * 8), a);
        unpack8(*reinterpret_cast<const u32x4*>(u + i * 8), b);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_t gp = gelu_erf_grad2(f32x2_t{b[e], b[e + 1]});
            a[e] *= gp.x; a[e + 1] *= gp.y;
        }
        *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(a);
    }
}

// ---- the classes beyond the last full 64-wide GEMM tile (V % 64 of them: 4 for the 28996-word vocabulary, 1 for the
// 1601 region classes): too few for a tile, so three sliver kernels keep them exact -----------------------------------
// logits[m][Vm + j] = t[m] . w[Vm + j] + b[Vm + j]      grid n, 4 waves; wave w takes classes w, w+4, ...
__global__ __launch_bounds__(256) void tail_logits_kernel(const bf16_t* __restrict__ t, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ b, bf16_t* __restrict__ logits,
                                                          int64_t ld, int H, int Vm, int V) {
    const int m = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bf16_t* tr = t + (int64_t)m * H;
    for (int j = Vm + wid; j < V; j += 4) {
        const bf16_t* wr = w + (int64_t)j * H;
        float s = 0.f;
        for (int c = lane * 8; c < H; c += 512) {
            float a[8], q[8];
            unpack8(*reinterpret_cast<const u32x4*>(tr + c), a);
            unpack8(*reinterpret_cast<const u32x4*>(wr + c), q);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += a[e] * q[e];
        }
        s = wave_sum(s);
        if (lane == 0) logits[(int64_t)m * ld + j] = f2bf(s + (b ? bf2f(b[j]) : 0.f));
    }
}
// dt[m][c] += sum_j dl[m][Vm + j] * w[Vm + j][c]          grid n, thread per column (strided)
__global__ __launch_bounds__(256) void tail_dgrad_kernel(const bf16_t* __restrict__ dl, int64_t ld, const bf16_t* __restrict__ w,
                                                         bf16_t* __restrict__ dt, int H, int Vm, int V) {
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < H; c += 256) {
        float s = bf2f(dt[(int64_t)m * H + c]);
        for (int j = Vm; j < V; ++j) s += bf2f(dl[(int64_t)m * ld + j]) * bf2f(w[(int64_t)j * H + c]);
        dt[(int64_t)m * H + c] = f2bf(s);
    }
}
// gw[Vm + j][c] += sum_m dl[m][Vm + j] * t[m][c] ; gb[Vm + j] += sum_m dl[m][Vm + j]        grid V - Vm
__global__ __launch_bounds__(256) void tail_wgrad_kernel(const bf16_t* __restrict__ dl, int64_t ld, const bf16_t* __restrict__ t,
                                                         bf16_t* __restrict__ gw, bf16_t* __restrict__ gb, int n, int H, int Vm) {
    __shared__ float red[4];
    const int j = Vm + blockIdx.x;
    for (int c = threadIdx.x; c < H; c += 256) {
        float s = 0.f;
        for (int m = 0; m < n; ++m) s += bf2f(dl[(int64_t)m * ld + j]) * bf2f(t[(int64_t)m * H + c]);
        if (gw) gw[(int64_t)j * H + c] = f2bf(bf2f(gw[(int64_t)j * H + c]) + s);
    }
    if (gb != nullptr) {
        float s = 0.f;
        for (int m = threadIdx.x; m < n; m += 256) s += bf2f(dl[(int64_t)m * ld + j]);
        s = blk_sum(s, red);
        if (threadIdx.x == 0) gb[j] = f2bf(bf2f(gb[j]) + s);
    }
}

// layout of the tensors the forward pass keeps for the backward pass (one caller-owned buffer)
struct HeadSave {
    size_t u, g, t, mean, rstd, lse, logits, total;
    int64_t ld;
};
HeadSave head_save(int64_t n, int64_t H, int64_t V) {
    HeadSave s;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    s.ld = (V + 63) / 64 * 64;
    size_t o = 0;
    s.u = o; o += al((size_t)n * H * 2);
    s.g = o; o += al((size_t)n * H * 2);
    s.t = o; o += al((size_t)n * H * 2);
    s.mean = o; o += al((size_t)n * 4);
    s.rstd = o; o += al((size_t)n * 4);
    s.lse = o; o += al((size_t)n * 4);
    s.logits = o; o += al((size_t)n * s.ld * 2);
    s.total = o;
    return s;
}
size_t head_gemm_ws(int64_t n, int64_t H, int64_t Vm) {
    size_t w = uh::gemm_wgrad_workspace_bytes(n, H, H);
    if (Vm > 0) {
        w = std::max(w, uh::gemm_dgrad_splitk_workspace_bytes(n, Vm, H));
        w = std::max(w, (size_t)0);      // the projection's weight gradient never splits (it carries the bias gradient)
    }
    w = std::max(w, uh::layernorm_bwd_workspace_bytes(n, H));
    return (w + 255) / 256 * 256;
}
int head_check(const UniterHeadParams* p, int64_t n, int64_t H, int64_t V) {
    if (p == nullptr || !p->dense_w || !p->dense_b || !p->ln_g || !p->ln_b || !p->proj_w) { uh_set_error("head: null parameter pointer"); return -1; }
    if (n <= 0 || V <= 0 || H <= 0 || H % 64 != 0 || H > 2048) { uh_set_error("head: need n > 0, V > 0, H %% 64 == 0, H <= 2048"); return -1; }
    return 0;
}

int check_rows(int64_t n, int64_t V, int64_t ld) {
    if (n <= 0 || V <= 0 || ld < V) { uh_set_error("cross entropy: need n > 0, V > 0, ld >= V"); return -1; }
    if (n > INT32_MAX || V > INT32_MAX) { uh_set_error("cross entropy: dimension too large"); return -1; }
    return 0;
}

}  // namespace

extern "C" {

int uniter_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* loss, float* lse,
                  int64_t n, int64_t V, void* stream) {
    UH_CHECK_ARG(logits && labels && loss && lse, "null pointer");
    if (check_rows(n, V, ld)) return -1;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)n), dim3(CT), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, labels,
                       loss, lse, (int)V);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_ce_bwd(void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* gout,
                  int64_t n, int64_t V, void* stream) {
    UH_CHECK_ARG(logits && labels && lse && gout, "null pointer");
    if (check_rows(n, V, ld)) return -1;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)n), dim3(CT), 0, (hipStream_t)stream, (bf16_t*)logits, ld, labels, lse,
                       gout, (int)V);
    UH_LAUNCH_CHECK();
    return 0;
}

size_t uniter_head_ce_save_bytes(int64_t n, int64_t H, int64_t V) { return head_save(n, H, V).total; }
size_t uniter_head_ce_workspace_bytes(int64_t n, int64_t H, int64_t V) {
    return (size_t)3 * (((size_t)n * H * 2 + 255) / 256 * 256) + head_gemm_ws(n, H, V / 64 * 64);
}

#define HRC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// shared halves: everything up to the logits, and everything after d loss / d logits sits in the logits buffer
static int head_fwd_logits(const UniterHeadParams* p, const void* x, char* S, const HeadSave& sv, int64_t n, int64_t H, int64_t V,
                           float eps, hipStream_t st) {
    const int64_t Vm = V / 64 * 64;
    const DropoutCfg nodrop = make_dropout(0.f, 0, 0);
    HRC(uh::gemm_fwd(uh::GEMM_EPI_BIAS_GELU, x, p->dense_w, p->dense_b, nullptr, S + sv.u, S + sv.g, n, H, H, nodrop, st));
    HRC(uh::layernorm_fwd(S + sv.g, p->ln_g, p->ln_b, S + sv.t, (float*)(S + sv.mean), (float*)(S + sv.rstd), n, H, eps, nodrop, st));
    if (Vm > 0)
        HRC(uh::gemm_fwd(uh::GEMM_EPI_BIAS, S + sv.t, p->proj_w, p->proj_b, nullptr, S + sv.logits, nullptr, n, Vm, H, nodrop, st, H, sv.ld));
    if (V > Vm) {
        hipLaunchKernelGGL(tail_logits_kernel, dim3((unsigned)n), dim3(256), 0, st, (const bf16_t*)(S + sv.t), (const bf16_t*)p->proj_w,
                           (const bf16_t*)p->proj_b, (bf16_t*)(S + sv.logits), sv.ld, (int)H, (int)Vm, (int)V);
        UH_LAUNCH_CHECK();
    }
    return 0;
}

static int head_bwd_from_dl(const UniterHeadParams* p, const void* x, void* dx, char* S, const HeadSave& sv, void* workspace,
                            size_t workspace_bytes, int64_t n, int64_t H, int64_t V, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t Vm = V / 64 * 64;
    const size_t act = ((size_t)n * H * 2 + 255) / 256 * 256;
    char* W = (char*)workspace;
    char *dt = W, *dg = W + act, *du = W + 2 * act, *gws = W + 3 * act;
    const size_t gws_bytes = workspace_bytes - 3 * act;
    bf16_t* dl = (bf16_t*)(S + sv.logits);
    const DropoutCfg nodrop = make_dropout(0.f, 0, 0);
    // dt = dl * W (split-K over the classes) + the sliver
    if (Vm > 0) HRC(uh::gemm_dgrad_splitk(dl, p->proj_w, dt, n, Vm, H, gws, gws_bytes, st, sv.ld));
    else UH_CHECK_HIP(hipMemsetAsync(dt, 0, (size_t)n * H * 2, st));
    if (V > Vm) {
        hipLaunchKernelGGL(tail_dgrad_kernel, dim3((unsigned)n), dim3(256), 0, st, (const bf16_t*)dl, sv.ld, (const bf16_t*)p->proj_w,
                           (bf16_t*)dt, (int)H, (int)Vm, (int)V);
        UH_LAUNCH_CHECK();
    }
    // projection weight gradient, bias gradient out of the same launch, then the sliver rows
    if (Vm > 0) HRC(uh::gemm_wgrad(dl, S + sv.t, p->g_proj_w, n, Vm, H, 1, gws, gws_bytes, st, sv.ld, H, p->g_proj_b));
    if (V > Vm) {
        hipLaunchKernelGGL(tail_wgrad_kernel, dim3((unsigned)(V - Vm)), dim3(256), 0, st, (const bf16_t*)dl, sv.ld,
                           (const bf16_t*)(S + sv.t), (bf16_t*)p->g_proj_w, (bf16_t*)p->g_proj_b, (int)n, (int)H, (int)Vm);
        UH_LAUNCH_CHECK();
    }
    // transform: LayerNorm, GELU, dense
    HRC(uh::layernorm_bwd(dt, nullptr, S + sv.g, (const float*)(S + sv.mean), (const float*)(S + sv.rstd), p->ln_g, dg, nullptr,
                          p->g_ln_g, p->g_ln_b, nullptr, n, H, 1, nodrop, 0, gws, gws_bytes, st));
    HRC(uniter_gelu_bwd(dg, S + sv.u, du, n * H, stream));
    if (dx != nullptr) HRC(uh::gemm_dgrad(uh::GEMM_EPI_RES, du, p->dense_w, nullptr, dx, n, H, H, st));
    HRC(uh::gemm_wgrad(du, x, p->g_dense_w, n, H, H, 1, gws, gws_bytes, st, 0, 0, p->g_dense_b));
    return 0;
}

static int head_bwd_check(const UniterHeadParams* p, int64_t n, int64_t H, int64_t V, size_t workspace_bytes) {
    if (head_check(p, n, H, V)) return -1;
    if (!p->g_dense_w || !p->g_dense_b || !p->g_ln_g || !p->g_ln_b || !p->g_proj_w) { uh_set_error("head: null gradient pointer"); return -1; }
    if (workspace_bytes < uniter_head_ce_workspace_bytes(n, H, V)) { uh_set_error("head: workspace too small"); return -1; }
    return 0;
}

int uniter_head_ce_fwd(const UniterHeadParams* p, const void* x, const int64_t* labels, float* loss, void* save,
                       int64_t n, int64_t H, int64_t V, float eps, void* stream) {
    UH_CHECK_ARG(x && labels && loss && save, "null pointer");
    if (head_check(p, n, H, V)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const HeadSave sv = head_save(n, H, V);
    char* S = (char*)save;
    HRC(head_fwd_logits(p, x, S, sv, n, H, V, eps, st));
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)n), dim3(CT), 0, st, (const bf16_t*)(S + sv.logits), sv.ld, labels, loss,
                       (float*)(S + sv.lse), (int)V);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_head_ce_bwd(const UniterHeadParams* p, const void* x, const int64_t* labels, const float* gloss, void* dx,
                       void* save, void* workspace, size_t workspace_bytes, int64_t n, int64_t H, int64_t V, void* stream) {
    UH_CHECK_ARG(x && labels && gloss && save && workspace, "null pointer");
    if (head_bwd_check(p, n, H, V, workspace_bytes)) return -1;
    const HeadSave sv = head_save(n, H, V);
    char* S = (char*)save;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)n), dim3(CT), 0, (hipStream_t)stream, (bf16_t*)(S + sv.logits), sv.ld, labels,
                       (const float*)(S + sv.lse), gloss, (int)V);
    UH_LAUNCH_CHECK();
    return head_bwd_from_dl(p, x, dx, S, sv, workspace, workspace_bytes, n, H, V, stream);
}

int uniter_head_kl_fwd(const UniterHeadParams* p, const void* x, const float* target, float* loss, void* save,
                       int64_t n, int64_t H, int64_t V, float eps, void* stream) {
    UH_CHECK_ARG(x && target && loss && save, "null pointer");
    if (head_check(p, n, H, V)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const HeadSave sv = head_save(n, H, V);
    char* S = (char*)save;
    HRC(head_fwd_logits(p, x, S, sv, n, H, V, eps, st));
    hipLaunchKernelGGL(kl_fwd_kernel, dim3((unsigned)n), dim3(CT), 0, st, (const bf16_t*)(S + sv.logits), sv.ld, target, loss,
                       (float*)(S + sv.lse), (int)V);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_head_kl_bwd(const UniterHeadParams* p, const void* x, const float* target, const float* gloss, void* dx,
                       void* save, void* workspace, size_t workspace_bytes, int64_t n, int64_t H, int64_t V, void* stream) {
    UH_CHECK_ARG(x && target && gloss && save && workspace, "null pointer");
    if (head_bwd_check(p, n, H, V, workspace_bytes)) return -1;
    const HeadSave sv = head_save(n, H, V);
    char* S = (char*)save;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)n), dim3(CT), 0, (hipStream_t)stream, (bf16_t*)(S + sv.logits), sv.ld, target,
                       (const float*)(S + sv.lse), gloss, (int)V);
    UH_LAUNCH_CHECK();
    return head_bwd_from_dl(p, x, dx, S, sv, workspace, workspace_bytes, n, H, V, stream);
}

int uniter_gelu_bwd(const void* dy, const void* u, void* dx, int64_t numel, void* stream) {
    UH_CHECK_ARG(dy && u && dx, "null pointer");
    if (numel <= 0 || numel % 8 != 0) { uh_set_error("gelu_bwd: need numel %% 8 == 0"); return -1; }
    const int64_t n8 = numel / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)u, (bf16_t*)dx, n8);
    UH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
