// pool.hip — attention pooling of the NLVR2 paired-attention head (reference model/nlvr2.py:110-125, AttentionPool):
//   score_t = relu(h_t . w + b) - 1e4 * pad_t ; p = dropout(softmax_t(score)) ; out = sum_t p_t h_t
// In PyTorch this is ~11 tiny launches forward and ~20 backward on [B, L, H] = [32, 96, 768]; here one workgroup per
// sequence does each direction (fp32 arithmetic on the bf16 inputs).  Saved for backward: raw (pre-mask) scores,
// softmax probabilities and the post-dropout weights, all [B, L] fp32.
#include "common.cuh"
#include "kernels.h"
#include "../../include/uniter_hip.h"

namespace {

constexpr int PT = 1024;                // threads per workgroup: 16 waves — one workgroup per SEQUENCE is all the parallelism there is (32 at the
                                        // benchmark shape), so its two passes over the 147 KB of a sequence are latency chains per wave: with 4
                                        // waves 16.6 / 16.5 us forward / backward, with 16 waves see profiles/r06_pool_threads_ab.txt
constexpr int PNW = PT / 64;            // waves
constexpr int PSL = PT / 128;           // row slices of the weighted-sum passes (thread = (slice, 8-column chunk))
constexpr int PMAXL = 512;            // max_position_embeddings: the longest sequence the encoder itself takes

struct PoolArgs {
    const bf16_t* x;        // [B, L, H]
    const uint8_t* pad;     // [B, L] 1 = padded slot (may be null)
    const bf16_t* w;        // [H]
    const bf16_t* b;        // [1]
    bf16_t* out;            // [B, H]
    float* raw;             // [B, L] relu input (h.w + b)
    float* sm;              // [B, L] softmax
    float* pw;              // [B, L] softmax * dropout multiplier
    const bf16_t* dout;     // [B, H]
    bf16_t* dx;             // [B, L, H]
    float* part;            // [B, H + 1] per-sequence partials of (dw, db)
    int B, L, H;
    DropoutCfg drop;
};

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < PNW; ++w) t += red[w];            // fixed order: deterministic
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int w = 1; w < PNW; ++w) t = fmaxf(t, red[w]);
    return t;
}

// dots of rows t0..t0+3 of x with a vector (H <= 1024: at most two 8-column chunks per lane); all eight row loads
// are issued before the first reduction, so a wave pays one memory round trip per FOUR tokens instead of per token
__device__ __forceinline__ void row_dot4(const bf16_t* x, const bf16_t* vec, int H, int L, int t0, int lane, float (&out)[4]) {
    u32x4 xv[4][2], vv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (lane + 64 * j) * 8;
        vv[j] = u32x4{0u, 0u, 0u, 0u};
        if (c < H) vv[j] = *reinterpret_cast<const u32x4*>(vec + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xv[k][j] = u32x4{0u, 0u, 0u, 0u};
            if (c < H && t0 + k < L) xv[k][j] = *reinterpret_cast<const u32x4*>(x + (int64_t)(t0 + k) * H + c);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float a[8], b[8];
            unpack8(xv[k][j], a);
            unpack8(vv[j], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += a[e] * b[e];
        }
        out[k] = wave_sum(s);
    }
}

__global__ __launch_bounds__(PT) void pool_fwd_kernel(const PoolArgs p) {
    __shared__ float sc[PMAXL];          // scores -> probabilities -> pooling weights
    __shared__ float red[PNW];
    __shared__ float acc2[PSL][1024];    // row slices of the weighted sum, H <= 1024
    const int b = blockIdx.x, L = p.L, H = p.H;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bf16_t* x = p.x + (int64_t)b * L * H;
    const float bias = p.b ? bf2f(p.b[0]) : 0.f;
    for (int t0 = wid * 4; t0 < L; t0 += (PT / 64) * 4) {
        float dots[4];
        row_dot4(x, p.w, H, L, t0, lane, dots);
        if (lane < 4 && t0 + lane < L) {
            const int t = t0 + lane;
            const float raw = (lane == 0 ? dots[0] : (lane == 1 ? dots[1] : (lane == 2 ? dots[2] : dots[3]))) + bias;
            p.raw[(int64_t)b * L + t] = raw;
            sc[t] = fmaxf(raw, 0.f) + ((p.pad && p.pad[(int64_t)b * L + t]) ? -1e4f : 0.f);
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < L; t += PT) mx = fmaxf(mx, sc[t]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int t = threadIdx.x; t < L; t += PT) { const float e = __expf(sc[t] - mx); sc[t] = e; sum += e; }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    for (int t = threadIdx.x; t < L; t += PT) {
        const float s = sc[t] * inv;
        float w = s;
        if (p.drop.p > 0.f) {
            const uint64_t idx = (uint64_t)b * (uint64_t)L + (uint64_t)t;
            w *= dropout_mult1(p.drop, idx >> 2, (int)(idx & 3));
        }
        p.sm[(int64_t)b * L + t] = s;
        p.pw[(int64_t)b * L + t] = w;
        sc[t] = w;
    }
    __syncthreads();
    // out[d] = sum_t w_t x[t][d]: thread = (row slice, 8-column chunk)
    const int nchunk = H >> 3;
    const int half = threadIdx.x / 128, c = threadIdx.x % 128;
    const int lq = (L + PSL - 1) / PSL;
    if (c < nchunk) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int t0 = min(L, half * lq), t1 = min(L, (half + 1) * lq);
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + (int64_t)t * H + c * 8), v);
            const float w = sc[t];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += w * v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc2[half][c * 8 + e] = a[e];
    }
    __syncthreads();
    if (threadIdx.x < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < PSL; ++q) t += acc2[q][threadIdx.x * 8 + e];
            o[e] = t;
        }
        *reinterpret_cast<u32x4*>(p.out + (int64_t)b * H + threadIdx.x * 8) = pack8(o);
    }
}

__global__ __launch_bounds__(PT) void pool_bwd_kernel(const PoolArgs p) {
    __shared__ float ds[PMAXL];          // d out / d (pre-relu score), then reused
    __shared__ float pwv[PMAXL];
    __shared__ float red[PNW];
    __shared__ float acc2[PSL][1024];
    const int b = blockIdx.x, L = p.L, H = p.H;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bf16_t* x = p.x + (int64_t)b * L * H;
    const bf16_t* dout = p.dout + (int64_t)b * H;
    // d pw_t = x_t . dout ; through dropout: d sm_t = d pw_t * (pw_t / sm_t)
    for (int t0 = wid * 4; t0 < L; t0 += (PT / 64) * 4) {
        float dots[4];
        row_dot4(x, dout, H, L, t0, lane, dots);
        if (lane < 4 && t0 + lane < L) {
            const int t = t0 + lane;
            const float d = lane == 0 ? dots[0] : (lane == 1 ? dots[1] : (lane == 2 ? dots[2] : dots[3]));
            const float s = p.sm[(int64_t)b * L + t], w = p.pw[(int64_t)b * L + t];
            ds[t] = (s > 0.f) ? d * (w / s) : 0.f;
            pwv[t] = w;
        }
    }
    __syncthreads();
    float dotp = 0.f;
    for (int t = threadIdx.x; t < L; t += PT) dotp += p.sm[(int64_t)b * L + t] * ds[t];
    dotp = block_sum(dotp, red);
    float dbias = 0.f;
    for (int t = threadIdx.x; t < L; t += PT) {
        const float s = p.sm[(int64_t)b * L + t];
        float g = s * (ds[t] - dotp);                                   // softmax backward
        if (!(p.raw[(int64_t)b * L + t] > 0.f)) g = 0.f;                // relu; the additive mask has no gradient
        ds[t] = g;
        dbias += g;
    }
    dbias = block_sum(dbias, red);
    // dx[t][d] = pw_t dout[d] + ds_t w[d] ; dw[d] partial = sum_t ds_t x[t][d]
    const int nchunk = H >> 3;
    const int half = threadIdx.x / 128, c = threadIdx.x % 128;
    const int lq = (L + PSL - 1) / PSL;
    if (c < nchunk) {
        float dov[8], wv[8], a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unpack8(*reinterpret_cast<const u32x4*>(dout + c * 8), dov);
        unpack8(*reinterpret_cast<const u32x4*>(p.w + c * 8), wv);
        const int t0 = min(L, half * lq), t1 = min(L, (half + 1) * lq);
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            float v[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + (int64_t)t * H + c * 8), v);
            const float w = pwv[t], g = ds[t];
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = w * dov[e] + g * wv[e]; a[e] += g * v[e]; }
            *reinterpret_cast<u32x4*>(p.dx + ((int64_t)b * L + t) * H + c * 8) = pack8(o);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc2[half][c * 8 + e] = a[e];
    }
    __syncthreads();
    float* part = p.part + (int64_t)b * (H + 1);
    for (int d = threadIdx.x; d < H; d += PT) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < PSL; ++q) t += acc2[q][d];
        part[d] = t;
    }
    if (threadIdx.x == 0) part[H] = dbias;
}

// dw[d] += sum_b part[b][d] ; db += sum_b part[b][H]   (fixed order: deterministic)
__global__ __launch_bounds__(PT) void pool_finalize_kernel(const float* __restrict__ part, int B, int H, bf16_t* dw, bf16_t* db) {
    const int d = blockIdx.x * PT + threadIdx.x;
    if (d > H) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[(int64_t)b * (H + 1) + d];
    if (d < H) { if (dw) dw[d] = f2bf(bf2f(dw[d]) + s); }
    else if (db) db[0] = f2bf(bf2f(db[0]) + s);
}

int check(int64_t B, int64_t L, int64_t H) {
    if (B <= 0 || L <= 0 || H <= 0) { uh_set_error("attn_pool: non-positive dimension"); return -1; }
    if (L > PMAXL || H > 1024 || H % 8 != 0) { uh_set_error("attn_pool: need L <= 512, H <= 1024, H %% 8 == 0"); return -1; }
    return 0;
}

}  // namespace

extern "C" {

size_t uniter_attn_pool_workspace_bytes(int64_t B, int64_t H) { return (size_t)B * (size_t)(H + 1) * sizeof(float); }

int uniter_attn_pool_fwd(const void* x, const uint8_t* pad, const void* w, const void* b, void* out,
                         float* raw, float* sm, float* pw, int64_t B, int64_t L, int64_t H,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(x && w && out && raw && sm && pw, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    if (check(B, L, H)) return -1;
    PoolArgs a{};
    a.x = (const bf16_t*)x; a.pad = pad; a.w = (const bf16_t*)w; a.b = (const bf16_t*)b; a.out = (bf16_t*)out;
    a.raw = raw; a.sm = sm; a.pw = pw; a.B = (int)B; a.L = (int)L; a.H = (int)H;
    a.drop = make_dropout(p_drop, seed, offset);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)B), dim3(PT), 0, (hipStream_t)stream, a);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_attn_pool_bwd(const void* x, const void* w, const float* raw, const float* sm, const float* pw,
                         const void* dout, void* dx, void* dw, void* db, int64_t B, int64_t L, int64_t H,
                         void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(x && w && raw && sm && pw && dout && dx && workspace, "null pointer");
    if (check(B, L, H)) return -1;
    UH_CHECK_ARG(workspace_bytes >= uniter_attn_pool_workspace_bytes(B, H), "workspace too small");
    PoolArgs a{};
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.raw = const_cast<float*>(raw); a.sm = const_cast<float*>(sm);
    a.pw = const_cast<float*>(pw); a.dout = (const bf16_t*)dout; a.dx = (bf16_t*)dx; a.part = (float*)workspace;
    a.B = (int)B; a.L = (int)L; a.H = (int)H;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)B), dim3(PT), 0, (hipStream_t)stream, a);
    UH_LAUNCH_CHECK();
    if (dw != nullptr || db != nullptr) {
        hipLaunchKernelGGL(pool_finalize_kernel, dim3((unsigned)((H + 1 + PT - 1) / PT)), dim3(PT), 0, (hipStream_t)stream,
                           (const float*)workspace, (int)B, (int)H, (bf16_t*)dw, (bf16_t*)db);
        UH_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
