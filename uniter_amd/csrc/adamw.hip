// adamw.hip — fused multi-tensor AdamW, global gradient norm and fused clipping.
//
// Reference: optim/adamw.py:40-103 (HF-style AdamW: per-parameter Python loop of ~6 elementwise
// kernels) and torch.nn.utils.clip_grad_norm_ as called at pretrain.py:329-331.
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; denom = sqrt(v) + eps
//   step_size = lr * sqrt(1-b2^t)/(1-b1^t)   (correct_bias)
//   p -= step_size * m/denom ; p -= lr*wd*p   (decay applied AFTER the Adam update, with the raw lr)
// One launch updates every tensor: the tensor table lives on the device, work is cut into fixed-size
// chunks (chunk -> tensor map built at plan creation).  HBM-bound: 4 fp32 streams read + 3 written
// (+ bf16 grad read / bf16 param written) per element, 16-byte accesses.
#include "common.cuh"
#include "kernels.h"
#include "../../include/uniter_hip.h"

#include <algorithm>
#include <mutex>
#include <vector>

namespace {

constexpr int CHUNK = 4096;          // elements per block-iteration (256 threads x 16)
constexpr int MAX_GROUPS = 16;

struct DevTensor {
    void* param;
    void* grad;
    float* master;
    float* m;
    float* v;
    int64_t numel;
    int32_t group;
    int32_t is_bf16;     // bit 0: bf16 parameter / gradient; bit 1 (DT_KEEP_GRAD): the fused zero_grad leaves this gradient alone — its
                         // producer overwrites it in the next backward pass (uniter_adamw_plan_keep_grads)
};
constexpr int32_t DT_BF16 = 1, DT_KEEP_GRAD = 2, DT_SKIP_NORM = 4;   // (bits 1, 2: uniter_adamw_plan_set_flags)
struct ChunkRef {
    int32_t tensor;
    int32_t chunk;     // chunk index inside the tensor
};
struct GroupHyper {
    float lr, beta1, beta2, eps, wd, step_size;
};
struct HyperTable {
    GroupHyper g[MAX_GROUPS];
};

struct Plan {
    std::vector<uintptr_t> chunk_addr;   // host copy: parameter address of every chunk, ascending (chunks are sorted by it)
    DevTensor* d_tensors = nullptr;
    ChunkRef* d_chunks = nullptr;
    float* d_partial = nullptr;      // per-block partial sums for the norm
    int64_t n_tensors = 0;
    int64_t n_chunks = 0;
    int norm_blocks = 0;
    std::vector<ChunkRef> h_chunks;  // host copy of d_chunks (uniter_adamw_plan_set_flags builds the filtered list from it)
    ChunkRef* d_norm_chunks = nullptr;   // the chunks of the tensors NOT flagged UNITER_ADAM_SKIP_NORM (nullptr: no tensor is flagged)
    int64_t n_norm_chunks = 0;
};

// `zero_grads`: the gradient element is overwritten with zero once it has been read — optimizer.zero_grad() folded into
// the update (same bytes written as the separate memset, one pass and one launch fewer).
// NT = the update's streams (every byte is touched exactly once) are loaded and stored with the non-temporal hint, so that they
// do not displace what the next forward pass wants to find in the L2s / the Infinity Cache.  Same arithmetic either way; the
// step ends on bit-identical parameters (tests/test_gpu_parity.py digest test) and is 1.3-1.5 % shorter (round 5, one box, A/B:
// 4.53 / 4.52 ms default policy, 4.46 ms non-temporal), so NT is what runs (the template parameter stays for an A/B build).
// (the table hands the kernel generic pointers; named as global at the access they become global_load / global_store
//  instead of flat instructions, like every other kernel of the library — common.cuh: ldg16)
template <bool NT, typename V>
__device__ __forceinline__ V adam_ld(const V* p) {
    typedef __attribute__((address_space(1))) V GV;
    if constexpr (NT) return __builtin_nontemporal_load((const GV*)p);
    else return *(const GV*)p;
}
template <bool NT, typename V>
__device__ __forceinline__ void adam_st(V* p, const V v) {
    typedef __attribute__((address_space(1))) V GV;
    if constexpr (NT) __builtin_nontemporal_store(v, (GV*)p);
    else *(GV*)p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void adamw_kernel(const DevTensor* __restrict__ tensors,
                                                    const ChunkRef* __restrict__ chunks, int64_t chunk_begin, int64_t n_chunks,
                                                    const HyperTable hyp, const GroupHyper* __restrict__ hyp_dev,
                                                    const float* __restrict__ clip_coef, const int zero_grads) {
    const float coef = clip_coef ? *clip_coef : 1.0f;
    for (int64_t ci = chunk_begin + blockIdx.x; ci < n_chunks; ci += gridDim.x) {
        const ChunkRef cr = chunks[ci];
        DevTensor t = tensors[cr.tensor];
        const bool zero_this = zero_grads && !(t.is_bf16 & DT_KEEP_GRAD);
        t.is_bf16 &= DT_BF16;
        const GroupHyper h = hyp_dev ? hyp_dev[t.group] : hyp.g[t.group];
        const int64_t base = (int64_t)cr.chunk * CHUNK;
#pragma unroll
        for (int it = 0; it < CHUNK / (256 * 4); ++it) {
            const int64_t idx = base + ((int64_t)it * 256 + threadIdx.x) * 4;
            if (idx >= t.numel) break;
            const int n = (t.numel - idx >= 4) ? 4 : (int)(t.numel - idx);
            float g[4] = {0.f, 0.f, 0.f, 0.f}, p[4] = {0.f, 0.f, 0.f, 0.f}, m[4], v[4];
            const bool vec = (n == 4);
            float* pm = t.is_bf16 ? t.master : (float*)t.param;
            if (vec) {
                if (t.is_bf16) unpack4(adam_ld<NT>(reinterpret_cast<const u32x2*>((const bf16_t*)t.grad + idx)), g);
                else { const f32x4 q = adam_ld<NT>(reinterpret_cast<const f32x4*>((const float*)t.grad + idx)); g[0] = q[0]; g[1] = q[1]; g[2] = q[2]; g[3] = q[3]; }
                const f32x4 pq = adam_ld<NT>(reinterpret_cast<const f32x4*>(pm + idx));
                const f32x4 mq = adam_ld<NT>(reinterpret_cast<const f32x4*>(t.m + idx));
                const f32x4 vq = adam_ld<NT>(reinterpret_cast<const f32x4*>(t.v + idx));
#pragma unroll
                for (int e = 0; e < 4; ++e) { p[e] = pq[e]; m[e] = mq[e]; v[e] = vq[e]; }
            } else {
                for (int e = 0; e < 4; ++e) {
                    m[e] = 0.f; v[e] = 0.f;
                    if (e < n) {
                        g[e] = t.is_bf16 ? bf2f(((const bf16_t*)t.grad)[idx + e]) : ((const float*)t.grad)[idx + e];
                        p[e] = pm[idx + e]; m[e] = t.m[idx + e]; v[e] = t.v[idx + e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = g[e] * coef;
                m[e] = h.beta1 * m[e] + (1.0f - h.beta1) * ge;
                v[e] = h.beta2 * v[e] + (1.0f - h.beta2) * ge * ge;
                const float denom = sqrtf(v[e]) + h.eps;
                p[e] = p[e] - h.step_size * (m[e] / denom);
                if (h.wd > 0.f) p[e] = p[e] - h.lr * h.wd * p[e];
            }
            if (vec) {
                adam_st<NT>(reinterpret_cast<f32x4*>(pm + idx), f32x4{p[0], p[1], p[2], p[3]});
                adam_st<NT>(reinterpret_cast<f32x4*>(t.m + idx), f32x4{m[0], m[1], m[2], m[3]});
                adam_st<NT>(reinterpret_cast<f32x4*>(t.v + idx), f32x4{v[0], v[1], v[2], v[3]});
                // (the bf16 weights are what the next forward pass reads, the zeroed gradients what the next backward pass
                //  accumulates into: both keep the default policy)
                if (t.is_bf16) adam_st<false>(reinterpret_cast<u32x2*>((bf16_t*)t.param + idx), pack4(p));
                if (zero_this) {
                    if (t.is_bf16) adam_st<false>(reinterpret_cast<u32x2*>((bf16_t*)t.grad + idx), u32x2{0u, 0u});
                    else adam_st<false>(reinterpret_cast<f32x4*>((float*)t.grad + idx), f32x4{0.f, 0.f, 0.f, 0.f});
                }
            } else {
                for (int e = 0; e < n; ++e) {
                    pm[idx + e] = p[e]; t.m[idx + e] = m[e]; t.v[idx + e] = v[e];
                    if (t.is_bf16) ((bf16_t*)t.param)[idx + e] = f2bf(p[e]);
                    if (zero_this) {
                        if (t.is_bf16) ((bf16_t*)t.grad)[idx + e] = f2bf(0.f);
                        else ((float*)t.grad)[idx + e] = 0.f;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void gradsq_kernel(const DevTensor* __restrict__ tensors,
                                                     const ChunkRef* __restrict__ chunks, int64_t n_chunks,
                                                     float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    // The chunk descriptor (two dependent loads) of the NEXT iteration is fetched while the current chunk streams, and the
    // four 8-byte loads of a bf16 chunk are issued together: per 8 KiB of gradients the kernel used to pay three
    // dependent memory round trips.
    int64_t ci = blockIdx.x;
    ChunkRef cr{};
    DevTensor t{};
    if (ci < n_chunks) { cr = chunks[ci]; t = tensors[cr.tensor]; t.is_bf16 &= DT_BF16; }
    while (ci < n_chunks) {
        const int64_t nci = ci + gridDim.x;
        ChunkRef ncr{};
        DevTensor nt{};
        if (nci < n_chunks) { ncr = chunks[nci]; nt = tensors[ncr.tensor]; nt.is_bf16 &= DT_BF16; }
        const int64_t base = (int64_t)cr.chunk * CHUNK;
        if (t.is_bf16) {
            u32x2 raw[CHUNK / (256 * 4)];
#pragma unroll
            for (int it = 0; it < CHUNK / (256 * 4); ++it) {
                const int64_t idx = base + ((int64_t)it * 256 + threadIdx.x) * 4;
                raw[it] = u32x2{0u, 0u};
                if (idx + 4 <= t.numel) raw[it] = adam_ld<false>(reinterpret_cast<const u32x2*>((const bf16_t*)t.grad + idx));
            }
#pragma unroll
            for (int it = 0; it < CHUNK / (256 * 4); ++it) {
                const int64_t idx = base + ((int64_t)it * 256 + threadIdx.x) * 4;
                float g[4];
                unpack4(raw[it], g);
                acc += (g[0] * g[0] + g[1] * g[1]) + (g[2] * g[2] + g[3] * g[3]);
                if (idx < t.numel && idx + 4 > t.numel) {            // ragged end of a tensor
                    for (int64_t e = idx; e < t.numel; ++e) { const float ge = bf2f(((const bf16_t*)t.grad)[e]); acc += ge * ge; }
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < CHUNK / (256 * 4); ++it) {
                const int64_t idx = base + ((int64_t)it * 256 + threadIdx.x) * 4;
                if (idx >= t.numel) break;
                const int n = (t.numel - idx >= 4) ? 4 : (int)(t.numel - idx);
                if (n == 4) {
                    const f32x4 q = adam_ld<false>(reinterpret_cast<const f32x4*>((const float*)t.grad + idx));
                    acc += (q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]);
                } else {
                    for (int e = 0; e < n; ++e) { const float ge = ((const float*)t.grad)[idx + e]; acc += ge * ge; }
                }
            }
        }
        ci = nci; cr = ncr; t = nt;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// norm_out[0] = sqrt(sum)*grad_scale ; norm_out[1] = grad_scale * min(1, max_norm/(norm+1e-6))
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ partial, int n,
                                                            float grad_scale, float max_norm, float* __restrict__ out,
                                                            const float* __restrict__ extra, int n_extra) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    for (int i = threadIdx.x; i < n_extra; i += 256) acc += extra[i];      // per-tile sums a producer kernel left (grad_norm_ex)
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (red[0] + red[1]) + (red[2] + red[3]);
        const float norm = sqrtf(total) * grad_scale;
        float coef = grad_scale;
        if (max_norm > 0.f) {
            const float c = max_norm / (norm + 1e-6f);
            if (c < 1.0f) coef = grad_scale * c;
        }
        out[0] = norm;
        out[1] = coef;
    }
}

}  // namespace

extern "C" {

int uniter_adamw_plan_create(const UniterAdamTensor* tensors, int64_t n_tensors, void** plan_out) {
    UH_CHECK_ARG(tensors != nullptr && plan_out != nullptr && n_tensors > 0, "null / empty tensor table");
    std::vector<DevTensor> dt((size_t)n_tensors);
    std::vector<ChunkRef> ch;
    for (int64_t i = 0; i < n_tensors; ++i) {
        const UniterAdamTensor& t = tensors[i];
        UH_CHECK_ARG(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.numel > 0, "tensor entry has null pointer / zero size");
        UH_CHECK_ARG(!t.param_is_bf16 || t.master != nullptr, "bf16 parameter needs an fp32 master copy");
        UH_CHECK_ARG(t.group >= 0 && t.group < MAX_GROUPS, "group index out of range (max 16 groups)");
        // vector accesses need 16-byte (fp32) / 8-byte (bf16) alignment of every stream
        UH_CHECK_ARG(((uintptr_t)t.exp_avg % 16 == 0) && ((uintptr_t)t.exp_avg_sq % 16 == 0), "optimizer state must be 16-byte aligned");
        UH_CHECK_ARG(((uintptr_t)t.param % (t.param_is_bf16 ? 8 : 16) == 0) && ((uintptr_t)t.grad % (t.param_is_bf16 ? 8 : 16) == 0),
                     "param / grad pointers must be 8-byte (bf16) / 16-byte (fp32) aligned");
        UH_CHECK_ARG(!t.param_is_bf16 || ((uintptr_t)t.master % 16 == 0), "master copy must be 16-byte aligned");
        dt[(size_t)i] = DevTensor{t.param, (void*)t.grad, t.master, t.exp_avg, t.exp_avg_sq, t.numel, t.group, t.param_is_bf16 ? DT_BF16 : 0};
    }
    // chunks in ascending parameter-address order: with the parameters in one arena laid out in module order
    // (utils/arena.py) that is the order the next forward pass needs them in, which is what the segmented asynchronous
    // step (uniter_adamw_step_async) relies on; for a single launch the order is irrelevant
    std::vector<int64_t> order((size_t)n_tensors);
    for (int64_t i = 0; i < n_tensors; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return (uintptr_t)tensors[a].param < (uintptr_t)tensors[b].param; });
    std::vector<uintptr_t> chunk_addr;
    for (int64_t oi = 0; oi < n_tensors; ++oi) {
        const int64_t i = order[(size_t)oi];
        const UniterAdamTensor& t = tensors[i];
        const int64_t nc = (t.numel + CHUNK - 1) / CHUNK;
        const size_t esz = t.param_is_bf16 ? 2 : 4;
        for (int64_t c = 0; c < nc; ++c) {
            ch.push_back(ChunkRef{(int32_t)i, (int32_t)c});
            chunk_addr.push_back((uintptr_t)t.param + (size_t)c * CHUNK * esz);
        }
    }
    Plan* p = new Plan();
    p->chunk_addr = chunk_addr;
    p->h_chunks = ch;
    p->n_tensors = n_tensors;
    p->n_chunks = (int64_t)ch.size();
    p->norm_blocks = (int)(p->n_chunks < 1024 ? p->n_chunks : 1024);
    hipError_t e;
    if ((e = hipMalloc(&p->d_tensors, dt.size() * sizeof(DevTensor))) != hipSuccess ||
        (e = hipMalloc(&p->d_chunks, ch.size() * sizeof(ChunkRef))) != hipSuccess ||
        (e = hipMalloc(&p->d_partial, 1024 * sizeof(float))) != hipSuccess ||
        (e = hipMemcpy(p->d_tensors, dt.data(), dt.size() * sizeof(DevTensor), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->d_chunks, ch.data(), ch.size() * sizeof(ChunkRef), hipMemcpyHostToDevice)) != hipSuccess) {
        uh_set_error("uniter_adamw_plan_create: %s", hipGetErrorString(e));
        if (p->d_tensors) (void)hipFree(p->d_tensors);
        if (p->d_chunks) (void)hipFree(p->d_chunks);
        if (p->d_partial) (void)hipFree(p->d_partial);
        delete p;
        return (int)e;
    }
    *plan_out = p;
    return 0;
}

int uniter_adamw_plan_set_flags(void* plan, const uint8_t* flags, int64_t n_tensors) {
    UH_CHECK_ARG(plan != nullptr && flags != nullptr, "null pointer");
    Plan* p = (Plan*)plan;
    UH_CHECK_ARG(n_tensors == p->n_tensors, "one flag byte per tensor of the plan");
    std::vector<DevTensor> dt((size_t)n_tensors);
    UH_CHECK_HIP(hipMemcpy(dt.data(), p->d_tensors, dt.size() * sizeof(DevTensor), hipMemcpyDeviceToHost));
    bool any_skip = false;
    for (int64_t i = 0; i < n_tensors; ++i) {
        dt[(size_t)i].is_bf16 = (dt[(size_t)i].is_bf16 & DT_BF16) | ((flags[i] & UNITER_ADAM_KEEP_GRAD) ? DT_KEEP_GRAD : 0) |
                                ((flags[i] & UNITER_ADAM_SKIP_NORM) ? DT_SKIP_NORM : 0);
        any_skip = any_skip || (flags[i] & UNITER_ADAM_SKIP_NORM);
    }
    UH_CHECK_HIP(hipMemcpy(p->d_tensors, dt.data(), dt.size() * sizeof(DevTensor), hipMemcpyHostToDevice));
    if (p->d_norm_chunks != nullptr) { (void)hipFree(p->d_norm_chunks); p->d_norm_chunks = nullptr; }
    p->n_norm_chunks = 0;
    if (any_skip) {
        std::vector<ChunkRef> sel;
        for (const ChunkRef& c : p->h_chunks)
            if (!(flags[c.tensor] & UNITER_ADAM_SKIP_NORM)) sel.push_back(c);
        UH_CHECK_HIP(hipMalloc(&p->d_norm_chunks, std::max<size_t>(sel.size(), 1) * sizeof(ChunkRef)));
        if (!sel.empty()) UH_CHECK_HIP(hipMemcpy(p->d_norm_chunks, sel.data(), sel.size() * sizeof(ChunkRef), hipMemcpyHostToDevice));
        p->n_norm_chunks = (int64_t)sel.size();
    }
    return 0;
}

int uniter_adamw_plan_destroy(void* plan) {
    if (plan == nullptr) return 0;
    Plan* p = (Plan*)plan;
    (void)hipFree(p->d_tensors);
    (void)hipFree(p->d_chunks);
    (void)hipFree(p->d_partial);
    if (p->d_norm_chunks != nullptr) (void)hipFree(p->d_norm_chunks);
    delete p;
    return 0;
}

int uniter_adamw_grad_norm_ex(void* plan, float grad_scale, float max_norm, float* norm_out, const float* extra, int32_t n_extra,
                              void* stream) {
    UH_CHECK_ARG(plan != nullptr && norm_out != nullptr, "null pointer");
    UH_CHECK_ARG(n_extra >= 0 && (n_extra == 0 || extra != nullptr), "bad list of extra partial sums");
    Plan* p = (Plan*)plan;
    hipStream_t st = (hipStream_t)stream;
    // extra partial sums stand for the tensors flagged UNITER_ADAM_SKIP_NORM: the reduction then walks the other tensors' chunks only
    const bool folded = n_extra > 0 && p->d_norm_chunks != nullptr;
    UH_CHECK_ARG(n_extra == 0 || folded, "extra partial sums need tensors flagged UNITER_ADAM_SKIP_NORM (uniter_adamw_plan_set_flags)");
    const ChunkRef* chunks = folded ? p->d_norm_chunks : p->d_chunks;
    const int64_t n_chunks = folded ? p->n_norm_chunks : p->n_chunks;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(n_chunks, p->norm_blocks));
    hipLaunchKernelGGL(gradsq_kernel, dim3(blocks), dim3(256), 0, st, (const DevTensor*)p->d_tensors, chunks, n_chunks, p->d_partial);
    UH_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)p->d_partial, blocks,
                       grad_scale, max_norm, norm_out, extra, (int)n_extra);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_adamw_grad_norm(void* plan, float grad_scale, float max_norm, float* norm_out, void* stream) {
    return uniter_adamw_grad_norm_ex(plan, grad_scale, max_norm, norm_out, nullptr, 0, stream);
}

static int fill_hyper(const UniterAdamGroup* groups, int32_t n_groups, HyperTable* ht) {
    for (int i = 0; i < n_groups; ++i) {
        const UniterAdamGroup& g = groups[i];
        UH_CHECK_ARG(g.step >= 1, "step must be >= 1");
        double step_size = g.lr;
        if (g.correct_bias) {
            const double bc1 = 1.0 - pow((double)g.beta1, (double)g.step);
            const double bc2 = 1.0 - pow((double)g.beta2, (double)g.step);
            step_size = step_size * sqrt(bc2) / bc1;
        }
        ht->g[i] = GroupHyper{g.lr, g.beta1, g.beta2, g.eps, g.weight_decay, (float)step_size};
    }
    return 0;
}

static int adamw_step_impl(void* plan, const UniterAdamGroup* groups, int32_t n_groups, const float* clip_coef, int zero_grads,
                           void* stream);

static void adamw_launch(int64_t blocks, hipStream_t st, const DevTensor* tensors, const ChunkRef* chunks, int64_t begin, int64_t end,
                         const HyperTable& ht, const GroupHyper* dev_hyper, const float* clip_coef, int zero_grads) {
    hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, tensors, chunks, begin, end, ht, dev_hyper,
                       clip_coef, zero_grads);
}

int uniter_adamw_step(void* plan, const UniterAdamGroup* groups, int32_t n_groups,
                      const float* clip_coef, void* stream) {
    return adamw_step_impl(plan, groups, n_groups, clip_coef, 0, stream);
}

int uniter_adamw_step_zero(void* plan, const UniterAdamGroup* groups, int32_t n_groups,
                           const float* clip_coef, void* stream) {
    return adamw_step_impl(plan, groups, n_groups, clip_coef, 1, stream);
}

static int adamw_step_impl(void* plan, const UniterAdamGroup* groups, int32_t n_groups, const float* clip_coef, int zero_grads,
                           void* stream) {
    UH_CHECK_ARG(plan != nullptr && groups != nullptr, "null pointer");
    UH_CHECK_ARG(n_groups > 0 && n_groups <= MAX_GROUPS, "1..16 parameter groups supported");
    Plan* p = (Plan*)plan;
    HyperTable ht{};
    { int rc = fill_hyper(groups, n_groups, &ht); if (rc) return rc; }
    // enough blocks to fill the chip several times over; chunks are grid-strided
    int64_t blocks = p->n_chunks < 8192 ? p->n_chunks : 8192;
    uh::LaunchTimer lt(uh::TIME_ADAMW, p->n_chunks, 0, 0, (hipStream_t)stream);
    adamw_launch(blocks, (hipStream_t)stream, (const DevTensor*)p->d_tensors, (const ChunkRef*)p->d_chunks, (int64_t)0, p->n_chunks, ht,
                 (const GroupHyper*)nullptr, clip_coef, zero_grads);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_adamw_step_dev(void* plan, const float* dev_hyper, int32_t n_groups, const float* clip_coef, void* stream) {
    UH_CHECK_ARG(plan != nullptr && dev_hyper != nullptr, "null pointer");
    UH_CHECK_ARG(n_groups > 0 && n_groups <= MAX_GROUPS, "1..16 parameter groups supported");
    Plan* p = (Plan*)plan;
    HyperTable ht{};
    int64_t blocks = p->n_chunks < 8192 ? p->n_chunks : 8192;
    uh::LaunchTimer lt(uh::TIME_ADAMW, p->n_chunks, 0, 0, (hipStream_t)stream);
    adamw_launch(blocks, (hipStream_t)stream, (const DevTensor*)p->d_tensors, (const ChunkRef*)p->d_chunks, (int64_t)0, p->n_chunks, ht,
                 (const GroupHyper*)dev_hyper, clip_coef, 0);
    UH_LAUNCH_CHECK();
    return 0;
}

// ---- asynchronous, segmented step -------------------------------------------------------------------------------------
// The update is HBM-bound (28 B per parameter) while the forward pass that follows it is MFMA / LDS bound, so the two
// overlap almost for free — but only if the forward pass can start before the last parameter is written.  The step is
// therefore cut at caller-given parameter addresses (ascending; typically the first parameter of every BertLayer) into
// segments that run back to back on an internal stream, each followed by an event; consumers call
// uniter_params_wait(address, stream) before they read a parameter (the encoder does it per layer) and
// uniter_params_wait_all(stream) before they write gradients again (the fused zero_grad also runs on that stream).
namespace {
struct ParamSegment { uintptr_t lo, hi; hipEvent_t ev; };
struct ParamTracker {
    hipStream_t stream = nullptr;
    hipEvent_t start = nullptr;
    std::vector<hipEvent_t> pool;
    std::vector<ParamSegment> pending;   // in issue order
};
constexpr int MAX_TRACKED_DEVICES = 16;
ParamTracker g_pts[MAX_TRACKED_DEVICES];   // one tracker (stream, events, pending segments) per device: events and streams
std::mutex g_pt_mu;                        // belong to the device they were created on.  Process-wide: the step is issued by
                                           // the training thread, waits may come from autograd worker threads

int tracker_get(ParamTracker** out) {
    int dev = 0;
    UH_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_TRACKED_DEVICES) { uh_set_error("adamw: device index %d out of range", dev); return -1; }
    ParamTracker& t = g_pts[dev];
    if (t.stream == nullptr) {
        UH_CHECK_HIP(hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking));
        UH_CHECK_HIP(hipEventCreateWithFlags(&t.start, hipEventDisableTiming));
    }
    *out = &t;
    return 0;
}
}  // namespace

int uniter_adamw_step_async(void* plan, const UniterAdamGroup* groups, int32_t n_groups, const float* clip_coef,
                            const void* const* bounds, int32_t n_bounds, int32_t zero_grads, void* stream) {
    UH_CHECK_ARG(plan != nullptr && groups != nullptr, "null pointer");
    UH_CHECK_ARG(n_groups > 0 && n_groups <= MAX_GROUPS, "1..16 parameter groups supported");
    UH_CHECK_ARG(n_bounds >= 0 && n_bounds <= 255 && (n_bounds == 0 || bounds != nullptr), "0..255 segment boundaries");
    Plan* p = (Plan*)plan;
    HyperTable ht{};
    { int rc = fill_hyper(groups, n_groups, &ht); if (rc) return rc; }
    std::lock_guard<std::mutex> lk(g_pt_mu);
    ParamTracker* ptp = nullptr;
    { int rc = tracker_get(&ptp); if (rc) return rc; }
    ParamTracker& g_pt = *ptp;
    hipStream_t main = (hipStream_t)stream, side = g_pt.stream;
    // every segment end is computed and checked before anything is launched: a bad boundary list must not leave a
    // partially applied step behind
    std::vector<int64_t> ends;
    {
        int64_t begin = 0;
        for (int sgm = 0; sgm <= n_bounds; ++sgm) {
            int64_t end = p->n_chunks;
            if (sgm < n_bounds) {
                const uintptr_t b = (uintptr_t)bounds[sgm];
                end = (int64_t)(std::lower_bound(p->chunk_addr.begin(), p->chunk_addr.end(), b) - p->chunk_addr.begin());
                if (end < begin) { uh_set_error("uniter_adamw_step_async: boundaries must ascend"); return -1; }
            }
            ends.push_back(end);
            begin = end;
        }
    }
    while (g_pt.pool.size() < ends.size()) {
        hipEvent_t e;
        UH_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g_pt.pool.push_back(e);
    }
    // an earlier asynchronous step may still be pending on the side stream: it is ordered before this one there
    UH_CHECK_HIP(hipEventRecord(g_pt.start, main));              // gradients + clip coefficient are final here
    UH_CHECK_HIP(hipStreamWaitEvent(side, g_pt.start, 0));
    g_pt.pending.clear();
    int64_t begin = 0;
    for (int sgm = 0; sgm <= n_bounds; ++sgm) {
        const int64_t end = ends[(size_t)sgm];
        if (end == begin) continue;
        const int64_t n = end - begin;
        // (throttled grids for the later segments were measured worse than flat-out ones, DESIGN.md / EXPERIMENTS.md round 2)
        const int64_t blocks = n < 8192 ? n : 8192;
        {
            uh::LaunchTimer lt(uh::TIME_ADAMW, n, 0, 0, side);
            adamw_launch(blocks, side, (const DevTensor*)p->d_tensors, (const ChunkRef*)p->d_chunks, begin, end, ht,
                         (const GroupHyper*)nullptr, clip_coef, (int)zero_grads);
            UH_LAUNCH_CHECK();
        }
        hipEvent_t ev = g_pt.pool[g_pt.pending.size()];
        UH_CHECK_HIP(hipEventRecord(ev, side));
        const uintptr_t lo = p->chunk_addr[(size_t)begin];
        const uintptr_t hi = end < p->n_chunks ? p->chunk_addr[(size_t)end] : ~(uintptr_t)0;
        g_pt.pending.push_back(ParamSegment{lo, hi, ev});
        begin = end;
    }
    return 0;
}

}  // extern "C"
namespace uh {
// an asynchronous optimizer step has segments the forward pass still has to wait for (encoder.hip: such a forward keeps its
// kernels in queue order — the per-layer stream waits must stay between them)
bool params_pending() {
    std::lock_guard<std::mutex> lk(g_pt_mu);
    ParamTracker* ptp = nullptr;
    if (tracker_get(&ptp)) return true;
    return !ptp->pending.empty();
}
}  // namespace uh
extern "C" {

int uniter_params_wait(const void* addr, void* stream) {
    std::lock_guard<std::mutex> lk(g_pt_mu);
    ParamTracker* ptp = nullptr;
    { int rc = tracker_get(&ptp); if (rc) return rc; }
    ParamTracker& g_pt = *ptp;
    if (g_pt.pending.empty()) return 0;
    const uintptr_t a = (uintptr_t)addr;
    for (const ParamSegment& sgm : g_pt.pending) {
        if (a >= sgm.lo && a < sgm.hi) {
            UH_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, sgm.ev, 0));   // segments complete in order
            return 0;
        }
    }
    return 0;          // not a parameter of the pending step
}

int uniter_params_wait_all(void* stream) {
    std::lock_guard<std::mutex> lk(g_pt_mu);
    ParamTracker* ptp = nullptr;
    { int rc = tracker_get(&ptp); if (rc) return rc; }
    ParamTracker& g_pt = *ptp;
    if (g_pt.pending.empty()) return 0;
    UH_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, g_pt.pending.back().ev, 0));
    g_pt.pending.clear();
    return 0;
}

}  // extern "C"
