// common.cuh — shared device helpers for the gfx950 (CDNA4, wave64) kernels of libuniter_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---------------------------------------------------------------------------------------------
// bf16 <-> fp32 (round-to-nearest-even; the compiler lowers the casts to v_cvt_pk_bf16_f32 on gfx950)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

__device__ __forceinline__ float bits2f_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bits2f_lo(uint32_t w) { return __uint_as_float(w << 16); }

// pack two floats into one dword of 2 x bf16 (lo = a, hi = b)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    bf16x2 t;
    t[0] = (bf16_t)a;
    t[1] = (bf16_t)b;
    return __builtin_bit_cast(uint32_t, t);
}

// 8-byte vector of 4 bf16 <-> 4 floats
__device__ __forceinline__ void unpack4(const u32x2 w, float (&f)[4]) {
    f[0] = bits2f_lo(w[0]); f[1] = bits2f_hi(w[0]);
    f[2] = bits2f_lo(w[1]); f[3] = bits2f_hi(w[1]);
}
__device__ __forceinline__ u32x2 pack4(const float (&f)[4]) {
    u32x2 w;
    w[0] = pack_bf16x2(f[0], f[1]);
    w[1] = pack_bf16x2(f[2], f[3]);
    return w;
}
__device__ __forceinline__ void unpack8(const u32x4 w, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bits2f_lo(w[i]); f[2 * i + 1] = bits2f_hi(w[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return w;
}

// ---------------------------------------------------------------------------------------------
// Global loads of data another CU of the same launch may have just written (persistent per-XCD forward): COH = true
// turns the load into an `nt` load, which the vector L1 never serves (measured, tests/native/xcd_probe.cpp: 0 stale words
// of 2e7 with the L1 holding the previous contents; a plain load was stale 43 % of the time); the XCD's L2 still does.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) u32x4 gmem_u32x4;      // the global address space, named at the access: pointers that
typedef __attribute__((address_space(1))) u32x2 gmem_u32x2;      // went through a function call are generic (flat) otherwise
template <bool COH>
__device__ __forceinline__ u32x4 ldg16(const void* p) {
    if constexpr (COH) return __builtin_nontemporal_load((const gmem_u32x4*)p);
    else return *(const gmem_u32x4*)p;
}
template <bool COH>
__device__ __forceinline__ u32x2 ldg8(const void* p) {
    if constexpr (COH) return __builtin_nontemporal_load((const gmem_u32x2*)p);
    else return *(const gmem_u32x2*)p;
}
// store for a consumer in this launch (COH: default policy, the line stays in the XCD's L2) or in a later one (nt)
template <bool COH>
__device__ __forceinline__ void stg8(void* p, const u32x2 v) {
    if constexpr (COH) *(gmem_u32x2*)p = v;
    else __builtin_nontemporal_store(v, (gmem_u32x2*)p);
}

// Store of a kernel's OUTPUT (read next by a later kernel, possibly on another XCD).  UNITER_STORE_POLICY: 0 = `nt`
// (the line stays dirty in this XCD's L2 until it is evicted or the end-of-kernel write-back flushes it), 1 = `sc1`
// (write-through: the bytes leave the L2 while the kernel is still running), 2 = default policy.
#ifndef UNITER_STORE_POLICY
#define UNITER_STORE_POLICY 0
#endif
__device__ __forceinline__ void out_store16(void* p, const u32x4 v) {
#if UNITER_STORE_POLICY == 1
    // (inline asm: the builtin stores have no cache-policy operand; the s_nop covers the data-register hazard of a 128-bit store)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((gmem_u32x4*)p), "v"(v) : "memory");
#elif UNITER_STORE_POLICY == 2
    *(gmem_u32x4*)p = v;
#else
    __builtin_nontemporal_store(v, (gmem_u32x4*)p);
#endif
}
__device__ __forceinline__ void out_store8(void* p, const u32x2 v) {
#if UNITER_STORE_POLICY == 1
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"((gmem_u32x2*)p), "v"(v) : "memory");
#elif UNITER_STORE_POLICY == 2
    *(gmem_u32x2*)p = v;
#else
    __builtin_nontemporal_store(v, (gmem_u32x2*)p);
#endif
}

// ---------------------------------------------------------------------------------------------
// Overlapped kernel chains (EXPERIMENTS.md section 10; feasibility: tests/native/anyorder_probe.cpp, profiles/r04_anyorder_probe.log)
// ---------------------------------------------------------------------------------------------
// The kernels of an encoder forward / backward call form a dependent chain in which every operation is local to a block of
// rows (tokens) — attention to an example, which at L % 32 == 0 is a whole number of 32-row units.  Launched in order, each
// kernel boundary costs ~1.5-2 us of idle chip (queue barrier, cache write-back / invalidate) plus the ramp of one kernel
// that cannot overlap the drain of the previous one.  A chain kernel is therefore dispatched WITHOUT the barrier packet bit
// (hipExtLaunchKernel + hipExtAnyOrderLaunch: it starts as soon as its predecessor has been dispatched, measured) and orders
// itself by flags: one counter per 32-row unit and kernel; a producer tile adds 1 to each unit it covers when its output has
// left the chip's caches, a consumer tile waits until the units under its rows have received `expect` contributions.
// Deadlock-free by construction: packets of one queue are dispatched in order, so every workgroup a consumer can wait for is
// already resident or done.  Coherence across XCDs without a kernel boundary (measured in the probe: 0 stale words of 3e8):
// producers store write-through (`sc1`) and wait for the acknowledgement (`s_waitcnt vmcnt(0)`) before they signal; consumers
// read with plain loads AFTER seeing the flag — and no buffer of a chain is written twice within a call (encoder.hip gives
// every layer its own), so no cache on the chip can hold an older version of a line a consumer asks for.
// A wait that does not complete within 50 ms sets *status and falls through: a broken chain fails a test, it never hangs the GPU.
struct ChainLink {
    const uint32_t* wait;      // the producer kernel's counters, one per 32-row unit (nullptr: inputs are complete at dispatch)
    uint32_t* signal;          // this kernel's counters (nullptr: consumers are ordered by a normal kernel boundary)
    uint32_t* status;          // device word, set to 1 by a wait that timed out
    uint32_t expect;           // contributions a producer unit receives (the producer's column tiles / heads / row groups)
    uint32_t pad;
};

// all threads of the workgroup; rows [row0, row0 + nrows) of the producer's output are about to be read (nrows <= 2048)
__device__ __forceinline__ void chain_wait(const ChainLink& c, const int row0, const int nrows) {
    if (c.wait == nullptr) return;                         // (uniform)
    if (threadIdx.x < 64) {
        const int u0 = row0 >> 5;
        const int nu = ((row0 + nrows + 31) >> 5) - u0;
        const int lane = (int)threadIdx.x;
        bool ok = lane >= nu;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 0;; ++spins) {
            if (!ok) ok = __hip_atomic_load(c.wait + u0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= c.expect;
            if (__all(ok)) break;
            if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) {      // 50 ms at 100 MHz
                if (lane == 0 && c.status != nullptr) __hip_atomic_store(c.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// all live threads of the workgroup, after their last output store; rows [row0, row0 + nrows) of this kernel's output are final
__device__ __forceinline__ void chain_signal(const ChainLink& c, const int row0, const int nrows) {
    if (c.signal == nullptr) return;                       // (uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my write-through stores have been acknowledged
    __syncthreads();
    const int u0 = row0 >> 5;
    const int nu = ((row0 + nrows + 31) >> 5) - u0;
    if ((int)threadIdx.x < nu) __hip_atomic_fetch_add(c.signal + u0 + (int)threadIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bijective XCD-aware block remap (cdna_hip_programming.md T1): consecutive hardware block ids go to
// different XCDs; give each XCD a contiguous range of logical tiles so neighbours share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + loc;
}

// output stores of a kernel that may run inside a chain: write-through when it signals, the kernel's usual policy otherwise
__device__ __forceinline__ void out_store16c(void* p, const u32x4 v, const bool wt) {
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((gmem_u32x4*)p), "v"(v) : "memory");
    else out_store16(p, v);
}
__device__ __forceinline__ void out_store8c(void* p, const u32x2 v, const bool wt) {
    if (wt) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"((gmem_u32x2*)p), "v"(v) : "memory");
    else out_store8(p, v);
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32 counter-based RNG (Salmon et al., SC'11).  One call gives 4 x 32 random bits for counter (c0..c3) under
// key (k0,k1).  Dropout draws its masks from the 7-round variant (the fewest rounds that pass BigCrush in that paper;
// 10 is the library default with extra margin) and spends 16 bits per element, so ONE call covers 8 consecutive
// elements: key = seed, counter = (element_index/8 lo, hi, site offset lo, hi); element e of the group keeps iff its
// 16-bit field (word e/2, half e%2) >= thresh16 = round(p * 65536), and survivors are scaled by 65536 / (65536 - thresh16)
// (exactly unbiased for the quantised p; p = 0.1 becomes 0.100006).  The multiplies are quarter-rate instructions and
// dominate a call, which is why the call count matters for the VALU-bound attention / LayerNorm kernels.
// ---------------------------------------------------------------------------------------------
constexpr int PHILOX_DROPOUT_ROUNDS = 7;

template <int ROUNDS>
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n1 = lo1;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        const uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    u32x4 out;
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    return out;
}

struct DropoutCfg {
    float    p;          // drop probability (0 => disabled)
    float    scale;      // 65536 / (65536 - thresh)
    uint32_t thresh;     // keep iff the element's 16-bit field >= thresh ; thresh = round(p * 65536)
    uint32_t seed_lo, seed_hi;
    uint32_t off_lo, off_hi;
    // optional device-resident base offset added to (off_hi:off_lo) at run time: lets a captured hipGraph draw fresh
    // masks on every replay (the host-side offset is baked into the graph, the device counter is not)
    const unsigned long long* off_ptr;
};

extern const unsigned long long* uh_drop_offset_ptr;   // set through uniter_hip_set_dropout_offset_ptr (capi.hip)

static inline DropoutCfg make_dropout(float p, uint64_t seed, uint64_t offset) {
    DropoutCfg d;
    d.p = p;
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    if (t < 0 || !(p > 0.f)) t = 0;
    d.thresh = (uint32_t)t;
    d.scale = p > 0.f ? 65536.0f / (65536.0f - (float)d.thresh) : 1.0f;
    d.seed_lo = (uint32_t)seed; d.seed_hi = (uint32_t)(seed >> 32);
    d.off_lo = (uint32_t)offset; d.off_hi = (uint32_t)(offset >> 32);
    d.off_ptr = p > 0.f ? uh_drop_offset_ptr : nullptr;
    return d;
}

__device__ __forceinline__ void dropout_offset(const DropoutCfg& d, uint32_t& lo, uint32_t& hi) {
    unsigned long long o = ((unsigned long long)d.off_hi << 32) | (unsigned long long)d.off_lo;
    if (d.off_ptr != nullptr) o += *d.off_ptr;
    lo = (uint32_t)o;
    hi = (uint32_t)(o >> 32);
}

// the 128 random bits of element group idx8 (= element index / 8)
__device__ __forceinline__ u32x4 dropout_bits8(const DropoutCfg& d, uint64_t idx8) {
    uint32_t olo, ohi;
    dropout_offset(d, olo, ohi);
    return philox4x32<PHILOX_DROPOUT_ROUNDS>((uint32_t)idx8, (uint32_t)(idx8 >> 32), olo, ohi, d.seed_lo, d.seed_hi);
}
__device__ __forceinline__ bool dropout_keep_field(const DropoutCfg& d, const u32x4& r, int e) {   // e = 0..7
    return ((r[e >> 1] >> ((e & 1) << 4)) & 0xffffu) >= d.thresh;
}
// Keep-multipliers (0 or scale) for the 8 consecutive elements of group `idx8`: one Philox call.
__device__ __forceinline__ void dropout_mult8(const DropoutCfg& d, uint64_t idx8, float (&m)[8]) {
    const u32x4 r = dropout_bits8(d, idx8);
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = dropout_keep_field(d, r, e) ? d.scale : 0.f;
}
// Keep-multipliers for the 4 consecutive elements of group `idx4` (= element index / 4): half of a call's bits.
__device__ __forceinline__ void dropout_mult4(const DropoutCfg& d, uint64_t idx4, float (&m)[4]) {
    const u32x4 r = dropout_bits8(d, idx4 >> 1);
    const int h = (int)(idx4 & 1) * 2;
    const uint32_t w0 = h ? r[2] : r[0], w1 = h ? r[3] : r[1];
    m[0] = (w0 & 0xffffu) >= d.thresh ? d.scale : 0.f;
    m[1] = (w0 >> 16) >= d.thresh ? d.scale : 0.f;
    m[2] = (w1 & 0xffffu) >= d.thresh ? d.scale : 0.f;
    m[3] = (w1 >> 16) >= d.thresh ? d.scale : 0.f;
}
// Keep bits (bit e set = element e of the group survives) for the 4 consecutive elements of group `idx4`.
__device__ __forceinline__ uint32_t dropout_keep4(const DropoutCfg& d, uint64_t idx4) {
    const u32x4 r = dropout_bits8(d, idx4 >> 1);
    const int h = (int)(idx4 & 1) * 2;
    const uint32_t w0 = h ? r[2] : r[0], w1 = h ? r[3] : r[1];
    return ((w0 & 0xffffu) >= d.thresh ? 1u : 0u) | ((w0 >> 16) >= d.thresh ? 2u : 0u) |
           ((w1 & 0xffffu) >= d.thresh ? 4u : 0u) | ((w1 >> 16) >= d.thresh ? 8u : 0u);
}
// Keep bits of all 8 elements of group idx8.
__device__ __forceinline__ uint32_t dropout_keep8(const DropoutCfg& d, uint64_t idx8) {
    const u32x4 r = dropout_bits8(d, idx8);
    uint32_t k = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) k |= dropout_keep_field(d, r, e) ? (1u << e) : 0u;
    return k;
}
// Single element `e` (0..3) of group idx4.
__device__ __forceinline__ float dropout_mult1(const DropoutCfg& d, uint64_t idx4, int e) {
    const u32x4 r = dropout_bits8(d, idx4 >> 1);
    return dropout_keep_field(d, r, (int)(idx4 & 1) * 4 + e) ? d.scale : 0.f;
}

// ---------------------------------------------------------------------------------------------
// exact erf GELU (model/layer.py:31-37) and its derivative
// ---------------------------------------------------------------------------------------------
// Phi(x) = 0.5*(1+erf(x/sqrt2)) and phi(x) = exp(-x^2/2)/sqrt(2 pi) from ONE exponential:
// erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1/(1 + p z)  (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7,
// far below the bf16 resolution of the stored result) with z = |x|/sqrt2, so exp(-z^2) = exp(-x^2/2) serves both.
// Two elements at a time: the GEMM epilogues that apply it are VALU-bound on it (a 3072 x 3072 output is 5-6 us of
// nothing but this on every SIMD), so the multiplies and fused multiply-adds are packed (v_pk_fma_f32 / v_pk_mul_f32: two
// elements per issue slot), the reciprocal is the hardware's v_rcp_f32 (1 ulp: __frcp_rn expands to a ten-instruction
// correctly rounded division) and the exponential is one multiply + v_exp_f32.  The scalar entry points run the same
// instruction sequence on a duplicated pair, so an element gets the same bits from every caller.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void normal_cdf_pdf2(const f32x2_t x, f32x2_t& cdf, f32x2_t& pdf) {
    // every multiply-add is spelled out (and nothing else may be contracted): the same bits wherever this is inlined
#pragma clang fp contract(off)
    const f32x2_t ax = __builtin_elementwise_abs(x);
    const f32x2_t h = (ax * ax) * -0.72134752044448170368f;              // -x^2/2 * log2(e)
    f32x2_t e, t;
    e.x = __builtin_amdgcn_exp2f(h.x);
    e.y = __builtin_amdgcn_exp2f(h.y);
    const f32x2_t d = __builtin_elementwise_fma(f32x2_t{0.23164189f, 0.23164189f}, ax, f32x2_t{1.0f, 1.0f});   // p / sqrt2 = 0.3275911 / 1.41421356
    t.x = __builtin_amdgcn_rcpf(d.x);
    t.y = __builtin_amdgcn_rcpf(d.y);
    f32x2_t poly = __builtin_elementwise_fma(t, f32x2_t{1.061405429f, 1.061405429f}, f32x2_t{-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(t, poly, f32x2_t{1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(t, poly, f32x2_t{-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(t, poly, f32x2_t{0.254829592f, 0.254829592f});
    poly = t * poly;
    const f32x2_t tail = (0.5f * poly) * e;                              // = 0.5 * erfc(|x| / sqrt2)
    const f32x2_t upper = 1.0f - tail;
    cdf.x = x.x >= 0.f ? upper.x : tail.x;
    cdf.y = x.y >= 0.f ? upper.y : tail.y;
    pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ f32x2_t gelu_erf2(const f32x2_t x) {
#pragma clang fp contract(off)
    f32x2_t cdf, pdf;
    normal_cdf_pdf2(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ f32x2_t gelu_erf_grad2(const f32x2_t x) {
    f32x2_t cdf, pdf;
    normal_cdf_pdf2(x, cdf, pdf);
    return __builtin_elementwise_fma(x, pdf, cdf);
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2(f32x2_t{x, x}).x; }
__device__ __forceinline__ float gelu_erf_grad(float x) { return gelu_erf_grad2(f32x2_t{x, x}).x; }

// hidden_act of the config (model/layer.py:44 ACT2FN): 0 = gelu (erf form), 1 = relu, 2 = swish (x * sigmoid(x))
enum { UH_ACT_GELU = 0, UH_ACT_RELU = 1, UH_ACT_SWISH = 2 };
__device__ __forceinline__ float act_fwd(int act, float x) {
#pragma clang fp contract(off)
    if (act == UH_ACT_RELU) return fmaxf(x, 0.f);
    if (act == UH_ACT_SWISH) return x * __frcp_rn(1.0f + __expf(-x));
    return gelu_erf(x);
}
__device__ __forceinline__ float act_grad(int act, float x) {
    if (act == UH_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    if (act == UH_ACT_SWISH) { const float s = __frcp_rn(1.0f + __expf(-x)); return s * (1.0f + x * (1.0f - s)); }
    return gelu_erf_grad(x);
}
// the same for two elements (the GEMM epilogues): GELU on the packed path, the other activations element by element
__device__ __forceinline__ f32x2_t act_fwd2(int act, const f32x2_t x) {
    if (act == UH_ACT_GELU) return gelu_erf2(x);
    return f32x2_t{act_fwd(act, x.x), act_fwd(act, x.y)};
}
__device__ __forceinline__ f32x2_t act_grad2(int act, const f32x2_t x) {
    if (act == UH_ACT_GELU) return gelu_erf_grad2(x);
    return f32x2_t{act_grad(act, x.x), act_grad(act, x.y)};
}

// The encoder's FFN1 epilogue saves act'(u) in place of u (UH_ACT_SAVE_GRAD or-ed into the activation code): the normal cdf and
// pdf that GELU needs give the derivative for one more fused multiply-add, and the backward's "x act'(u)" epilogue (FFN2 data
// gradient, model/layer.py:139-142) becomes one multiply per element instead of an exponential, a reciprocal and a degree-5
// polynomial (alone at 3072 x 3072 x 768, profiles/r06_save_act_grad_ab.txt: 29.1 -> 24.6 us with the derivative read, 21.2 without any epilogue operand).  y has the bits of act_fwd2.
enum { UH_ACT_SAVE_GRAD = 0x100, UH_ACT_MASK = 0xff };
__device__ __forceinline__ void act_fwd_grad2(int act, const f32x2_t x, f32x2_t& y, f32x2_t& dy) {
    if (act == UH_ACT_GELU) {
#pragma clang fp contract(off)
        f32x2_t cdf, pdf;
        normal_cdf_pdf2(x, cdf, pdf);
        y = x * cdf;
        dy = __builtin_elementwise_fma(x, pdf, cdf);
        return;
    }
    y = f32x2_t{act_fwd(act, x.x), act_fwd(act, x.y)};
    dy = f32x2_t{act_grad(act, x.x), act_grad(act, x.y)};
}

// ---------------------------------------------------------------------------------------------
// host-side status plumbing for the C ABI
// ---------------------------------------------------------------------------------------------
void uh_set_error(const char* fmt, ...);
#define UH_CHECK_ARG(cond, msg)                                                         \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            uh_set_error("%s: argument error: %s (%s)", __func__, msg, #cond);          \
            return -1;                                                                  \
        }                                                                               \
    } while (0)
#define UH_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            uh_set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(_e));       \
            return (int)_e;                                                             \
        }                                                                               \
    } while (0)
#define UH_LAUNCH_CHECK() UH_CHECK_HIP(hipGetLastError())
