// test_kernels.cpp — native (no Python) GPU check + micro-benchmark of libuniter_hip.so.
//
// Build:  python tests/native/build.py        Run (on a gfx950 box):  tests/native/build/test_kernels [--bench]
// Every check compares the HIP kernel with a plain fp32 host loop on bf16-rounded inputs (the host
// reference lives in THIS file, it is test code).  Exit code = number of failed checks.
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/uniter_hip.h"
#include "../../include/uniter_hip_test.h"

#define HIPCHK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(99);                                                                    \
        }                                                                                \
    } while (0)
#define UHCHK(x)                                                                         \
    do {                                                                                 \
        int r_ = (x);                                                                    \
        if (r_ != 0) {                                                                   \
            fprintf(stderr, "uniter error %d (%s) at %s:%d\n", r_, uniter_hip_last_error(), __FILE__, __LINE__); \
            exit(98);                                                                    \
        }                                                                                \
    } while (0)

static int g_fail = 0;

// ---------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------
static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd32() {
    g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
    return (uint32_t)(g_rng >> 32);
}
static inline float rndu() { return (rnd32() >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f; }   // [-1,1)

struct HostBf {   // host mirror of a bf16 device tensor (values kept as the rounded floats)
    std::vector<float> v;
    std::vector<uint16_t> raw;
    void fill(size_t n, float scale) {
        v.resize(n); raw.resize(n);
        for (size_t i = 0; i < n; ++i) { raw[i] = f2bf(rndu() * scale); v[i] = bf2f(raw[i]); }
    }
};

template <typename T>
static T* dalloc(size_t n) {
    T* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
    return p;
}
static uint16_t* upload(const HostBf& h) {
    uint16_t* d = dalloc<uint16_t>(h.raw.size());
    HIPCHK(hipMemcpy(d, h.raw.data(), h.raw.size() * 2, hipMemcpyHostToDevice));
    return d;
}
static std::vector<float> download_bf(const uint16_t* d, size_t n) {
    std::vector<uint16_t> r(n);
    HIPCHK(hipMemcpy(r.data(), d, n * 2, hipMemcpyDeviceToHost));
    std::vector<float> f(n);
    for (size_t i = 0; i < n; ++i) f[i] = bf2f(r[i]);
    return f;
}
static std::vector<float> download_f(const float* d, size_t n) {
    std::vector<float> f(n);
    HIPCHK(hipMemcpy(f.data(), d, n * 4, hipMemcpyDeviceToHost));
    return f;
}

// compare with mixed tolerance |a-b| <= atol + rtol*|b|
static bool check(const char* name, const std::vector<float>& got, const std::vector<float>& ref, float atol, float rtol) {
    double maxerr = 0, maxref = 0;
    size_t bad = 0, worst = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
        const double e = fabs((double)got[i] - (double)ref[i]);
        if (!(e <= atol + rtol * fabs(ref[i]))) { if (bad == 0 || e > maxerr) worst = i; ++bad; }
        if (e > maxerr || e != e) maxerr = e;
        maxref = std::max(maxref, (double)fabs(ref[i]));
    }
    const bool ok = bad == 0;
    printf("[%s] %-52s n=%zu max|err|=%.3e max|ref|=%.3e bad=%zu", ok ? " OK " : "FAIL", name, ref.size(), maxerr, maxref, bad);
    if (!ok) printf("  (worst idx %zu got %.6f ref %.6f)", worst, got[worst], ref[worst]);
    printf("\n");
    if (!ok) ++g_fail;
    return ok;
}

// host Philox4x32-7 and the dropout field layout, identical to common.cuh
static void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 7; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static float drop_mult(float p, uint64_t seed, uint64_t offset, uint64_t elem) {
    if (p <= 0.f) return 1.f;
    const uint32_t thresh = (uint32_t)std::min((double)p * 65536.0 + 0.5, 65535.0);
    uint32_t r[4];
    const uint64_t idx8 = elem >> 3;
    philox((uint32_t)idx8, (uint32_t)(idx8 >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int e = (int)(elem & 7);
    const uint32_t field = (r[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
    return field >= thresh ? 65536.0f / (65536.0f - (float)thresh) : 0.f;
}
// element index of attention probability (row = (b,h,query), key j) in the dropout stream: one 8-element group holds the
// 4 keys 4g..4g+3 of both tiles of a key-tile pair (attention.hip)
static uint64_t attn_drop_elem(uint64_t row, int j, int L) {
    const int u = j >> 5, hf = (j >> 4) & 1, g = (j & 15) >> 2, r = j & 3;
    const uint64_t pairs = L > 256 ? 16 : 8;      // element groups per query row (attention.hip pair_stride)
    return ((row * pairs + (uint64_t)u) * 4 + (uint64_t)g) * 8 + (uint64_t)(4 * hf + r);
}
static float gelu_h(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
static float gelu_grad_h(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); }
    template <typename F>
    double run(F f, int warm = 3, int iters = 20) {
        for (int i = 0; i < warm; ++i) f();
        HIPCHK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) f();
        HIPCHK(hipEventRecord(b, 0));
        HIPCHK(hipEventSynchronize(b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        return ms * 1000.0 / iters;   // microseconds
    }
};

// ---------------------------------------------------------------------------------------------
// probe kernels: MFMA fragment layout and ds_read_b64_tr_b16 semantics
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void probe_mfma(const uint16_t* A, const uint16_t* B, float* D) {
    // A [16][32] row-major, B [32][16] row-major; hypothesis: lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15];
    // D[4*(l>>4)+r][l&15]
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[i * 32 + 8 * g + j]; b[j] = (short)B[(8 * g + j) * 16 + i]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}
__global__ void probe_tr(const uint16_t* src, uint16_t* out) {
    // LDS block [4][16] row-major; lane i of each 16-lane group gives the address of row (i>>2), cols 4*(i&3)..;
    // hypothesis: lane i receives column i (rows 0..3).  Four groups read four different blocks.
    __shared__ __attribute__((aligned(16))) uint16_t lds[4 * 64];
    const int l = threadIdx.x;
    for (int k = l; k < 256; k += 64) lds[k] = src[k];
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const uint16_t* p = lds + g * 64 + (i >> 2) * 16 + (i & 3) * 4;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

static void run_probes() {
    printf("== probes ==\n");
    {
        HostBf A, B;
        A.fill(16 * 32, 1.f); B.fill(32 * 16, 1.f);
        uint16_t* dA = upload(A); uint16_t* dB = upload(B);
        float* dD = dalloc<float>(256);
        hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        HIPCHK(hipDeviceSynchronize());
        std::vector<float> got = download_f(dD, 256), ref(256);
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                float s = 0;
                for (int k = 0; k < 32; ++k) s += A.v[m * 32 + k] * B.v[k * 16 + n];
                ref[m * 16 + n] = s;
            }
        check("mfma_f32_16x16x32_bf16 fragment layout", got, ref, 1e-4f, 1e-4f);
    }
    {
        std::vector<uint16_t> src(256);
        for (int k = 0; k < 256; ++k) src[k] = (uint16_t)k;
        uint16_t* dS = dalloc<uint16_t>(256);
        uint16_t* dO = dalloc<uint16_t>(256);
        HIPCHK(hipMemcpy(dS, src.data(), 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, dS, dO);
        HIPCHK(hipDeviceSynchronize());
        std::vector<uint16_t> o(256);
        HIPCHK(hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost));
        std::vector<float> got(256), ref(256);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                got[l * 4 + j] = o[l * 4 + j];
                ref[l * 4 + j] = (float)((l >> 4) * 64 + j * 16 + (l & 15));
            }
        if (!check("ds_read_b64_tr_b16 semantics", got, ref, 0.f, 0.f)) {
            printf("   raw dump (lane: 4 values):\n");
            for (int l = 0; l < 64; ++l) printf("   %2d: %3d %3d %3d %3d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GEMM family
// ---------------------------------------------------------------------------------------------
static void host_gemm_nt(const std::vector<float>& X, const std::vector<float>& W, std::vector<float>& Y, int M, int N, int K) {
    Y.assign((size_t)M * N, 0.f);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float s = 0;
            const float* x = &X[(size_t)m * K];
            const float* w = &W[(size_t)n * K];
            for (int k = 0; k < K; ++k) s += x[k] * w[k];
            Y[(size_t)m * N + n] = s;
        }
}

// mirror of kTiles in gemm.hip (tile index -> BM x BN)
static const int kTileBM[] = {128, 128, 64, 64, 96, 192, 192, 128, 96, 128, 96, 96, 192, 128, 64, 64, 96, 96, 96, 128,
                              128, 96, 192, 128, 64, 96, 128, 64, 96, 64,
                              96, 96, 128, 128, 64, 96, 128, 64, 96, 192,
                              96, 128, 192, 128,
                              128, 128, 192, 96, 192, 128, 192, 128, 64, 192,
                              96, 96, 128, 128,
                              256, 192,
                              192, 96, 192};
static const int kTileBN[] = {128, 64, 128, 64, 192, 96, 128, 192, 128, 96, 96, 64, 64, 64, 128, 64, 96, 64, 128, 96,
                              128, 192, 96, 64, 128, 96, 64, 128, 96, 64,
                              64, 128, 96, 128, 64, 64, 64, 128, 96, 64,
                              128, 96, 64, 128,
                              128, 128, 192, 192, 96, 192, 128, 64, 128, 64,
                              128, 128, 96, 96,
                              256, 192,
                              192, 192, 96};
static const int kNumTiles = 63;     // 13..19 are 3-stage rings, 20..29 wave-specialised (4 compute + 4 loader waves), 58 the eight-phase 256x256 tile
static const int kTileG8 = 58, kTileG6 = 59;   // the deep-pipelined 256x256 / 192x192 tiles

static void test_gemm(int M, int N, int K, int cfg, int splits) {
    char tag[128];
    const bool g8 = cfg == kTileG8 || cfg == kTileG6;      // deep-pipelined tiles: contraction % 64 == 0, output columns % 256 == 0, no bias gradient output
    const bool pow2_bn = cfg < 0 || g8 || kTileBN[cfg] == 64 || kTileBN[cfg] == 128 || kTileBN[cfg] == 192;
    const bool pow2_bm = cfg < 0 || g8 || kTileBM[cfg] == 64 || kTileBM[cfg] == 128;
    if (g8 && K % 64 != 0) { printf("  (skip cfg%d for K=%d)\n", cfg, K); return; }
    if (cfg >= 0 && N % kTileBN[cfg] != 0) { printf("  (skip cfg%d for N=%d)\n", cfg, N); return; }
    HostBf X, W, Bv, R, DY;
    X.fill((size_t)M * K, 1.f); W.fill((size_t)N * K, 0.5f); Bv.fill(N, 1.f); R.fill((size_t)M * N, 1.f); DY.fill((size_t)M * N, 1.f);
    uint16_t *dX = upload(X), *dW = upload(W), *dB = upload(Bv), *dR = upload(R), *dDY = upload(DY);
    uint16_t* dY = dalloc<uint16_t>((size_t)M * N);
    uint16_t* dY2 = dalloc<uint16_t>((size_t)M * N);
    uniter_gemm_debug_force(cfg, splits);

    std::vector<float> acc;
    host_gemm_nt(X.v, W.v, acc, M, N, K);
    // --- fwd bias ---
    UHCHK(uniter_gemm_bias_fwd(dX, dW, dB, dY, M, N, K, 0));
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)M * N);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) ref[(size_t)m * N + n] = acc[(size_t)m * N + n] + Bv.v[n];
    snprintf(tag, sizeof tag, "gemm_bias_fwd M%d N%d K%d cfg%d", M, N, K, cfg);
    check(tag, download_bf(dY, (size_t)M * N), ref, 0.02f * sqrtf((float)K) * 0.05f + 0.02f, 0.01f);
    // --- fwd bias + gelu ---
    UHCHK(uniter_gemm_bias_gelu_fwd(dX, dW, dB, dY, dY2, M, N, K, 0));
    HIPCHK(hipDeviceSynchronize());
    {
        std::vector<float> gu = download_bf(dY, (size_t)M * N), gg = download_bf(dY2, (size_t)M * N), rg((size_t)M * N);
        for (size_t i = 0; i < rg.size(); ++i) rg[i] = gelu_h(gu[i]);   // g must be gelu of the stored (rounded) u
        snprintf(tag, sizeof tag, "gemm_bias_gelu_fwd(u) M%d N%d K%d cfg%d", M, N, K, cfg);
        check(tag, gu, ref, 0.05f, 0.01f);
        snprintf(tag, sizeof tag, "gemm_bias_gelu_fwd(g=gelu(u)) cfg%d", cfg);
        check(tag, gg, rg, 0.01f, 0.01f);
        // the encoder's form: the first output is gelu'(u) of the same rounded u, the second output keeps its bits
        uniter_gemm_debug_act_flags(0x100);
        UHCHK(uniter_gemm_bias_gelu_fwd(dX, dW, dB, dY, dY2, M, N, K, 0));
        HIPCHK(hipDeviceSynchronize());
        uniter_gemm_debug_act_flags(0);
        std::vector<float> gd = download_bf(dY, (size_t)M * N), gg2 = download_bf(dY2, (size_t)M * N), rd((size_t)M * N);
        for (size_t i = 0; i < rd.size(); ++i) rd[i] = gelu_grad_h(gu[i]);
        snprintf(tag, sizeof tag, "gemm_bias_gelu_fwd(d=gelu'(u)) cfg%d", cfg);
        check(tag, gd, rd, 0.006f, 0.005f);
        snprintf(tag, sizeof tag, "gemm_bias_gelu_fwd(g, saved-derivative form) bit-identical cfg%d", cfg);
        check(tag, gg2, gg, 0.f, 0.f);
    }
    // --- fwd bias + dropout + residual ---
    for (float p : {0.0f, 0.25f}) {
        const uint64_t seed = 1234, off = 77;
        UHCHK(uniter_gemm_bias_dropout_residual_fwd(dX, dW, dB, dR, dY, M, N, K, p, seed, off, 0));
        HIPCHK(hipDeviceSynchronize());
        std::vector<float> rz((size_t)M * N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                const size_t i = (size_t)m * N + n;
                rz[i] = ref[i] * drop_mult(p, seed, off, i) + R.v[i];
            }
        snprintf(tag, sizeof tag, "gemm_bias_dropout_residual_fwd p=%.2f cfg%d", p, cfg);
        check(tag, download_bf(dY, (size_t)M * N), rz, 0.06f, 0.01f);
    }
    // --- dgrad: dx[M,K] = dy[M,N] * w[N,K] (+resid) ---  (needs K % 64 == 0; the K-strided weight side needs a 64/128 tile)
    if (K % 64 == 0 && pow2_bn && (cfg < 0 || K % kTileBN[cfg] == 0) && (!g8 || N % 64 == 0)) {
        uint16_t* dDX = dalloc<uint16_t>((size_t)M * K);
        HostBf RX, U;
        RX.fill((size_t)M * K, 1.f); U.fill((size_t)M * K, 2.f);
        uint16_t *dRX = upload(RX), *dU = upload(U);
        std::vector<float> dxr((size_t)M * K, 0.f);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                const float d = DY.v[(size_t)m * N + n];
                const float* w = &W.v[(size_t)n * K];
                float* o = &dxr[(size_t)m * K];
                for (int k = 0; k < K; ++k) o[k] += d * w[k];
            }
        UHCHK(uniter_gemm_dgrad(dDY, dW, nullptr, dDX, M, N, K, 0));
        HIPCHK(hipDeviceSynchronize());
        snprintf(tag, sizeof tag, "gemm_dgrad M%d N%d K%d cfg%d", M, N, K, cfg);
        check(tag, download_bf(dDX, (size_t)M * K), dxr, 0.02f * sqrtf((float)N) * 0.05f + 0.03f, 0.01f);
        UHCHK(uniter_gemm_dgrad(dDY, dW, dRX, dDX, M, N, K, 0));
        HIPCHK(hipDeviceSynchronize());
        std::vector<float> r2(dxr);
        for (size_t i = 0; i < r2.size(); ++i) r2[i] += RX.v[i];
        snprintf(tag, sizeof tag, "gemm_dgrad+resid cfg%d", cfg);
        check(tag, download_bf(dDX, (size_t)M * K), r2, 0.05f, 0.01f);
        UHCHK(uniter_gemm_dgrad_gelu(dDY, dW, dU, dDX, M, N, K, 0));
        HIPCHK(hipDeviceSynchronize());
        for (size_t i = 0; i < r2.size(); ++i) r2[i] = dxr[i] * gelu_grad_h(U.v[i]);
        snprintf(tag, sizeof tag, "gemm_dgrad_gelu cfg%d", cfg);
        check(tag, download_bf(dDX, (size_t)M * K), r2, 0.05f, 0.01f);
        uniter_gemm_debug_act_flags(0x100);                      // `u` holds the derivative itself
        UHCHK(uniter_gemm_dgrad_gelu(dDY, dW, dU, dDX, M, N, K, 0));
        HIPCHK(hipDeviceSynchronize());
        uniter_gemm_debug_act_flags(0);
        for (size_t i = 0; i < r2.size(); ++i) r2[i] = dxr[i] * U.v[i];
        snprintf(tag, sizeof tag, "gemm_dgrad_gelu (saved derivative) cfg%d", cfg);
        check(tag, download_bf(dDX, (size_t)M * K), r2, 0.05f, 0.01f);
        HIPCHK(hipFree(dDX)); HIPCHK(hipFree(dRX)); HIPCHK(hipFree(dU));
    }
    // --- wgrad: dw[N,K] (+)= dy^T x ; db = colsum(dy) ---
    if (K % 64 == 0 && pow2_bn && pow2_bm && (cfg < 0 || (N % kTileBM[cfg] == 0 && K % kTileBN[cfg] == 0)) && (!g8 || M % 64 == 0)) {
        const size_t wsb = uniter_gemm_wgrad_workspace_bytes(M, N, K);
        void* ws = dalloc<char>(wsb);
        HostBf Old, OldB;
        Old.fill((size_t)N * K, 1.f); OldB.fill(N, 1.f);
        uint16_t *dDW = upload(Old), *dDB = upload(OldB);
        std::vector<float> dwr((size_t)N * K, 0.f), dbr(N, 0.f);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                const float d = DY.v[(size_t)m * N + n];
                dbr[n] += d;
                const float* x = &X.v[(size_t)m * K];
                float* o = &dwr[(size_t)n * K];
                for (int k = 0; k < K; ++k) o[k] += d * x[k];
            }
        uint16_t* dDBarg = g8 ? nullptr : dDB;
        UHCHK(uniter_gemm_wgrad(dDY, dX, dDW, dDBarg, M, N, K, 0, ws, wsb, 0));
        HIPCHK(hipDeviceSynchronize());
        snprintf(tag, sizeof tag, "gemm_wgrad M%d N%d K%d cfg%d splits%d", M, N, K, cfg, splits);
        const float tol = 0.02f * sqrtf((float)M) + 0.05f;
        check(tag, download_bf(dDW, (size_t)N * K), dwr, tol, 0.01f);
        if (!g8) check("  wgrad bias (colsum)", download_bf(dDB, N), dbr, tol, 0.01f);
        // accumulate on top of the result just written
        std::vector<float> cur = download_bf(dDW, (size_t)N * K), curb = download_bf(dDB, N);
        UHCHK(uniter_gemm_wgrad(dDY, dX, dDW, dDBarg, M, N, K, 1, ws, wsb, 0));
        HIPCHK(hipDeviceSynchronize());
        for (size_t i = 0; i < cur.size(); ++i) cur[i] += dwr[i];
        for (int n = 0; n < N; ++n) curb[n] += dbr[n];
        check("  wgrad accumulate", download_bf(dDW, (size_t)N * K), cur, 2 * tol, 0.02f);
        if (!g8) check("  wgrad bias accumulate", download_bf(dDB, N), curb, 2 * tol, 0.02f);
        HIPCHK(hipFree(ws)); HIPCHK(hipFree(dDW)); HIPCHK(hipFree(dDB));
    }
    uniter_gemm_debug_force(-1, -1);
    HIPCHK(hipFree(dX)); HIPCHK(hipFree(dW)); HIPCHK(hipFree(dB)); HIPCHK(hipFree(dR)); HIPCHK(hipFree(dDY));
    HIPCHK(hipFree(dY)); HIPCHK(hipFree(dY2));
}

// ---------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------
static void test_attention(int B, int L, int heads, float p) {
    const int H = heads * 64;
    const size_t T = (size_t)B * L;
    char tag[128];
    HostBf QKV, DO;
    QKV.fill(T * 3 * H, 1.5f); DO.fill(T * H, 1.f);
    std::vector<float> mask(B * L);
    for (int b = 0; b < B; ++b) {
        const int valid = L - (b * 7) % (L / 2);
        for (int k = 0; k < L; ++k) mask[b * L + k] = k < valid ? 0.f : -10000.f;
    }
    uint16_t *dQKV = upload(QKV), *dDO = upload(DO);
    float* dMask = dalloc<float>(B * L);
    HIPCHK(hipMemcpy(dMask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    uint16_t* dCtx = dalloc<uint16_t>(T * H);
    uint16_t* dDQKV = dalloc<uint16_t>(T * 3 * H);
    float* dLse = dalloc<float>((size_t)B * heads * L);
    HIPCHK(hipMemset(dDQKV, 0xff, T * 3 * H * 2));
    const uint64_t seed = 99, off = 5;
    UHCHK(uniter_attention_fwd(dQKV, dMask, dCtx, dLse, B, L, heads, p, seed, off, 0));
    const size_t awsb = uniter_attention_bwd_workspace_bytes(B, L, heads);
    void* aws = dalloc<char>(awsb + 16);
    UHCHK(uniter_attention_bwd_ws(dQKV, dMask, nullptr, dCtx, dLse, dDO, dDQKV, B, L, heads, p, seed, off, aws, awsb, 0));
    HIPCHK(hipDeviceSynchronize());

    std::vector<float> ctx_ref(T * H, 0.f), lse_ref((size_t)B * heads * L), dqkv_ref(T * 3 * H, 0.f);
    std::vector<float> P((size_t)L * L), Pd((size_t)L * L), dPd((size_t)L * L);
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < heads; ++h) {
            const int bh = b * heads + h;
            auto q = [&](int i, int d) { return QKV.v[((size_t)b * L + i) * 3 * H + h * 64 + d]; };
            auto k = [&](int i, int d) { return QKV.v[((size_t)b * L + i) * 3 * H + H + h * 64 + d]; };
            auto v = [&](int i, int d) { return QKV.v[((size_t)b * L + i) * 3 * H + 2 * H + h * 64 + d]; };
            for (int i = 0; i < L; ++i) {
                float mx = -INFINITY;
                for (int j = 0; j < L; ++j) {
                    float s = 0;
                    for (int d = 0; d < 64; ++d) s += q(i, d) * k(j, d);
                    s = s * 0.125f + mask[b * L + j];
                    P[(size_t)i * L + j] = s;
                    mx = fmaxf(mx, s);
                }
                float sum = 0;
                for (int j = 0; j < L; ++j) { P[(size_t)i * L + j] = expf(P[(size_t)i * L + j] - mx); sum += P[(size_t)i * L + j]; }
                lse_ref[(size_t)bh * L + i] = mx + logf(sum);
                for (int j = 0; j < L; ++j) {
                    P[(size_t)i * L + j] /= sum;
                    const uint64_t elem = attn_drop_elem((uint64_t)bh * L + i, j, L);
                    Pd[(size_t)i * L + j] = P[(size_t)i * L + j] * drop_mult(p, seed, off, elem);
                }
                for (int d = 0; d < 64; ++d) {
                    float o = 0;
                    for (int j = 0; j < L; ++j) o += rbf(Pd[(size_t)i * L + j]) * v(j, d);
                    ctx_ref[((size_t)b * L + i) * H + h * 64 + d] = o;
                }
            }
            // backward
            for (int i = 0; i < L; ++i) {
                float Di = 0;
                for (int d = 0; d < 64; ++d) Di += DO.v[((size_t)b * L + i) * H + h * 64 + d] * rbf(ctx_ref[((size_t)b * L + i) * H + h * 64 + d]);
                for (int j = 0; j < L; ++j) {
                    float dp = 0;
                    for (int d = 0; d < 64; ++d) dp += DO.v[((size_t)b * L + i) * H + h * 64 + d] * v(j, d);
                    const uint64_t elem = attn_drop_elem((uint64_t)bh * L + i, j, L);
                    const float mult = drop_mult(p, seed, off, elem);
                    dPd[(size_t)i * L + j] = P[(size_t)i * L + j] * (dp * mult - Di) * 0.125f;   // dS * scale
                }
            }
            for (int i = 0; i < L; ++i)
                for (int d = 0; d < 64; ++d) {
                    float dq = 0;
                    for (int j = 0; j < L; ++j) dq += dPd[(size_t)i * L + j] * k(j, d);
                    dqkv_ref[((size_t)b * L + i) * 3 * H + h * 64 + d] = dq;
                }
            for (int j = 0; j < L; ++j)
                for (int d = 0; d < 64; ++d) {
                    float dk = 0, dv = 0;
                    for (int i = 0; i < L; ++i) {
                        dk += dPd[(size_t)i * L + j] * q(i, d);
                        dv += Pd[(size_t)i * L + j] * DO.v[((size_t)b * L + i) * H + h * 64 + d];
                    }
                    dqkv_ref[((size_t)b * L + j) * 3 * H + H + h * 64 + d] = dk;
                    dqkv_ref[((size_t)b * L + j) * 3 * H + 2 * H + h * 64 + d] = dv;
                }
        }
    snprintf(tag, sizeof tag, "attention_fwd ctx B%d L%d heads%d p=%.2f", B, L, heads, p);
    check(tag, download_bf(dCtx, T * H), ctx_ref, 0.03f, 0.02f);
    snprintf(tag, sizeof tag, "attention_fwd lse B%d L%d", B, L);
    check(tag, download_f(dLse, (size_t)B * heads * L), lse_ref, 1e-3f, 1e-4f);
    snprintf(tag, sizeof tag, "attention_bwd dqkv B%d L%d heads%d p=%.2f", B, L, heads, p);
    check(tag, download_bf(dDQKV, T * 3 * H), dqkv_ref, 0.06f, 0.03f);
    HIPCHK(hipFree(dQKV)); HIPCHK(hipFree(dDO)); HIPCHK(hipFree(dMask)); HIPCHK(hipFree(dCtx)); HIPCHK(hipFree(dDQKV)); HIPCHK(hipFree(dLse));
}

// ---------------------------------------------------------------------------------------------
// LayerNorm / colsum
// ---------------------------------------------------------------------------------------------
// grouped weight gradients == the individual launches
static void test_wgrad_group(int M) {
    const int n = 4;
    const int64_t N[4] = {128, 256, 128, 384}, K[4] = {256, 128, 128, 128};
    HostBf DY[4], X[4], W0[4], B0[4];
    uint16_t *dDY[4], *dX[4], *dW1[4], *dW2[4], *dB1[4], *dB2[4];
    size_t wsb = 0;
    for (int q = 0; q < n; ++q) {
        DY[q].fill((size_t)M * N[q], 1.f); X[q].fill((size_t)M * K[q], 1.f); W0[q].fill((size_t)N[q] * K[q], 0.5f);
        dDY[q] = upload(DY[q]); dX[q] = upload(X[q]); dW1[q] = upload(W0[q]); dW2[q] = upload(W0[q]);
        B0[q].fill((size_t)N[q], 0.5f); dB1[q] = upload(B0[q]); dB2[q] = upload(B0[q]);
        wsb = std::max(wsb, uniter_gemm_wgrad_workspace_bytes(M, N[q], K[q]));
    }
    void* ws = dalloc<char>(wsb);
    uniter_gemm_debug_force(3, 1);                       // 64x64, no split: the same summation order as the grouped default
    for (int q = 0; q < n; ++q) UHCHK(uniter_gemm_wgrad(dDY[q], dX[q], dW1[q], dB1[q], M, N[q], K[q], 1, ws, wsb, 0));
    uniter_gemm_debug_force(-1, -1);
    const void* dyp[4] = {dDY[0], dDY[1], dDY[2], dDY[3]};
    const void* xp[4] = {dX[0], dX[1], dX[2], dX[3]};
    void* dwp[4] = {dW2[0], dW2[1], dW2[2], dW2[3]};
    void* dbp[4] = {dB2[0], dB2[1], nullptr, dB2[3]};   // member 2 without a bias gradient: its buffer must stay untouched
    UHCHK(uniter_gemm_wgrad_group(n, dyp, nullptr, xp, nullptr, dwp, dbp, M, N, K, 1, 0));
    HIPCHK(hipDeviceSynchronize());
    for (int q = 0; q < n; ++q) {
        char tag[128];
        snprintf(tag, sizeof(tag), "wgrad group member %d (M%d N%lld K%lld, accumulate)", q, M, (long long)N[q], (long long)K[q]);
        check(tag, download_bf(dW2[q], (size_t)N[q] * K[q]), download_bf(dW1[q], (size_t)N[q] * K[q]), 0.13f, 0.02f);
        snprintf(tag, sizeof(tag), "wgrad group member %d bias gradient (column sums of dy, accumulate)", q);
        if (q == 2) check(tag, download_bf(dB2[q], (size_t)N[q]), B0[q].v, 0.f, 0.f);
        else check(tag, download_bf(dB2[q], (size_t)N[q]), download_bf(dB1[q], (size_t)N[q]), 0.13f, 0.02f);
    }
}

// grouped weight gradients on the eight-phase tile (one and two K slices, bias gradients from the appended strips) == the
// individual launches on a 64x64 tile
static void test_wgrad_group_g8(int M, int splits, int tile = kTileG8) {
    const int n = 4;
    const int64_t e = tile == kTileG8 ? 256 : 192;
    const int64_t N[4] = {e, 2 * e, e, 3 * e}, K[4] = {2 * e, e, e, e};
    HostBf DY[4], X[4], W0[4], B0[4];
    uint16_t *dDY[4], *dX[4], *dW1[4], *dW2[4], *dB1[4], *dB2[4];
    size_t wsb = 0;
    for (int q = 0; q < n; ++q) {
        DY[q].fill((size_t)M * N[q], 1.f); X[q].fill((size_t)M * K[q], 1.f); W0[q].fill((size_t)N[q] * K[q], 0.5f);
        dDY[q] = upload(DY[q]); dX[q] = upload(X[q]); dW1[q] = upload(W0[q]); dW2[q] = upload(W0[q]);
        B0[q].fill((size_t)N[q], 0.5f); dB1[q] = upload(B0[q]); dB2[q] = upload(B0[q]);
        wsb = std::max(wsb, uniter_gemm_wgrad_workspace_bytes(M, N[q], K[q]));
    }
    void* ws = dalloc<char>(wsb);
    uniter_gemm_debug_force(3, 1);
    for (int q = 0; q < n; ++q) UHCHK(uniter_gemm_wgrad(dDY[q], dX[q], dW1[q], dB1[q], M, N[q], K[q], 1, ws, wsb, 0));
    uniter_gemm_debug_force(-1, -1);
    const void* dyp[4] = {dDY[0], dDY[1], dDY[2], dDY[3]};
    const void* xp[4] = {dX[0], dX[1], dX[2], dX[3]};
    void* dwp[4] = {dW2[0], dW2[1], dW2[2], dW2[3]};
    void* dbp[4] = {dB2[0], dB2[1], nullptr, dB2[3]};
    const size_t gwb = uniter_gemm_wgrad_group_workspace_bytes(n, N, K);
    void* gws = dalloc<char>(gwb);
    for (int rep = 0; rep < 2; ++rep) {                  // twice: the tile counters must come back to zero
        UHCHK(uniter_gemm_wgrad_group_ws(n, dyp, nullptr, xp, nullptr, dwp, dbp, M, N, K, 1, gws, gwb, tile, splits, 0));
        HIPCHK(hipDeviceSynchronize());
        if (rep == 0) {
            for (int q = 0; q < n; ++q) {
                char tag[160];
                snprintf(tag, sizeof(tag), "deep-pipelined wgrad group, tile %d, %d slice(s), member %d (M%d N%lld K%lld, accumulate)", tile, splits, q, M, (long long)N[q], (long long)K[q]);
                check(tag, download_bf(dW2[q], (size_t)N[q] * K[q]), download_bf(dW1[q], (size_t)N[q] * K[q]), 0.13f, 0.02f);
                snprintf(tag, sizeof(tag), "eight-phase wgrad group member %d bias gradient (appended column-sum strips)", q);
                if (q == 2) check(tag, download_bf(dB2[q], (size_t)N[q]), B0[q].v, 0.f, 0.f);
                else check(tag, download_bf(dB2[q], (size_t)N[q]), download_bf(dB1[q], (size_t)N[q]), 0.13f, 0.02f);
            }
        }
    }
    // second pass accumulated once more on top: 2x the gradient + the start value
    for (int q = 0; q < n; ++q) UHCHK(uniter_gemm_wgrad(dDY[q], dX[q], dW1[q], dB1[q], M, N[q], K[q], 1, ws, wsb, 0));   // (tuned/default tile)
    HIPCHK(hipDeviceSynchronize());
    for (int q = 0; q < n; ++q) {
        char tag[160];
        snprintf(tag, sizeof(tag), "eight-phase wgrad group, %d slice(s), member %d, second accumulation", splits, q);
        check(tag, download_bf(dW2[q], (size_t)N[q] * K[q]), download_bf(dW1[q], (size_t)N[q] * K[q]), 0.3f, 0.03f);
    }
    for (int q = 0; q < n; ++q) { HIPCHK(hipFree(dDY[q])); HIPCHK(hipFree(dX[q])); HIPCHK(hipFree(dW1[q])); HIPCHK(hipFree(dW2[q])); HIPCHK(hipFree(dB1[q])); HIPCHK(hipFree(dB2[q])); }
    HIPCHK(hipFree(ws)); HIPCHK(hipFree(gws));
}

static void test_layernorm(int rows, int H, float p, int post) {
    char tag[128];
    HostBf Z, G, Bt, DY;
    Z.fill((size_t)rows * H, 2.f); G.fill(H, 1.f); Bt.fill(H, 1.f); DY.fill((size_t)rows * H, 1.f);
    for (auto& g : G.v) g += 1.0f;
    for (size_t i = 0; i < G.v.size(); ++i) { G.raw[i] = f2bf(G.v[i]); G.v[i] = bf2f(G.raw[i]); }
    uint16_t *dZ = upload(Z), *dG = upload(G), *dB = upload(Bt), *dDY = upload(DY);
    uint16_t* dY = dalloc<uint16_t>((size_t)rows * H);
    uint16_t* dDZ = dalloc<uint16_t>((size_t)rows * H);
    uint16_t* dDD = dalloc<uint16_t>((size_t)rows * H);
    uint16_t *dDG = dalloc<uint16_t>(H), *dDB = dalloc<uint16_t>(H), *dDBias = dalloc<uint16_t>(H);
    float *dMean = dalloc<float>(rows), *dRstd = dalloc<float>(rows);
    const float eps = 1e-12f;
    const uint64_t seed = 7, off = 3;
    UHCHK(uniter_layernorm_fwd(dZ, dG, dB, dY, dMean, dRstd, rows, H, eps, post ? p : 0.f, seed, off, 0));
    const size_t wsb = uniter_layernorm_bwd_workspace_bytes(rows, H);
    void* ws = dalloc<char>(wsb);
    UHCHK(uniter_layernorm_bwd(dDY, nullptr, dZ, dMean, dRstd, dG, dDZ, dDD, dDG, dDB, dDBias, rows, H, 0, p, seed, off, post, ws, wsb, 0));
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> yr((size_t)rows * H), dzr((size_t)rows * H), ddr((size_t)rows * H), dgr(H, 0.f), dbr(H, 0.f), dbias(H, 0.f);
    for (int r = 0; r < rows; ++r) {
        double mean = 0, var = 0;
        for (int c = 0; c < H; ++c) mean += Z.v[(size_t)r * H + c];
        mean /= H;
        for (int c = 0; c < H; ++c) { const double d = Z.v[(size_t)r * H + c] - mean; var += d * d; }
        var /= H;
        const float rstd = (float)(1.0 / sqrt(var + eps));
        double c1 = 0, c2 = 0;
        std::vector<float> xh(H), gy(H), dyv(H);
        for (int c = 0; c < H; ++c) {
            const size_t i = (size_t)r * H + c;
            xh[c] = (float)((Z.v[i] - mean) * rstd);
            float y = xh[c] * G.v[c] + Bt.v[c];
            if (post && p > 0) y = rbf(y) * drop_mult(p, seed, off, i);
            yr[i] = y;
            dyv[c] = DY.v[i] * (post ? drop_mult(p, seed, off, i) : 1.f);
            gy[c] = dyv[c] * G.v[c];
            c1 += gy[c]; c2 += gy[c] * xh[c];
            dgr[c] += dyv[c] * xh[c]; dbr[c] += dyv[c];
        }
        c1 /= H; c2 /= H;
        for (int c = 0; c < H; ++c) {
            const size_t i = (size_t)r * H + c;
            dzr[i] = rstd * (gy[c] - (float)c1 - xh[c] * (float)c2);
            ddr[i] = rbf(dzr[i]) * (post ? 1.f : drop_mult(p, seed, off, i));
            dbias[c] += ddr[i];
        }
    }
    snprintf(tag, sizeof tag, "layernorm_fwd rows%d H%d p=%.2f post=%d", rows, H, p, post);
    check(tag, download_bf(dY, (size_t)rows * H), yr, 0.03f, 0.01f);
    snprintf(tag, sizeof tag, "layernorm_bwd dz rows%d H%d", rows, H);
    check(tag, download_bf(dDZ, (size_t)rows * H), dzr, 0.03f, 0.02f);
    if (p > 0 && !post) check("  layernorm_bwd dd (dropout-masked)", download_bf(dDD, (size_t)rows * H), ddr, 0.03f, 0.02f);
    const float tol = 0.02f * sqrtf((float)rows) + 0.05f;
    check("  layernorm_bwd dgamma", download_bf(dDG, H), dgr, tol, 0.01f);
    check("  layernorm_bwd dbeta", download_bf(dDB, H), dbr, tol, 0.01f);
    if (!post) check("  layernorm_bwd dbias", download_bf(dDBias, H), dbias, tol, 0.01f);
    HIPCHK(hipFree(dZ)); HIPCHK(hipFree(dG)); HIPCHK(hipFree(dB)); HIPCHK(hipFree(dDY)); HIPCHK(hipFree(dY)); HIPCHK(hipFree(dDZ));
    HIPCHK(hipFree(dDD)); HIPCHK(hipFree(dDG)); HIPCHK(hipFree(dDB)); HIPCHK(hipFree(dDBias)); HIPCHK(hipFree(dMean)); HIPCHK(hipFree(dRstd)); HIPCHK(hipFree(ws));
}

// ---------------------------------------------------------------------------------------------
// AdamW
// ---------------------------------------------------------------------------------------------
static void test_adamw() {
    const int NT = 3;
    const int64_t numel[NT] = {4096 * 3 + 5, 768, 100003};
    std::vector<UniterAdamTensor> tab(NT);
    std::vector<std::vector<float>> hp(NT), hg(NT), hm(NT), hv(NT);
    std::vector<HostBf> hgbf(NT);
    for (int t = 0; t < NT; ++t) {
        const size_t n = (size_t)numel[t];
        hp[t].resize(n); hm[t].assign(n, 0.f); hv[t].assign(n, 0.f);
        for (auto& x : hp[t]) x = rndu();
        const bool bf = (t != 1);
        float* master = dalloc<float>(n);
        HIPCHK(hipMemcpy(master, hp[t].data(), n * 4, hipMemcpyHostToDevice));
        float *m = dalloc<float>(n), *v = dalloc<float>(n);
        HIPCHK(hipMemset(m, 0, n * 4)); HIPCHK(hipMemset(v, 0, n * 4));
        tab[t].numel = numel[t]; tab[t].group = t == 2 ? 1 : 0; tab[t].param_is_bf16 = bf;
        tab[t].exp_avg = m; tab[t].exp_avg_sq = v;
        if (bf) {
            hgbf[t].fill(n, 0.1f);
            hg[t] = hgbf[t].v;
            tab[t].grad = upload(hgbf[t]);
            tab[t].param = dalloc<uint16_t>(n);
            tab[t].master = master;
        } else {
            hg[t].resize(n);
            for (auto& x : hg[t]) x = rndu() * 0.1f;
            float* g = dalloc<float>(n);
            HIPCHK(hipMemcpy(g, hg[t].data(), n * 4, hipMemcpyHostToDevice));
            tab[t].grad = g; tab[t].param = master; tab[t].master = nullptr;
        }
    }
    void* plan = nullptr;
    UHCHK(uniter_adamw_plan_create(tab.data(), NT, &plan));
    float* dnorm = dalloc<float>(2);
    UniterAdamGroup groups[2] = {{3e-3f, 0.9f, 0.98f, 1e-6f, 0.01f, 1, 1}, {1e-3f, 0.9f, 0.98f, 1e-6f, 0.0f, 1, 1}};
    const float max_norm = 2.0f, gscale = 0.5f;
    double sq = 0;
    for (int t = 0; t < NT; ++t) for (float g : hg[t]) sq += (double)g * g;
    const float norm = (float)sqrt(sq) * gscale;
    float coef = gscale;
    if (max_norm / (norm + 1e-6f) < 1.f) coef = gscale * max_norm / (norm + 1e-6f);
    for (int step = 1; step <= 2; ++step) {
        groups[0].step = groups[1].step = step;
        UHCHK(uniter_adamw_grad_norm(plan, gscale, max_norm, dnorm, 0));
        UHCHK(uniter_adamw_step(plan, groups, 2, dnorm + 1, 0));
        for (int t = 0; t < NT; ++t) {
            const UniterAdamGroup& g = groups[tab[t].group];
            const double ss = g.lr * sqrt(1.0 - pow((double)g.beta2, step)) / (1.0 - pow((double)g.beta1, step));
            for (size_t i = 0; i < hp[t].size(); ++i) {
                const float ge = hg[t][i] * coef;
                hm[t][i] = g.beta1 * hm[t][i] + (1.f - g.beta1) * ge;
                hv[t][i] = g.beta2 * hv[t][i] + (1.f - g.beta2) * ge * ge;
                hp[t][i] = hp[t][i] - (float)ss * (hm[t][i] / (sqrtf(hv[t][i]) + g.eps));
                if (g.weight_decay > 0) hp[t][i] = hp[t][i] - g.lr * g.weight_decay * hp[t][i];
            }
        }
    }
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> nr = download_f(dnorm, 2);
    check("adamw grad_norm / clip coef", nr, std::vector<float>{norm, coef}, 1e-4f, 1e-4f);
    for (int t = 0; t < NT; ++t) {
        char tag[64];
        const size_t n = (size_t)numel[t];
        snprintf(tag, sizeof tag, "adamw tensor %d fp32 state (p)", t);
        check(tag, download_f(tab[t].param_is_bf16 ? tab[t].master : (float*)tab[t].param, n), hp[t], 1e-6f, 1e-5f);
        snprintf(tag, sizeof tag, "adamw tensor %d exp_avg_sq", t);
        check(tag, download_f(tab[t].exp_avg_sq, n), hv[t], 1e-9f, 1e-5f);
        if (tab[t].param_is_bf16) {
            std::vector<float> r(n);
            for (size_t i = 0; i < n; ++i) r[i] = rbf(hp[t][i]);
            snprintf(tag, sizeof tag, "adamw tensor %d bf16 copy", t);
            check(tag, download_bf((uint16_t*)tab[t].param, n), r, 1e-2f, 1e-2f);
        }
    }
    UHCHK(uniter_adamw_plan_destroy(plan));
}

// ---------------------------------------------------------------------------------------------
// benchmark: kernels at the UNITER-base 60+36 / batch-32 shapes, and the whole 12-layer encoder
// ---------------------------------------------------------------------------------------------
static void bench(int B, int L, int H, int heads, int I, int layers) {
    printf("== bench B%d L%d H%d I%d layers%d ==\n", B, L, H, I, layers);
    const int64_t T = (int64_t)B * L;
    Timer tm;
    HostBf X, W, Bv;
    X.fill((size_t)T * I, 1.f); W.fill((size_t)I * H * 3, 0.05f); Bv.fill(I, 0.1f);
    uint16_t *dX = upload(X), *dW = upload(W), *dB = upload(Bv);
    uint16_t* dY = dalloc<uint16_t>((size_t)T * I);
    uint16_t* dY2 = dalloc<uint16_t>((size_t)T * I);
    uint16_t* dG = dalloc<uint16_t>((size_t)I * H * 3);
    const size_t wsb = (size_t)8 * I * H * 4 * 3;
    void* ws = dalloc<char>(wsb);
    struct Shape { const char* name; int64_t M, N, K; } shapes[] = {
        {"qkv   ", T, 3 * (int64_t)H, H}, {"out   ", T, H, H}, {"ffn1  ", T, I, H}, {"ffn2  ", T, H, I}};
    for (auto& s : shapes) {
        const double fl = 2.0 * s.M * s.N * s.K;
        for (int cfg = -1; cfg < kNumTiles; ++cfg) {
            const int bm = cfg < 0 ? 0 : kTileBM[cfg], bn = cfg < 0 ? 0 : kTileBN[cfg];
            const bool p2n = cfg < 0 || bn == 64 || bn == 128 || bn == 192, p2m = cfg < 0 || bm == 64 || bm == 128;
            printf("  %s cfg%2d %3dx%3d", s.name, cfg, bm, bn);
            if (cfg < 0 || s.N % bn == 0) {
                uniter_gemm_debug_force(cfg, -1);
                double t1 = tm.run([&] { UHCHK(uniter_gemm_bias_fwd(dX, dW, dB, dY, s.M, s.N, s.K, 0)); });
                printf("  fwd %6.1f us %6.1f TF", t1, fl / t1 * 1e-6);
            } else printf("  fwd      -            ");
            if (p2n && (cfg < 0 || s.K % bn == 0)) {
                uniter_gemm_debug_force(cfg, -1);
                double t2 = tm.run([&] { UHCHK(uniter_gemm_dgrad(dY, dW, nullptr, dY2, s.M, s.N, s.K, 0)); });
                printf(" | dgrad %6.1f us %6.1f TF", t2, fl / t2 * 1e-6);
            } else printf(" | dgrad      -            ");
            if (p2n && p2m && (cfg < 0 || (s.N % bm == 0 && s.K % bn == 0))) {
                for (int sp : {1, 2, 4}) {
                    uniter_gemm_debug_force(cfg, cfg < 0 ? -1 : sp);
                    double t3 = tm.run([&] { UHCHK(uniter_gemm_wgrad(dY, dX, dG, nullptr, s.M, s.N, s.K, 1, ws, wsb, 0)); });
                    printf(" | wgrad s%d %6.1f us %5.1f TF", cfg < 0 ? -1 : sp, t3, fl / t3 * 1e-6);
                    if (cfg < 0) break;
                }
            }
            printf("\n");
        }
    }
    uniter_gemm_debug_force(-1, -1);
    {
        double t1 = tm.run([&] { UHCHK(uniter_gemm_bias_gelu_fwd(dX, dW, dB, dY, dY2, T, I, H, 0)); });
        double t2 = tm.run([&] { UHCHK(uniter_gemm_bias_dropout_residual_fwd(dX, dW, dB, dY2, dY, T, H, I, 0.1f, 1, 2, 0)); });
        double t3 = tm.run([&] { UHCHK(uniter_gemm_dgrad_gelu(dY, dW, dX, dY2, T, H, I, 0)); });
        printf("  ffn1+gelu %.1f us | ffn2+drop+res %.1f us | dgrad_gelu %.1f us\n", t1, t2, t3);
    }
    {
        float* dMask = dalloc<float>((size_t)B * L);
        HIPCHK(hipMemset(dMask, 0, (size_t)B * L * 4));
        float* dLse = dalloc<float>((size_t)B * heads * L);
        for (float p : {0.f, 0.1f}) {
            double t1 = tm.run([&] { UHCHK(uniter_attention_fwd(dX, dMask, dY, dLse, B, L, heads, p, 1, 2, 0)); });
            double t2 = tm.run([&] { UHCHK(uniter_attention_bwd(dX, dMask, dY, dLse, dY2, dW, B, L, heads, p, 1, 2, 0)); });
            const double fl = 4.0 * T * L * H;
            printf("  attention p=%.1f fwd %.1f us (%.1f TF) | bwd %.1f us (%.1f TF)\n", p, t1, fl / t1 * 1e-6, t2, 2.5 * fl / t2 * 1e-6);
        }
        float *dMean = dalloc<float>(T), *dRstd = dalloc<float>(T);
        double t1 = tm.run([&] { UHCHK(uniter_layernorm_fwd(dX, dB, dB, dY, dMean, dRstd, T, H, 1e-12f, 0.f, 0, 0, 0)); });
        const size_t lws = uniter_layernorm_bwd_workspace_bytes(T, H);
        double t2 = tm.run([&] { UHCHK(uniter_layernorm_bwd(dX, nullptr, dY, dMean, dRstd, dB, dY2, dY2 + T * H, dG, dG + H, dG + 2 * H, T, H, 1, 0.1f, 1, 2, 0, ws, lws, 0)); });
        double t3 = tm.run([&] { UHCHK(uniter_colsum(dX, dG, T, I, 1, ws, wsb, 0)); });
        printf("  layernorm fwd %.1f us (%.0f GB/s) | bwd %.1f us (%.0f GB/s) | colsum[T,I] %.1f us (%.0f GB/s)\n", t1,
               2.0 * T * H * 2 / t1 * 1e-3, t2, 4.0 * T * H * 2 / t2 * 1e-3, t3, (double)T * I * 2 / t3 * 1e-3);
    }
    // whole encoder
    {
        UniterEncoderShape sh{B, L, H, heads, I, 0.1f, 0.1f, 1e-12f, 1};
        const size_t act = uniter_encoder_layer_act_bytes(&sh), scr = uniter_encoder_scratch_bytes(&sh);
        char* acts = dalloc<char>(act * layers);
        char* scratch = dalloc<char>(scr);
        const size_t per = (size_t)3 * H * H + 3 * H + (size_t)H * H + H + 2 * H + (size_t)I * H + I + (size_t)H * I + H + 2 * H;
        HostBf P;
        P.fill(per, 0.03f);
        std::vector<UniterLayerParams> lp(layers);
        const bool share_w = getenv("UNITER_BENCH_SHARE_WEIGHTS") != nullptr;   // experiment: every layer reads the same (cache-hot) weights
        uint16_t* p_shared = share_w ? upload(P) : nullptr;
        for (int l = 0; l < layers; ++l) {
            uint16_t* p = share_w ? p_shared : upload(P);
            uint16_t* g = dalloc<uint16_t>(per);
            HIPCHK(hipMemset(g, 0, per * 2));
            size_t o = 0;
            auto nxt = [&](size_t n) { size_t r = o; o += n; return r; };
            size_t o_wqkv = nxt((size_t)3 * H * H), o_bqkv = nxt(3 * H), o_wo = nxt((size_t)H * H), o_bo = nxt(H), o_g1 = nxt(H), o_b1n = nxt(H);
            size_t o_w1 = nxt((size_t)I * H), o_b1 = nxt(I), o_w2 = nxt((size_t)H * I), o_b2 = nxt(H), o_g2 = nxt(H), o_b2n = nxt(H);
            lp[l] = UniterLayerParams{p + o_wqkv, p + o_bqkv, p + o_wo, p + o_bo, p + o_g1, p + o_b1n, p + o_w1, p + o_b1, p + o_w2, p + o_b2, p + o_g2, p + o_b2n,
                                      g + o_wqkv, g + o_bqkv, g + o_wo, g + o_bo, g + o_g1, g + o_b1n, g + o_w1, g + o_b1, g + o_w2, g + o_b2, g + o_g2, g + o_b2n};
        }
        float* dMask = dalloc<float>((size_t)B * L);
        HIPCHK(hipMemset(dMask, 0, (size_t)B * L * 4));
        uint16_t* dDx = dalloc<uint16_t>((size_t)T * H);
        {
            double tf0 = tm.run([&] { UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0)); }, 2, 10);
            double tb0 = tm.run([&] { UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0)); }, 2, 10);
            printf("  (cost-model tiles: fwd %.1f us, bwd %.1f us)\n", tf0, tb0);
            UHCHK(uniter_encoder_debug_tune_in_situ(0));
            UHCHK(uniter_encoder_autotune(&sh, 0));
            double tf1 = tm.run([&] { UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0)); }, 2, 10);
            double tb1 = tm.run([&] { UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0)); }, 2, 10);
            printf("  (isolated-sweep tiles: fwd %.1f us, bwd %.1f us)\n", tf1, tb1);
            UHCHK(uniter_encoder_debug_tune_in_situ(1));
            UHCHK(uniter_encoder_autotune(&sh, 0));
            if (const char* ov = getenv("UNITER_BENCH_SET_TUNED")) {   // experiment: "kind,N,K,cfg;kind,N,K,cfg;..." overrides after tuning
                std::string sv(ov);
                size_t pos = 0;
                while (pos < sv.size()) {
                    size_t end = sv.find(';', pos);
                    if (end == std::string::npos) end = sv.size();
                    int kd = 0, cf = 0; long long nn = 0, kk = 0;
                    if (sscanf(sv.substr(pos, end - pos).c_str(), "%d,%lld,%lld,%d", &kd, &nn, &kk, &cf) == 4) {
                        UHCHK(uniter_gemm_set_tuned(kd, T, nn, kk, cf, 1));
                        printf("  (override: kind %d N%lld K%lld -> config %d)\n", kd, nn, kk, cf);
                    }
                    pos = end + 1;
                }
            }
            const int64_t shp[4][2] = {{3 * (int64_t)H, H}, {H, H}, {I, H}, {H, I}};
            const char* kn[3] = {"fwd", "dgrad", "wgrad"};
            for (int kind = 0; kind < 3; ++kind)
                for (int gi = 0; gi < 4; ++gi) {
                    int32_t ch[2];
                    UHCHK(uniter_gemm_tuned_choice(kind, T, shp[gi][0], shp[gi][1], ch));
                    printf("  autotune %-5s N%lld K%lld -> tile %dx%d splits %d\n", kn[kind], (long long)shp[gi][0], (long long)shp[gi][1],
                           ch[0] >= 0 ? kTileBM[ch[0]] : -1, ch[0] >= 0 ? kTileBN[ch[0]] : -1, ch[1]);
                }
        }
        {
            int32_t ch[2];
            UHCHK(uniter_gemm_tuned_choice(3, T, 5 * (int64_t)H + I, 3 * (int64_t)H + I, ch));
            printf("  autotune grouped wgrad -> tile %dx%d (config %d)\n", ch[0] >= 0 ? kTileBM[ch[0]] : -1, ch[0] >= 0 ? kTileBN[ch[0]] : -1, ch[0]);
        }
        // headline numbers: best of 5 repeats of 20 passes each (single 10-pass averages move by +-3 % on one box)
        double tf = 1e30, tb = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            tf = std::min(tf, tm.run([&] { UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0)); }, 2, 20));
            tb = std::min(tb, tm.run([&] { UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0)); }, 2, 20));
        }
        uniter_encoder_debug_side_stream(0);
        double tb0 = tm.run([&] { UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0)); }, 2, 10);
        uniter_encoder_debug_side_stream(1);
        printf("  (backward with the wgrad side stream disabled: %.1f us)\n", tb0);
        {   // in-situ per-launch durations (HIP events around every launch) of one forward + backward
            static const char* kinds[] = {"gemm fwd +bias", "gemm fwd +gelu", "gemm fwd +drop+res", "gemm dgrad", "gemm dgrad gelu'", "gemm wgrad",
                                          "attn fwd", "attn bwd", "ln fwd", "ln bwd rows", "colsum", "adamw", "ln bwd cols", "gemm wgrad group"};
            const int nkinds = (int)(sizeof(kinds) / sizeof(kinds[0]));
            UniterTimingRecord rec[64];
            int32_t nrec = 0;
            HIPCHK(hipDeviceSynchronize());
            UHCHK(uniter_hip_timing_begin());
            UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0));
            UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0));
            UHCHK(uniter_hip_timing_end(rec, 64, &nrec));
            for (int i = 0; i < nrec && i < 64; ++i)
                printf("  in-situ %-18s M%-5lld N%-5lld K%-5lld x%-3d avg %7.2f us\n", rec[i].kind >= 0 && rec[i].kind < nkinds ? kinds[rec[i].kind] : "?", (long long)rec[i].M,
                       (long long)rec[i].N, (long long)rec[i].K, rec[i].calls, rec[i].total_us / rec[i].calls);
        }
        {   // experiment: two half batches on two streams (phases of the two kernel chains are not aligned)
            UniterEncoderShape sh2 = sh;
            sh2.B = B / 2;
            UHCHK(uniter_encoder_autotune(&sh2, 0));
            const size_t act2 = uniter_encoder_layer_act_bytes(&sh2);
            char* actsA = dalloc<char>(act2 * layers);
            char* actsB = dalloc<char>(act2 * layers);
            hipStream_t sA, sB;
            HIPCHK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
            HIPCHK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
            const size_t half_x = (size_t)(T / 2) * H;
            auto both = [&] {
                UHCHK(uniter_encoder_forward(&sh2, lp.data(), 0, layers, dX, dMask, actsA, scratch, 1, 0, sA));
                UHCHK(uniter_encoder_forward(&sh2, lp.data(), 0, layers, dX + half_x, dMask + (B / 2) * L, actsB, scratch, 1, 0, sB));
            };
            for (int i = 0; i < 3; ++i) both();
            HIPCHK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < 10; ++i) both();
            HIPCHK(hipDeviceSynchronize());
            auto t1 = std::chrono::high_resolution_clock::now();
            const double us2 = std::chrono::duration<double, std::micro>(t1 - t0).count() / 10;
            double one = tm.run([&] { UHCHK(uniter_encoder_forward(&sh2, lp.data(), 0, layers, dX, dMask, actsA, scratch, 1, 0, 0)); }, 2, 10);
            printf("  (experiment: forward of two B=%d halves on two streams: %.1f us; one half alone: %.1f us; full batch: %.1f us)\n", B / 2, us2, one, tf);
        }
        {   // word-embedding scatter-add of a 32 x 60 batch (uniter_embed_txt_bwd, word table only)
            const int Bq = 32, Lt = 60, V = 28996;
            std::vector<int64_t> ids((size_t)Bq * Lt), pos(Lt);
            for (auto& v : ids) v = 1000 + (int64_t)(rndu() * 0.5f * 27000 + 13500) % 27000;
            for (int b2 = 0; b2 < Bq; ++b2) { ids[(size_t)b2 * Lt] = 101; ids[(size_t)b2 * Lt + Lt - 1] = 102; }
            for (int t2 = 0; t2 < Lt; ++t2) pos[t2] = t2;
            int64_t* dIds = dalloc<int64_t>(ids.size()); int64_t* dPos = dalloc<int64_t>(Lt);
            HIPCHK(hipMemcpy(dIds, ids.data(), ids.size() * 8, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(dPos, pos.data(), Lt * 8, hipMemcpyHostToDevice));
            uint16_t* dzq = dalloc<uint16_t>((size_t)Bq * Lt * H);
            HIPCHK(hipMemset(dzq, 0x3c, (size_t)Bq * Lt * H * 2));
            uint16_t* gword = dalloc<uint16_t>((size_t)V * H);
            HIPCHK(hipMemset(gword, 0, (size_t)V * H * 2));
            double ts = tm.run([&] { UHCHK(uniter_embed_txt_bwd(dIds, dPos, nullptr, dzq, gword, nullptr, nullptr, Bq, Lt, H, V, 512, 2, 0)); }, 3, 20);
            printf("  embedding scatter-add (32 x 60 ids -> [28996, %d] gradient): %.1f us\n", H, ts);
        }
        {   // grouped weight gradients of one layer vs the four separate launches
            const int64_t Ng[4] = {H, I, H, 3 * (int64_t)H}, Kg[4] = {I, H, H, H};
            HostBf big; big.fill((size_t)T * I, 1.f);
            uint16_t* dyb[4]; uint16_t* xb[4]; uint16_t* dwb[4];
            for (int q = 0; q < 4; ++q) {
                dyb[q] = dalloc<uint16_t>((size_t)T * Ng[q]); xb[q] = dalloc<uint16_t>((size_t)T * Kg[q]); dwb[q] = dalloc<uint16_t>((size_t)Ng[q] * Kg[q]);
                HIPCHK(hipMemset(dyb[q], 0x3c, (size_t)T * Ng[q] * 2)); HIPCHK(hipMemset(xb[q], 0x3c, (size_t)T * Kg[q] * 2));
                HIPCHK(hipMemset(dwb[q], 0, (size_t)Ng[q] * Kg[q] * 2));
            }
            const size_t wsb2 = uniter_gemm_wgrad_workspace_bytes(T, I, H);
            void* ws2 = dalloc<char>(wsb2);
            const void* dyp[4] = {dyb[0], dyb[1], dyb[2], dyb[3]};
            const void* xp[4] = {xb[0], xb[1], xb[2], xb[3]};
            void* dwp[4] = {dwb[0], dwb[1], dwb[2], dwb[3]};
            uint16_t* dbb[4];
            for (int q = 0; q < 4; ++q) { dbb[q] = dalloc<uint16_t>((size_t)Ng[q]); HIPCHK(hipMemset(dbb[q], 0, (size_t)Ng[q] * 2)); }
            void* dbp[4] = {dbb[0], dbb[1], dbb[2], dbb[3]};
            double sep = tm.run([&] { for (int q = 0; q < 4; ++q) UHCHK(uniter_gemm_wgrad(dyb[q], xb[q], dwb[q], nullptr, T, Ng[q], Kg[q], 1, ws2, wsb2, 0)); }, 3, 20);
            double grp = tm.run([&] { UHCHK(uniter_gemm_wgrad_group(4, dyp, nullptr, xp, nullptr, dwp, nullptr, T, Ng, Kg, 1, 0)); }, 3, 20);
            double grb = tm.run([&] { UHCHK(uniter_gemm_wgrad_group(4, dyp, nullptr, xp, nullptr, dwp, dbp, T, Ng, Kg, 1, 0)); }, 3, 20);
            printf("  (four weight gradients of a layer: separate tuned launches %.1f us, one grouped launch %.1f us, grouped with the four bias gradients %.1f us)\n", sep, grp, grb);
        }
        const double flf = (double)layers * (24.0 * T * H * H + 4.0 * T * L * H);
        printf("  ENCODER fwd %.1f us (%.1f TF) | bwd %.1f us (%.1f TF) | fwd+bwd %.1f us = %.1f TF = %.1f%% of 2.5 PF ; %.0f ex/s\n", tf,
               flf / tf * 1e-6, tb, 2 * flf / tb * 1e-6, tf + tb, 3 * flf / (tf + tb) * 1e-6, 3 * flf / (tf + tb) * 1e-6 / 2500 * 100,
               B / ((tf + tb) * 1e-6));
    }
}


// ---------------------------------------------------------------------------------------------
// --enc: the 12-layer encoder alone with the shipped tile table (uniter_amd/tuned/gfx950.json), no sweeps: the quick
// A/B harness for scheduling experiments (UNITER_AMD_GROUP_PERSIST, UNITER_AMD_WGRAD_DEFER, ...)
// ---------------------------------------------------------------------------------------------
static int load_tuned_json(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    std::string txt;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) txt.append(buf, n);
    fclose(f);
    int cnt = 0;
    size_t pos = 0;
    while ((pos = txt.find("\"kind\"", pos)) != std::string::npos) {        // scanf's blanks match any run of white space (either json.dump layout)
        int kind = 0, cfg = 0, sp = 1;
        long long M = 0, N = 0, K = 0;
        if (sscanf(txt.c_str() + pos, "\"kind\" : %d , \"M\" : %lld , \"N\" : %lld , \"K\" : %lld , \"cfg\" : %d , \"splits\" : %d", &kind, &M, &N, &K, &cfg, &sp) == 6) {
            if (uniter_gemm_set_tuned(kind, M, N, K, cfg, sp) == 0) ++cnt;
        }
        ++pos;
    }
    return cnt;
}

// Deferred weight gradients (uniter_encoder_set_wgrad_stage: one launch for all layers of a backward call, also as two calls
// that alternate halves of a double-size stage) against the per-layer grouped launches: every parameter gradient of every
// layer and the input gradient, same inputs, dropout on, gradients zeroed before each run.
static void test_deferred_wgrad(int B, int L, int H, int heads, int I, int layers) {
    const int64_t T = (int64_t)B * L;
    HostBf X, DYh;
    X.fill((size_t)T * H, 1.f); DYh.fill((size_t)T * H, 1.f);
    uint16_t *dX = upload(X), *dY = upload(DYh);
    UniterEncoderShape sh{B, L, H, heads, I, 0.1f, 0.1f, 1e-12f, 1};
    const size_t act = uniter_encoder_layer_act_bytes(&sh), scr = uniter_encoder_scratch_bytes(&sh);
    char* acts = dalloc<char>(act * layers);
    char* scratch = dalloc<char>(scr);
    const size_t per = (size_t)3 * H * H + 3 * H + (size_t)H * H + H + 2 * H + (size_t)I * H + I + (size_t)H * I + H + 2 * H;
    HostBf P;
    P.fill(per, 0.05f);
    std::vector<UniterLayerParams> lp(layers);
    std::vector<uint16_t*> gbase(layers);
    for (int l = 0; l < layers; ++l) {
        uint16_t* p = upload(P);
        uint16_t* g = dalloc<uint16_t>(per);
        gbase[l] = g;
        size_t o = 0;
        auto nxt = [&](size_t n) { size_t r = o; o += n; return r; };
        size_t o_wqkv = nxt((size_t)3 * H * H), o_bqkv = nxt(3 * H), o_wo = nxt((size_t)H * H), o_bo = nxt(H), o_g1 = nxt(H), o_b1n = nxt(H);
        size_t o_w1 = nxt((size_t)I * H), o_b1 = nxt(I), o_w2 = nxt((size_t)H * I), o_b2 = nxt(H), o_g2 = nxt(H), o_b2n = nxt(H);
        lp[l] = UniterLayerParams{p + o_wqkv, p + o_bqkv, p + o_wo, p + o_bo, p + o_g1, p + o_b1n, p + o_w1, p + o_b1, p + o_w2, p + o_b2, p + o_g2, p + o_b2n,
                                  g + o_wqkv, g + o_bqkv, g + o_wo, g + o_bo, g + o_g1, g + o_b1n, g + o_w1, g + o_b1, g + o_w2, g + o_b2, g + o_g2, g + o_b2n};
    }
    float* dMask = dalloc<float>((size_t)B * L);
    HIPCHK(hipMemset(dMask, 0, (size_t)B * L * 4));
    uint16_t* dDx = dalloc<uint16_t>((size_t)T * H);
    uint16_t* dMid = dalloc<uint16_t>((size_t)T * H);
    const size_t stb = uniter_encoder_wgrad_stage_bytes(&sh, layers);
    char* stage = dalloc<char>(2 * stb);
    UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 7, 3, 0));
    const size_t out_off = uniter_encoder_layer_out_offset(&sh);
    std::vector<std::vector<float>> ref(layers);
    std::vector<float> ref_dx;
    const char* names[3] = {"per-layer grouped launches (no stage)", "one deferred launch for the whole call", "two calls alternating halves of the stage"};
    for (int pass = 0; pass < 3; ++pass) {
        for (int l = 0; l < layers; ++l) HIPCHK(hipMemset(gbase[l], 0, per * 2));
        if (pass == 0) UHCHK(uniter_encoder_set_wgrad_stage(nullptr, 0));
        if (pass == 1) UHCHK(uniter_encoder_set_wgrad_stage(stage, stb));
        if (pass < 2) {
            UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 7, 3, 0));
        } else {                                            // the stack cut in two ranges, as a gradient-bucket hook does
            const int cut = layers / 2;
            const size_t half_bytes = 2 * uniter_encoder_wgrad_stage_bytes(&sh, layers - cut);
            UHCHK(uniter_encoder_set_wgrad_stage(stage, half_bytes));
            UHCHK(uniter_encoder_backward(&sh, lp.data(), cut, layers, acts + (size_t)(cut - 1) * act + out_off, dMask, dY, dMid, acts, scratch, 7, 3, 0));
            UHCHK(uniter_encoder_set_wgrad_stage(stage, 2 * uniter_encoder_wgrad_stage_bytes(&sh, cut)));
            UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, cut, dX, dMask, dMid, dDx, acts, scratch, 7, 3, 0));
        }
        HIPCHK(hipDeviceSynchronize());
        size_t nbad = 0;
        double maxd = 0, maxr = 0;
        for (int l = 0; l < layers; ++l) {
            std::vector<float> got = download_bf(gbase[l], per);
            if (pass == 0) { ref[l] = got; continue; }
            for (size_t k = 0; k < per; ++k) maxr = std::max(maxr, fabs((double)ref[l][k]));
            for (size_t k = 0; k < per; ++k) {
                const double d = fabs((double)got[k] - ref[l][k]);
                maxd = std::max(maxd, d);
                if (!(d <= 0.02 * fabs(ref[l][k]) + 0.004 * maxr + 1e-6)) {
                    if (nbad < 12 || (nbad % 397) == 0) {
                        const size_t seg[13] = {0, (size_t)3 * H * H, (size_t)3 * H * H + 3 * H, (size_t)3 * H * H + 3 * H + (size_t)H * H, (size_t)3 * H * H + 3 * H + (size_t)H * H + H,
                                                (size_t)3 * H * H + 3 * H + (size_t)H * H + 2 * H, (size_t)3 * H * H + 3 * H + (size_t)H * H + 3 * H,
                                                (size_t)3 * H * H + 3 * H + (size_t)H * H + 3 * H + (size_t)I * H, (size_t)3 * H * H + 3 * H + (size_t)H * H + 3 * H + (size_t)I * H + I,
                                                (size_t)3 * H * H + 3 * H + (size_t)H * H + 3 * H + (size_t)I * H + I + (size_t)H * I, (size_t)3 * H * H + 3 * H + (size_t)H * H + 3 * H + (size_t)I * H + I + (size_t)H * I + H,
                                                (size_t)3 * H * H + 3 * H + (size_t)H * H + 3 * H + (size_t)I * H + I + (size_t)H * I + 2 * H, per};
                        const char* sn[12] = {"wqkv", "bqkv", "wo", "bo", "ln1.g", "ln1.b", "w1", "b1", "w2", "b2", "ln2.g", "ln2.b"};
                        int q = 0;
                        while (q < 11 && k >= seg[q + 1]) ++q;
                        printf("    layer %d %s[%zu]: got %.4f ref %.4f\n", l, sn[q], k - seg[q], got[k], ref[l][k]);
                    }
                    ++nbad;
                }
            }
        }
        std::vector<float> dxv = download_bf(dDx, (size_t)T * H);
        if (pass == 0) { ref_dx = dxv; continue; }
        size_t ndx = 0;
        for (size_t k = 0; k < dxv.size(); ++k) if (dxv[k] != ref_dx[k]) ++ndx;          // the data-gradient chain is untouched: bit-identical
        printf("[%s] deferred weight gradients, %s (B%d L%d H%d I%d, %d layers) == %s: max |d| %.4g of max |ref| %.4g, %zu outside tolerance; dx differs in %zu elements\n",
               (nbad || ndx) ? "FAIL" : " OK ", names[pass], B, L, H, I, layers, names[0], maxd, maxr, nbad, ndx);
        if (nbad || ndx) ++g_fail;
    }
    UHCHK(uniter_encoder_set_wgrad_stage(nullptr, 0));
    for (int l = 0; l < layers; ++l) { HIPCHK(hipFree(gbase[l])); HIPCHK(hipFree((void*)lp[l].wqkv)); }
    HIPCHK(hipFree(acts)); HIPCHK(hipFree(scratch)); HIPCHK(hipFree(dX)); HIPCHK(hipFree(dY)); HIPCHK(hipFree(dMask)); HIPCHK(hipFree(dDx));
    HIPCHK(hipFree(dMid)); HIPCHK(hipFree(stage));
}

static void bench_encoder(int B, int L, int H, int heads, int I, int layers) {
    printf("== encoder B%d L%d H%d I%d layers%d ==\n", B, L, H, I, layers);
    const int64_t T = (int64_t)B * L;
    Timer tm;
    HostBf X;
    X.fill((size_t)T * H, 1.f);
    uint16_t* dX = upload(X);
    uint16_t* dY = upload(X);
    const float pdrop = getenv("UNITER_BENCH_NODROP") ? 0.f : 0.1f;
    UniterEncoderShape sh{B, L, H, heads, I, pdrop, pdrop, 1e-12f, 1};
    const size_t act = uniter_encoder_layer_act_bytes(&sh), scr = uniter_encoder_scratch_bytes(&sh);
    char* acts = dalloc<char>(act * layers);
    char* scratch = dalloc<char>(scr);
    const size_t per = (size_t)3 * H * H + 3 * H + (size_t)H * H + H + 2 * H + (size_t)I * H + I + (size_t)H * I + H + 2 * H;
    HostBf P;
    P.fill(per, 0.03f);
    std::vector<UniterLayerParams> lp(layers);
    std::vector<uint16_t*> gbase(layers);
    const bool share_w = getenv("UNITER_BENCH_SHARE_WEIGHTS") != nullptr;   // experiment: every layer reads the same (cache-hot) weights
    uint16_t* p_shared = share_w ? upload(P) : nullptr;
    for (int l = 0; l < layers; ++l) {
        uint16_t* p = share_w ? p_shared : upload(P);
        uint16_t* g = dalloc<uint16_t>(per);
        gbase[l] = g;
        HIPCHK(hipMemset(g, 0, per * 2));
        size_t o = 0;
        auto nxt = [&](size_t n) { size_t r = o; o += n; return r; };
        size_t o_wqkv = nxt((size_t)3 * H * H), o_bqkv = nxt(3 * H), o_wo = nxt((size_t)H * H), o_bo = nxt(H), o_g1 = nxt(H), o_b1n = nxt(H);
        size_t o_w1 = nxt((size_t)I * H), o_b1 = nxt(I), o_w2 = nxt((size_t)H * I), o_b2 = nxt(H), o_g2 = nxt(H), o_b2n = nxt(H);
        lp[l] = UniterLayerParams{p + o_wqkv, p + o_bqkv, p + o_wo, p + o_bo, p + o_g1, p + o_b1n, p + o_w1, p + o_b1, p + o_w2, p + o_b2, p + o_g2, p + o_b2n,
                                  g + o_wqkv, g + o_bqkv, g + o_wo, g + o_bo, g + o_g1, g + o_b1n, g + o_w1, g + o_b1, g + o_w2, g + o_b2, g + o_g2, g + o_b2n};
    }
    float* dMask = dalloc<float>((size_t)B * L);
    HIPCHK(hipMemset(dMask, 0, (size_t)B * L * 4));
    uint16_t* dDx = dalloc<uint16_t>((size_t)T * H);
    const char* tj = getenv("UNITER_TUNED_JSON");
    const int nt = load_tuned_json(tj ? tj : "uniter_amd/tuned/gfx950.json");
    printf("  tile table: %d entries\n", nt);
    if (nt == 0) { UHCHK(uniter_encoder_autotune(&sh, 0)); }
    if (!getenv("UNITER_BENCH_SKIP_CHAIN_CHECK")) {
        // the overlapped kernel chain (no queue barrier between dependent kernels, row-block flags) against the in-order launches:
        // every saved activation of every layer, bit for bit — same kernels, same arithmetic, only the dispatch differs
        std::vector<unsigned char> ref(act * layers), got(act * layers);
        UHCHK(uniter_encoder_debug_chain(0));
        HIPCHK(hipMemset(acts, 0, act * layers));
        UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(ref.data(), acts, ref.size(), hipMemcpyDeviceToHost));
        UHCHK(uniter_encoder_debug_chain(1));
        size_t bad = 0, first = 0;
        int32_t st_word = 0;
        for (int rep = 0; rep < 4; ++rep) {                 // (several launches: a race would not show every time)
            HIPCHK(hipMemset(acts, 0, act * layers));
            UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0));
            int32_t w = 0;
            UHCHK(uniter_encoder_chain_status(&sh, scratch, &w));
            st_word |= w;
            HIPCHK(hipMemcpy(got.data(), acts, got.size(), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < ref.size(); ++k) if (ref[k] != got[k]) { if (!bad) first = k; ++bad; }
        }
        printf("[%s] overlapped-chain forward == in-order forward (4 launches): %zu of %zu bytes differ, status word %d", (bad || st_word) ? "FAIL" : " OK ", bad, 4 * ref.size(), st_word);
        if (bad) printf(" (first at layer %zu, offset %zu of %zu)", first / act, first % act, act);
        if (bad || st_word) ++g_fail;
        printf("\n");
        if (bad) {          // which saved tensor of which layer (the block layout of encoder.hip: 256-byte aligned fields in this order)
            static const char* names[] = {"qkv", "lse", "ctx", "z1", "mean1", "rstd1", "a", "u", "g", "z2", "mean2", "rstd2", "y"};
            const size_t sizes[] = {(size_t)T * 3 * H * 2, (size_t)B * heads * L * 4, (size_t)T * H * 2, (size_t)T * H * 2, (size_t)T * 4, (size_t)T * 4, (size_t)T * H * 2,
                                    (size_t)T * I * 2, (size_t)T * I * 2, (size_t)T * H * 2, (size_t)T * 4, (size_t)T * 4, (size_t)T * H * 2};
            for (int l = 0; l < layers && l < 2; ++l) {
                size_t o = 0;
                for (int f = 0; f < 13; ++f) {
                    size_t nb = 0, fb = 0;
                    for (size_t k = 0; k < sizes[f]; ++k) if (ref[l * act + o + k] != got[l * act + o + k]) { if (!nb) fb = k; ++nb; }
                    if (nb) printf("      layer %d %-6s %zu of %zu bytes differ, first at byte %zu\n", l, names[f], nb, sizes[f], fb);
                    o += (sizes[f] + 255) & ~(size_t)255;
                }
            }
        }
        double t0 = 1e30, t1 = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            UHCHK(uniter_encoder_debug_chain(0));
            t0 = std::min(t0, tm.run([&] { UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0)); }, 2, 20));
            UHCHK(uniter_encoder_debug_chain(1));
            t1 = std::min(t1, tm.run([&] { UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0)); }, 2, 20));
        }
        printf("  forward: in-order launches %.1f us | overlapped chain %.1f us\n", t0, t1);
    }
    if (getenv("UNITER_BENCH_CHAIN_ONLY")) return;
    // deferred weight gradients (one launch for all layers of the call) against the per-layer grouped launches: every
    // parameter gradient of every layer, same inputs, gradients zeroed before each run
    const size_t stb = uniter_encoder_wgrad_stage_bytes(&sh, layers);
    char* stage = stb ? dalloc<char>(stb) : nullptr;
    if (stage != nullptr && !getenv("UNITER_BENCH_NO_STAGE")) {
        UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0));
        std::vector<std::vector<float>> ref(layers);
        for (int pass = 0; pass < 2; ++pass) {
            for (int l = 0; l < layers; ++l) HIPCHK(hipMemset(gbase[l], 0, per * 2));
            UHCHK(uniter_encoder_set_wgrad_stage(pass == 0 ? nullptr : stage, pass == 0 ? 0 : stb));
            UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0));
            HIPCHK(hipDeviceSynchronize());
            size_t nbad = 0;
            double maxd = 0, maxr = 0;
            for (int l = 0; l < layers; ++l) {
                std::vector<float> got = download_bf(gbase[l], per);
                if (pass == 0) { ref[l] = got; continue; }
                for (size_t k = 0; k < per; ++k) {
                    const double d = fabs((double)got[k] - ref[l][k]);
                    maxd = std::max(maxd, d); maxr = std::max(maxr, fabs((double)ref[l][k]));
                    if (!(d <= 0.02 * fabs(ref[l][k]) + 0.004 * maxr + 1e-6)) ++nbad;
                }
            }
            if (pass == 1) {
                printf("[%s] deferred weight gradients (one launch, %d layers) == per-layer grouped launches: max |d| %.4g of max |ref| %.4g, %zu of %zu outside tolerance\n",
                       nbad ? "FAIL" : " OK ", layers, maxd, maxr, nbad, per * (size_t)layers);
                if (nbad) ++g_fail;
            }
        }
        for (int l = 0; l < layers; ++l) HIPCHK(hipMemset(gbase[l], 0, per * 2));
        // the backward data-gradient chain overlapped (row-block flags) against in-order launches, stage registered in both: dx and
        // every parameter gradient of every layer, bit for bit
        std::vector<std::vector<uint16_t>> gref(layers);
        std::vector<uint16_t> dxref((size_t)T * H), dxgot((size_t)T * H), ggot(per);
        size_t bad = 0;
        int32_t st_word = 0;
        for (int pass = 0; pass < 4; ++pass) {              // pass 0: in order; 1..3: chained
            UHCHK(uniter_encoder_debug_chain(pass == 0 ? 0 : 1));
            for (int l = 0; l < layers; ++l) HIPCHK(hipMemset(gbase[l], 0, per * 2));
            HIPCHK(hipMemset(dDx, 0, (size_t)T * H * 2));
            UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0));
            HIPCHK(hipDeviceSynchronize());
            if (pass > 0) { int32_t w = 0; UHCHK(uniter_encoder_chain_status(&sh, scratch, &w)); st_word |= w; }
            HIPCHK(hipMemcpy(pass == 0 ? dxref.data() : dxgot.data(), dDx, (size_t)T * H * 2, hipMemcpyDeviceToHost));
            if (pass > 0) for (size_t k = 0; k < dxref.size(); ++k) bad += dxref[k] != dxgot[k];
            for (int l = 0; l < layers; ++l) {
                if (pass == 0) { gref[l].resize(per); HIPCHK(hipMemcpy(gref[l].data(), gbase[l], per * 2, hipMemcpyDeviceToHost)); continue; }
                HIPCHK(hipMemcpy(ggot.data(), gbase[l], per * 2, hipMemcpyDeviceToHost));
                for (size_t k = 0; k < per; ++k) bad += gref[l][k] != ggot[k];
            }
        }
        printf("[%s] overlapped-chain backward == in-order backward (3 launches): %zu elements of dx / parameter gradients differ, status word %d\n",
               (bad || st_word) ? "FAIL" : " OK ", bad, st_word);
        if (bad || st_word) ++g_fail;
        for (int l = 0; l < layers; ++l) HIPCHK(hipMemset(gbase[l], 0, per * 2));
    }
    if (getenv("UNITER_BENCH_NO_STAGE")) UHCHK(uniter_encoder_set_wgrad_stage(nullptr, 0));
    double tf = 1e30, tb = 1e30, tf0 = 1e30, tb0 = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        UHCHK(uniter_encoder_debug_chain(0));
        tf0 = std::min(tf0, tm.run([&] { UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0)); }, 2, 20));
        tb0 = std::min(tb0, tm.run([&] { UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0)); }, 2, 20));
        UHCHK(uniter_encoder_debug_chain(1));
        tf = std::min(tf, tm.run([&] { UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0)); }, 2, 20));
        tb = std::min(tb, tm.run([&] { UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0)); }, 2, 20));
    }
    printf("  in-order launches: fwd %.1f us | bwd %.1f us ;  overlapped chains (opt-in): fwd %.1f us | bwd %.1f us\n", tf0, tb0, tf, tb);
    UHCHK(uniter_encoder_debug_chain(0));                  // the library's default from here on
    tf = tf0; tb = tb0;                                     // the ENCODER line below is the default path
    {
        static const char* kinds[] = {"gemm fwd +bias", "gemm fwd +gelu", "gemm fwd +drop+res", "gemm dgrad", "gemm dgrad gelu'", "gemm wgrad",
                                      "attn fwd", "attn bwd", "ln fwd", "ln bwd rows", "colsum", "adamw", "ln bwd cols", "gemm wgrad group"};
        const int nkinds = (int)(sizeof(kinds) / sizeof(kinds[0]));
        UniterTimingRecord rec[64];
        int32_t nrec = 0;
        HIPCHK(hipDeviceSynchronize());
        UHCHK(uniter_hip_timing_begin());
        UHCHK(uniter_encoder_forward(&sh, lp.data(), 0, layers, dX, dMask, acts, scratch, 1, 0, 0));
        UHCHK(uniter_encoder_backward(&sh, lp.data(), 0, layers, dX, dMask, dY, dDx, acts, scratch, 1, 0, 0));
        UHCHK(uniter_hip_timing_end(rec, 64, &nrec));
        for (int i = 0; i < nrec && i < 64; ++i)
            printf("  in-situ %-18s M%-5lld N%-5lld K%-5lld x%-3d avg %7.2f us\n", rec[i].kind >= 0 && rec[i].kind < nkinds ? kinds[rec[i].kind] : "?", (long long)rec[i].M,
                   (long long)rec[i].N, (long long)rec[i].K, rec[i].calls, rec[i].total_us / rec[i].calls);
    }
    const double flf = (double)layers * (24.0 * T * H * H + 4.0 * T * L * H);
    printf("  ENCODER fwd %.1f us (%.1f TF) | bwd %.1f us (%.1f TF) | fwd+bwd %.1f us = %.1f TF = %.1f%% of 2.5 PF ; %.0f ex/s\n", tf,
           flf / tf * 1e-6, tb, 2 * flf / tb * 1e-6, tf + tb, 3 * flf / (tf + tb) * 1e-6, 3 * flf / (tf + tb) * 1e-6 / 2500 * 100,
           B / ((tf + tb) * 1e-6));
}

#ifdef UNITER_GEMM_PROBE
extern "C" int uniter_gemm_debug_probe(unsigned long long* dev);
#endif
// --one <fwd|gelu|dgrad|wgrad> M N K cfg splits iters : launch one GEMM flavour repeatedly (for rocprofv3 --pmc runs)
static int run_one(int argc, char** argv, int at) {
    if (at + 7 > argc) { fprintf(stderr, "usage: --one kind M N K cfg splits iters\n"); return 2; }
    const std::string kind = argv[at];
    const int64_t M = atoll(argv[at + 1]), N = atoll(argv[at + 2]), K = atoll(argv[at + 3]);
    const int cfg = atoi(argv[at + 4]), splits = atoi(argv[at + 5]), iters = atoi(argv[at + 6]);
    HostBf X, W, Bv;
    const size_t big = (size_t)std::max(M, std::max(N, K));
    X.fill((size_t)M * big, 1.f); W.fill((size_t)N * big, 0.05f); Bv.fill(big, 0.1f);
    uint16_t *dX = upload(X), *dW = upload(W), *dB = upload(Bv);
    uint16_t* dY = dalloc<uint16_t>((size_t)M * big);
    uint16_t* dY2 = dalloc<uint16_t>((size_t)M * big);
    const size_t wsb = uniter_gemm_wgrad_workspace_bytes(M, N, K);
    void* ws = dalloc<char>(wsb);
    uniter_gemm_debug_force(cfg, splits);
    Timer tm;
    auto fn = [&] {
        if (kind == "fwd") UHCHK(uniter_gemm_bias_fwd(dX, dW, dB, dY, M, N, K, 0));
        else if (kind == "gelu") UHCHK(uniter_gemm_bias_gelu_fwd(dX, dW, dB, dY, dY2, M, N, K, 0));
        else if (kind == "dgrad") UHCHK(uniter_gemm_dgrad(dX, dW, nullptr, dY, M, N, K, 0));
        else UHCHK(uniter_gemm_wgrad(dX, dY, dW, nullptr, M, N, K, 1, ws, wsb, 0));
    };
#ifdef UNITER_GEMM_PROBE
    {
        const int nblk = 4096;
        const size_t pr_total = (size_t)nblk * 64 * 5 + (size_t)nblk * 2 + (size_t)nblk * 64 * 4;
        unsigned long long* dpr = dalloc<unsigned long long>(pr_total);
        for (int i = 0; i < 3; ++i) fn();
        HIPCHK(hipMemset(dpr, 0, pr_total * 8));
        uniter_gemm_debug_probe(dpr);
        fn();
        HIPCHK(hipDeviceSynchronize());
        uniter_gemm_debug_probe(nullptr);
        std::vector<unsigned long long> h(pr_total);
        HIPCHK(hipMemcpy(h.data(), dpr, h.size() * 8, hipMemcpyDeviceToHost));
        const int nkt = (int)std::min<int64_t>((kind == "wgrad" ? M : (kind == "dgrad" ? N : K)) / 64, 63);
        double ph[4] = {0, 0, 0, 0}, tot = 0, gap = 0; long cnt = 0, gcnt = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < nblk; ++b) {
            if (h[(size_t)b * 320] == 0) continue;
            for (int k = 0; k < nkt; ++k) {
                const unsigned long long* r = &h[((size_t)b * 64 + k) * 5];
                if (r[4] == 0) continue;
                for (int q = 0; q < 4; ++q) ph[q] += (double)(r[q + 1] - r[q]);
                tot += (double)(r[4] - r[0]); ++cnt;
                if (k + 1 < nkt && r[9] != 0) { gap += (double)(r[5] - r[4]); ++gcnt; }
                tmin = std::min(tmin, r[0]); tmax = std::max(tmax, r[4]);
            }
        }
        printf("   probe: %ld compute-wave iterations; cycles/iter: mma step 0 (+ issue of step-1 reads) %.0f | lgkm + barrier wait %.0f | mma step 1 (+ next tile reads) %.0f | total %.0f (+%.0f between)\n",
               cnt, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, tot / cnt, gcnt ? gap / gcnt : 0.0);
        {   // loader wave 0
            const unsigned long long* L = &h[(size_t)nblk * 64 * 5 + (size_t)nblk * 2];
            double lw[3] = {0, 0, 0}, ltot = 0; long lc = 0;
            for (int b = 0; b < nblk; ++b)
                for (int k = 1; k + 1 < nkt; ++k) {          // steady state: skip the first and last tile
                    const unsigned long long* r = &L[((size_t)b * 64 + k) * 4];
                    const unsigned long long* rn = &L[((size_t)b * 64 + k + 1) * 4];
                    if (!r[0] || !r[3] || !rn[0]) continue;
                    lw[0] += (double)(r[1] - r[0]); lw[1] += (double)(r[2] - r[1]); lw[2] += (double)(r[3] - r[2]);
                    ltot += (double)(rn[0] - r[0]); ++lc;
                }
            if (lc) printf("   probe: %ld loader-wave iterations; cycles/iter: wait for DMA %.0f | wait at barrier %.0f | issue next tile %.0f | total %.0f\n",
                           lc, lw[0] / lc, lw[1] / lc, lw[2] / lc, ltot / lc);
        }
        // per-block span distribution
        std::vector<double> spans;
        for (int b = 0; b < nblk; ++b) {
            const unsigned long long* r0 = &h[(size_t)b * 320];
            const unsigned long long* r1 = &h[((size_t)b * 64 + nkt - 1) * 5];
            if (r0[0] && r1[4]) spans.push_back((double)(r1[4] - r0[0]));
        }
        std::sort(spans.begin(), spans.end());
        if (!spans.empty()) printf("   probe: main-loop span per block: min %.0f median %.0f max %.0f cycles (%zu blocks)\n", spans.front(), spans[spans.size() / 2], spans.back(), spans.size());
        {   // workgroup life cycle: entry -> first tile usable -> main loop end -> epilogue stored
            const unsigned long long* life = &h[(size_t)nblk * 64 * 5];
            unsigned long long t0 = ~0ull, t1 = 0;
            std::vector<double> pro, epi, startoff, whole;
            for (int b = 0; b < nblk; ++b) if (life[b * 2] && life[b * 2 + 1]) { t0 = std::min(t0, life[b * 2]); t1 = std::max(t1, life[b * 2 + 1]); }
            for (int b = 0; b < nblk; ++b) {
                if (!life[b * 2] || !life[b * 2 + 1]) continue;
                const unsigned long long* r0 = &h[(size_t)b * 320];
                const unsigned long long* r1 = &h[((size_t)b * 64 + nkt - 1) * 5];
                if (!r0[1] || !r1[4]) continue;
                pro.push_back((double)(r0[1] - life[b * 2]));
                epi.push_back((double)(life[b * 2 + 1] - r1[4]));
                startoff.push_back((double)(life[b * 2] - t0));
                whole.push_back((double)(life[b * 2 + 1] - life[b * 2]));
            }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
            auto mx = [](std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
            printf("   probe: workgroup life (cycles): prologue median %.0f max %.0f | epilogue median %.0f max %.0f | whole median %.0f | entry offset median %.0f max %.0f | kernel first-entry..last-exit %.0f\n",
                   med(pro), mx(pro), med(epi), mx(epi), med(whole), med(startoff), mx(startoff), (double)(t1 - t0));
        }
    }
#endif
    const double us = tm.run(fn, 3, iters);
    printf("%s M%lld N%lld K%lld cfg%d splits%d: %.2f us  %.1f TF\n", kind.c_str(), (long long)M, (long long)N, (long long)K, cfg,
           splits, us, 2.0 * M * N * K / us * 1e-6);
    return 0;
}

// --g8: the eight-phase 256x256 tile (cfg 58) at full size: agreement with the 128x128 tile, a race screen (repeated
// launches must be bit-identical), timings against the tuned choice of the older tile family, and the grouped weight
// gradients of a layer with one / two K slices
static int run_g8(int argc, char** argv, int at) {
    const bool big = !(at < argc && (!strcmp(argv[at], "quick") || !strcmp(argv[at], "short")));
    load_tuned_json(getenv("UNITER_TUNED_JSON") ? getenv("UNITER_TUNED_JSON") : "uniter_amd/tuned/gfx950.json");
    Timer tm;
    struct Case { const char* kind; int64_t M, N, K; int splits; };
    std::vector<Case> cases = {
        {"fwd", 4096, 4096, 4096, 1}, {"fwd", 3072, 2304, 768, 1}, {"gelu", 3072, 3072, 768, 1}, {"dgelu", 3072, 768, 3072, 1},
        {"dgrad", 3072, 2304, 768, 1}, {"dgrad", 3072, 3072, 768, 1}, {"fwd", 3072, 768, 3072, 1}, {"fwd", 3072, 768, 768, 1},
        {"wgrad", 3072, 3072, 768, 1}, {"wgrad", 3072, 3072, 768, 2}, {"wgrad", 3072, 768, 3072, 2}, {"wgrad", 3072, 2304, 768, 2},
        {"fwd", 3072, 3072, 1024, 1}, {"gelu", 3072, 4096, 1024, 1}, {"dgelu", 3072, 1024, 4096, 1}, {"fwd", 5696, 3072, 1024, 1},
        {"gelu", 5696, 4096, 1024, 1}, {"wgrad", 5696, 4096, 1024, 2}, {"wgrad", 5696, 4096, 1024, 1}};
    if (big) cases.push_back({"fwd", 8192, 8192, 8192, 1});
    const bool shortlist = at < argc && !strcmp(argv[at], "short");      // epilogue experiments: the 3072 x 3072 x 768 trio only
    if (shortlist) cases = {{"fwd", 3072, 3072, 768, 1}, {"gelu", 3072, 3072, 768, 1}, {"dgelu", 3072, 768, 3072, 1}, {"fwd", 3072, 2304, 768, 1}};
    for (const Case& c : cases) {
        const std::string kind = c.kind;
        const int64_t M = c.M, N = c.N, K = c.K;
        // buffers sized for every role: x [M,K] / dy [M,N], w [N,K], out [M,N] / [M,K] / [N,K]
        const size_t e_mk = (size_t)M * K, e_mn = (size_t)M * N, e_nk = (size_t)N * K;
        HostBf A, B, Bv;
        A.fill(std::max(e_mk, e_mn), 1.f); B.fill(std::max(e_nk, e_mk), 0.05f); Bv.fill((size_t)std::max(N, K), 0.1f);
        uint16_t *dA = upload(A), *dB = upload(B), *dBias = upload(Bv);
        const size_t e_out = std::max(std::max(e_mn, e_mk), e_nk);
        uint16_t *dO = dalloc<uint16_t>(e_out), *dO2 = dalloc<uint16_t>(e_out), *dU = upload(A);
        const size_t wsb = uniter_gemm_wgrad_workspace_bytes(M, N, K);
        void* ws = kind == "wgrad" ? dalloc<char>(wsb) : nullptr;
        auto fn = [&](uint16_t* out) {
            if (kind == "fwd") UHCHK(uniter_gemm_bias_fwd(dA, dB, dBias, out, M, N, K, 0));
            else if (kind == "gelu") UHCHK(uniter_gemm_bias_gelu_fwd(dA, dB, dBias, out, dO2, M, N, K, 0));
            else if (kind == "dgrad") UHCHK(uniter_gemm_dgrad(dA, dB, nullptr, out, M, N, K, 0));
            else if (kind == "dgelu") UHCHK(uniter_gemm_dgrad_gelu(dA, dB, dU, out, M, N, K, 0));
            else UHCHK(uniter_gemm_wgrad(dA, dB, out, nullptr, M, N, K, 0, ws, wsb, 0));      // dA = dy [M,N], dB = x [M,K]
        };
        const size_t n_out = kind == "wgrad" ? e_nk : ((kind == "dgrad" || kind == "dgelu") ? e_mk : e_mn);
        const double fl = 2.0 * M * N * K;
        // reference: the tuned / default tile of the older family
        uniter_gemm_debug_force(-1, -1);
        uint16_t* dRef = dalloc<uint16_t>(n_out);
        fn(dRef);
        HIPCHK(hipDeviceSynchronize());
        const double t_old = tm.run([&] { fn(dRef); }, 3, 20);
        for (int tile : {kTileG8, kTileG6}) {
        const int64_t edge = tile == kTileG8 ? 256 : 192;
        const bool legal = (kind == "fwd" || kind == "gelu") ? N % edge == 0 : ((kind == "dgrad" || kind == "dgelu") ? K % edge == 0 : (N % edge == 0 && K % edge == 0));
        if (!legal) continue;
        uniter_gemm_debug_force(tile, c.splits);
        fn(dO);
        HIPCHK(hipDeviceSynchronize());
        std::vector<uint16_t> r0(n_out), r1(n_out), rr(n_out);
        HIPCHK(hipMemcpy(r0.data(), dO, n_out * 2, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(rr.data(), dRef, n_out * 2, hipMemcpyDeviceToHost));
        double maxd = 0, maxr = 0; size_t nbad = 0;
        for (size_t k = 0; k < n_out; ++k) {
            const float a = bf2f(r0[k]), b = bf2f(rr[k]);
            const double d = fabs((double)a - b);
            maxd = std::max(maxd, d); maxr = std::max(maxr, (double)fabsf(b));
            if (!(d <= 0.02 * fabs(b) + 0.02 * sqrt((double)(kind == "wgrad" ? M : (kind == "fwd" || kind == "gelu" ? K : N))) * 0.05 + 0.05)) ++nbad;
        }
        char tag[200];
        snprintf(tag, sizeof tag, "tile %lld %s M%lld N%lld K%lld s%d == older tile family (max |d| %.4f of max |ref| %.2f, %zu outside tolerance)", (long long)edge, c.kind,
                 (long long)M, (long long)N, (long long)K, c.splits, maxd, maxr, nbad);
        printf("[%s] %s\n", nbad ? "FAIL" : " OK ", tag);
        if (nbad) ++g_fail;
        // race screen: 24 more launches, every one bit-identical to the first
        size_t ndiff = 0;
        for (int rep = 0; rep < 24; ++rep) {
            HIPCHK(hipMemsetAsync(dO, 0xff, n_out * 2, 0));
            fn(dO);
            if (rep % 8 == 7) {
                HIPCHK(hipMemcpy(r1.data(), dO, n_out * 2, hipMemcpyDeviceToHost));
                for (size_t k = 0; k < n_out; ++k) if (r1[k] != r0[k]) ++ndiff;
            }
        }
        printf("[%s] tile %lld %s repeated launches bit-identical (%zu differing elements)\n", ndiff ? "FAIL" : " OK ", (long long)edge, c.kind, ndiff);
        if (ndiff) ++g_fail;
        const double t_new = tm.run([&] { fn(dO); }, 3, 20);
        printf("  TIME %-5s M%-5lld N%-5lld K%-5lld splits%d : older family %7.2f us %7.1f TF | %lldx%lld deep-pipelined %7.2f us %7.1f TF  (x%.2f)\n", c.kind, (long long)M,
               (long long)N, (long long)K, c.splits, t_old, fl / t_old * 1e-6, (long long)edge, (long long)edge, t_new, fl / t_new * 1e-6, t_old / t_new);
        }
        uniter_gemm_debug_force(-1, -1);
        HIPCHK(hipFree(dA)); HIPCHK(hipFree(dB)); HIPCHK(hipFree(dBias)); HIPCHK(hipFree(dO)); HIPCHK(hipFree(dO2)); HIPCHK(hipFree(dU)); HIPCHK(hipFree(dRef));
        if (ws) HIPCHK(hipFree(ws));
    }
    if (shortlist) { printf("== %d check(s) failed ==\n", g_fail); return g_fail; }
    // the four weight gradients (+ bias gradients) of a layer as one launch
    struct GCase { const char* name; int64_t T, H, I; } gcases[] = {{"base-96", 3072, 768, 3072}, {"large-96", 3072, 1024, 4096}, {"large-178", 5696, 1024, 4096}};
    for (const GCase& gc : gcases) {
        const int64_t T = gc.T, H = gc.H, I = gc.I;
        const int64_t gN[4] = {H, I, H, 3 * H}, gK[4] = {I, H, H, H};
        uint16_t *dy[4], *x[4], *dw[4], *dwr[4], *db[4], *dbr[4];
        for (int q = 0; q < 4; ++q) {
            HostBf a, b;
            a.fill((size_t)T * gN[q], 1.f); b.fill((size_t)T * gK[q], 1.f);
            dy[q] = upload(a); x[q] = upload(b);
            dw[q] = dalloc<uint16_t>((size_t)gN[q] * gK[q]); dwr[q] = dalloc<uint16_t>((size_t)gN[q] * gK[q]);
            db[q] = dalloc<uint16_t>(gN[q]); dbr[q] = dalloc<uint16_t>(gN[q]);
        }
        const void* dyp[4] = {dy[0], dy[1], dy[2], dy[3]};
        const void* xp[4] = {x[0], x[1], x[2], x[3]};
        void* dwp[4] = {dw[0], dw[1], dw[2], dw[3]};
        void* dbp[4] = {db[0], db[1], db[2], db[3]};
        void* dwrp[4] = {dwr[0], dwr[1], dwr[2], dwr[3]};
        void* dbrp[4] = {dbr[0], dbr[1], dbr[2], dbr[3]};
        const size_t gwb = uniter_gemm_wgrad_group_workspace_bytes(4, gN, gK);
        void* gws = dalloc<char>(gwb);
        double fl = 0;
        for (int q = 0; q < 4; ++q) fl += 2.0 * T * gN[q] * gK[q];
        UHCHK(uniter_gemm_wgrad_group_ws(4, dyp, nullptr, xp, nullptr, dwrp, dbrp, T, gN, gK, 0, nullptr, 0, 33, 1, 0));   // 128x128, 4+4 waves
        const double t_old = tm.run([&] { UHCHK(uniter_gemm_wgrad_group_ws(4, dyp, nullptr, xp, nullptr, dwrp, dbrp, T, gN, gK, 0, nullptr, 0, 33, 1, 0)); }, 3, 20);
        for (int var = 0; var < 3; ++var) {
            const int sp = var == 1 ? 2 : 1, kTileX = var == 2 ? kTileG6 : kTileG8;
            if (var == 2 && (H % 192 != 0 || I % 192 != 0)) continue;
            UHCHK(uniter_gemm_wgrad_group_ws(4, dyp, nullptr, xp, nullptr, dwp, dbp, T, gN, gK, 0, gws, gwb, kTileX, sp, 0));
            HIPCHK(hipDeviceSynchronize());
            size_t nbad = 0, ndiff = 0;
            double maxd = 0;
            std::vector<std::vector<uint16_t>> first(4);
            for (int q = 0; q < 4; ++q) {
                const size_t ne = (size_t)gN[q] * gK[q];
                std::vector<float> got = download_bf(dw[q], ne), ref = download_bf(dwr[q], ne);
                for (size_t k = 0; k < ne; ++k) { const double d = fabs((double)got[k] - ref[k]); maxd = std::max(maxd, d); if (!(d <= 0.02 * fabs(ref[k]) + 0.02 * sqrt((double)T) + 0.05)) ++nbad; }
                std::vector<float> gb = download_bf(db[q], gN[q]), rb = download_bf(dbr[q], gN[q]);
                for (int64_t k = 0; k < gN[q]; ++k) if (!(fabs((double)gb[k] - rb[k]) <= 0.02 * fabs(rb[k]) + 0.02 * sqrt((double)T) + 0.05)) ++nbad;
                first[q].resize(ne);
                HIPCHK(hipMemcpy(first[q].data(), dw[q], ne * 2, hipMemcpyDeviceToHost));
            }
            printf("[%s] tile %d group %s, %d slice(s) == 128x128 grouped launch (max |d| %.3f, %zu outside tolerance)\n", nbad ? "FAIL" : " OK ", kTileX, gc.name, sp, maxd, nbad);
            if (nbad) ++g_fail;
            for (int rep = 0; rep < 16; ++rep) UHCHK(uniter_gemm_wgrad_group_ws(4, dyp, nullptr, xp, nullptr, dwp, dbp, T, gN, gK, 0, gws, gwb, kTileX, sp, 0));
            HIPCHK(hipDeviceSynchronize());
            for (int q = 0; q < 4; ++q) {
                const size_t ne = (size_t)gN[q] * gK[q];
                std::vector<uint16_t> now(ne);
                HIPCHK(hipMemcpy(now.data(), dw[q], ne * 2, hipMemcpyDeviceToHost));
                for (size_t k = 0; k < ne; ++k) if (now[k] != first[q][k]) ++ndiff;
            }
            printf("[%s] tile %d group %s, %d slice(s): 17th launch bit-identical to the first (%zu differing)\n", ndiff ? "FAIL" : " OK ", kTileX, gc.name, sp, ndiff);
            if (ndiff) ++g_fail;
            const double t_new = tm.run([&] { UHCHK(uniter_gemm_wgrad_group_ws(4, dyp, nullptr, xp, nullptr, dwp, dbp, T, gN, gK, 0, gws, gwb, kTileX, sp, 0)); }, 3, 20);
            printf("  TIME group %-9s: 128x128 grouped %7.2f us %7.1f TF | tile %d, %d slice(s) %7.2f us %7.1f TF (x%.2f)\n", gc.name, t_old, fl / t_old * 1e-6, kTileX, sp,
                   t_new, fl / t_new * 1e-6, t_old / t_new);
        }
        for (int q = 0; q < 4; ++q) { HIPCHK(hipFree(dy[q])); HIPCHK(hipFree(x[q])); HIPCHK(hipFree(dw[q])); HIPCHK(hipFree(dwr[q])); HIPCHK(hipFree(db[q])); HIPCHK(hipFree(dbr[q])); }
        HIPCHK(hipFree(gws));
    }
    printf("== %d check(s) failed ==\n", g_fail);
    return g_fail;
}


// --roofs [iters] [M] : the eight GEMMs of a UNITER-base layer's forward / data-gradient chain, each alone on hot operands with the
// tile the shipped table picks (or UNITER_ROOFS_CFG=<tile> for all of them), `iters` launches each.  Prints one TIME line per
// shape; meant to run under `rocprofv3 --pmc ... --kernel-trace` (scripts/gpu_r5_record.sh), where the dispatches of the CSV
// appear in this order, 3 warm-up + iters per shape.
static int run_roofs(int argc, char** argv, int at) {
    const int iters = at < argc ? atoi(argv[at]) : 10;
    const int64_t M = at + 1 < argc ? atoll(argv[at + 1]) : 3072;
    const int64_t H = 768, I = 3072;
    load_tuned_json(getenv("UNITER_TUNED_JSON") ? getenv("UNITER_TUNED_JSON") : "uniter_amd/tuned/gfx950.json");
    const char* fc = getenv("UNITER_ROOFS_CFG");
    uniter_gemm_debug_force(fc ? atoi(fc) : -1, fc ? 1 : -1);
    HostBf A, W, Bv;
    A.fill((size_t)M * I, 1.f); W.fill((size_t)I * I, 0.05f); Bv.fill((size_t)I, 0.1f);
    uint16_t *dA = upload(A), *dW = upload(W), *dB = upload(Bv), *dR = upload(A);
    uint16_t *dO = dalloc<uint16_t>((size_t)M * I), *dO2 = dalloc<uint16_t>((size_t)M * I);
    struct Shape { const char* name; int kind; int64_t N, K; } shapes[] = {
        {"qkv_fwd", 0, 3 * H, H}, {"out_fwd", 1, H, H}, {"ffn1_fwd_gelu", 2, I, H}, {"ffn2_fwd", 1, H, I},
        {"ffn2_dgrad_gelu", 3, H, I}, {"ffn1_dgrad", 4, I, H}, {"out_dgrad", 5, H, H}, {"qkv_dgrad", 4, 3 * H, H},
        {"ffn2_dgrad_plain", 5, H, I},         // (not a launch of the model: the x gelu' shape without its epilogue, beside the vendor yardstick)
        {"ffn1_fwd_gelu_d", 6, I, H}, {"ffn2_dgrad_saved_d", 7, H, I}};   // the encoder's forms: FFN1 saves gelu'(u), the data gradient multiplies by it
    Timer tm;
    for (const Shape& s : shapes) {
        const int64_t N = s.N, K = s.K;
        auto fn = [&] {
            switch (s.kind) {
                case 0: UHCHK(uniter_gemm_bias_fwd(dA, dW, dB, dO, M, N, K, 0)); break;
                case 1: UHCHK(uniter_gemm_bias_dropout_residual_fwd(dA, dW, dB, dR, dO, M, N, K, 0.1f, 1234u, 0u, 0)); break;
                case 2: UHCHK(uniter_gemm_bias_gelu_fwd(dA, dW, dB, dO, dO2, M, N, K, 0)); break;
                case 3: UHCHK(uniter_gemm_dgrad_gelu(dA, dW, dR, dO, M, N, K, 0)); break;
                case 4: UHCHK(uniter_gemm_dgrad(dA, dW, dR, dO, M, N, K, 0)); break;
                case 6: UHCHK(uniter_gemm_bias_gelu_fwd(dA, dW, dB, dO, dO2, M, N, K, 0)); break;
                case 7: UHCHK(uniter_gemm_dgrad_gelu(dA, dW, dR, dO, M, N, K, 0)); break;
                default: UHCHK(uniter_gemm_dgrad(dA, dW, nullptr, dO, M, N, K, 0)); break;
            }
        };
        uniter_gemm_debug_act_flags(s.kind >= 6 ? 0x100 : 0);
        const double us = tm.run(fn, 3, iters);
        uniter_gemm_debug_act_flags(0);
        int32_t ch[2] = {-1, -1};
        uniter_gemm_tuned_choice(s.kind == 0 || s.kind == 1 || s.kind == 2 || s.kind == 6 ? 0 : 1, M, N, K, ch);
        printf("  ROOF %-16s M%lld N%lld K%lld tile %d : %7.2f us  %7.1f TF\n", s.name, (long long)M, (long long)N, (long long)K, fc ? atoi(fc) : ch[0], us,
               2.0 * M * N * K / us * 1e-6);
    }
    uniter_gemm_debug_force(-1, -1);
    return 0;
}

// --sweep [iters] : every tile of the table on each of the eight chain shapes (alone, hot operands), one line per (shape, tile);
// the best tile per shape at the end.  In-process (one allocation, no per-tile process start): ~10 s for all 67 tiles.
static int run_sweep(int argc, char** argv, int at) {
    const int iters = at < argc ? atoi(argv[at]) : 10;
    const int64_t M = 3072, H = 768, I = 3072;
    HostBf A, W, Bv;
    A.fill((size_t)M * I, 1.f); W.fill((size_t)I * I, 0.05f); Bv.fill((size_t)I, 0.1f);
    uint16_t *dA = upload(A), *dW = upload(W), *dB = upload(Bv), *dR = upload(A);
    uint16_t *dO = dalloc<uint16_t>((size_t)M * I), *dO2 = dalloc<uint16_t>((size_t)M * I);
    struct Shape { const char* name; int kind; int64_t N, K; };
    std::vector<Shape> shapes = {
        {"qkv_fwd", 0, 3 * H, H}, {"out_fwd", 1, H, H}, {"ffn1_fwd_gelu", 2, I, H}, {"ffn2_fwd", 1, H, I},
        {"ffn2_dgrad_gelu", 3, H, I}, {"ffn1_dgrad", 4, I, H}, {"out_dgrad", 5, H, H}, {"qkv_dgrad", 4, 3 * H, H}};
    // `--sweep iters head`: the two un-grouped GEMMs of the NLVR2 paired-attention head instead (Linear(2H, H) + ReLU + Dropout,
    // model/nlvr2.py:138-141): forward 3072 x 768 x 1536, data gradient 3072 x 1536 from a 768-wide dy
    if (at + 1 < argc && !strcmp(argv[at + 1], "head")) shapes = {{"head_fc_fwd", 1, H, 2 * H}, {"head_fc_dgrad", 5, H, 2 * H}};
    Timer tm;
    for (const Shape& s : shapes) {
        const int64_t N = s.N, K = s.K;
        const bool dgrad = s.kind >= 3;
        const int64_t out_cols = dgrad ? K : N;             // width of the output (tile columns)
        auto fn = [&]() -> int {
            switch (s.kind) {
                case 0: return uniter_gemm_bias_fwd(dA, dW, dB, dO, M, N, K, 0);
                case 1: return uniter_gemm_bias_dropout_residual_fwd(dA, dW, dB, dR, dO, M, N, K, 0.1f, 1234u, 0u, 0);
                case 2: return uniter_gemm_bias_gelu_fwd(dA, dW, dB, dO, dO2, M, N, K, 0);
                case 3: return uniter_gemm_dgrad_gelu(dA, dW, dR, dO, M, N, K, 0);
                case 4: return uniter_gemm_dgrad(dA, dW, dR, dO, M, N, K, 0);
                default: return uniter_gemm_dgrad(dA, dW, nullptr, dO, M, N, K, 0);
            }
        };
        double best = 1e30; int best_cfg = -1;
        for (int cfg = 0; cfg < kNumTiles; ++cfg) {
            const int bn = kTileBN[cfg];
            if (out_cols % bn != 0) continue;
            if (dgrad && !(bn == 64 || bn == 128 || bn == 192 || cfg == kTileG8 || cfg == kTileG6)) continue;   // K-strided N-side operand
            uniter_gemm_debug_force(cfg, 1);
            if (fn() != 0) { HIPCHK(hipDeviceSynchronize()); continue; }
            HIPCHK(hipDeviceSynchronize());
            const double us = tm.run([&] { UHCHK(fn()); }, 2, iters);
            printf("  SWEEP %-16s tile %2d %3dx%3d : %7.2f us\n", s.name, cfg, kTileBM[cfg], bn, us);
            if (us < best) { best = us; best_cfg = cfg; }
        }
        printf("  BEST  %-16s tile %2d %3dx%3d : %7.2f us  %7.1f TF\n", s.name, best_cfg, kTileBM[best_cfg], kTileBN[best_cfg], best, 2.0 * M * N * K / best * 1e-6);
    }
    uniter_gemm_debug_force(-1, -1);
    return 0;
}

static void on_segv(int) {                 // where did it die: raw return addresses + symbols to stderr
    void* bt[48];
    const int n = backtrace(bt, 48);
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}


// ---------------------------------------------------------------------------------------------
// attention alone: forward / backward launch time at a given shape (--attn B L heads p)
// ---------------------------------------------------------------------------------------------
static int run_attn(int argc, char** argv, int at) {
    const int B = at < argc ? atoi(argv[at]) : 32, L = at + 1 < argc ? atoi(argv[at + 1]) : 96, heads = at + 2 < argc ? atoi(argv[at + 2]) : 12;
    const float p = at + 3 < argc ? (float)atof(argv[at + 3]) : 0.1f;
    const int H = heads * 64;
    const size_t T = (size_t)B * L;
    HostBf QKV, DO;
    QKV.fill(T * 3 * H, 1.5f); DO.fill(T * H, 1.f);
    std::vector<float> mask(B * L, 0.f);
    uint16_t *dQKV = upload(QKV), *dDO = upload(DO);
    float* dMask = dalloc<float>(B * L);
    HIPCHK(hipMemcpy(dMask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    uint16_t* dCtx = dalloc<uint16_t>(T * H);
    uint16_t* dDQKV = dalloc<uint16_t>(T * 3 * H);
    float* dLse = dalloc<float>((size_t)B * heads * L);
    const size_t awsb = uniter_attention_bwd_workspace_bytes(B, L, heads);
    const size_t stamp_words = (size_t)B * heads * 64;
    void* aws = dalloc<char>(awsb + 16 + stamp_words * 8);
    HIPCHK(hipMemset(aws, 0, awsb + 16 + stamp_words * 8));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto timeit = [&](const char* what, auto&& fn) {
        for (int i = 0; i < 5; ++i) fn();
        float best = 1e9f, sum = 0.f;
        for (int rep = 0; rep < 5; ++rep) {
            HIPCHK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) fn();
            HIPCHK(hipEventRecord(e1, 0));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 20.f); sum += ms / 20.f;
        }
        printf("attn %s B%d L%d heads%d p=%.2f: best %.2f us, mean %.2f us (back-to-back launches)\n", what, B, L, heads, p, best * 1e3f, sum / 5.f * 1e3f);
    };
    timeit("fwd", [&] { UHCHK(uniter_attention_fwd(dQKV, dMask, dCtx, dLse, B, L, heads, p, 99, 5, 0)); });
    timeit("bwd", [&] { UHCHK(uniter_attention_bwd_ws(dQKV, dMask, nullptr, dCtx, dLse, dDO, dDQKV, B, L, heads, p, 99, 5, aws, awsb + 16 + stamp_words * 8, 0)); });
    if (getenv("UNITER_AMD_ATTN_DBG") && (atoi(getenv("UNITER_AMD_ATTN_DBG")) & 8)) {
        HIPCHK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(stamp_words);
        HIPCHK(hipMemcpy(st.data(), aws, stamp_words * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t w = 0; w < stamp_words / 8; ++w) if (st[w * 8] && (w & 7) < 6) { t0 = std::min(t0, st[w * 8]); t1 = std::max(t1, st[w * 8 + 6]); }
        printf("stamps (last launch): first wave start -> last wave end %llu ticks\n", t1 - t0);
        for (int bh : {0, 1, 100, 255, 256, 300, 383}) {
            if (bh >= B * heads) continue;
            for (int w = 0; w < 6; w += 5) {
                const unsigned long long* q = &st[((size_t)bh * 8 + w) * 8];
                printf("  bh %3d wave %d: start +%6llu | loads+commit %5llu | barrier %5llu | query sweep %5llu | barrier(s) %5llu | key sweep %5llu | drain %5llu\n", bh, w,
                       q[0] - t0, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5]);
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// fused QKV projection + attention forward (uniter_qkv_attention_fwd) against the two launches it replaces: qkv, ctx and lse must be
// the same BITS (same MFMA order in the projection, same attention body); --qkvattn [B heads p] also times both forms
// ---------------------------------------------------------------------------------------------
static int test_qkv_attention(int B, int heads, float p, bool timing) {
    const int L = 96, H = heads * 64;
    const size_t T = (size_t)B * L;
    HostBf X, W, Bq;
    X.fill(T * H, 1.f); W.fill((size_t)3 * H * H, 0.05f); Bq.fill((size_t)3 * H, 0.1f);
    std::vector<float> mask((size_t)B * L, 0.f);
    for (int b = 0; b < B; ++b)
        for (int k = L - (b % 7) * 5; k < L; ++k) mask[(size_t)b * L + k] = -10000.f;      // ragged key padding
    uint16_t *dX = upload(X), *dW = upload(W), *dB = upload(Bq);
    float* dMask = dalloc<float>((size_t)B * L);
    HIPCHK(hipMemcpy(dMask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    uint16_t *dQ1 = dalloc<uint16_t>(T * 3 * H), *dQ2 = dalloc<uint16_t>(T * 3 * H), *dC1 = dalloc<uint16_t>(T * H), *dC2 = dalloc<uint16_t>(T * H);
    float *dL1 = dalloc<float>((size_t)B * heads * L), *dL2 = dalloc<float>((size_t)B * heads * L);
    HIPCHK(hipMemset(dQ2, 0xFF, T * 3 * H * 2)); HIPCHK(hipMemset(dC2, 0xFF, T * H * 2)); HIPCHK(hipMemset(dL2, 0xFF, (size_t)B * heads * L * 4));
    UHCHK(uniter_gemm_bias_fwd(dX, dW, dB, dQ1, (int64_t)T, 3 * H, H, 0));
    UHCHK(uniter_attention_fwd(dQ1, dMask, dC1, dL1, B, L, heads, p, 99, 5, 0));
    UHCHK(uniter_qkv_attention_fwd(dX, dW, dB, dMask, dQ2, dC2, dL2, B, L, heads, p, 99, 5, 0));
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint16_t> q1(T * 3 * H), q2(T * 3 * H), c1(T * H), c2(T * H);
    std::vector<float> l1((size_t)B * heads * L), l2((size_t)B * heads * L);
    HIPCHK(hipMemcpy(q1.data(), dQ1, q1.size() * 2, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(q2.data(), dQ2, q2.size() * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c1.data(), dC1, c1.size() * 2, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(c2.data(), dC2, c2.size() * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(l1.data(), dL1, l1.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(l2.data(), dL2, l2.size() * 4, hipMemcpyDeviceToHost));
    size_t dq = 0, dc = 0, dl = 0;
    for (size_t k = 0; k < q1.size(); ++k) dq += q1[k] != q2[k];
    for (size_t k = 0; k < c1.size(); ++k) dc += c1[k] != c2[k];
    for (size_t k = 0; k < l1.size(); ++k) dl += memcmp(&l1[k], &l2[k], 4) != 0;
    const bool ok = dq == 0 && dc == 0 && dl == 0;
    printf("[%s] fused qkv + attention fwd B%d heads%d p=%.2f vs two launches: differing qkv %zu / %zu, ctx %zu / %zu, lse %zu / %zu\n",
           ok ? " OK " : "FAIL", B, heads, p, dq, q1.size(), dc, c1.size(), dl, l1.size());
    if (!ok) ++g_fail;
    if (timing) {
        Timer tm;
        const double two = tm.run([&] { UHCHK(uniter_gemm_bias_fwd(dX, dW, dB, dQ1, (int64_t)T, 3 * H, H, 0)); UHCHK(uniter_attention_fwd(dQ1, dMask, dC1, dL1, B, L, heads, p, 99, 5, 0)); }, 3, 20);
        const double one = tm.run([&] { UHCHK(uniter_qkv_attention_fwd(dX, dW, dB, dMask, dQ2, dC2, dL2, B, L, heads, p, 99, 5, 0)); }, 3, 20);
        const double gem = tm.run([&] { UHCHK(uniter_gemm_bias_fwd(dX, dW, dB, dQ1, (int64_t)T, 3 * H, H, 0)); }, 3, 20);
        printf("  TIME qkv projection + attention fwd, B%d x 96 tokens, %d heads, p=%.2f: two launches %.2f us (projection alone %.2f), fused %.2f us\n", B, heads, p, two, gem, one);
    }
    for (void* q : {(void*)dX, (void*)dW, (void*)dB, (void*)dMask, (void*)dQ1, (void*)dQ2, (void*)dC1, (void*)dC2, (void*)dL1, (void*)dL2}) HIPCHK(hipFree(q));
    return 0;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);   // keep the log complete if a later test dies
    signal(SIGSEGV, on_segv);
    bool do_bench = false, quick = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--bench")) do_bench = true;
        if (!strcmp(argv[i], "--quick")) quick = true;
        if (!strcmp(argv[i], "--enc")) {
            int32_t inf[4];
            UHCHK(uniter_hip_device_info(inf));
            printf("== grouped weight gradients ==\n");
            if (!getenv("UNITER_BENCH_XCD_ONLY")) {
                test_wgrad_group(320);
                test_wgrad_group(300);
            }
            if (i + 1 < argc && !strcmp(argv[i + 1], "large")) bench_encoder(32, 96, 1024, 16, 4096, 24);
            else if (i + 1 < argc && !strcmp(argv[i + 1], "large178")) bench_encoder(32, 178, 1024, 16, 4096, 24);
            else bench_encoder(32, 96, 768, 12, 3072, 12);
            printf("== %d check(s) failed ==\n", g_fail);
            return g_fail;
        }
        if (!strcmp(argv[i], "--g8")) {
            int32_t inf[4];
            UHCHK(uniter_hip_device_info(inf));
            return run_g8(argc, argv, i + 1);
        }
        if (!strcmp(argv[i], "--attn")) {
            int32_t inf[4];
            UHCHK(uniter_hip_device_info(inf));
            return run_attn(argc, argv, i + 1);
        }
        if (!strcmp(argv[i], "--qkvattn")) {
            int32_t inf[4];
            UHCHK(uniter_hip_device_info(inf));
            load_tuned_json(getenv("UNITER_TUNED_JSON") ? getenv("UNITER_TUNED_JSON") : "uniter_amd/tuned/gfx950.json");
            const int B = i + 1 < argc ? atoi(argv[i + 1]) : 32, heads = i + 2 < argc ? atoi(argv[i + 2]) : 12;
            const float p = i + 3 < argc ? (float)atof(argv[i + 3]) : 0.1f;
            test_qkv_attention(B, heads, p, true);
            printf("== %d check(s) failed ==\n", g_fail);
            return g_fail;
        }
        if (!strcmp(argv[i], "--sweep")) {
            int32_t inf[4];
            UHCHK(uniter_hip_device_info(inf));
            return run_sweep(argc, argv, i + 1);
        }
        if (!strcmp(argv[i], "--roofs")) {
            int32_t inf[4];
            UHCHK(uniter_hip_device_info(inf));
            return run_roofs(argc, argv, i + 1);
        }
        if (!strcmp(argv[i], "--one")) {
            int32_t inf[4];
            UHCHK(uniter_hip_device_info(inf));
            return run_one(argc, argv, i + 1);
        }
    }
    int32_t info[4];
    UHCHK(uniter_hip_device_info(info));
    printf("device: %d CUs, wave %d, LDS/CU %d, gfx%d, abi %d\n", info[0], info[1], info[2], info[3], uniter_hip_abi_version());
    run_probes();
    printf("== gemm ==\n");
    for (int cfg = 0; cfg < 4; ++cfg) test_gemm(200, 256, 128, cfg, cfg == 0 ? 1 : 2);
    for (int cfg = 4; cfg < kNumTiles; ++cfg) test_gemm(cfg >= 20 ? 320 : 300, 384, 128, cfg, 1);   // WS: contraction % 64 == 0 (wgrad contracts over M)
    for (int cfg = 0; cfg < kNumTiles; ++cfg)            // 192-wide K-strided N-side operand (dgrad / wgrad): two column tiles
        if (kTileBN[cfg] == 192) test_gemm(320, 384, 384, cfg, 1);
    // the eight-phase 256x256 tile: partial last row tile, odd / even K tile counts, one K tile, one and two slices (in-launch
    // combination) and four (fp32 partials + reduce kernel)
    test_gemm(640, 512, 256, kTileG8, 1);
    test_gemm(640, 512, 256, kTileG8, 2);
    test_gemm(576, 256, 512, kTileG8, 2);
    test_gemm(300, 256, 64, kTileG8, 1);
    test_gemm(1024, 256, 256, kTileG8, 4);
    test_gemm(640, 384, 192, kTileG6, 1);
    test_gemm(640, 384, 192, kTileG6, 2);
    test_gemm(576, 192, 384, kTileG6, 2);
    test_gemm(300, 192, 64, kTileG6, 1);
    test_gemm(77, 128, 192, 3, 3);
    test_gemm(384, 384, 320, -1, -1);
    if (!quick) test_gemm(1000, 768, 768, -1, -1);
    printf("== attention ==\n");
    test_attention(2, 96, 2, 0.f);
    test_attention(3, 40, 1, 0.f);
    test_attention(2, 96, 2, 0.2f);
    test_attention(2, 178, 2, 0.1f);
    test_attention(1, 250, 1, 0.f);
    test_attention(2, 17, 1, 0.1f);
    test_attention(2, 260, 2, 0.1f);          // 60 text + 2 x 100 regions (the NLVR2 triplet format): the split backward
    test_attention(1, 384, 1, 0.f);
    test_attention(1, 512, 2, 0.1f);
    printf("== fused qkv projection + attention forward ==\n");
    test_qkv_attention(2, 2, 0.f, false);
    test_qkv_attention(5, 4, 0.2f, false);
    test_qkv_attention(32, 12, 0.1f, false);          // the benchmark shape (UNITER-base, 32 x 96 tokens)
    test_qkv_attention(16, 16, 0.1f, false);          // UNITER-large heads
    printf("== layernorm ==\n");
    printf("== grouped weight gradients ==\n");
    test_wgrad_group(320);
    test_wgrad_group(300);
    test_wgrad_group_g8(320, 1);
    test_wgrad_group_g8(320, 2);
    test_wgrad_group_g8(576, 2);
    test_wgrad_group_g8(320, 1, kTileG6);
    test_wgrad_group_g8(576, 1, kTileG6);
    test_layernorm(37, 128, 0.f, 0);
    test_layernorm(300, 768, 0.2f, 0);
    test_layernorm(300, 768, 0.2f, 1);
    test_layernorm(129, 1024, 0.f, 0);
    test_layernorm(64, 2048, 0.1f, 0);
    printf("== deferred weight gradients ==\n");
    test_deferred_wgrad(4, 64, 256, 4, 512, 4);
    test_deferred_wgrad(2, 96, 768, 12, 3072, 3);
    printf("== adamw ==\n");
    test_adamw();
    if (do_bench) {
        bench(32, 96, 768, 12, 3072, 12);
    }
    printf("== %d check(s) failed ==\n", g_fail);
    return g_fail;
}
