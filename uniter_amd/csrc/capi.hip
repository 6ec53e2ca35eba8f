// capi.hip — extern "C" entry points of libuniter_hip.so for the single-kernel operations
// (GEMM family, attention, LayerNorm, column sums) plus status plumbing.  See include/uniter_hip.h.
#include "common.cuh"
#include "kernels.h"
#include "../../include/uniter_hip.h"
#include "../../include/uniter_hip_test.h"

#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {
thread_local char g_err[512] = {0};

__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) { *c += inc; }
}

const unsigned long long* uh_drop_offset_ptr = nullptr;

// ---- per-launch timing: event pairs recorded on the launch stream, resolved at uniter_hip_timing_end ----
namespace uh {
bool g_timing_on = false;
namespace {
struct TimedLaunch { int kind; int64_t M, N, K; hipEvent_t e0, e1; };
std::vector<TimedLaunch> g_timed;
std::vector<hipEvent_t> g_event_pool;
std::mutex g_timing_mu;
hipEvent_t take_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace
void timing_mark(int kind, int64_t M, int64_t N, int64_t K, hipStream_t st, bool begin) {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (begin) {
        TimedLaunch t{kind, M, N, K, take_event(), take_event()};
        if (t.e0 == nullptr || t.e1 == nullptr) return;
        (void)hipEventRecord(t.e0, st);
        g_timed.push_back(t);
    } else {
        for (size_t i = g_timed.size(); i-- > 0;) {         // innermost open bracket of this kind
            TimedLaunch& t = g_timed[i];
            if (t.kind == kind && t.M == M && t.N == N && t.K == K) { (void)hipEventRecord(t.e1, st); break; }
        }
    }
}
}  // namespace uh

void uh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

extern "C" {

int uniter_hip_abi_version(void) { return UNITER_HIP_ABI_VERSION; }
const char* uniter_hip_last_error(void) { return g_err; }

int uniter_hip_device_info(int32_t out[4]) {
    UH_CHECK_ARG(out != nullptr, "null pointer");
    int dev = 0;
    UH_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    UH_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    out[0] = prop.multiProcessorCount;
    out[1] = prop.warpSize;
    out[2] = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    int arch = 0;
    const char* n = prop.gcnArchName;   // "gfx950:sramecc+:xnack-"
    if (n[0] == 'g' && n[1] == 'f' && n[2] == 'x') {
        for (const char* c = n + 3; *c && *c != ':'; ++c) {
            int d = (*c >= '0' && *c <= '9') ? (*c - '0') : ((*c >= 'a' && *c <= 'f') ? (*c - 'a' + 10) : -1);
            if (d < 0) break;
            arch = arch * (d > 9 ? 16 : 10) + d;
        }
    }
    out[3] = arch;
    uh::gemm_set_num_cus(prop.multiProcessorCount);
    return 0;
}

int uniter_hip_set_dropout_offset_ptr(const uint64_t* dev_counter) {
    uh_drop_offset_ptr = (const unsigned long long*)dev_counter;
    return 0;
}

int uniter_hip_counter_add(uint64_t* dev_counter, uint64_t inc, void* stream) {
    UH_CHECK_ARG(dev_counter != nullptr, "null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)dev_counter,
                       (unsigned long long)inc);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_hip_timing_begin(void) {
    std::lock_guard<std::mutex> lk(uh::g_timing_mu);
    for (auto& t : uh::g_timed) { uh::g_event_pool.push_back(t.e0); uh::g_event_pool.push_back(t.e1); }
    uh::g_timed.clear();
    uh::g_timing_on = true;
    return 0;
}

int uniter_hip_timing_end(UniterTimingRecord* out, int32_t cap, int32_t* n_out) {
    UH_CHECK_ARG(n_out != nullptr && (out != nullptr || cap == 0), "null pointer");
    uh::g_timing_on = false;
    UH_CHECK_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(uh::g_timing_mu);
    std::map<std::tuple<int, int64_t, int64_t, int64_t>, std::pair<int, double>> agg;
    for (auto& t : uh::g_timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
            auto& a = agg[std::make_tuple(t.kind, t.M, t.N, t.K)];
            a.first += 1;
            a.second += (double)ms * 1e3;
        }
        uh::g_event_pool.push_back(t.e0);
        uh::g_event_pool.push_back(t.e1);
    }
    (void)hipGetLastError();
    uh::g_timed.clear();
    int32_t n = 0;
    for (auto& kv : agg) {
        if (n < cap) {
            out[n].kind = std::get<0>(kv.first);
            out[n].calls = kv.second.first;
            out[n].M = std::get<1>(kv.first); out[n].N = std::get<2>(kv.first); out[n].K = std::get<3>(kv.first);
            out[n].total_us = kv.second.second;
        }
        ++n;
    }
    *n_out = n;
    return 0;
}

// test / tuning hook: force a GEMM tile config (0..3, -1 = heuristic) and split count (-1 = heuristic)
int uniter_gemm_debug_force(int cfg, int splits) {
    uh::gemm_debug_force(cfg, splits);
    return 0;
}

int uniter_gemm_autotune(int kind, int64_t M, int64_t N, int64_t K, void* stream) {
    return uh::gemm_autotune(kind, M, N, K, (hipStream_t)stream);
}

int uniter_gemm_set_tuned(int kind, int64_t M, int64_t N, int64_t K, int32_t cfg, int32_t splits) {
    return uh::gemm_set_tuned(kind, M, N, K, cfg, splits);
}

int uniter_gemm_tile_count(void) { return uh::gemm_tile_count(); }

int uniter_gemm_tuned_choice(int kind, int64_t M, int64_t N, int64_t K, int32_t out[2]) {
    UH_CHECK_ARG(out != nullptr, "null pointer");
    int cfg = -1, sp = -1;
    if (uh::gemm_tuned_choice(kind, M, N, K, &cfg, &sp)) { out[0] = -1; out[1] = -1; return 0; }
    out[0] = cfg; out[1] = sp;
    return 0;
}

int uniter_gemm_bias_fwd(const void* x, const void* w, const void* bias, void* y,
                         int64_t M, int64_t N, int64_t K, void* stream) {
    UH_CHECK_ARG(x && w && y, "null pointer");
    return uh::gemm_fwd(uh::GEMM_EPI_BIAS, x, w, bias, nullptr, y, nullptr, M, N, K, make_dropout(0.f, 0, 0), (hipStream_t)stream);
}

int uniter_gemm_bias_fwd_ld(const void* x, int64_t ldx, const void* w, const void* bias, void* y, int64_t ldy,
                            int64_t M, int64_t N, int64_t K, void* stream) {
    UH_CHECK_ARG(x && w && y, "null pointer");
    return uh::gemm_fwd(uh::GEMM_EPI_BIAS, x, w, bias, nullptr, y, nullptr, M, N, K, make_dropout(0.f, 0, 0), (hipStream_t)stream, ldx, ldy);
}

int uniter_gemm_bias_fwd_group(int32_t n, const void* const* x, const int64_t* ldx, const void* const* w, const void* const* bias,
                               void* const* y, const int64_t* ldy, int64_t M, const int64_t* N, int64_t K, void* stream) {
    UH_CHECK_ARG(x && w && y && N, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int rc = uh::gemm_fwd_group(n, x, ldx, w, bias, y, ldy, M, N, K, st);
    if (rc != 1) return rc;
    for (int q = 0; q < n; ++q) {                            // no grouped tile for these shapes: one launch per problem
        const int r = uh::gemm_fwd(uh::GEMM_EPI_BIAS, x[q], w[q], bias ? bias[q] : nullptr, nullptr, y[q], nullptr, M, N[q], K, make_dropout(0.f, 0, 0), st,
                                   ldx ? ldx[q] : 0, ldy ? ldy[q] : 0);
        if (r) return r;
    }
    return 0;
}

int uniter_gemm_dgrad_group(int32_t n, const void* const* dy, const int64_t* lddy, const void* const* w, const void* const* resid,
                            void* const* dx, int64_t M, const int64_t* N, int64_t K, void* stream) {
    UH_CHECK_ARG(dy && w && dx && N, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int rc = uh::gemm_dgrad_group(n, dy, lddy, w, resid, dx, M, N, K, st);
    if (rc != 1) return rc;
    for (int q = 0; q < n; ++q) {
        const int r = uh::gemm_dgrad(uh::GEMM_EPI_RES, dy[q], w[q], resid ? resid[q] : nullptr, dx[q], M, N[q], K, st, lddy ? lddy[q] : 0);
        if (r) return r;
    }
    return 0;
}

static int g_debug_act_flags = 0;      // test hook (include/uniter_hip_test.h): the encoder's saved-derivative mode through the two entry points below
int uniter_gemm_debug_act_flags(int flags) { g_debug_act_flags = flags & UH_ACT_SAVE_GRAD; return 0; }

int uniter_gemm_bias_gelu_fwd(const void* x, const void* w, const void* bias, void* u, void* g,
                              int64_t M, int64_t N, int64_t K, void* stream) {
    UH_CHECK_ARG(x && w && u && g, "null pointer");
    return uh::gemm_fwd(uh::GEMM_EPI_BIAS_GELU, x, w, bias, nullptr, u, g, M, N, K, make_dropout(0.f, 0, 0), (hipStream_t)stream,
                        0, 0, UH_ACT_GELU | g_debug_act_flags);
}

int uniter_gemm_bias_dropout_residual_fwd(const void* x, const void* w, const void* bias,
                                          const void* resid, void* z, int64_t M, int64_t N, int64_t K,
                                          float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(x && w && z, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::gemm_fwd(uh::GEMM_EPI_BIAS_DROP_RES, x, w, bias, resid, z, nullptr, M, N, K, make_dropout(p_drop, seed, offset),
                        (hipStream_t)stream);
}

int uniter_gemm_dgrad(const void* dy, const void* w, const void* resid, void* dx,
                      int64_t M, int64_t N, int64_t K, void* stream) {
    UH_CHECK_ARG(dy && w && dx, "null pointer");
    return uh::gemm_dgrad(uh::GEMM_EPI_RES, dy, w, resid, dx, M, N, K, (hipStream_t)stream);
}

int uniter_gemm_dgrad_ld(const void* dy, int64_t lddy, const void* w, const void* resid, void* dx,
                         int64_t M, int64_t N, int64_t K, void* stream) {
    UH_CHECK_ARG(dy && w && dx, "null pointer");
    return uh::gemm_dgrad(uh::GEMM_EPI_RES, dy, w, resid, dx, M, N, K, (hipStream_t)stream, lddy);
}

int uniter_gemm_dgrad_gelu(const void* dy, const void* w, const void* u, void* dpre,
                           int64_t M, int64_t N, int64_t K, void* stream) {
    UH_CHECK_ARG(dy && w && u && dpre, "null pointer");
    return uh::gemm_dgrad(uh::GEMM_EPI_GELU_BWD, dy, w, u, dpre, M, N, K, (hipStream_t)stream, 0, UH_ACT_GELU | g_debug_act_flags);
}

size_t uniter_gemm_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    size_t a = uh::gemm_wgrad_workspace_bytes(M, N, K);
    size_t b = uh::colsum_workspace_bytes(M, N);
    return a > b ? a : b;
}

int uniter_gemm_wgrad(const void* dy, const void* x, void* dw, void* db,
                      int64_t M, int64_t N, int64_t K, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(dy && x && dw, "null pointer");
    RC(uh::gemm_wgrad(dy, x, dw, M, N, K, accumulate, workspace, workspace_bytes, (hipStream_t)stream));
    if (db != nullptr) {
        UH_CHECK_ARG(workspace != nullptr, "bias gradient needs a workspace");
        RC(uh::colsum(dy, db, M, N, accumulate, workspace, workspace_bytes, (hipStream_t)stream));
    }
    return 0;
}

int uniter_gemm_wgrad_ld(const void* dy, int64_t lddy, const void* x, int64_t ldx, void* dw, void* db,
                         int64_t M, int64_t N, int64_t K, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(dy && x && dw, "null pointer");
    return uh::gemm_wgrad(dy, x, dw, M, N, K, accumulate, workspace, workspace_bytes, (hipStream_t)stream, lddy, ldx, db);
}

size_t uniter_gemm_dgrad_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    return uh::gemm_dgrad_splitk_workspace_bytes(M, N, K);
}
int uniter_gemm_dgrad_splitk(const void* dy, int64_t lddy, const void* w, void* dx, int64_t M, int64_t N, int64_t K,
                             void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(dy && w && dx && workspace, "null pointer");
    return uh::gemm_dgrad_splitk(dy, w, dx, M, N, K, workspace, workspace_bytes, (hipStream_t)stream, lddy);
}

int uniter_gemm_wgrad_group(int32_t n, const void* const* dy, const int64_t* lddy, const void* const* x, const int64_t* ldx,
                            void* const* dw, void* const* db, int64_t M, const int64_t* N, const int64_t* K, int accumulate,
                            void* stream) {
    UH_CHECK_ARG(dy && x && dw && N && K, "null pointer");
    return uh::gemm_wgrad_group(n, dy, x, dw, db, M, N, K, accumulate, (hipStream_t)stream, -1, lddy, ldx);
}

size_t uniter_gemm_wgrad_group_workspace_bytes(int32_t n, const int64_t* N, const int64_t* K) {
    if (N == nullptr || K == nullptr || n < 1 || n > 4) return 0;
    return uh::gemm_wgrad_group_workspace_bytes(n, N, K);
}
int uniter_gemm_wgrad_group_ws(int32_t n, const void* const* dy, const int64_t* lddy, const void* const* x, const int64_t* ldx,
                               void* const* dw, void* const* db, int64_t M, const int64_t* N, const int64_t* K, int accumulate,
                               void* workspace, size_t workspace_bytes, int cfg, int splits, void* stream) {
    UH_CHECK_ARG(dy && x && dw && N && K, "null pointer");
    UH_CHECK_ARG(splits >= 0 && splits <= 2, "splits must be 0 (tuned), 1 or 2");
    return uh::gemm_wgrad_group(n, dy, x, dw, db, M, N, K, accumulate, (hipStream_t)stream, cfg, lddy, ldx, workspace, workspace_bytes, splits);
}

int uniter_gemm_wgrad_group_autotune(int32_t n, int64_t M, const int64_t* N, const int64_t* K, void* stream) {
    UH_CHECK_ARG(N && K, "null pointer");
    return uh::gemm_group_autotune(n, M, N, K, (hipStream_t)stream);
}

int uniter_attention_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse,
                         int64_t B, int64_t L, int64_t heads,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(qkv && mask_bias && ctx, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::attention_fwd(qkv, mask_bias, ctx, lse, B, L, heads, make_dropout(p_drop, seed, offset), (hipStream_t)stream);
}

int uniter_qkv_attention_fwd(const void* x, const void* wqkv, const void* bqkv, const float* mask_bias, void* qkv, void* ctx, float* lse,
                             int64_t B, int64_t L, int64_t heads, float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(x && wqkv && mask_bias && qkv && ctx, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    const int rc = uh::qkv_attention_fwd(x, wqkv, bqkv, mask_bias, qkv, ctx, lse, B, L, heads, make_dropout(p_drop, seed, offset), (hipStream_t)stream);
    if (rc != 1) return rc;
    // shapes the fused tile does not cover: the two launches it replaces
    const int64_t H = heads * 64;
    const int r1 = uh::gemm_fwd(uh::GEMM_EPI_BIAS, x, wqkv, bqkv, nullptr, qkv, nullptr, B * L, 3 * H, H, make_dropout(0.f, 0, 0), (hipStream_t)stream);
    if (r1) return r1;
    return uh::attention_fwd(qkv, mask_bias, ctx, lse, B, L, heads, make_dropout(p_drop, seed, offset), (hipStream_t)stream);
}

int uniter_attention_bwd(const void* qkv, const float* mask_bias, const void* ctx, const float* lse,
                         const void* dctx, void* dqkv, int64_t B, int64_t L, int64_t heads,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(qkv && mask_bias && ctx && lse && dctx && dqkv, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::attention_bwd(qkv, mask_bias, ctx, lse, dctx, dqkv, B, L, heads, make_dropout(p_drop, seed, offset),
                             (hipStream_t)stream);
}

size_t uniter_attention_bwd_workspace_bytes(int64_t B, int64_t L, int64_t heads) {
    return uh::attention_bwd_workspace_bytes(B, L, heads);
}

int uniter_attention_bwd_ws(const void* qkv, const float* mask_bias, const int32_t* cu_seqlens, const void* ctx, const float* lse,
                            const void* dctx, void* dqkv, int64_t B, int64_t L, int64_t heads,
                            float p_drop, uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(qkv && (mask_bias || cu_seqlens) && ctx && lse && dctx && dqkv, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    UH_CHECK_ARG(workspace_bytes >= uh::attention_bwd_workspace_bytes(B, L, heads), "workspace too small");
    return uh::attention_bwd(qkv, cu_seqlens ? nullptr : mask_bias, ctx, lse, dctx, dqkv, B, L, heads, make_dropout(p_drop, seed, offset),
                             (hipStream_t)stream, cu_seqlens, workspace_bytes ? workspace : nullptr);
}

int uniter_attention_fwd_packed(const void* qkv, const int32_t* cu_seqlens, void* ctx, float* lse,
                                int64_t B, int64_t max_len, int64_t heads,
                                float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(qkv && cu_seqlens && ctx, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::attention_fwd(qkv, nullptr, ctx, lse, B, max_len, heads, make_dropout(p_drop, seed, offset), (hipStream_t)stream, cu_seqlens);
}

int uniter_attention_bwd_packed(const void* qkv, const int32_t* cu_seqlens, const void* ctx, const float* lse,
                                const void* dctx, void* dqkv, int64_t B, int64_t max_len, int64_t heads,
                                float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(qkv && cu_seqlens && ctx && lse && dctx && dqkv, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::attention_bwd(qkv, nullptr, ctx, lse, dctx, dqkv, B, max_len, heads, make_dropout(p_drop, seed, offset),
                             (hipStream_t)stream, cu_seqlens);
}

int uniter_layernorm_fwd(const void* z, const void* gamma, const void* beta, void* y,
                         float* mean, float* rstd, int64_t rows, int64_t H, float eps,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream) {
    UH_CHECK_ARG(z && gamma && beta && y, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::layernorm_fwd(z, gamma, beta, y, mean, rstd, rows, H, eps, make_dropout(p_drop, seed, offset), (hipStream_t)stream);
}

size_t uniter_layernorm_bwd_workspace_bytes(int64_t rows, int64_t H) { return uh::layernorm_bwd_workspace_bytes(rows, H); }

int uniter_layernorm_bwd(const void* dy, const void* dy_extra, const void* z, const float* mean,
                         const float* rstd, const void* gamma,
                         void* dz, void* dd, void* dgamma, void* dbeta, void* dbias,
                         int64_t rows, int64_t H, int accumulate,
                         float p_drop, uint64_t seed, uint64_t offset, int drop_on_output,
                         void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(dy && z && mean && rstd && gamma && dz && workspace, "null pointer");
    UH_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0,1)");
    return uh::layernorm_bwd(dy, dy_extra, z, mean, rstd, gamma, dz, dd, dgamma, dbeta, dbias, rows, H, accumulate,
                             make_dropout(p_drop, seed, offset), drop_on_output, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t uniter_colsum_workspace_bytes(int64_t rows, int64_t N) { return uh::colsum_workspace_bytes(rows, N); }

int uniter_colsum(const void* a, void* out, int64_t rows, int64_t N, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(a && out && workspace, "null pointer");
    return uh::colsum(a, out, rows, N, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
