// kernels.h — internal C++ interface between the .hip translation units of libuniter_hip.so.
// The public C ABI (include/uniter_hip.h) is implemented in capi.hip on top of these.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.cuh"

struct DropoutCfg;
struct UniterEncoderShape;
struct UniterLayerParams;

namespace uh {

// ---- overlapped kernel chains (common.cuh: ChainLink; encoder.hip builds the chain) ----
// One launch of a chain.  A launcher that is handed a step puts `link` into its kernel's arguments, dispatches without the
// queue barrier when `anyorder` is set, and reports in `produced` how many contributions each 32-row unit of link.signal
// receives from this launch (its column tiles / heads / row groups) — the next step's link.expect.
struct ChainStep {
    ChainLink link{nullptr, nullptr, nullptr, 0, 0};
    int anyorder = 0;
    uint32_t produced = 0;
};
// launch with or without the barrier bit of the dispatch packet
template <typename K, typename... Args>
inline void chain_launch(const ChainStep* cs, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    if (cs != nullptr && cs->anyorder) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

// ---- capi.hip: optional per-launch timing (uniter_hip_timing_begin / _end) ----
// Kinds of timed launches; (kind, M, N, K) identifies one row of the report.
enum { TIME_GEMM_FWD_BIAS = 0, TIME_GEMM_FWD_GELU = 1, TIME_GEMM_FWD_DROP_RES = 2, TIME_GEMM_DGRAD = 3,
       TIME_GEMM_DGRAD_GELU = 4, TIME_GEMM_WGRAD = 5, TIME_ATTN_FWD = 6, TIME_ATTN_BWD = 7, TIME_LN_FWD = 8,
       TIME_LN_BWD = 9, TIME_COLSUM = 10, TIME_ADAMW = 11, TIME_LN_BWD_COLS = 12, TIME_GEMM_WGRAD_GROUP = 13 };
extern bool g_timing_on;
void timing_mark(int kind, int64_t M, int64_t N, int64_t K, hipStream_t st, bool begin);
// Brackets everything launched in its scope with two events on `st` while timing is enabled (no-op otherwise).
struct LaunchTimer {
    int kind; int64_t M, N, K; hipStream_t st; bool on;
    LaunchTimer(int kind_, int64_t M_, int64_t N_, int64_t K_, hipStream_t st_)
        : kind(kind_), M(M_), N(N_), K(K_), st(st_), on(g_timing_on) { if (on) timing_mark(kind, M, N, K, st, true); }
    ~LaunchTimer() { if (on) timing_mark(kind, M, N, K, st, false); }
};

// ---- adamw.hip ----
bool params_pending();      // segments of an asynchronous optimizer step are outstanding on this device

// ---- gemm.hip ----
enum { GEMM_EPI_BIAS = 0, GEMM_EPI_BIAS_GELU = 1, GEMM_EPI_BIAS_DROP_RES = 2, GEMM_EPI_RES = 3, GEMM_EPI_GELU_BWD = 4 };
int gemm_fwd(int epi, const void* x, const void* w, const void* bias, const void* resid, void* y, void* y2,
             int64_t M, int64_t N, int64_t K, const DropoutCfg& drop, hipStream_t st, int64_t ldx = 0, int64_t ldy = 0,
             int relu = 0, ChainStep* chain = nullptr);
int gemm_dgrad(int epi, const void* dy, const void* w, const void* aux, void* dx,
               int64_t M, int64_t N, int64_t K, hipStream_t st, int64_t lddy = 0, int act = 0, ChainStep* chain = nullptr);
// fused Q / K / V projection + self-attention forward in one launch (dense L == 96, 64-wide heads); 1 = shape not covered (nothing launched)
bool qkv_attention_fused_ok(int64_t B, int64_t L, int64_t heads, int64_t H);
int qkv_attention_fwd(const void* x, const void* wqkv, const void* bqkv, const float* mask_bias, void* qkv, void* ctx, float* lse,
                      int64_t B, int64_t L, int64_t heads, const DropoutCfg& drop, hipStream_t st);
// up to four forward / data-gradient problems over the same rows in one launch; 1 = no grouped tile fits (nothing was launched)
int gemm_fwd_group(int n, const void* const* x, const int64_t* ldx, const void* const* w, const void* const* bias, void* const* y,
                   const int64_t* ldy, int64_t M, const int64_t* N, int64_t K, hipStream_t st);
int gemm_dgrad_group(int n, const void* const* dy, const int64_t* lddy, const void* const* w, const void* const* resid, void* const* dx,
                     int64_t M, const int64_t* N, int64_t K, hipStream_t st);
size_t gemm_dgrad_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gemm_dgrad_splitk(const void* dy, const void* w, void* dx, int64_t M, int64_t N, int64_t K, void* workspace,
                      size_t ws_bytes, hipStream_t st, int64_t lddy = 0);
size_t gemm_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gemm_wgrad(const void* dy, const void* x, void* dw, int64_t M, int64_t N, int64_t K, int accumulate,
               void* workspace, size_t ws_bytes, hipStream_t st, int64_t lddy = 0, int64_t ldx = 0, void* db = nullptr);
int gemm_wgrad_group(int n, const void* const* dy, const void* const* x, void* const* dw, void* const* db, int64_t M,
                     const int64_t* N, const int64_t* K, int accumulate, hipStream_t st, int cfg_override = -1,
                     const int64_t* lddy = nullptr, const int64_t* ldx = nullptr, void* workspace = nullptr, size_t ws_bytes = 0,
                     int splits_override = 0);
// fp32 slabs of the eight-phase tile's two-slice form (4 bytes per weight element); without it the grouped launch runs one slice
size_t gemm_wgrad_group_workspace_bytes(int n, const int64_t* N, const int64_t* K);
// n problems (the weight gradients of several layers) in ONE launch on the 256 x 256 tile; 1 = not applicable (shape, capture).
// n_ln LayerNorm parameter-gradient jobs (dgamma, dbeta: column sums over `rows` tokens) ride along as extra workgroups.
struct LnColsJob { const void* dy; const void* z; const float* mean; const float* rstd; void* dgamma; void* dbeta; int64_t rows, H; };
// Gradient buckets of a multi launch (data-parallel steps): problems / LayerNorm jobs carry a bucket id (non-decreasing along
// the problem list, < 24); the launch finishes the buckets in order and writes `epoch` to flag[k] (signal memory, one word per
// bucket: hipExtMallocWithFlags(hipMallocSignalMemory)) when bucket k is complete.  count: nb zeroed device words.
struct MultiBuckets { int nb; const int* prob_bucket; const int* ln_bucket; unsigned* const* flag; unsigned* count; unsigned epoch; };
int gemm_wgrad_multi(int n, const void* const* dy, const void* const* x, void* const* dw, void* const* db, int64_t M,
                     const int64_t* N, const int64_t* K, int accumulate, hipStream_t st, int n_ln = 0, const LnColsJob* ln = nullptr,
                     const MultiBuckets* buckets = nullptr, float** sq_out = nullptr, int* sq_n = nullptr);
// (sq_out / sq_n: when given and the launch is not bucketed, *sq_out receives a library-owned device array of *sq_n floats that the
//  launch fills with one sum of squares of the stored gradient values per weight tile — the weights' share of the gradient norm)
int gemm_group_autotune(int n, int64_t M, const int64_t* N, const int64_t* K, hipStream_t st);
void gemm_debug_force(int cfg, int splits);
int gemm_autotune(int kind, int64_t M, int64_t N, int64_t K, hipStream_t st);
int gemm_tuned_choice(int kind, int64_t M, int64_t N, int64_t K, int* cfg, int* splits);
int gemm_set_tuned(int kind, int64_t M, int64_t N, int64_t K, int cfg, int splits);
int gemm_autotune_candidates(int kind, int64_t M, int64_t N, int64_t K, int* cfgs, int* splits, int cap);
void gemm_set_num_cus(int n);
int gemm_tile_count();

// ---- attention.hip ----
int attention_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse,
                  int64_t B, int64_t L, int64_t heads, const DropoutCfg& drop, hipStream_t st,
                  const int32_t* cu = nullptr, ChainStep* chain = nullptr);
// whether attention can be a link of an overlapped chain at this shape (dense examples of whole 32-row units, one launch)
bool attention_chainable(int64_t L, const int32_t* cu);
size_t attention_bwd_workspace_bytes(int64_t B, int64_t L, int64_t heads);
int attention_bwd(const void* qkv, const float* mask_bias, const void* ctx, const float* lse,
                  const void* dctx, void* dqkv, int64_t B, int64_t L, int64_t heads,
                  const DropoutCfg& drop, hipStream_t st, const int32_t* cu = nullptr, void* workspace = nullptr,
                  ChainStep* chain = nullptr);

// ---- layernorm.hip ----
int layernorm_fwd(const void* z, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                  int64_t rows, int64_t H, float eps, const DropoutCfg& drop, hipStream_t st, ChainStep* chain = nullptr);
size_t layernorm_bwd_workspace_bytes(int64_t rows, int64_t H);
int layernorm_bwd(const void* dy, const void* dy_extra, const void* z, const float* mean, const float* rstd,
                  const void* gamma, void* dz, void* dd, void* dgamma, void* dbeta, void* dbias,
                  int64_t rows, int64_t H, int accumulate, const DropoutCfg& drop, int post_drop,
                  void* workspace, size_t ws_bytes, hipStream_t st);
int layernorm_bwd_rows(const void* dy, const void* dy_extra, const void* z, const float* mean, const float* rstd,
                       const void* gamma, void* dz, void* dd, int64_t rows, int64_t H, const DropoutCfg& drop,
                       int post_drop, hipStream_t st, ChainStep* chain = nullptr);
int layernorm_bwd_cols(const void* dy, const void* dy_extra, const void* z, const float* mean, const float* rstd,
                       const void* dz, const void* dd, void* dgamma, void* dbeta, void* dbias,
                       int64_t rows, int64_t H, int accumulate, const DropoutCfg& drop, int post_drop,
                       void* workspace, size_t ws_bytes, hipStream_t st);
size_t colsum_workspace_bytes(int64_t rows, int64_t N);
int colsum(const void* a, void* out, int64_t rows, int64_t N, int accumulate,
           void* workspace, size_t ws_bytes, hipStream_t st);

}  // namespace uh
