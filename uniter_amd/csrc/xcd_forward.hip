// xcd_forward.hip — the forward pass of a range of BertLayers as ONE persistent launch, partitioned over the XCDs.
//
// Reference control flow: UniterEncoder.forward (model/model.py:282-292) -> BertLayer.forward (model/layer.py:166-170);
// the arithmetic of every phase is the one of the per-operation kernels (gemm.hip, attention.hip, layernorm.hip) —
// results are bit-identical to uniter_encoder_forward's kernel-per-operation path, which stays the fallback.
//
// Why: at the benchmark shape (32 x 96 tokens, H = 768) a forward layer is seven kernels of 5-30 us whose fixed cost
// (boundary, pipeline ramp, drain: ~5 us each) is a third of their duration.  Every dependency inside a
// BertLayer is local to an example, so the batch is cut into one group of consecutive examples per XCD and the 32
// workgroups that share an XCD (one per CU) walk their group through all phases of all layers:
//     QKV GEMM -> attention -> out-proj + dropout + residual -> LayerNorm -> FFN1 + act -> FFN2 + dropout + residual -> LayerNorm
// Phases meet at a barrier among the workgroups of ONE XCD: a counter in that XCD's L2 and an L1-bypassing poll, 0.9 us
// (tests/native/xcd_probe.cpp, profiles/r02_xcd_probe.log).  Hand-off between phases: plain stores, `s_waitcnt vmcnt(0)`,
// the barrier, then loads that the reader's L1 cannot serve (LDS-DMA with sc1, `nt` register loads) — 0 stale words of 2e7
// per flavour inside an XCD in that probe; no fence, no L2 write-back, the tiles stay in the XCD's L2.  Nothing crosses
// XCDs inside the launch.  Which XCD a workgroup runs on is read from HW_REG_XCC_ID, never inferred from blockIdx: a
// workgroup joins the team of the XCD it finds itself on, teams take the groups in XCD order, so any placement the
// dispatcher chooses gives the same results (an XCD that received no workgroup simply owns no group).  All spins are
// bounded; a timeout makes every workgroup leave and is reported to the host through a mapped flag.
//
// GEMM phases: 96-row tiles (one example of the benchmark shape per tile row), 4 MFMA waves + 4 LDS-DMA loader waves, one
// CONTINUOUS ring of K steps over all tiles a workgroup owns in the phase: the loaders run NSTAGE-1 steps ahead across
// tile boundaries, so only the first tile of a phase pays a pipeline ramp, and the epilogue (registers -> global, no LDS)
// of one tile overlaps the loads of the next.
#include "common.cuh"
#include "kernels.h"
#include "gemm_lds.cuh"
#include "attention_fwd.cuh"
#include "layernorm_fwd.cuh"
#include "../../include/uniter_hip.h"

#include <map>
#include <mutex>

namespace {

constexpr int XT = 512;                    // threads per workgroup: 8 waves
constexpr int X_MAX_LAYERS = 32;
constexpr unsigned X_SPIN_LIMIT = 1u << 21;
constexpr int X_SMEM = 147456;             // 3 x (96 + 288) x 64 x 2 = 4 x (96 + 192) x 64 x 2 = 6 x (96 + 96) x 64 x 2

struct XCtl {                              // zeroed before every launch
    unsigned joined[8];                    // workgroups that found themselves on XCD x
    unsigned total;
    unsigned timeout;
    unsigned pad[22];
    unsigned bar[8 * 32];                  // one 128-byte line per XCD: arrivals so far
};

struct XLayer { const bf16_t *wqkv, *bqkv, *wo, *bo, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b; };

struct XArgs {
    XCtl* ctl;
    unsigned* host_flag;                   // mapped host word: set to 1 on a timeout
    const bf16_t* x_in;
    const float* mask_bias;
    char* acts;
    size_t act_stride;
    uh::XcdActOffsets o;
    const int32_t* cu;
    int B, L, Lp, H, heads, I;
    int layer_begin, layer_end;
    int act;
    int dbg;
    float eps;
    DropoutCfg d_hidden, d_attn;           // offsets = the call's base offset; layer l, site s adds l*8 + s
    unsigned long long* probe;             // debug: wall-clock stamps [workgroup][layer][phase 0..6][work done, barrier passed] (null = off)
    XLayer layer[X_MAX_LAYERS];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Team {
    XCtl* ctl;
    unsigned* host_flag;
    unsigned* cnt;          // this XCD's arrival counter
    unsigned target;        // arrivals after the next barrier
    unsigned nx;            // workgroups of this XCD
    unsigned role;          // 0 .. nx-1
    unsigned* s_ok;         // LDS word
};

__device__ __forceinline__ void flag_timeout(const Team& tm) {
    st_agent(&tm.ctl->timeout, 1u);
    __hip_atomic_store(tm.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Barrier among the workgroups of this XCD.  Every wave first waits for its own stores (they are then in the XCD's L2).
__device__ __forceinline__ bool xbar(Team& tm) {
    tm.target += tm.nx;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(tm.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1;
        for (unsigned spins = 0;; ++spins) {
            if ((int)(ld_agent(tm.cnt) - tm.target) >= 0) break;
            if (spins > X_SPIN_LIMIT || ((spins & 1023u) == 1023u && ld_agent(&tm.ctl->timeout) != 0u)) {
                flag_timeout(tm);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *tm.s_ok = ok;
    }
    __syncthreads();
    return *tm.s_ok != 0u;
}

// ---- GEMM phase ------------------------------------------------------------------------------------------------------
enum { XEPI_BIAS = 0, XEPI_BIAS_ACT = 1, XEPI_BIAS_DROP_RES = 2 };

struct XGemm {                             // dense operands: A [M][K], W [N][K], res / C / C2 [M][N]
    const bf16_t* A;                       // activations of this group (written earlier in this launch)
    const bf16_t* W;                       // weights
    const bf16_t* bias;                    // [N]
    const bf16_t* res;                     // residual (XEPI_BIAS_DROP_RES)
    bf16_t* C; bf16_t* C2;                 // C2: activation output (XEPI_BIAS_ACT)
    int M, N, K;
    int act;
    int dbg;                               // experiment switches (UNITER_AMD_XCD_DBG)
    unsigned long long* pr;                // debug stamps of this workgroup for this phase (8 words) or null
    int row_base;                          // row of the whole batch that row 0 of the group is (dropout element index)
    DropoutCfg drop;
};

// The phases are real (non-inlined) functions so that each gets its own register allocation; their arguments arrive in
// VGPRs although they are wave-uniform: pass them through readfirstlane so that addresses and loop control stay scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
    return (T*)(((uint64_t)hi << 32) | (uint64_t)lo);
}
__device__ __forceinline__ DropoutCfg uni(const DropoutCfg& d) {
    DropoutCfg r;
    r.p = uni(d.p); r.scale = uni(d.scale); r.thresh = uni(d.thresh);
    r.seed_lo = uni(d.seed_lo); r.seed_hi = uni(d.seed_hi); r.off_lo = uni(d.off_lo); r.off_hi = uni(d.off_hi);
    r.off_ptr = uni(d.off_ptr);
    return r;
}
// pointers that went through a call are generic: name the global address space at the access
typedef __attribute__((address_space(1))) u32x2 g_u32x2;
__device__ __forceinline__ u32x2 gload8(const void* p) { return *(const g_u32x2*)p; }
__device__ __forceinline__ u32x2 gload8_nt(const void* p) { return __builtin_nontemporal_load((const g_u32x2*)p); }
__device__ __forceinline__ void gstore8(void* p, const u32x2 v) { *(g_u32x2*)p = v; }

template <int NSTAGE, int G>
__device__ __forceinline__ void xwait(int younger) {       // this wave's share of a step has landed; `younger` steps stay in flight
    static_assert(4 * G <= 63, "vmcnt is a 6-bit counter");
    if (NSTAGE >= 6 && younger >= 4)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * G) : "memory");
    else if (NSTAGE >= 5 && younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * G) : "memory");
    else if (NSTAGE >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
    else if (NSTAGE >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else                                  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// C[M,N] = epilogue(A[M,K] * W[N,K]^T): tiles role, role+nx, ... of the group's tile list (M fastest), as one stream of K steps.
template <int BM, int BN, int NSTAGE, int EPI>
__device__ __noinline__ void xgemm(const XGemm q_in, const int role_in, const int nx_in, bf16_t* smem_in) {
#pragma clang fp contract(off)          // the epilogue arithmetic of gemm.hip, rounding for rounding
    XGemm q;
    q.A = uni(q_in.A); q.W = uni(q_in.W); q.bias = uni(q_in.bias); q.res = uni(q_in.res); q.C = uni(q_in.C); q.C2 = uni(q_in.C2);
    q.M = uni(q_in.M); q.N = uni(q_in.N); q.K = uni(q_in.K); q.act = uni(q_in.act); q.dbg = uni(q_in.dbg); q.pr = uni(q_in.pr); q.row_base = uni(q_in.row_base);
    q.drop = uni(q_in.drop);
    const int role = uni(role_in), nx = uni(nx_in);
    bf16_t* smem = uni(smem_in);
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % 32 == 0 && BN % 32 == 0, "tile shape");
    constexpr int TILE_R = BM * 64, TILE_C = BN * 64, STAGE = TILE_R + TILE_C;
    constexpr int G = BM / 32 + BN / 32;
    static_assert(NSTAGE * STAGE * 2 <= X_SMEM, "ring must fit the launch's LDS");
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int tiles_m = (q.M + BM - 1) / BM, tiles_n = q.N / BN, ntiles = tiles_m * tiles_n;
    const int ntl = role < ntiles ? (ntiles - role + nx - 1) / nx : 0;
    const int nk = q.K >> 6;
    const int nsteps = ntl * nk;

    if (wid >= 4) {
        // ---- loader waves: step s goes to ring slot s % NSTAGE; one s_barrier per step, shared with the MFMA waves ----
        const int lw = wid - 4;
        const bool rec = q.pr != nullptr && t == 256;
        if (rec) q.pr[4] = wall_clock64();
        int lt = role, lkt = 0, slot = 0, issued = 0;
        auto issue = [&]() {
            const int tm = lt % tiles_m, tn = lt / tiles_m;
            bf16_t* tr_ = smem + slot * STAGE;
            if (!(q.dbg & 8)) {
                if ((q.dbg & 3) == 0)      glds_kc<BM, 16>(tr_, q.A, q.K, tm * BM, q.M, lkt * 64, lw, lane);
                else if ((q.dbg & 3) == 1) glds_kc<BM, 0>(tr_, q.A, q.K, tm * BM, q.M, lkt * 64, lw, lane);
                else                       glds_kc<BM, 2>(tr_, q.A, q.K, tm * BM, q.M, lkt * 64, lw, lane);
                if (q.dbg & 16) glds_kc<BN, 2>(tr_ + TILE_R, q.W, q.K, tn * BN, q.N, lkt * 64, lw, lane);
                else            glds_kc<BN, 0>(tr_ + TILE_R, q.W, q.K, tn * BN, q.N, lkt * 64, lw, lane);
            }
            slot = (slot + 1 == NSTAGE) ? 0 : slot + 1;
            if (++lkt == nk) { lkt = 0; lt += nx; }
            ++issued;
        };
#pragma unroll
        for (int d = 0; d < NSTAGE - 1; ++d)
            if (issued < nsteps) issue();
        if (rec) q.pr[5] = wall_clock64();
        for (int j = 0; j < nsteps; ++j) {
            if (j == 1 && rec) q.pr[6] = wall_clock64();
            if (q.dbg & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else xwait<NSTAGE, G>(issued - 1 - j);
            __builtin_amdgcn_s_barrier();      // step j is complete in LDS; every MFMA wave is done with step j-1's slot
            if (issued < nsteps) issue();
        }
        if (rec) q.pr[7] = wall_clock64();
        return;
    }

    // ---- MFMA waves (2 x 2) ----
    const int g = lane >> 4, i = lane & 15;
    const int wm = wid >> 1, wn = wid & 1;
    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias and residual of a tile are fetched at the tile's FIRST K step: their latency hides under the main loop instead of
    // being paid once per 16x16 block in the epilogue (measured: 10-30 us per phase when loaded where they are used)
    u32x2 bias_r[NI];
    u32x2 res_r[NI][MI];
    auto prefetch = [&](const int tile) {
        const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int n = n0 + wn * WN + a * 16 + 4 * g;
            bias_r[a] = q.bias != nullptr ? gload8(q.bias + n) : u32x2{0u, 0u};
            if constexpr (EPI == XEPI_BIAS_DROP_RES) {
#pragma unroll
                for (int b = 0; b < MI; ++b) {
                    int m = m0 + wm * WM + b * 16 + i;
                    m = m < q.M ? m : q.M - 1;
                    res_r[a][b] = gload8_nt(q.res + (int64_t)m * q.N + n);
                }
            }
        }
    };

    const bool rec = q.pr != nullptr && t == 0;
    if (rec) q.pr[0] = wall_clock64();
    int slot = 0, kt = 0, ct = role;
    for (int j = 0; j < nsteps; ++j) {
        if (kt == 0) prefetch(ct);
        __builtin_amdgcn_s_barrier();
        if (j == 0 && rec) q.pr[1] = wall_clock64();
        const bf16_t* tr = smem + slot * STAGE;
        const bf16_t* tc = tr + TILE_R;
        if (!(q.dbg & 4))
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fr[MI], fc[NI];
#pragma unroll
            for (int b = 0; b < MI; ++b) fr[b] = frag_kc(tr, wm * WM + b * 16 + i, ks, g);
#pragma unroll
            for (int a = 0; a < NI; ++a) fc[a] = frag_kc(tc, wn * WN + a * 16 + i, ks, g);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[a], fr[b], acc[a][b], 0, 0, 0);
        }
        slot = (slot + 1 == NSTAGE) ? 0 : slot + 1;
        if (++kt < nk) continue;
        if (j == nsteps - 1 && rec) q.pr[2] = wall_clock64();
        // ---- epilogue of tile ct, straight from the accumulators: lane (g, i) holds C[m][n .. n+3] ----
        kt = 0;
        const int m0 = (ct % tiles_m) * BM, n0 = (ct / tiles_m) * BN;
        ct += nx;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int n = n0 + wn * WN + a * 16 + 4 * g;
            float bv[4];
            unpack4(bias_r[a], bv);
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int m = m0 + wm * WM + b * 16 + i;
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m >= q.M) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bv[e];              // (a null bias was fetched as zeros: + 0.0f changes nothing but -0.0f, which bf16 GEMM outputs never keep apart)
                bf16_t* cptr = q.C + (int64_t)m * q.N + n;
                if constexpr (EPI == XEPI_BIAS_ACT) {
                    const u32x2 ub = pack4(v);
                    gstore8(cptr, ub);                                            // u (pre-activation)
                    float uq[4], gq[4];
                    unpack4(ub, uq);                                              // the activation sees the bf16-rounded u, as backward will
#pragma unroll
                    for (int e = 0; e < 4; ++e) gq[e] = act_fwd(q.act, uq[e]);
                    gstore8(q.C2 + (int64_t)m * q.N + n, pack4(gq));
                } else {
                    if constexpr (EPI == XEPI_BIAS_DROP_RES) {
                        if (q.drop.p > 0.f) {
                            float mv[4];
                            dropout_mult4(q.drop, ((uint64_t)(q.row_base + m) * (uint64_t)q.N + (uint64_t)n) >> 2, mv);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= mv[e];
                        }
                        float rv[4];
                        unpack4(res_r[a][b], rv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    gstore8(cptr, pack4(v));
                }
            }
        }
    }
    if (rec) q.pr[3] = wall_clock64();
}

__device__ __forceinline__ DropoutCfg site_dropout(const DropoutCfg& base, int layer, int site) {
    DropoutCfg d = base;
    const unsigned long long o = (((unsigned long long)base.off_hi << 32) | base.off_lo) + (unsigned long long)layer * 8ull + (unsigned long long)site;
    d.off_lo = (uint32_t)o;
    d.off_hi = (uint32_t)(o >> 32);
    return d;
}

template <int NC>
__device__ __noinline__ void xlayernorm(const bf16_t* z_in, const bf16_t* gamma_in, const bf16_t* beta_in, bf16_t* y_in, float* mean_in, float* rstd_in,
                                        int r0_in, int r1_in, int H_in, float eps_in, int role_in, int nx_in) {
    const bf16_t* z = uni(z_in); const bf16_t* gamma = uni(gamma_in); const bf16_t* beta = uni(beta_in);
    bf16_t* y = uni(y_in); float* mean = uni(mean_in); float* rstd = uni(rstd_in);
    const int r0 = uni(r0_in), r1 = uni(r1_in), H = uni(H_in), role = uni(role_in), nx = uni(nx_in);
    const float eps = uni(eps_in);
    DropoutCfg nodrop{};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int row = r0 + role * 8 + wid; row < r1; row += nx * 8)
        ln_fwd_row<NC, true>(z, gamma, beta, y, mean, rstd, row, H, eps, nodrop, lane);
}

__global__ __launch_bounds__(XT) void xcd_forward_kernel(const XArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ unsigned s_role, s_nx, s_gi, s_ng, s_ok;
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) {
        s_role = __hip_atomic_fetch_add(&a.ctl->joined[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.ctl->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1;
        for (unsigned spins = 0;; ++spins) {                 // until every workgroup of the grid has joined its team
            if (ld_agent(&a.ctl->total) == gridDim.x) break;
            if (spins > X_SPIN_LIMIT || ((spins & 1023u) == 1023u && ld_agent(&a.ctl->timeout) != 0u)) {
                st_agent(&a.ctl->timeout, 1u);
                __hip_atomic_store(a.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        unsigned ng = 0, gi = 0;
        for (unsigned x = 0; x < 8; ++x) {
            const unsigned n = ld_agent(&a.ctl->joined[x]);
            if (n != 0u) { if (x < xcc) ++gi; ++ng; }
        }
        s_nx = ld_agent(&a.ctl->joined[xcc]);
        s_gi = gi;
        s_ng = ng;
        s_ok = ok;
    }
    __syncthreads();
    if (s_ok == 0u) return;
    Team tm{a.ctl, a.host_flag, &a.ctl->bar[xcc * 32], 0u, s_nx, s_role, &s_ok};
    const int role = (int)tm.role, nx = (int)tm.nx;

    // this team's group of consecutive examples and its rows
    const int b0 = (int)(((int64_t)a.B * s_gi) / s_ng), b1 = (int)(((int64_t)a.B * (s_gi + 1)) / s_ng);
    const int r0 = a.cu ? a.cu[b0] : b0 * a.L, r1 = a.cu ? a.cu[b1] : b1 * a.L;
    const int R = r1 - r0;
    if (R <= 0) return;                                        // the whole team leaves: nobody waits for it
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int H = a.H, I = a.I;
    DropoutCfg nodrop = a.d_hidden;
    nodrop.p = 0.f;

    auto stamp = [&](int l, int phase, int which) {
        if (a.probe != nullptr && threadIdx.x == 0)
            a.probe[(((size_t)blockIdx.x * X_MAX_LAYERS + l) * 8 + phase) * 2 + which] = wall_clock64();
    };
    stamp(a.layer_begin, 7, 1);                                // start of the walk
    auto gpr = [&](int l, int k) -> unsigned long long* {      // GEMM-internal stamps of layer 5: behind the phase stamps
        if (a.probe == nullptr || l != 5) return nullptr;
        return a.probe + (size_t)256 * X_MAX_LAYERS * 8 * 2 + ((size_t)blockIdx.x * 4 + k) * 8;
    };
    const bf16_t* x = a.x_in;
    for (int l = a.layer_begin; l < a.layer_end; ++l) {
        const XLayer& P = a.layer[l];
        char* A = a.acts + (size_t)l * a.act_stride;
        bf16_t* qkv = reinterpret_cast<bf16_t*>(A + a.o.qkv);
        bf16_t* ctx = reinterpret_cast<bf16_t*>(A + a.o.ctx);
        bf16_t* z1 = reinterpret_cast<bf16_t*>(A + a.o.z1);
        bf16_t* av = reinterpret_cast<bf16_t*>(A + a.o.a);
        bf16_t* u = reinterpret_cast<bf16_t*>(A + a.o.u);
        bf16_t* gq = reinterpret_cast<bf16_t*>(A + a.o.g);
        bf16_t* z2 = reinterpret_cast<bf16_t*>(A + a.o.z2);
        bf16_t* y = reinterpret_cast<bf16_t*>(A + a.o.y);

        // ---- model/layer.py:76-78: Q, K, V projections as one [3H, H] GEMM ----
        {
            const XGemm q{x + (int64_t)r0 * H, P.wqkv, P.bqkv, nullptr, qkv + (int64_t)r0 * 3 * H, nullptr, R, 3 * H, H, 0, a.dbg, gpr(l, 0), r0, nodrop};
            xgemm<96, 288, 3, XEPI_BIAS>(q, role, nx, smem);
        }
        stamp(l, 0, 0);
        if (!xbar(tm)) return;
        stamp(l, 0, 1);
        // ---- model/layer.py:80-100: attention, one (example, head) unit at a time ----
        {
            AttnArgs p{};
            p.qkv = qkv; p.mask_bias = a.cu ? nullptr : a.mask_bias; p.ctx = ctx; p.lse = reinterpret_cast<float*>(A + a.o.lse);
            p.B = a.B; p.L = a.L; p.heads = a.heads; p.Lp = a.Lp; p.cu = a.cu;
            p.drop = site_dropout(a.d_attn, l, 0);
            const int units = (b1 - b0) * a.heads;
            for (int un = role; un < units; un += nx) {
                attn_fwd_unit<8, true>(p, (b0 + un / a.heads) * a.heads + un % a.heads, smem_raw);
                __syncthreads();
            }
        }
        stamp(l, 1, 0);
        if (!xbar(tm)) return;
        stamp(l, 1, 1);
        // ---- model/layer.py:112-114: dense + dropout + residual ----
        {
            const XGemm q{ctx + (int64_t)r0 * H, P.wo, P.bo, x + (int64_t)r0 * H, z1 + (int64_t)r0 * H, nullptr, R, H, H, 0, a.dbg, gpr(l, 1), r0,
                          site_dropout(a.d_hidden, l, 1)};
            xgemm<96, 96, 6, XEPI_BIAS_DROP_RES>(q, role, nx, smem);
        }
        stamp(l, 2, 0);
        if (!xbar(tm)) return;
        stamp(l, 2, 1);
        if (H == 768) xlayernorm<3>(z1, P.ln1_g, P.ln1_b, av, reinterpret_cast<float*>(A + a.o.mean1), reinterpret_cast<float*>(A + a.o.rstd1), r0, r1, H, a.eps, role, nx);
        else          xlayernorm<4>(z1, P.ln1_g, P.ln1_b, av, reinterpret_cast<float*>(A + a.o.mean1), reinterpret_cast<float*>(A + a.o.rstd1), r0, r1, H, a.eps, role, nx);
        stamp(l, 3, 0);
        if (!xbar(tm)) return;
        stamp(l, 3, 1);
        // ---- model/layer.py:140-141: dense + activation ----
        {
            const XGemm q{av + (int64_t)r0 * H, P.w1, P.b1, nullptr, u + (int64_t)r0 * I, gq + (int64_t)r0 * I, R, I, H, a.act, a.dbg, gpr(l, 2), r0, nodrop};
            xgemm<96, 192, 4, XEPI_BIAS_ACT>(q, role, nx, smem);
        }
        stamp(l, 4, 0);
        if (!xbar(tm)) return;
        stamp(l, 4, 1);
        // ---- model/layer.py:153-155: dense + dropout + residual ----
        {
            const XGemm q{gq + (int64_t)r0 * I, P.w2, P.b2, av + (int64_t)r0 * H, z2 + (int64_t)r0 * H, nullptr, R, H, I, 0, a.dbg, gpr(l, 3), r0,
                          site_dropout(a.d_hidden, l, 2)};
            xgemm<96, 96, 6, XEPI_BIAS_DROP_RES>(q, role, nx, smem);
        }
        stamp(l, 5, 0);
        if (!xbar(tm)) return;
        stamp(l, 5, 1);
        if (H == 768) xlayernorm<3>(z2, P.ln2_g, P.ln2_b, y, reinterpret_cast<float*>(A + a.o.mean2), reinterpret_cast<float*>(A + a.o.rstd2), r0, r1, H, a.eps, role, nx);
        else          xlayernorm<4>(z2, P.ln2_g, P.ln2_b, y, reinterpret_cast<float*>(A + a.o.mean2), reinterpret_cast<float*>(A + a.o.rstd2), r0, r1, H, a.eps, role, nx);
        stamp(l, 6, 0);
        if (l + 1 < a.layer_end && !xbar(tm)) return;
        stamp(l, 6, 1);
        x = y;
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct XResources {
    XCtl* ctl = nullptr;
    unsigned* host_flag = nullptr;       // hipHostMalloc'ed, mapped
    unsigned* host_flag_dev = nullptr;
};
std::mutex g_x_mu;
std::map<std::pair<int, hipStream_t>, XResources> g_x_res;
int g_x_enable = [] { const char* e = getenv("UNITER_AMD_XCD_FWD"); return e ? atoi(e) : 0; }();   // off until it beats the per-operation path
bool g_x_lds_set = false;
unsigned long long* g_x_probe = nullptr;

}  // namespace

namespace uh {

void xcd_forward_enable(int on) { g_x_enable = on; }
void xcd_forward_probe(void* dev) { g_x_probe = (unsigned long long*)dev; }

bool xcd_forward_eligible(const UniterEncoderShape& s, int n_layers) {
    if (!g_x_enable) return false;
    if (s.H != 768 && s.H != 1024) return false;                       // LayerNorm instantiations
    if (s.H % 96 != 0 || (3 * s.H) % 288 != 0 || s.I % 192 != 0) return false;   // tile shapes of the GEMM phases
    if (s.L > 128 || n_layers > X_MAX_LAYERS || n_layers <= 0) return false;
    if (s.B < 8) return false;                                         // fewer examples than XCDs: teams would idle
    return true;
}

int xcd_forward(const UniterEncoderShape* s, const UniterLayerParams* layers, int layer_begin, int layer_end, const void* x_in,
                const float* mask_bias, void* acts, size_t act_stride, const XcdActOffsets& o, uint64_t seed, uint64_t offset,
                hipStream_t st) {
    int dev = 0;
    UH_CHECK_HIP(hipGetDevice(&dev));
    XResources res;
    {
        std::lock_guard<std::mutex> lk(g_x_mu);
        XResources& r = g_x_res[{dev, st}];
        if (r.ctl == nullptr) {
            UH_CHECK_HIP(hipMalloc(&r.ctl, sizeof(XCtl)));
            UH_CHECK_HIP(hipHostMalloc(&r.host_flag, sizeof(unsigned), hipHostMallocMapped));
            *r.host_flag = 0u;
            UH_CHECK_HIP(hipHostGetDevicePointer((void**)&r.host_flag_dev, r.host_flag, 0));
        }
        if (!g_x_lds_set) {
            UH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xcd_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X_SMEM));
            g_x_lds_set = true;
        }
        res = r;
    }
    if (*res.host_flag != 0u) {
        *res.host_flag = 0u;
        uh_set_error("xcd_forward: a previous persistent forward launch on this stream timed out at a team barrier (its outputs are invalid)");
        return -1;
    }
    int cus = 0;
    UH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    XArgs a{};
    a.ctl = res.ctl;
    a.host_flag = res.host_flag_dev;
    a.x_in = (const bf16_t*)x_in;
    a.mask_bias = mask_bias;
    a.acts = (char*)acts;
    a.act_stride = act_stride;
    a.o = o;
    a.cu = s->total_tokens > 0 ? s->cu_seqlens : nullptr;
    a.B = (int)s->B; a.L = (int)s->L; a.Lp = (int)((s->L + 31) / 32 * 32); a.H = (int)s->H; a.heads = (int)s->heads; a.I = (int)s->I;
    a.layer_begin = layer_begin; a.layer_end = layer_end;
    a.act = s->hidden_act;
    a.eps = s->ln_eps;
    const bool tr = s->training != 0;
    a.d_hidden = make_dropout(tr ? s->p_hidden : 0.f, seed, offset);
    a.d_attn = make_dropout(tr ? s->p_attn : 0.f, seed, offset);
    a.probe = g_x_probe;
    { const char* e = getenv("UNITER_AMD_XCD_DBG"); a.dbg = e ? atoi(e) : 0; }
    for (int l = layer_begin; l < layer_end; ++l) {
        const UniterLayerParams& P = layers[l];
        a.layer[l] = XLayer{(const bf16_t*)P.wqkv, (const bf16_t*)P.bqkv, (const bf16_t*)P.wo, (const bf16_t*)P.bo, (const bf16_t*)P.ln1_g,
                            (const bf16_t*)P.ln1_b, (const bf16_t*)P.w1, (const bf16_t*)P.b1, (const bf16_t*)P.w2, (const bf16_t*)P.b2,
                            (const bf16_t*)P.ln2_g, (const bf16_t*)P.ln2_b};
    }
    UH_CHECK_HIP(hipMemsetAsync(res.ctl, 0, sizeof(XCtl), st));
    hipLaunchKernelGGL(xcd_forward_kernel, dim3((unsigned)cus), dim3(XT), X_SMEM, st, a);
    UH_LAUNCH_CHECK();
    return 0;
}

}  // namespace uh
