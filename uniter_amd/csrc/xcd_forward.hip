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
// One tile shape for every GEMM of a layer (96 x 96 x 64, six-stage ring = 147 KB of LDS) and ONE inlined instance of the
// code: the phases differ only in run-time arguments.  The activation of the FFN is its own element-wise phase so that
// all eight waves of a workgroup share its transcendental work instead of the four MFMA waves alone.
constexpr int XBM = 96, XBN = 96, XNS = 6;
constexpr int XSTAGE = (XBM + XBN) * 64;           // bf16 elements per ring slot
static_assert(XNS * XSTAGE * 2 <= X_SMEM, "ring must fit the launch's LDS");

struct XGemm {                             // dense operands: A [M][K], W [N][K], res / C [M][N]
    const bf16_t* A;                       // activations of this group (written earlier in this launch)
    const bf16_t* W;                       // weights
    const bf16_t* bias;                    // [N]
    const bf16_t* res;                     // residual: non-null selects bias + dropout + residual, null bias only
    bf16_t* C;
    int M, N, K;
    int row_base;                          // row of the whole batch that row 0 of the group is (dropout element index)
    int dbg;                               // experiment switches (UNITER_AMD_XCD_DBG): 4 = no MFMA, 8 = no DMA
    unsigned long long* pr;                // debug stamps of this workgroup for this phase (8 words) or null
    DropoutCfg drop;
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int G>
__device__ __forceinline__ void xwait(int younger) {       // this wave's share of a step has landed; `younger` steps stay in flight
    static_assert(4 * G <= 63, "vmcnt is a 6-bit counter");
    if (younger >= 4)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * G) : "memory");
    else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * G) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// C[M,N] = epilogue(A[M,K] * W[N,K]^T): tiles role, role+nx, ... of the group's tile list (M fastest), as one stream of K steps.
__device__ __forceinline__ void xgemm(const XGemm& q, const int role, const int nx, bf16_t* smem) {
#pragma clang fp contract(off)          // the epilogue arithmetic of gemm.hip, rounding for rounding
    constexpr int WM = XBM / 2, WN = XBN / 2, MI = WM / 16, NI = WN / 16;
    constexpr int TILE_R = XBM * 64;
    constexpr int NP = XBM / 32;                         // LDS-DMA pieces per loader wave, operand and K step
    constexpr int G = 2 * NP;
    static_assert(XBM == XBN, "one piece table serves both operands");
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int tiles_m = (q.M + XBM - 1) / XBM, tiles_n = q.N / XBN, ntiles = tiles_m * tiles_n;
    const int ntl = role < ntiles ? (ntiles - role + nx - 1) / nx : 0;
    const int nk = q.K >> 6;
    const int nsteps = ntl * nk;

    if (wid >= 4) {
        // ---- loader waves: step s goes to ring slot s % XNS; one s_barrier per step, shared with the MFMA waves ----
        // The per-lane part of a source address (row, swizzled 16-byte chunk) is a 32-bit byte offset computed once per
        // tile; per K step a piece is "uniform origin + that offset" — the scalar unit advances the origin.
        const int lw = wid - 4;
        const bool rec = q.pr != nullptr && t == 256;
        if (rec) q.pr[4] = wall_clock64();
        uint32_t offw[NP], offa[NP];
        int prow[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int j = it * 4 + lw;
            const int r = 8 * j + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            prow[it] = r;
            offw[it] = (uint32_t)((r * q.K + c * 8) * 2);
            offa[it] = offw[it];
        }
        int lt = role, lkt = 0, slot = 0, issued = 0;
        const char* oa = nullptr;
        const char* ow = nullptr;
        auto new_tile = [&]() {
            const int tm = lt % tiles_m, tn = lt / tiles_m;
            const int m0 = tm * XBM;
            oa = reinterpret_cast<const char*>(q.A) + (int64_t)m0 * q.K * 2;
            ow = reinterpret_cast<const char*>(q.W) + (int64_t)tn * XBN * q.K * 2;
            if (m0 + XBM > q.M) {                         // rows beyond the group: clamp to its last row (never stored)
#pragma unroll
                for (int it = 0; it < NP; ++it) {
                    const int r = prow[it] < q.M - m0 ? prow[it] : q.M - m0 - 1;
                    offa[it] = offw[it] - (uint32_t)((prow[it] - r) * q.K * 2);
                }
            } else {
#pragma unroll
                for (int it = 0; it < NP; ++it) offa[it] = offw[it];
            }
        };
        auto issue = [&]() {
            if (lkt == 0) new_tile();
            bf16_t* tr_ = smem + slot * XSTAGE;
            if (!(q.dbg & 8)) {
#pragma unroll
                for (int it = 0; it < NP; ++it)
                    glds16<16>(reinterpret_cast<const bf16_t*>(oa + offa[it]), tr_ + (it * 4 + lw) * 512);
#pragma unroll
                for (int it = 0; it < NP; ++it)
                    glds16<0>(reinterpret_cast<const bf16_t*>(ow + offw[it]), tr_ + TILE_R + (it * 4 + lw) * 512);
            }
            oa += 128; ow += 128;
            slot = (slot + 1 == XNS) ? 0 : slot + 1;
            if (++lkt == nk) { lkt = 0; lt += nx; }
            ++issued;
        };
#pragma unroll 1
        for (int d = 0; d < XNS - 1; ++d)
            if (issued < nsteps) issue();
        if (rec) q.pr[5] = wall_clock64();
#pragma unroll 1
        for (int j = 0; j < nsteps; ++j) {
            if (j == 1 && rec) q.pr[6] = wall_clock64();
            if (q.dbg & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else xwait<G>(issued - 1 - j);
            __builtin_amdgcn_s_barrier();      // step j is complete in LDS; every MFMA wave is done with step j-1's slot
            if (issued < nsteps) issue();
        }
        if (rec) q.pr[7] = wall_clock64();
        return;
    }

    // ---- MFMA waves (2 x 2) ----
    const int g = lane >> 4, i = lane & 15;
    const int wm = wid >> 1, wn = wid & 1;
    const bool with_res = q.res != nullptr;
    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias and residual of a tile are fetched at the tile's FIRST K step: their latency hides under the main loop
    u32x2 bias_r[NI];
    u32x2 res_r[NI][MI];
    auto prefetch = [&](const int tile) {
        const int m0 = (tile % tiles_m) * XBM, n0 = (tile / tiles_m) * XBN;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int n = n0 + wn * WN + a * 16 + 4 * g;
            bias_r[a] = ldg8<false>(q.bias + n);
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                int m = m0 + wm * WM + b * 16 + i;
                m = m < q.M ? m : q.M - 1;
                res_r[a][b] = with_res ? ldg8<true>(q.res + (int64_t)m * q.N + n) : u32x2{0u, 0u};
            }
        }
    };

    const bool rec = q.pr != nullptr && t == 0;
    if (rec) q.pr[0] = wall_clock64();
    int slot = 0, kt = 0, ct = role;
#pragma unroll 1
    for (int j = 0; j < nsteps; ++j) {
        if (kt == 0) prefetch(ct);
        __builtin_amdgcn_s_barrier();
        if (j == 0 && rec) q.pr[1] = wall_clock64();
        const bf16_t* tr = smem + slot * XSTAGE;
        const bf16_t* tc = tr + TILE_R;
        if (!(q.dbg & 4)) {
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
        }
        slot = (slot + 1 == XNS) ? 0 : slot + 1;
        if (++kt < nk) continue;
        if (j == nsteps - 1 && rec) q.pr[2] = wall_clock64();
        // ---- epilogue of tile ct, straight from the accumulators: lane (g, i) holds C[m][n .. n+3] ----
        kt = 0;
        const int m0 = (ct % tiles_m) * XBM, n0 = (ct / tiles_m) * XBN;
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
                for (int e = 0; e < 4; ++e) v[e] += bv[e];
                if (with_res) {
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
                stg8<true>(q.C + (int64_t)m * q.N + n, pack4(v));
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
    Team tm{a.ctl, a.host_flag, &a.ctl->bar[xcc * 32], 0u, (unsigned)uni((int)s_nx), (unsigned)uni((int)s_role), &s_ok};
    const int role = (int)tm.role, nx = (int)tm.nx;
    const int gi = uni((int)s_gi), ng = uni((int)s_ng);

    // this team's group of consecutive examples and its rows
    const int b0 = (int)(((int64_t)a.B * gi) / ng), b1 = (int)(((int64_t)a.B * (gi + 1)) / ng);
    const int r0 = a.cu ? a.cu[b0] : b0 * a.L, r1 = a.cu ? a.cu[b1] : b1 * a.L;
    const int R = r1 - r0;
    if (R <= 0) return;                                        // the whole team leaves: nobody waits for it
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int H = a.H, I = a.I;
    DropoutCfg nodrop = a.d_hidden;
    nodrop.p = 0.f;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

    auto stamp = [&](int l, int phase, int which) {
        if (a.probe != nullptr && threadIdx.x == 0)
            a.probe[(((size_t)blockIdx.x * X_MAX_LAYERS + l) * 8 + phase) * 2 + which] = wall_clock64();
    };
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
#pragma unroll 1
        for (int ph = 0; ph < 8; ++ph) {
            if (ph == 0 || ph == 2 || ph == 4 || ph == 6) {
                XGemm q;
                q.dbg = a.dbg; q.row_base = r0; q.M = R; q.drop = nodrop; q.res = nullptr;
                q.pr = gpr(l, ph >> 1);
                if (ph == 0) {            // model/layer.py:76-78: Q, K, V projections as one [3H, H] GEMM
                    q.A = x + (int64_t)r0 * H; q.W = P.wqkv; q.bias = P.bqkv; q.C = qkv + (int64_t)r0 * 3 * H; q.N = 3 * H; q.K = H;
                } else if (ph == 2) {     // model/layer.py:112-114: dense + dropout + residual
                    q.A = ctx + (int64_t)r0 * H; q.W = P.wo; q.bias = P.bo; q.res = x + (int64_t)r0 * H; q.C = z1 + (int64_t)r0 * H; q.N = H; q.K = H;
                    q.drop = site_dropout(a.d_hidden, l, 1);
                } else if (ph == 4) {     // model/layer.py:140: dense (its activation is phase 5)
                    q.A = av + (int64_t)r0 * H; q.W = P.w1; q.bias = P.b1; q.C = u + (int64_t)r0 * I; q.N = I; q.K = H;
                } else {                  // model/layer.py:153-155: dense + dropout + residual
                    q.A = gq + (int64_t)r0 * I; q.W = P.w2; q.bias = P.b2; q.res = av + (int64_t)r0 * H; q.C = z2 + (int64_t)r0 * H; q.N = H; q.K = I;
                    q.drop = site_dropout(a.d_hidden, l, 2);
                }
                xgemm(q, role, nx, smem);
            } else if (ph == 1) {         // model/layer.py:80-100: attention, one (example, head) unit at a time
                AttnArgs p{};
                p.qkv = qkv; p.mask_bias = a.cu ? nullptr : a.mask_bias; p.ctx = ctx; p.lse = reinterpret_cast<float*>(A + a.o.lse);
                p.B = a.B; p.L = a.L; p.heads = a.heads; p.Lp = a.Lp; p.cu = a.cu;
                p.drop = site_dropout(a.d_attn, l, 0);
                const int units = (b1 - b0) * a.heads;
                for (int un = role; un < units; un += nx) {
                    attn_fwd_unit<8, true>(p, (b0 + un / a.heads) * a.heads + un % a.heads, smem_raw);
                    __syncthreads();
                }
            } else if (ph == 5) {         // model/layer.py:141: activation, on the bf16-rounded pre-activation as backward will see it
                const int64_t nchunk = (int64_t)R * I / 8;
                const bf16_t* ub = u + (int64_t)r0 * I;
                bf16_t* gb = gq + (int64_t)r0 * I;
                const int64_t stride = (int64_t)nx * XT;
                for (int64_t c = (int64_t)role * XT + threadIdx.x; c < nchunk; c += 3 * stride) {
                    u32x4 v[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (c + k * stride < nchunk) v[k] = ldg16<true>(ub + (c + k * stride) * 8);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (c + k * stride >= nchunk) continue;
                        float f[8];
                        unpack8(v[k], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = act_fwd(a.act, f[e]);
                        *(gmem_u32x4*)(gb + (c + k * stride) * 8) = pack8(f);
                    }
                }
            } else {                      // LayerNorm of the attention block (3) / of the layer (7)
                const bf16_t* z = ph == 3 ? z1 : z2;
                const bf16_t* gam = ph == 3 ? P.ln1_g : P.ln2_g;
                const bf16_t* bet = ph == 3 ? P.ln1_b : P.ln2_b;
                bf16_t* out = ph == 3 ? av : y;
                float* mean = reinterpret_cast<float*>(A + (ph == 3 ? a.o.mean1 : a.o.mean2));
                float* rstd = reinterpret_cast<float*>(A + (ph == 3 ? a.o.rstd1 : a.o.rstd2));
                for (int row = r0 + role * 8 + wid; row < r1; row += nx * 8) {
                    if (H == 768) ln_fwd_row<3, true>(z, gam, bet, out, mean, rstd, row, H, a.eps, nodrop, lane);
                    else          ln_fwd_row<4, true>(z, gam, bet, out, mean, rstd, row, H, a.eps, nodrop, lane);
                }
            }
            stamp(l, ph, 0);
            if (ph == 7 && l + 1 == a.layer_end) break;
            if (!xbar(tm)) return;
            stamp(l, ph, 1);
        }
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
