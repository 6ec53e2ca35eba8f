// attention_fwd.cuh — LDS tile helpers of the attention kernels and the forward pass of one (example, head) unit,
// shared by attention.hip (one workgroup per unit) and xcd_forward.hip (persistent per-XCD forward).
// Reference arithmetic: model/layer.py:75-101 (BertSelfAttention.forward).
#pragma once
#include "common.cuh"

namespace {

constexpr int DH = 64;
constexpr int LMAX = 256;      // one workgroup holds Q, K, V, dO of a head in LDS up to here (backward); dropout group stride
constexpr int LLONG = 512;     // forward and the split backward (attn_bwd_dq_kernel / attn_bwd_dkv_kernel) reach this

__device__ __forceinline__ int at_off8(int r, int ch8) { return r * 64 + ((ch8 ^ (((r >> 1) & 3) << 2)) << 2); }

__device__ __forceinline__ bf16x8 at_frag(const bf16_t* tile, int row, int ks, int g) {
    return *reinterpret_cast<const bf16x8*>(tile + at_off8(row, 2 * (ks * 4 + g)));
}
__device__ __forceinline__ s16x4 at_read_tr(const bf16_t* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}
// transposed fragment: 8 "row" values (rows 32u+4g+{0..3} and 32u+16+4g+{0..3}) of column dt*16+i
__device__ __forceinline__ bf16x8 at_frag_tr(const bf16_t* tile, int u, int dt, int g, int i) {
    const int j = i >> 2, q = i & 3;
    const int r0 = 32 * u + 4 * g + j;
    const int ch = dt * 4 + q;
    const s16x4 lo = at_read_tr(tile + at_off8(r0, ch));
    const s16x4 hi = at_read_tr(tile + at_off8(r0 + 16, ch));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_frag(const float (&a)[4], const float (&b)[4]) {
    u32x4 w;
    w[0] = pack_bf16x2(a[0], a[1]); w[1] = pack_bf16x2(a[2], a[3]);
    w[2] = pack_bf16x2(b[0], b[1]); w[3] = pack_bf16x2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, w);
}

// A wave holds a [16 rows x 64 columns] fp32 block of one head as o[dt][r] = column 16 dt + 4 g + r of row i.  Stored as it sits that
// is 8 bytes per lane: four store instructions, each 16 row pieces of 32 bytes — and the store tail of these kernels is issue-bound
// (MI355X_MICROARCH.md, "attention epilogue store tail").  Lanes g and g ^ 1 of a row swap one packed half per block pair, so that
// every lane stores 16 bytes: two instructions of 16 x 64-byte row pieces, the same bits at the same addresses.
// row = column 0 of this lane's row (of this head); all 64 lanes must call (the exchange is a wave operation), `valid` gates the store.
__device__ __forceinline__ void store_rows64(bf16_t* row, const f32x4 (&o)[4], const int g, const bool valid, const bool wt) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const float va[4] = {o[2 * pr][0], o[2 * pr][1], o[2 * pr][2], o[2 * pr][3]};
        const float vb[4] = {o[2 * pr + 1][0], o[2 * pr + 1][1], o[2 * pr + 1][2], o[2 * pr + 1][3]};
        const u32x2 a = pack4(va), b = pack4(vb);
        const bool odd = (g & 1) != 0;
        const u32x2 send = odd ? a : b;                      // even g keeps block 2 pr (and takes the partner's), odd g block 2 pr + 1
        u32x2 recv;
        recv[0] = (unsigned)__shfl_xor((int)send[0], 16, WAVE);
        recv[1] = (unsigned)__shfl_xor((int)send[1], 16, WAVE);
        const u32x4 w = odd ? u32x4{recv[0], recv[1], b[0], b[1]} : u32x4{a[0], a[1], recv[0], recv[1]};
        const int col = odd ? (2 * pr + 1) * 16 + 4 * (g - 1) : (2 * pr) * 16 + 4 * g;
        if (valid) out_store16c(row + col, w, wt);
    }
}

// copy rows [0,Lp) x 64 columns starting at src (row stride ld) into an LDS tile, zero beyond L
__device__ __forceinline__ void load_tile(bf16_t* tile, const bf16_t* src, int64_t ld, int L, int Lp) {
    for (int idx = threadIdx.x; idx < Lp * 8; idx += blockDim.x) {
        const int row = idx >> 3, c = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < L) v = *reinterpret_cast<const u32x4*>(src + (int64_t)row * ld + c * 8);
        *reinterpret_cast<u32x4*>(tile + at_off8(row, 2 * c)) = v;
    }
}

// A workgroup stages its [Lp x 64] operand tiles with at most TILE_IT 16-byte chunks per thread and tile (the launch
// picks the wave count so that this holds).  ALL global loads of the prologue are issued before the first LDS write:
// a load -> wait -> write loop would pay one full memory round trip per iteration and tile (measured: 10 serialized
// round trips were most of the backward kernel's 20 us).
constexpr int TILE_IT = 4;
// (tid / nthr: the calling team — the whole workgroup, or the threads of one head when a workgroup holds two)
template <int NIT, bool COH = false>
__device__ __forceinline__ void tile_fetch(u32x4 (&r)[NIT], const bf16_t* src, int64_t ld, int L, int Lp, int tid, int nthr) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * nthr;
        const int row = idx >> 3, c = idx & 7;
        r[it] = u32x4{0u, 0u, 0u, 0u};
        if (idx < Lp * 8 && row < L) r[it] = ldg16<COH>(src + (int64_t)row * ld + c * 8);
    }
}
template <int NIT, bool COH = false>
__device__ __forceinline__ void tile_fetch(u32x4 (&r)[NIT], const bf16_t* src, int64_t ld, int L, int Lp) {
    tile_fetch<NIT, COH>(r, src, ld, L, Lp, (int)threadIdx.x, (int)blockDim.x);
}
template <int NIT>
__device__ __forceinline__ void tile_commit(bf16_t* tile, const u32x4 (&r)[NIT], int Lp, int tid, int nthr) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * nthr;
        if (idx < Lp * 8) *reinterpret_cast<u32x4*>(tile + at_off8(idx >> 3, 2 * (idx & 7))) = r[it];
    }
}
template <int NIT>
__device__ __forceinline__ void tile_commit(bf16_t* tile, const u32x4 (&r)[NIT], int Lp) {
    tile_commit<NIT>(tile, r, Lp, (int)threadIdx.x, (int)blockDim.x);
}

// dropout element groups per query row: one per pair of 16-key tiles, 8 up to L = 256 (the layout every shorter sequence
// has always used), 16 beyond
__host__ __device__ __forceinline__ int pair_stride(int Lm) { return Lm > LMAX ? LLONG / 32 : LMAX / 32; }

struct AttnArgs {
    const bf16_t* qkv;
    const float* mask_bias;
    bf16_t* ctx;        // fwd: out ; bwd: forward output (for D = rowsum(dO*O))
    float* lse;
    float* dsum;        // long backward: D[q] = sum_d dO*O, written by the dQ kernel for the dK/dV kernel  [B*heads, L]
    const bf16_t* dctx;
    bf16_t* dqkv;
    int B, L, heads, Lp;   // L = rows per example (dense) or the longest example (packed); Lp = L rounded up to 32
    const int32_t* cu;     // packed mode: example b owns rows cu[b] .. cu[b+1]-1 of qkv / ctx (NULL = dense [B, L])
    DropoutCfg drop;
    int dbg;               // profiling builds of the harness only (UNITER_AMD_ATTN_DBG): bit 0 = no output stores, 1 = no key sweep, 2 = no query sweep
    ChainLink chain;       // overlapped kernel chain (common.cuh); only dense launches with L % 32 == 0 take part
};

// chain helpers for kernels that run HP (example, head) units per workgroup: the workgroup waits for the rows of the examples its
// units belong to, and every live unit contributes 1 to each 32-row unit of its example (a consumer expects `heads`)
template <int HP>
__device__ __forceinline__ void attn_chain_wait(const AttnArgs& p) {
    if (p.chain.wait == nullptr) return;
    const int last = p.B * p.heads - 1;
    const int bh0 = min((int)blockIdx.x * HP, last), bh1 = min((int)blockIdx.x * HP + HP - 1, last);
    const int b0 = bh0 / p.heads, b1 = bh1 / p.heads;
    chain_wait(p.chain, b0 * p.L, (b1 - b0 + 1) * p.L);
}
template <int HP>
__device__ __forceinline__ void attn_chain_signal(const AttnArgs& p) {
    if (p.chain.signal == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // write-through stores acknowledged
    __syncthreads();
    const int nu = p.L >> 5;
    const int t = (int)threadIdx.x;
    if (t < HP * nu) {
        const int slot = t / nu, u = t - slot * nu;
        const int bh = (int)blockIdx.x * HP + slot;
        if (bh < p.B * p.heads)
            __hip_atomic_fetch_add(p.chain.signal + (bh / p.heads) * nu + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// The query sweep of one (example, head) unit: K, V (swizzled [Lp x 64] tiles) and the additive key mask are in LDS, wave `wid` of
// `nw` takes query tiles wid, wid + nw, ...; fetch_q(qt, qf) supplies a tile's two Q fragments — from global memory (attn_fwd_unit
// below; the first tile's are prefetched with the prologue loads) or from an LDS tile (the fused QKV projection + attention of
// gemm.hip, whose epilogue leaves Q, K, V of its unit in LDS).  ctx_base = first output element of the unit (row 0, column h * 64);
// lse is indexed [bh * Lm + q].  One body for both callers: the same bits from either.
template <int MAXKT, bool COH, typename FetchQ>
__device__ __forceinline__ void attn_fwd_core(const bf16_t* Ks, const bf16_t* Vs, const float* mb, FetchQ&& fetch_q, bf16x8 (&qf)[2],
                                              const bool q_prefetched, const int L, const int Lm, const int Lp, const int bh, const int H,
                                              const DropoutCfg& dcfg, float* lse, bf16_t* ctx_base, const bool wt,
                                              const int wid, const int nw, const int g, const int i) {
#pragma clang fp contract(off)
    const int nkt = Lp >> 4;           // key tiles (even)
    const int nqt = (L + 15) >> 4;
    for (int qt = wid; qt < nqt; qt += nw) {
        const int q = qt * 16 + i;
        if (!(q_prefetched && qt == wid)) fetch_q(qt, qf);
        // S^T tiles: lane holds keys kt*16+4g+{0..3} of query i
        f32x4 s[MAXKT];
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Ks, kt * 16 + i, ks, g), qf[ks], s[kt], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            if (kt < nkt) {
                const f32x4 mv = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = s[kt][r] * 0.125f + mv[r];
                    mx = fmaxf(mx, s[kt][r]);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
        mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[kt][r] - mx);
                    s[kt][r] = e;
                    sum += e;
                }
            }
        }
        sum += __shfl_xor(sum, 16, WAVE);
        sum += __shfl_xor(sum, 32, WAVE);
        const float inv = 1.0f / sum;
        if (g == 0 && q < L && lse != nullptr) lse[(int64_t)bh * Lm + q] = mx + __logf(sum);

        // dropout: one Philox call covers this lane's 4 keys in BOTH tiles of a key-tile pair (element group
        // ((b,h,q) * 8 + pair) * 4 + g; fields 0-3 = tile 2u, 4-7 = tile 2u+1) — the backward pass indexes the same way
        const bool drop = dcfg.p > 0.f;
        const uint64_t drow = ((uint64_t)bh * (uint64_t)Lm + (uint64_t)q) * (uint64_t)pair_stride(Lm);
#pragma unroll
        for (int u = 0; u < MAXKT / 2; ++u) {
            if (2 * u < nkt) {
                float mult[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
                if (drop) dropout_mult8(dcfg, (drow + (uint64_t)u) * 4 + (uint64_t)g, mult);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[2 * u][r] = s[2 * u][r] * inv * mult[r];
                    s[2 * u + 1][r] = s[2 * u + 1][r] * inv * mult[4 + r];
                }
            }
        }
        // O^T[d][query] = sum_keys V^T[d][key] * P^T[key][query]
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < MAXKT / 2; ++u) {
            if (2 * u < nkt) {
                const float a4[4] = {s[2 * u][0], s[2 * u][1], s[2 * u][2], s[2 * u][3]};
                const float b4[4] = {s[2 * u + 1][0], s[2 * u + 1][1], s[2 * u + 1][2], s[2 * u + 1][3]};
                const bf16x8 pf = pack_frag(a4, b4);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Vs, u, dt, g, i), pf, o[dt], 0, 0, 0);
            }
        }
        if constexpr (COH) {
            if (q < L) {
                bf16_t* dst = ctx_base + (int64_t)q * H + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float v[4] = {o[dt][0], o[dt][1], o[dt][2], o[dt][3]};
                    stg8<true>(dst + dt * 16, pack4(v));
                }
            }
        } else {
            store_rows64(ctx_base + (int64_t)q * H, o, g, q < L, wt);
        }
    }
}

// One (example, head) unit, run by all waves of the calling workgroup; the caller owns Lp*64*2*2 + Lp*4 bytes of LDS at
// smem_raw and must put a __syncthreads() between two units that share it.  COH: the unit's qkv rows were written by
// another CU of this XCD inside the same launch (persistent per-XCD forward, xcd_forward.hip) — loads must not be
// served from this CU's L1 (nt loads are L2-served), and the outputs are stored with the default policy so that the
// consuming CU finds them in the XCD's L2.
template <int MAXKT, bool COH>
__device__ __forceinline__ void attn_fwd_unit(const AttnArgs& p, const int bh, char* smem_raw) {
#pragma clang fp contract(off)          // the same bits from every kernel this is inlined into
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Vs = Ks + p.Lp * 64;
    float* mb = reinterpret_cast<float*>(Vs + p.Lp * 64);

    const int b = bh / p.heads, h = bh % p.heads;
    const int H = p.heads * DH;
    const int Lm = p.L, Lp = p.Lp;
    const int64_t row0 = p.cu ? (int64_t)p.cu[b] : (int64_t)b * Lm;
    const int L = p.cu ? (p.cu[b + 1] - p.cu[b]) : Lm;         // real rows of this example
    const int64_t ld = 3 * (int64_t)H;
    const bf16_t* base = p.qkv + row0 * ld + h * DH;

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int nqt = (L + 15) >> 4;

    // prologue: the wave's first Q fragment, the K and V tiles and the mask all leave in one burst
    auto fetch_q = [&](int qt, bf16x8 (&qf)[2]) {
        const int q = qt * 16 + i;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < L) v = ldg16<COH>(base + (int64_t)q * ld + ks * 32 + g * 8);
            qf[ks] = __builtin_bit_cast(bf16x8, v);
        }
    };
    bf16x8 qf[2] = {};
    if (wid < nqt) fetch_q(wid, qf);
    {
        constexpr int NIT = MAXKT > 16 ? 2 * TILE_IT : TILE_IT;
        u32x4 rk[NIT], rv[NIT];
        tile_fetch<NIT, COH>(rk, base + H, ld, L, Lp);
        tile_fetch<NIT, COH>(rv, base + 2 * H, ld, L, Lp);
        float mbv[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = threadIdx.x + it * blockDim.x;
            mbv[it] = (k < L) ? (p.mask_bias ? p.mask_bias[(int64_t)b * Lm + k] : 0.f) : -INFINITY;
        }
        tile_commit(Ks, rk, Lp);
        tile_commit(Vs, rv, Lp);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = threadIdx.x + it * blockDim.x;
            if (k < Lp) mb[k] = mbv[it];
        }
    }
    __syncthreads();

    attn_fwd_core<MAXKT, COH>(Ks, Vs, mb, fetch_q, qf, true, L, Lm, Lp, bh, H, p.drop, p.lse, p.ctx + row0 * H + h * DH,
                              p.chain.signal != nullptr, wid, nw, g, i);
}

}  // namespace
