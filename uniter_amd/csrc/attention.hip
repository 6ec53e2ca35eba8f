// attention.hip — fused BERT self-attention forward / backward for gfx950 (head_dim 64, L <= 256).
//
// Reference arithmetic: model/layer.py:75-101 (BertSelfAttention.forward)
//   S = Q K^T / sqrt(dh) + mask ; P = dropout(softmax(S)) ; ctx = P V, heads merged back to [B,L,H].
//
// One workgroup per (batch, head).  K and V of that head live in LDS for the whole block; each wave
// owns 16-query tiles.  Scores are computed TRANSPOSED (S^T = K Q^T) so that the softmax axis (keys)
// runs over a lane's registers plus the 4 lane groups, and the accumulator registers of two adjacent
// key tiles are directly the MFMA operand of the P·V product (no LDS round trip for P).  V^T / K^T /
// Q^T / dO^T operands come from ds_read_b64_tr_b16.  The backward pass recomputes P from the saved
// log-sum-exp and makes two sweeps: query-tile owners produce dQ, key-tile owners produce dK and dV
// (scores recomputed in the other orientation instead of transposing through LDS).
//
// LDS tile layout (all tiles): [rows][64] bf16, 8-byte chunk ch of row r stored at ch ^ (((r>>1)&3)<<2):
// conflict-free for both ds_read_b128 row fragments and the transposed 8-byte reads used here.
#include "common.cuh"
#include "kernels.h"
#include "attention_fwd.cuh"

namespace {

// Profiling switches of the backward kernels (skip a sweep, skip the stores, cycle stamps through AttnArgs::dsum) exist only in
// builds with -DUNITER_ATTN_PROBE (tests/native/build_probe.sh); in the product library a stray UNITER_AMD_ATTN_DBG cannot
// silently drop gradients or make a kernel write stamps through a null workspace.
#ifdef UNITER_ATTN_PROBE
#define ATTN_DBG(p) ((p).dbg)
#else
#define ATTN_DBG(p) 0
#endif

// ------------------------------------------------------------------------------------------------
// forward: one workgroup per (example, head) — the unit itself lives in attention_fwd.cuh
// ------------------------------------------------------------------------------------------------
template <int MAXKT, int MAXT = 512>
__global__ __launch_bounds__(MAXT) void attn_fwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    attn_chain_wait<1>(p);
    attn_fwd_unit<MAXKT, false>(p, (int)blockIdx.x, smem_raw);
    attn_chain_signal<1>(p);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// HP = (example, head) units per workgroup.  One unit at L = 96 is six 16-row tiles = six waves, and a six-wave workgroup
// leaves two of a CU's four SIMDs with one wave while the other two carry two (the kernel is VALU-issue bound: ~3 000
// instructions per wave, cycle stamps in the harness), and a second workgroup does not fit beside it — 384 workgroups ran as
// two rounds of 256 / 128.  Two units per workgroup = 12 waves = three per SIMD on 192 CUs in ONE round.
template <int HP, int MAXT = (HP == 2 ? 768 : 512)>
__global__ __launch_bounds__(MAXT) void attn_bwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int Lp = p.Lp, Lm = p.L;
    const int nthr = (int)blockDim.x / HP;                     // threads of one unit's team
    const int slot = HP == 2 ? (int)(threadIdx.x >= (unsigned)nthr) : 0;
    const int tid = (int)threadIdx.x - slot * nthr;
    const size_t unit_bytes = ((size_t)Lp * 64 * 2 * 4 + (size_t)Lp * 4 * 3 + (size_t)Lp * (Lp / 4) + 15) & ~(size_t)15;
    char* smem_raw = smem_all + (size_t)slot * unit_bytes;
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Ks = Qs + Lp * 64;
    bf16_t* Vs = Ks + Lp * 64;
    bf16_t* Os = Vs + Lp * 64;            // dO
    float* mb = reinterpret_cast<float*>(Os + Lp * 64);
    float* lse_s = mb + Lp;
    float* D_s = lse_s + Lp;
    uint8_t* keep_s = reinterpret_cast<uint8_t*>(D_s + Lp);     // dropout keep nibbles [q][Lp/4], written by sweep 1
    const int kstride = Lp >> 2;

    const int bh_raw = (int)blockIdx.x * HP + slot;
    const bool live = bh_raw < p.B * p.heads;                   // (an odd unit count leaves the last workgroup's second team idle)
    const int bh = live ? bh_raw : 0;
    const int b = bh / p.heads, h = bh % p.heads;
    const int H = p.heads * DH;
    const int64_t ld = 3 * (int64_t)H;
    const int64_t row0 = p.cu ? (int64_t)p.cu[b] : (int64_t)b * Lm;
    const int L = p.cu ? (p.cu[b + 1] - p.cu[b]) : Lm;         // real rows of this example
    const bf16_t* base = p.qkv + row0 * ld + h * DH;
    const bf16_t* dO = p.dctx + row0 * H + h * DH;
    const bf16_t* O = p.ctx + row0 * H + h * DH;
    unsigned long long* stamp = ((ATTN_DBG(p) & 8) && live) ? reinterpret_cast<unsigned long long*>(p.dsum) + ((size_t)bh * 8 + (tid >> 6)) * 8 : nullptr;
    if (stamp && (tid & 63) == 0) stamp[0] = __builtin_readcyclecounter();
    attn_chain_wait<HP>(p);
    const bool wt = p.chain.signal != nullptr;

    if (live)
    {
        // prologue: Q, K, V, dO, O, the mask and lse all leave in one burst, then go to LDS (see tile_fetch)
        u32x4 rq[TILE_IT], rk[TILE_IT], rv[TILE_IT], rdo[TILE_IT], ro[TILE_IT];
        tile_fetch(rq, base, ld, L, Lp, tid, nthr);
        tile_fetch(rk, base + H, ld, L, Lp, tid, nthr);
        tile_fetch(rv, base + 2 * H, ld, L, Lp, tid, nthr);
        tile_fetch(rdo, dO, H, L, Lp, tid, nthr);
        tile_fetch(ro, O, H, L, Lp, tid, nthr);
        float mbv[2], lsv[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * nthr;
            mbv[it] = (k < L) ? (p.mask_bias ? p.mask_bias[(int64_t)b * Lm + k] : 0.f) : -INFINITY;
            lsv[it] = (k < L) ? p.lse[(int64_t)bh * Lm + k] : INFINITY;
        }
        tile_commit(Qs, rq, Lp, tid, nthr);
        tile_commit(Ks, rk, Lp, tid, nthr);
        tile_commit(Vs, rv, Lp, tid, nthr);
        tile_commit(Os, rdo, Lp, tid, nthr);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * nthr;
            if (k < Lp) { mb[k] = mbv[it]; lse_s[k] = lsv[it]; }
        }
        // D[q] = sum_d dO[q][d] * O[q][d]   (rows >= L were fetched as zeros; whole waves are in or out of range)
#pragma unroll
        for (int it = 0; it < TILE_IT; ++it) {
            const int idx = tid + it * nthr;
            if (idx < Lp * 8) {
                float a[8], o[8];
                unpack8(rdo[it], a);
                unpack8(ro[it], o);
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) part += a[e] * o[e];
                part += __shfl_xor(part, 1, WAVE);
                part += __shfl_xor(part, 2, WAVE);
                part += __shfl_xor(part, 4, WAVE);
                if ((idx & 7) == 0) D_s[idx >> 3] = part;
            }
        }
    }
    if (stamp && (tid & 63) == 0) stamp[1] = __builtin_readcyclecounter();
    __syncthreads();
    if (stamp && (tid & 63) == 0) stamp[2] = __builtin_readcyclecounter();

    const int lane = tid & 63, wid = tid >> 6, nw = nthr >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int npair = Lp >> 5;             // pairs of 16-row tiles
    const int nt = (L + 15) >> 4;          // tiles that contain real rows
    const bool drop = p.drop.p > 0.f;

    // ---- sweep 1: query-tile owners -> dQ ----
    // The row term D[q] = sum_k P~ dP~ is only known once the whole key row has been seen, and this sweep streams over the keys.
    // It runs against D0 (the prologue's dot product with the stored bf16 O), sums the exact D in fp32 on the side and carries
    // one more accumulator PK = P K: dS = dS0 - P (D - D0) / 8, so dQ = dS0 K - (D - D0) / 8 * PK.  The exact D replaces D0 in
    // LDS for the key-owner sweep (see attn_bwd_share_kernel's header for why the stored O is not good enough).
    if (live && !(ATTN_DBG(p) & 4))
    for (int qt = wid; qt < nt; qt += nw) {
        const int q = qt * 16 + i;
        const bf16x8 qf0 = at_frag(Qs, q, 0, g), qf1 = at_frag(Qs, q, 1, g);
        const bf16x8 of0 = at_frag(Os, q, 0, g), of1 = at_frag(Os, q, 1, g);
        const float lse_q = lse_s[q], D_q = D_s[q];
        const uint64_t drow = ((uint64_t)bh * (uint64_t)Lm + (uint64_t)q) * (uint64_t)pair_stride(Lm);
        f32x4 dq[4], pk[4];
        float dacc = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; pk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int u = 0; u < npair; ++u) {
            float ds[2][4], pv[2][4];
            // one Philox call per key-tile pair (same element groups as the forward kernel)
            const uint32_t keep8 = drop ? dropout_keep8(p.drop, (drow + (uint64_t)u) * 4 + (uint64_t)g) : 0xffu;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int kt = 2 * u + hf;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Ks, kt * 16 + i, 0, g), qf0, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Ks, kt * 16 + i, 1, g), qf1, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Vs, kt * 16 + i, 0, g), of0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Vs, kt * 16 + i, 1, g), of1, dp, 0, 0, 0);
                const f32x4 mv = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
                float mult[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop) {
                    // the keep bits are parked in LDS for the key-owner sweep
                    const uint32_t keep = (keep8 >> (4 * hf)) & 0xfu;
                    keep_s[q * kstride + kt * 4 + g] = (uint8_t)keep;
#pragma unroll
                    for (int r = 0; r < 4; ++r) mult[r] = ((keep >> r) & 1u) ? p.drop.scale : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __expf(s[r] * 0.125f + mv[r] - lse_q);
                    const float dpm = dp[r] * mult[r];
                    pv[hf][r] = pr;
                    dacc = fmaf(pr, dpm, dacc);
                    ds[hf][r] = pr * (dpm - D_q) * 0.125f;
                }
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
            const bf16x8 prf = pack_frag(pv[0], pv[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kfr = at_frag_tr(Ks, u, dt, g, i);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, dsf, dq[dt], 0, 0, 0);
                pk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, prf, pk[dt], 0, 0, 0);
            }
        }
        dacc += __shfl_xor(dacc, 16, WAVE);
        dacc += __shfl_xor(dacc, 32, WAVE);
        const float shift = (dacc - D_q) * 0.125f;
        if (g == 0) D_s[q] = dacc;             // (each query row belongs to exactly one wave; the barrier below publishes it)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dq[dt][r] = fmaf(-shift, pk[dt][r], dq[dt][r]);
        store_rows64(p.dqkv + (row0 + q) * ld + h * DH, dq, g, q < L && !((ATTN_DBG(p) & 1) && dq[0][0] != 12345.f), wt);
    }

    if (stamp && lane == 0) { asm volatile("s_nop 0" ::: "memory"); stamp[3] = __builtin_readcyclecounter(); }
    __syncthreads();                   // keep_s and the exact D_s complete before the key-owner sweep reads them
    if (stamp && lane == 0) stamp[4] = __builtin_readcyclecounter();
    // ---- sweep 2: key-tile owners -> dK, dV ----
    if (live && !(ATTN_DBG(p) & 2))
    for (int kt = wid; kt < nt; kt += nw) {
        const int key = kt * 16 + i;
        const bf16x8 kf0 = at_frag(Ks, key, 0, g), kf1 = at_frag(Ks, key, 1, g);
        const bf16x8 vf0 = at_frag(Vs, key, 0, g), vf1 = at_frag(Vs, key, 1, g);
        const float mb_k = mb[key];
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int u = 0; u < npair; ++u) {
            float ds[2][4], pd[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int qt = 2 * u + hf;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                // S[query 4g+r][key i]
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Qs, qt * 16 + i, 0, g), kf0, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Qs, qt * 16 + i, 1, g), kf1, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Os, qt * 16 + i, 0, g), vf0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Os, qt * 16 + i, 1, g), vf1, dp, 0, 0, 0);
                const int qb = qt * 16 + 4 * g;
                const f32x4 lv = *reinterpret_cast<const f32x4*>(lse_s + qb);
                const f32x4 Dv = *reinterpret_cast<const f32x4*>(D_s + qb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mult = 1.f;
                    if (drop) mult = ((keep_s[(qb + r) * kstride + (key >> 2)] >> (key & 3)) & 1u) ? p.drop.scale : 0.f;
                    const float pr = __expf(s[r] * 0.125f + mb_k - lv[r]);
                    pd[hf][r] = pr * mult;
                    ds[hf][r] = pr * (dp[r] * mult - Dv[r]) * 0.125f;
                }
            }
            const bf16x8 pdf = pack_frag(pd[0], pd[1]);
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Os, u, dt, g, i), pdf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Qs, u, dt, g, i), dsf, dk[dt], 0, 0, 0);
            }
        }
        {
            const bool st_ok = key < L && !((ATTN_DBG(p) & 1) && dk[0][0] != 12345.f);
            bf16_t* dst = p.dqkv + (row0 + key) * ld + h * DH;
            store_rows64(dst + H, dk, g, st_ok, wt);
            store_rows64(dst + 2 * H, dv, g, st_ok, wt);
        }
        if (stamp && lane == 0) { asm volatile("s_nop 0" ::: "memory"); stamp[5] = __builtin_readcyclecounter(); }
    }
    if (stamp && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[6] = __builtin_readcyclecounter(); }
    attn_chain_signal<HP>(p);
}

// ------------------------------------------------------------------------------------------------
// backward, L <= 128: the probabilities and score gradients computed ONCE
// ------------------------------------------------------------------------------------------------
// attn_bwd_kernel's key-owner sweep recomputes S, P, dP and dS in the other orientation (24 more MFMAs, 48 more exps and the
// dropout-bit lookups per wave at L = 96) — and the kernel is bound by the instructions a wave issues, not by the matrix pipe.
// Here the query-tile owners keep their tile's P~ = dropout(P) and dS rows as packed bf16 (the rounding the MFMA operands
// get anyway), and once every wave is done with K and V they write them over those two tiles (P~) and into one more tile
// pair (dS) as [query][key] panels of 64 keys in the operand tile layout; the key-tile owners then read their operands with
// the same transposing loads that fetch Q^T and dO^T.  dK / dV see exactly the P~ / dS that dQ saw.
//
// Row term D[q] of the softmax backward (round 4).  dS = P (dP - D) with D = sum_k P~[q][k] dP~[q][k] (= dO[q] . O[q] in exact
// arithmetic).  Flash-style kernels take the dot product from the STORED bf16 O; where the softmax is nearly uniform dP - D is a
// small difference in which the 2^-9 rounding of O shows up amplified (top layers of a 24-layer model: query / key gradients at
// 3x the error of unfused bf16 ops).  The query-tile owner has the whole key row of its queries in registers, so it keeps P and
// dP~ of all its key tiles in fp32, sums P~ dP~ exactly as torch's softmax backward does, and forms dS once, from unrounded
// operands, after the row sum is known — O is not read at all.
// NKT = key tiles a query row can have (6: L <= 96, the two-unit workgroup; 8: L <= 128).
template <int HP, int NKT>
__global__ __launch_bounds__(HP == 2 ? 768 : 512) void attn_bwd_share_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int Lp = p.Lp, Lm = p.L;
    const int nthr = (int)blockDim.x / HP;                     // threads of one unit's team
    const int slot = HP == 2 ? (int)(threadIdx.x >= (unsigned)nthr) : 0;
    const int tid = (int)threadIdx.x - slot * nthr;
    const int npanel = (Lp + 63) >> 6;                         // 64-key panels of the P~ / dS matrices (<= 2)
    const size_t unit_bytes = ((size_t)Lp * 64 * 2 * (4 + npanel) + (size_t)Lp * 4 * 3 + 15) & ~(size_t)15;
    char* smem_raw = smem_all + (size_t)slot * unit_bytes;
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Ks = Qs + Lp * 64;
    bf16_t* Vs = Ks + Lp * 64;
    bf16_t* Os = Vs + Lp * 64;            // dO
    bf16_t* DSs = Os + Lp * 64;           // dS [query][key], npanel panels of [Lp][64]
    bf16_t* PTs = Ks;                     // P~ [query][key], written over K (and V) after the barrier that ends their use
    float* mb = reinterpret_cast<float*>(DSs + (size_t)npanel * Lp * 64);
    float* lse_s = mb + Lp;
    float* D_s = lse_s + Lp;

    const int bh_raw = (int)blockIdx.x * HP + slot;
    const bool live = bh_raw < p.B * p.heads;                   // (an odd unit count leaves the last workgroup's second team idle)
    const int bh = live ? bh_raw : 0;
    const int b = bh / p.heads, h = bh % p.heads;
    const int H = p.heads * DH;
    const int64_t ld = 3 * (int64_t)H;
    const int64_t row0 = p.cu ? (int64_t)p.cu[b] : (int64_t)b * Lm;
    const int L = p.cu ? (p.cu[b + 1] - p.cu[b]) : Lm;         // real rows of this example
    const bf16_t* base = p.qkv + row0 * ld + h * DH;
    const bf16_t* dO = p.dctx + row0 * H + h * DH;
    unsigned long long* stamp = ((ATTN_DBG(p) & 8) && live) ? reinterpret_cast<unsigned long long*>(p.dsum) + ((size_t)bh * 8 + (tid >> 6)) * 8 : nullptr;
    if (stamp && (tid & 63) == 0) stamp[0] = __builtin_readcyclecounter();
    attn_chain_wait<HP>(p);
    const bool wt = p.chain.signal != nullptr;

    if (live) {
        // prologue: Q, K, V, dO, the mask and lse all leave in one burst, then go to LDS (see tile_fetch)
        u32x4 rq[TILE_IT], rk[TILE_IT], rv[TILE_IT], rdo[TILE_IT];
        tile_fetch(rq, base, ld, L, Lp, tid, nthr);
        tile_fetch(rk, base + H, ld, L, Lp, tid, nthr);
        tile_fetch(rv, base + 2 * H, ld, L, Lp, tid, nthr);
        tile_fetch(rdo, dO, H, L, Lp, tid, nthr);
        float mbv[2], lsv[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * nthr;
            mbv[it] = (k < L) ? (p.mask_bias ? p.mask_bias[(int64_t)b * Lm + k] : 0.f) : -INFINITY;
            lsv[it] = (k < L) ? p.lse[(int64_t)bh * Lm + k] : INFINITY;
        }
        tile_commit(Qs, rq, Lp, tid, nthr);
        tile_commit(Ks, rk, Lp, tid, nthr);
        tile_commit(Vs, rv, Lp, tid, nthr);
        tile_commit(Os, rdo, Lp, tid, nthr);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * nthr;
            if (k < Lp) { mb[k] = mbv[it]; lse_s[k] = lsv[it]; }
        }
    }
    if (stamp && (tid & 63) == 0) stamp[1] = __builtin_readcyclecounter();
    __syncthreads();
    if (stamp && (tid & 63) == 0) stamp[2] = __builtin_readcyclecounter();

    const int lane = tid & 63, wid = tid >> 6, nw = nthr >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int npair = Lp >> 5;             // pairs of 16-row tiles (<= 4)
    const int nt = (L + 15) >> 4;          // tiles that contain real rows; the launch gives every one of them its own wave
    const bool drop = p.drop.p > 0.f;
    const bool owner = live && wid < nt;

    // ---- query-tile owners: P~, dS (kept in registers) and dQ ----
    u32x2 ppk[NKT], dspk[NKT];             // this wave's query rows x key tile kt: lane holds keys kt*16 + 4g .. + 3 of query i
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) { ppk[kt] = u32x2{0u, 0u}; dspk[kt] = u32x2{0u, 0u}; }
    if (owner) {
        const int q = wid * 16 + i;
        const bf16x8 qf0 = at_frag(Qs, q, 0, g), qf1 = at_frag(Qs, q, 1, g);
        const bf16x8 of0 = at_frag(Os, q, 0, g), of1 = at_frag(Os, q, 1, g);
        const float lse_q = lse_s[q];
        const uint64_t drow = ((uint64_t)bh * (uint64_t)Lm + (uint64_t)q) * (uint64_t)pair_stride(Lm);
        float dacc = 0.f;                  // sum over this lane's keys of P~ dP~ (fp32, unrounded operands)
        f32x4 pvf[NKT], dpm[NKT];          // P and the dropout-scaled dP of every key tile, until the row sum is known
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            if (u < npair) {
                // one Philox call per key-tile pair (same element groups as the forward kernel)
                const uint32_t keep8 = drop ? dropout_keep8(p.drop, (drow + (uint64_t)u) * 4 + (uint64_t)g) : 0xffu;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int kt = 2 * u + hf;
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Ks, kt * 16 + i, 0, g), qf0, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Ks, kt * 16 + i, 1, g), qf1, s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Vs, kt * 16 + i, 0, g), of0, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Vs, kt * 16 + i, 1, g), of1, dp, 0, 0, 0);
                    const f32x4 mv = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
                    const uint32_t keep = (keep8 >> (4 * hf)) & 0xfu;
                    float pd[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float mult = drop ? (((keep >> r) & 1u) ? p.drop.scale : 0.f) : 1.f;
                        const float pr = __expf(s[r] * 0.125f + mv[r] - lse_q);
                        pvf[kt][r] = pr;
                        dpm[kt][r] = dp[r] * mult;
                        pd[r] = pr * mult;
                        dacc = fmaf(pd[r], dp[r], dacc);
                    }
                    ppk[kt] = pack4(pd);
                }
            }
        }
        // the row term: the other three lane groups hold the rest of query i's keys
        dacc += __shfl_xor(dacc, 16, WAVE);
        dacc += __shfl_xor(dacc, 32, WAVE);
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            if (u < npair) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int kt = 2 * u + hf;
                    float a[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = pvf[kt][r] * (dpm[kt][r] - dacc) * 0.125f;
                    dspk[kt] = pack4(a);
                }
                u32x4 w;
                w[0] = dspk[2 * u][0]; w[1] = dspk[2 * u][1]; w[2] = dspk[2 * u + 1][0]; w[3] = dspk[2 * u + 1][1];
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, w);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Ks, u, dt, g, i), dsf, dq[dt], 0, 0, 0);
            }
        }
        store_rows64(p.dqkv + (row0 + q) * ld + h * DH, dq, g, q < L, wt);
    }
    if (stamp && lane == 0) { asm volatile("s_nop 0" ::: "memory"); stamp[3] = __builtin_readcyclecounter(); }
    __syncthreads();                       // nobody reads K or V any more
    if (live) {
        // rows of this wave's query tile; tiles beyond the real rows (padding up to Lp) are cleared by whoever gets there first
        for (int qt = wid; qt < 2 * npair; qt += nw) {
            const int q = qt * 16 + i;
            const bool mine = (qt == wid) && owner;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt < 2 * npair) {
                    const int off = (kt >> 2) * Lp * 64 + at_off8(q, (kt & 3) * 4 + g);
                    *reinterpret_cast<u32x2*>(PTs + off) = mine ? ppk[kt] : u32x2{0u, 0u};
                    *reinterpret_cast<u32x2*>(DSs + off) = mine ? dspk[kt] : u32x2{0u, 0u};
                }
            }
        }
    }
    __syncthreads();
    if (stamp && lane == 0) stamp[4] = __builtin_readcyclecounter();

    // ---- key-tile owners -> dK, dV ----
    if (live)
    for (int kt = wid; kt < nt; kt += nw) {
        const int key = kt * 16 + i;
        const bf16_t* Pp = PTs + (kt >> 2) * Lp * 64;
        const bf16_t* Dp = DSs + (kt >> 2) * Lp * 64;
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            if (u < npair) {
                const bf16x8 pdf = at_frag_tr(Pp, u, kt & 3, g, i);
                const bf16x8 dsf = at_frag_tr(Dp, u, kt & 3, g, i);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Os, u, dt, g, i), pdf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Qs, u, dt, g, i), dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        {
            bf16_t* dst = p.dqkv + (row0 + key) * ld + h * DH;
            store_rows64(dst + H, dk, g, key < L, wt);
            store_rows64(dst + 2 * H, dv, g, key < L, wt);
        }
    }
    if (stamp && lane == 0) { asm volatile("s_nop 0" ::: "memory"); stamp[5] = __builtin_readcyclecounter(); }
    if (stamp && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[6] = __builtin_readcyclecounter(); }
    attn_chain_signal<HP>(p);
}

// ------------------------------------------------------------------------------------------------
// backward for 256 < L <= 512: Q, K, V and dO of a head no longer fit one CU's LDS together, so the two sweeps of
// attn_bwd_kernel become two launches that each keep only the operands they re-read in LDS (K, V / Q, dO: 128 KiB at
// L = 512) and take their own 16-row tile straight from global memory as MFMA fragments.  The dropout keep bits are
// regenerated from Philox in both (the L x L/4 nibble cache would not fit either).  Same arithmetic per element as the
// one-launch kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void frag_global(const bf16_t* base, int64_t ld, int row, int L, int g, bf16x8 (&f)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < L) v = *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + ks * 32 + g * 8);
        f[ks] = __builtin_bit_cast(bf16x8, v);
    }
}

// query-tile owners -> dQ (and D[q] for the second launch)
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int Lp = p.Lp, Lm = p.L;
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Vs = Ks + Lp * 64;
    float* mb = reinterpret_cast<float*>(Vs + Lp * 64);
    const int bh = (int)blockIdx.x;
    const int b = bh / p.heads, h = bh % p.heads;
    const int H = p.heads * DH;
    const int64_t ld = 3 * (int64_t)H;
    const int64_t row0 = p.cu ? (int64_t)p.cu[b] : (int64_t)b * Lm;
    const int L = p.cu ? (p.cu[b + 1] - p.cu[b]) : Lm;
    const bf16_t* base = p.qkv + row0 * ld + h * DH;
    const bf16_t* dO = p.dctx + row0 * H + h * DH;
    const bf16_t* O = p.ctx + row0 * H + h * DH;
    {
        u32x4 rk[2 * TILE_IT], rv[2 * TILE_IT];
        tile_fetch(rk, base + H, ld, L, Lp);
        tile_fetch(rv, base + 2 * H, ld, L, Lp);
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
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int npair = Lp >> 5;
    const int nt = (L + 15) >> 4;
    const bool drop = p.drop.p > 0.f;
    for (int qt = wid; qt < nt; qt += nw) {
        const int q = qt * 16 + i;
        bf16x8 qf[2], of[2], oo[2];
        frag_global(base, ld, q, L, g, qf);
        frag_global(dO, H, q, L, g, of);
        frag_global(O, H, q, L, g, oo);
        // D0[q] = sum_d dO*O from the stored bf16 O: this lane holds 16 of the 64 columns of row q, the other three lane groups
        // the rest.  The sweep runs against D0 and is corrected by the exact row term at its end (see attn_bwd_kernel).
        float D_q = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) D_q += (float)of[ks][e] * (float)oo[ks][e];
        D_q += __shfl_xor(D_q, 16, WAVE);
        D_q += __shfl_xor(D_q, 32, WAVE);
        const float lse_q = (q < L) ? p.lse[(int64_t)bh * Lm + q] : INFINITY;
        const uint64_t drow = ((uint64_t)bh * (uint64_t)Lm + (uint64_t)q) * (uint64_t)pair_stride(Lm);
        f32x4 dq[4], pk[4];
        float dacc = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; pk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int u = 0; u < npair; ++u) {
            float ds[2][4], pv[2][4];
            const uint32_t keep8 = drop ? dropout_keep8(p.drop, (drow + (uint64_t)u) * 4 + (uint64_t)g) : 0xffu;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int kt = 2 * u + hf;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Ks, kt * 16 + i, 0, g), qf[0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Ks, kt * 16 + i, 1, g), qf[1], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Vs, kt * 16 + i, 0, g), of[0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Vs, kt * 16 + i, 1, g), of[1], dp, 0, 0, 0);
                const f32x4 mv = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
                const uint32_t keep = (keep8 >> (4 * hf)) & 0xfu;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float mult = drop ? (((keep >> r) & 1u) ? p.drop.scale : 0.f) : 1.f;
                    const float pr = __expf(s[r] * 0.125f + mv[r] - lse_q);
                    const float dpm = dp[r] * mult;
                    pv[hf][r] = pr;
                    dacc = fmaf(pr, dpm, dacc);
                    ds[hf][r] = pr * (dpm - D_q) * 0.125f;
                }
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
            const bf16x8 prf = pack_frag(pv[0], pv[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kfr = at_frag_tr(Ks, u, dt, g, i);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, dsf, dq[dt], 0, 0, 0);
                pk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, prf, pk[dt], 0, 0, 0);
            }
        }
        dacc += __shfl_xor(dacc, 16, WAVE);
        dacc += __shfl_xor(dacc, 32, WAVE);
        const float shift = (dacc - D_q) * 0.125f;
        if (g == 0 && q < L) p.dsum[(int64_t)bh * Lm + q] = dacc;     // the exact row term, for the dK / dV launch
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dq[dt][r] = fmaf(-shift, pk[dt][r], dq[dt][r]);
        if (q < L) {
            bf16_t* dst = p.dqkv + (row0 + q) * ld + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const float v[4] = {dq[dt][0], dq[dt][1], dq[dt][2], dq[dt][3]};
                __builtin_nontemporal_store(pack4(v), reinterpret_cast<u32x2*>(dst + dt * 16));
            }
        }
    }
}

// key-tile owners -> dK, dV
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int Lp = p.Lp, Lm = p.L;
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* Os = Qs + Lp * 64;            // dO
    float* lse_s = reinterpret_cast<float*>(Os + Lp * 64);
    float* D_s = lse_s + Lp;
    const int bh = (int)blockIdx.x;
    const int b = bh / p.heads, h = bh % p.heads;
    const int H = p.heads * DH;
    const int64_t ld = 3 * (int64_t)H;
    const int64_t row0 = p.cu ? (int64_t)p.cu[b] : (int64_t)b * Lm;
    const int L = p.cu ? (p.cu[b + 1] - p.cu[b]) : Lm;
    const bf16_t* base = p.qkv + row0 * ld + h * DH;
    const bf16_t* dO = p.dctx + row0 * H + h * DH;
    {
        u32x4 rq[2 * TILE_IT], rdo[2 * TILE_IT];
        tile_fetch(rq, base, ld, L, Lp);
        tile_fetch(rdo, dO, H, L, Lp);
        float lsv[2], dv2[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = threadIdx.x + it * blockDim.x;
            lsv[it] = (k < L) ? p.lse[(int64_t)bh * Lm + k] : INFINITY;
            dv2[it] = (k < L) ? p.dsum[(int64_t)bh * Lm + k] : 0.f;
        }
        tile_commit(Qs, rq, Lp);
        tile_commit(Os, rdo, Lp);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = threadIdx.x + it * blockDim.x;
            if (k < Lp) { lse_s[k] = lsv[it]; D_s[k] = dv2[it]; }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int npair = Lp >> 5;
    const int nt = (L + 15) >> 4;
    const bool drop = p.drop.p > 0.f;
    const int pstride = pair_stride(Lm);
    for (int kt = wid; kt < nt; kt += nw) {
        const int key = kt * 16 + i;
        bf16x8 kf[2], vf[2];
        frag_global(base + H, ld, key, L, g, kf);
        frag_global(base + 2 * H, ld, key, L, g, vf);
        const float mb_k = (key < L) ? (p.mask_bias ? p.mask_bias[(int64_t)b * Lm + key] : 0.f) : -INFINITY;
        // position of this lane's key inside the forward's dropout groups: pair kt/2, lane group (key%16)/4, field
        const uint64_t kgrp = (uint64_t)(kt >> 1);
        const uint32_t kg = (uint32_t)((key & 15) >> 2);
        const int kfield = (kt & 1) * 4 + (key & 3);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int u = 0; u < npair; ++u) {
            float ds[2][4], pd[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int qt = 2 * u + hf;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Qs, qt * 16 + i, 0, g), kf[0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Qs, qt * 16 + i, 1, g), kf[1], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Os, qt * 16 + i, 0, g), vf[0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag(Os, qt * 16 + i, 1, g), vf[1], dp, 0, 0, 0);
                const int qb = qt * 16 + 4 * g;
                const f32x4 lv = *reinterpret_cast<const f32x4*>(lse_s + qb);
                const f32x4 Dv = *reinterpret_cast<const f32x4*>(D_s + qb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mult = 1.f;
                    if (drop) {
                        const uint64_t drow = ((uint64_t)bh * (uint64_t)Lm + (uint64_t)(qb + r)) * (uint64_t)pstride;
                        const u32x4 bits = dropout_bits8(p.drop, (drow + kgrp) * 4 + (uint64_t)kg);
                        mult = dropout_keep_field(p.drop, bits, kfield) ? p.drop.scale : 0.f;
                    }
                    const float pr = __expf(s[r] * 0.125f + mb_k - lv[r]);
                    pd[hf][r] = pr * mult;
                    ds[hf][r] = pr * (dp[r] * mult - Dv[r]) * 0.125f;
                }
            }
            const bf16x8 pdf = pack_frag(pd[0], pd[1]);
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Os, u, dt, g, i), pdf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_frag_tr(Qs, u, dt, g, i), dsf, dk[dt], 0, 0, 0);
            }
        }
        if (key < L) {
            bf16_t* dst = p.dqkv + (row0 + key) * ld + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const float kv[4] = {dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]};
                const float vv[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
                __builtin_nontemporal_store(pack4(kv), reinterpret_cast<u32x2*>(dst + H + dt * 16));
                __builtin_nontemporal_store(pack4(vv), reinterpret_cast<u32x2*>(dst + 2 * H + dt * 16));
            }
        }
    }
}

int attn_dbg() {
#ifdef UNITER_ATTN_PROBE
    static const int v = [] { const char* e = getenv("UNITER_AMD_ATTN_DBG"); return e ? atoi(e) : 0; }();
    return v;
#else
    return 0;
#endif
}

int pick_waves(int nt, int maxw = 8) {
    const int rounds = (nt + maxw - 1) / maxw;
    for (int w = 1; w <= maxw; ++w)
        if ((nt + w - 1) / w == rounds) return w < 2 ? 2 : w;
    return maxw;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { uh_set_error("attention: hipFuncSetAttribute(%zu) -> %s", bytes, hipGetErrorString(e)); return (int)e; }
    }
    return 0;
}

}  // namespace

namespace uh {

static int check(int64_t B, int64_t L, int64_t heads) {
    if (B <= 0 || L <= 0 || heads <= 0) { uh_set_error("attention: non-positive dimension"); return -1; }
    if (L > LLONG) { uh_set_error("attention: L=%lld exceeds the supported maximum %d", (long long)L, LLONG); return -1; }
    if (B * heads > INT32_MAX) { uh_set_error("attention: grid too large"); return -1; }
    return 0;
}

bool attention_chainable(int64_t L, const int32_t* cu) { return cu == nullptr && L % 32 == 0 && L <= LMAX; }

// binds a chain step to a launch's arguments; steps this shape cannot honour are turned into plain in-order launches
static void attn_bind(AttnArgs& a, ChainStep*& chain, int64_t L, const int32_t* cu, int64_t heads) {
    a.chain = ChainLink{nullptr, nullptr, nullptr, 0, 0};
    if (chain == nullptr) return;
    if (!attention_chainable(L, cu)) { chain->anyorder = 0; chain->produced = 0; return; }
    a.chain = chain->link;
    chain->produced = (uint32_t)heads;
}

int attention_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse,
                  int64_t B, int64_t L, int64_t heads, const DropoutCfg& drop, hipStream_t st, const int32_t* cu, ChainStep* chain) {
    if (check(B, L, heads)) return -1;
    LaunchTimer lt(TIME_ATTN_FWD, B, L, heads, st);
    AttnArgs a{};
    a.qkv = (const bf16_t*)qkv; a.mask_bias = mask_bias; a.ctx = (bf16_t*)ctx; a.lse = lse;
    a.dctx = nullptr; a.dqkv = nullptr;
    a.B = (int)B; a.L = (int)L; a.heads = (int)heads; a.Lp = (int)((L + 31) / 32 * 32);
    a.drop = drop;
    a.cu = cu;
    a.dbg = attn_dbg();
    attn_bind(a, chain, L, cu, heads);
    if (cu == nullptr && mask_bias == nullptr) { uh_set_error("attention: dense mode needs mask_bias"); return -1; }
    const int nkt = a.Lp / 16;
    const int nqt_all = (int)((L + 15) / 16);
    // (one 16-query tile per wave up to 12 tiles: see attention_bwd)
    const int nw = L > LMAX ? 8 : ((nqt_all > 8 && nqt_all <= 12) ? pick_waves(nqt_all, 12) : pick_waves(nqt_all));
    const int nit = L > LMAX ? 2 * TILE_IT : TILE_IT;
    if (a.Lp * 8 > nit * nw * 64 || a.Lp > 2 * nw * 64) { uh_set_error("attention: prologue staging does not cover L=%lld with %d waves", (long long)L, nw); return -1; }
    const size_t lds = (size_t)a.Lp * 64 * 2 * 2 + (size_t)a.Lp * 4;
    dim3 grid((unsigned)(B * heads)), block(nw * 64);
    int rc;
    if (nkt > 16) {            // 256 < L <= 512: the whole score row of a query tile in registers (128 accumulators)
        if ((rc = set_lds(attn_fwd_kernel<32>, lds))) return rc;
        chain_launch(chain, attn_fwd_kernel<32>, grid, block, lds, st, a);
    } else if (nkt <= 6) {
        if ((rc = set_lds(attn_fwd_kernel<6>, lds))) return rc;
        chain_launch(chain, attn_fwd_kernel<6>, grid, block, lds, st, a);
    } else if (nkt <= 8) {
        if ((rc = set_lds(attn_fwd_kernel<8>, lds))) return rc;
        chain_launch(chain, attn_fwd_kernel<8>, grid, block, lds, st, a);
    } else if (nkt <= 12 && nw > 8) {
        if ((rc = set_lds(attn_fwd_kernel<12, 768>, lds))) return rc;
        chain_launch(chain, attn_fwd_kernel<12, 768>, grid, block, lds, st, a);
    } else if (nkt <= 12) {
        if ((rc = set_lds(attn_fwd_kernel<12>, lds))) return rc;
        chain_launch(chain, attn_fwd_kernel<12>, grid, block, lds, st, a);
    } else {
        if ((rc = set_lds(attn_fwd_kernel<16>, lds))) return rc;
        chain_launch(chain, attn_fwd_kernel<16>, grid, block, lds, st, a);
    }
    UH_LAUNCH_CHECK();
    return 0;
}

size_t attention_bwd_workspace_bytes(int64_t B, int64_t L, int64_t heads) {
    return L > LMAX ? (size_t)B * (size_t)heads * (size_t)L * sizeof(float) : 0;
}

int attention_bwd(const void* qkv, const float* mask_bias, const void* ctx, const float* lse,
                  const void* dctx, void* dqkv, int64_t B, int64_t L, int64_t heads,
                  const DropoutCfg& drop, hipStream_t st, const int32_t* cu, void* workspace, ChainStep* chain) {
    if (check(B, L, heads)) return -1;
    LaunchTimer lt(TIME_ATTN_BWD, B, L, heads, st);
    AttnArgs a{};
    a.qkv = (const bf16_t*)qkv; a.mask_bias = mask_bias; a.ctx = (bf16_t*)const_cast<void*>(ctx);
    a.lse = const_cast<float*>(lse); a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
    a.B = (int)B; a.L = (int)L; a.heads = (int)heads; a.Lp = (int)((L + 31) / 32 * 32);
    a.drop = drop;
    a.cu = cu;
    a.dbg = attn_dbg();
    attn_bind(a, chain, L, cu, heads);
    if (cu == nullptr && mask_bias == nullptr) { uh_set_error("attention: dense mode needs mask_bias"); return -1; }
    if (L > LMAX) {            // two launches: dQ (+ D) with K, V in LDS, then dK / dV with Q, dO in LDS
        if (workspace == nullptr) { uh_set_error("attention_bwd: L > %d needs the workspace of uniter_attention_bwd_workspace_bytes", LMAX); return -1; }
        a.dsum = (float*)workspace;
        const size_t lds1 = (size_t)a.Lp * 64 * 2 * 2 + (size_t)a.Lp * 4;
        const size_t lds2 = (size_t)a.Lp * 64 * 2 * 2 + (size_t)a.Lp * 4 * 2;
        int rc2;
        if ((rc2 = set_lds(attn_bwd_dq_kernel, lds1))) return rc2;
        if ((rc2 = set_lds(attn_bwd_dkv_kernel, lds2))) return rc2;
        hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(B * heads)), dim3(512), lds1, st, a);
        UH_LAUNCH_CHECK();
        hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(B * heads)), dim3(512), lds2, st, a);
        UH_LAUNCH_CHECK();
        return 0;
    }
    if (a.dbg & 8) {                                    // probe builds only: cycle stamps, [B*heads][8 waves][8]
        if (workspace == nullptr) { uh_set_error("attention_bwd: the stamp probe needs a workspace"); return -1; }
        a.dsum = (float*)workspace;
    }
    // up to 12 waves (three per SIMD at the kernel's register count) when that gives every 16-row tile its own wave: at
    // L = 178 twelve tiles on six waves left two SIMDs with four tile-sweeps and two with two
    const int nt_all = (int)((L + 15) / 16);
    const int nw = (nt_all > 8 && nt_all <= 12) ? pick_waves(nt_all, 12) : pick_waves(nt_all);
    if (a.Lp * 8 > TILE_IT * nw * 64 || a.Lp > 2 * nw * 64) { uh_set_error("attention: prologue staging does not cover L=%lld with %d waves", (long long)L, nw); return -1; }
    const size_t lds = (size_t)a.Lp * 64 * 2 * 4 + (size_t)a.Lp * 4 * 3 + (size_t)a.Lp * (a.Lp / 4);
    int rc;
    const int64_t units = B * heads;
    const int nt = (int)((L + 15) / 16);
    if (a.Lp <= 128 && nt <= nw && !(a.dbg & 16)) {
        // L <= 128: P~ and dS are computed once and handed from the query-tile owners to the key-tile owners through LDS
        const int npanel = (a.Lp + 63) / 64;
        const size_t ub = ((size_t)a.Lp * 64 * 2 * (4 + npanel) + (size_t)a.Lp * 4 * 3 + 15) & ~(size_t)15;
        if (2 * nw <= 12 && 2 * ub <= 156 * 1024 && a.Lp <= 96) {   // two units per workgroup: 12 waves, three per SIMD (see attn_bwd_kernel)
            if ((rc = set_lds(attn_bwd_share_kernel<2, 6>, 2 * ub))) return rc;
            chain_launch(chain, attn_bwd_share_kernel<2, 6>, dim3((unsigned)((units + 1) / 2)), dim3(2 * nw * 64), 2 * ub, st, a);
        } else {
            if ((rc = set_lds(attn_bwd_share_kernel<1, 8>, ub))) return rc;
            chain_launch(chain, attn_bwd_share_kernel<1, 8>, dim3((unsigned)units), dim3(nw * 64), ub, st, a);
        }
        UH_LAUNCH_CHECK();
        return 0;
    }
    // two units per workgroup while that keeps it at 12 waves or fewer (three per SIMD at the kernel's register count)
    const size_t unit_bytes = (lds + 15) & ~(size_t)15;
    if (2 * nw <= 12 && 2 * unit_bytes <= 156 * 1024) {     // (and both units' tiles fit the CU's 160 KiB of LDS)
        if ((rc = set_lds(attn_bwd_kernel<2>, 2 * unit_bytes))) return rc;
        chain_launch(chain, attn_bwd_kernel<2>, dim3((unsigned)((units + 1) / 2)), dim3(2 * nw * 64), 2 * unit_bytes, st, a);
    } else {
        if (nw > 8) {
            if ((rc = set_lds(attn_bwd_kernel<1, 768>, lds))) return rc;
            chain_launch(chain, attn_bwd_kernel<1, 768>, dim3((unsigned)units), dim3(nw * 64), lds, st, a);
        } else {
            if ((rc = set_lds(attn_bwd_kernel<1>, lds))) return rc;
            chain_launch(chain, attn_bwd_kernel<1>, dim3((unsigned)units), dim3(nw * 64), lds, st, a);
        }
    }
    UH_LAUNCH_CHECK();
    return 0;
}

}  // namespace uh
