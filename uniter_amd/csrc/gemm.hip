// gemm.hip — bf16 MFMA GEMM family for gfx950 with fused epilogues.
//
// One kernel template covers the three contraction layouts the encoder needs
// (reference: nn.Linear forward model/layer.py:76-78,112,140,153 and its autograd dgrad/wgrad):
//
//   C[M,N] = sum_k R(m,k) * Cc(n,k)
//     fwd   : R = x  [M][K]   (k contiguous)          Cc = w  [N][K]   (k contiguous)
//     dgrad : R = dy [M][K]   (k contiguous)          Cc = w  [K][N]   (n contiguous)  -> TRB
//     wgrad : R = dy [K][M]   (m contiguous) -> TRA   Cc = x  [K][N]   (n contiguous)  -> TRB
//
// Tile BM x BN x 64, 256 threads = 4 waves (2 x 2), v_mfma_f32_16x16x32_bf16.  The N side feeds the
// MFMA "A" operand and the M side the "B" operand, so every lane ends up with 4 consecutive output
// columns of one output row (8-byte bf16 stores, 16-byte fp32 partial stores).
// LDS: double buffered, filled by direct global->LDS DMA (global_load_lds_dwordx4; partial K tiles fall back
// to register staging with zero fill), one barrier per K tile with the next tile's DMA in flight under the
// MFMAs.  K-contiguous tiles are stored [rows][64] with a 16-byte-chunk XOR swizzle so ds_read_b128
// fragment reads are bank-conflict free; K-strided ("transposed") tiles are stored [64][W] and read
// with ds_read_b64_tr_b16, with a 32-byte-block XOR swizzle that makes those reads conflict free.
#include "common.cuh"
#include "kernels.h"

namespace {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_DROP_RES = 2, EPI_RES = 3, EPI_GELU_BWD = 4, EPI_WGRAD = 5 };

struct GemmArgs {
    const bf16_t* R;      // M-side operand
    const bf16_t* Cc;     // N-side operand
    int64_t ldr, ldcc;    // leading dimensions (elements)
    bf16_t* C;            // output [M][N]
    bf16_t* C2;           // second output (EPI_BIAS_GELU: g)
    int64_t ldc;
    const bf16_t* bias;   // [N] or nullptr
    const bf16_t* aux;    // residual [M][N] / pre-activation u [M][N] / nullptr
    int64_t ldaux;
    float* partial;       // split-K fp32 partials [splits][M][N] (nullptr when splits == 1)
    int M, N, K;          // K = contraction length
    int k_per_split;      // multiple of 64
    int accumulate;       // EPI_WGRAD, splits == 1: C += result
    DropoutCfg drop;
};

// ---- LDS layouts -------------------------------------------------------------------------------
// K-contiguous tile: [rows][64] bf16, 8 chunks of 16 B per row, chunk c stored at c ^ ((row>>1)&7).
__device__ __forceinline__ int kc_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3);
}
// K-strided tile: [64][W] bf16 (W = 64 or 128).  8-byte chunk ch of row r stored at ch ^ (h(r) << 2).
template <int W>
__device__ __forceinline__ int ks_swz(int r) {
    if (W == 128) return (r & 3) | (((r >> 3) & 1) << 2);
    return ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
}
template <int W>
__device__ __forceinline__ int ks_off8(int r, int ch8) {   // element offset of 8-byte chunk ch8 of row r
    return r * W + ((ch8 ^ (ks_swz<W>(r) << 2)) << 2);
}

__device__ __forceinline__ bf16x8 lds_read_b128(const bf16_t* p) {
    return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ s16x4 lds_read_tr(const bf16_t* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

// fragment (8 k-values for row/col `i` of a 16-wide sub-tile) from a K-contiguous tile
__device__ __forceinline__ bf16x8 frag_kc(const bf16_t* tile, int row, int ks, int g) {
    return lds_read_b128(tile + kc_off(row, ks * 4 + g));
}
// same from a K-strided tile: lane (g, i = 4j+q) supplies row ks*32+8g+j (+4), cols cb+4q..
template <int W>
__device__ __forceinline__ bf16x8 frag_ks(const bf16_t* tile, int cb, int ks, int g, int i) {
    const int j = i >> 2, q = i & 3;
    const int r0 = ks * 32 + 8 * g + j;
    const int ch = (cb >> 2) + q;
    const s16x4 lo = lds_read_tr(tile + ks_off8<W>(r0, ch));
    const s16x4 hi = lds_read_tr(tile + ks_off8<W>(r0 + 4, ch));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// ---- staging -------------------------------------------------------------------------------------
// K-contiguous tile of ROWS rows: ROWS/32 16-byte loads per thread.
template <int ROWS>
struct StageKC {
    static constexpr int NL = ROWS / 32;
    u32x4 v[NL];
    __device__ __forceinline__ void load(const bf16_t* base, int64_t ld, int row0, int rows_total,
                                         int k0, int k_end, int t) {
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int idx = it * 256 + t;
            const int row = idx >> 3, c = idx & 7;
            const int gr = row0 + row, gk = k0 + c * 8;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (gr < rows_total && gk < k_end)
                z = *reinterpret_cast<const u32x4*>(base + (int64_t)gr * ld + gk);
            v[it] = z;
        }
    }
    __device__ __forceinline__ void store(bf16_t* tile, int t) const {
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int idx = it * 256 + t;
            const int row = idx >> 3, c = idx & 7;
            *reinterpret_cast<u32x4*>(tile + kc_off(row, c)) = v[it];
        }
    }
};
// K-strided tile [64][W]: W/32 16-byte loads per thread.
template <int W>
struct StageKS {
    static constexpr int NL = W / 32;
    static constexpr int CPR = W / 8;   // 16-byte chunks per row
    u32x4 v[NL];
    __device__ __forceinline__ void load(const bf16_t* base, int64_t ld, int col0, int k0, int k_end, int t) {
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int idx = it * 256 + t;
            const int row = idx / CPR, c = idx % CPR;
            const int gk = k0 + row;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (gk < k_end) z = *reinterpret_cast<const u32x4*>(base + (int64_t)gk * ld + col0 + c * 8);
            v[it] = z;
        }
    }
    __device__ __forceinline__ void store(bf16_t* tile, int t) const {
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int idx = it * 256 + t;
            const int row = idx / CPR, c = idx % CPR;
            *reinterpret_cast<u32x4*>(tile + ks_off8<W>(row, 2 * c)) = v[it];
        }
    }
};

// ---- direct global -> LDS staging (global_load_lds_dwordx4) ---------------------------------------
// One wave instruction moves 64 lanes x 16 B = 1 KiB to LDS base + lane*16 (the destination is lane-linear by
// hardware), so the XOR swizzle is applied on the SOURCE address: lane l, which lands in physical 16-byte
// chunk c' of row r, fetches the logical chunk c = c' ^ swz(r).  Lanes of one row still read one contiguous
// 128/256-byte row segment, so HBM/L2 coalescing is unchanged.  Rows beyond the matrix are clamped to the last
// valid row (their products land in output rows the epilogue never stores); partial K tiles do not use this
// path (they need zero fill).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

__device__ __forceinline__ void glds16(const bf16_t* src, bf16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((global_void_t*)src, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

// K-contiguous tile [ROWS][64]: instruction j covers rows 8j..8j+7.
template <int ROWS>
__device__ __forceinline__ void glds_kc(bf16_t* tile, const bf16_t* base, int64_t ld, int row0, int rows_total,
                                        int k0, int wid, int lane) {
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it) {
        const int j = it * 4 + wid;
        const int r = 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;
        glds16(base + (int64_t)gr * ld + k0 + c * 8, tile + j * 512);
    }
}
// K-strided tile [64][W]: W = 128 -> 4 rows per instruction, W = 64 -> 8 rows per instruction.
template <int W>
__device__ __forceinline__ void glds_ks(bf16_t* tile, const bf16_t* base, int64_t ld, int col0, int k0, int wid, int lane) {
    constexpr int RPI = 1024 / (2 * W);          // rows per instruction
    constexpr int NI = 64 / RPI;                 // instructions per tile
    constexpr int CPR = W / 8;                   // 16-byte chunks per row
#pragma unroll
    for (int it = 0; it < NI / 4; ++it) {
        const int j = it * 4 + wid;
        const int r = RPI * j + lane / CPR;
        const int c = (lane % CPR) ^ (ks_swz<W>(r) << 1);
        glds16(base + (int64_t)(k0 + r) * ld + col0 + c * 8, tile + j * 512);
    }
}

template <int ROWS, bool TR>
struct Stage;
template <int ROWS>
struct Stage<ROWS, false> : StageKC<ROWS> {};
template <int ROWS>
struct Stage<ROWS, true> : StageKS<ROWS> {};

// bijective XCD-aware block remap (cdna_hip_programming.md T1): consecutive hardware block ids go to
// different XCDs; give each XCD a contiguous range of logical tiles so neighbours share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + loc;
}

template <int BM, int BN, bool TRA, bool TRB, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    constexpr int TILE_R = BM * 64, TILE_C = BN * 64;   // elements per LDS tile
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (TILE_R + TILE_C)];

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int k_begin = blockIdx.y * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    const int nk = (k_end - k_begin + 63) >> 6;

    Stage<BM, TRA> sr;
    Stage<BN, TRB> sc;

    auto do_load = [&](int kt) {               // register-staged path (partial K tiles only)
        const int k0 = k_begin + kt * 64;
        if constexpr (TRA) sr.load(p.R, p.ldr, m0, k0, k_end, t);
        else               sr.load(p.R, p.ldr, m0, p.M, k0, k_end, t);
        if constexpr (TRB) sc.load(p.Cc, p.ldcc, n0, k0, k_end, t);
        else               sc.load(p.Cc, p.ldcc, n0, p.N, k0, k_end, t);
    };
    auto do_store = [&](int buf) {
        sr.store(smem + buf * (TILE_R + TILE_C), t);
        sc.store(smem + buf * (TILE_R + TILE_C) + TILE_R, t);
    };
    auto do_glds = [&](int kt, int buf) {       // direct-to-LDS path (full K tiles)
        const int k0 = k_begin + kt * 64;
        bf16_t* tr_ = smem + buf * (TILE_R + TILE_C);
        bf16_t* tc_ = tr_ + TILE_R;
        if constexpr (TRA) glds_ks<BM>(tr_, p.R, p.ldr, m0, k0, wid, lane);
        else               glds_kc<BM>(tr_, p.R, p.ldr, m0, p.M, k0, wid, lane);
        if constexpr (TRB) glds_ks<BN>(tc_, p.Cc, p.ldcc, n0, k0, wid, lane);
        else               glds_kc<BN>(tc_, p.Cc, p.ldcc, n0, p.N, k0, wid, lane);
    };
    auto is_full = [&](int kt) { return k_begin + kt * 64 + 64 <= k_end; };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        if (is_full(0)) {
            do_glds(0, 0);
        } else {
            do_load(0);
            do_store(0);
        }
    }

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed (the barrier's fence drains this wave's outstanding LDS-DMA) and every wave is done
        // reading the buffer the next tile is about to overwrite
        __syncthreads();
        const bool more = (kt + 1 < nk);
        const bool next_full = more && is_full(kt + 1);
        if (more) {
            if (next_full) do_glds(kt + 1, (kt + 1) & 1);
            else           do_load(kt + 1);
        }

        const bf16_t* tr = smem + (kt & 1) * (TILE_R + TILE_C);
        const bf16_t* tc = tr + TILE_R;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fr[MI], fc[NI];
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                if constexpr (TRA) fr[b] = frag_ks<BM>(tr, wm * WM + b * 16, ks, g, i);
                else               fr[b] = frag_kc(tr, wm * WM + b * 16 + i, ks, g);
            }
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                if constexpr (TRB) fc[a] = frag_ks<BN>(tc, wn * WN + a * 16, ks, g, i);
                else               fc[a] = frag_kc(tc, wn * WN + a * 16 + i, ks, g);
            }
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[a], fr[b], acc[a][b], 0, 0, 0);
        }

        if (more && !next_full) do_store((kt + 1) & 1);
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = m0 + wm*WM + b*16 + i, n = n0 + wn*WN + a*16 + 4g ----
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = m0 + wm * WM + b * 16 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int n = n0 + wn * WN + a * 16 + 4 * g;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};

            if (EPI == EPI_WGRAD && p.partial != nullptr) {
                float* dst = p.partial + ((int64_t)blockIdx.y * p.M + m) * p.N + n;
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                continue;
            }
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_DROP_RES) {
                if (p.bias != nullptr) {
                    float bv[4];
                    unpack4(*reinterpret_cast<const u32x2*>(p.bias + n), bv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                }
            }
            bf16_t* cptr = p.C + (int64_t)m * p.ldc + n;
            if (EPI == EPI_BIAS_GELU) {
                *reinterpret_cast<u32x2*>(cptr) = pack4(v);   // u (pre-activation)
                // activation is applied to the bf16-rounded u so that backward (which only has u) is consistent
                float gq[4], uq[4];
                unpack4(pack4(v), uq);
#pragma unroll
                for (int e = 0; e < 4; ++e) gq[e] = gelu_erf(uq[e]);
                *reinterpret_cast<u32x2*>(p.C2 + (int64_t)m * p.ldc + n) = pack4(gq);
                continue;
            }
            if (EPI == EPI_BIAS_DROP_RES) {
                if (p.drop.p > 0.f) {
                    float mult[4];
                    dropout_mult4(p.drop, ((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 2, mult);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= mult[e];
                }
            }
            if (EPI == EPI_BIAS_DROP_RES || EPI == EPI_RES) {
                if (p.aux != nullptr) {
                    float rv[4];
                    unpack4(*reinterpret_cast<const u32x2*>(p.aux + (int64_t)m * p.ldaux + n), rv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
                }
            }
            if (EPI == EPI_GELU_BWD) {
                float uv[4];
                unpack4(*reinterpret_cast<const u32x2*>(p.aux + (int64_t)m * p.ldaux + n), uv);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(uv[e]);
            }
            if (EPI == EPI_WGRAD && p.accumulate) {
                float ov[4];
                unpack4(*reinterpret_cast<const u32x2*>(cptr), ov);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += ov[e];
            }
            *reinterpret_cast<u32x2*>(cptr) = pack4(v);
        }
    }
}

// out[M*N] bf16 (+)= sum_s partial[s][M*N]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, bf16_t* out,
                                                            int64_t mn, int splits, int accumulate) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= mn) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(partial + idx);
    for (int k = 1; k < splits; ++k) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(partial + (int64_t)k * mn + idx);
        s += q;
    }
    float v[4] = {s[0], s[1], s[2], s[3]};
    if (accumulate) {
        float ov[4];
        unpack4(*reinterpret_cast<const u32x2*>(out + idx), ov);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += ov[e];
    }
    *reinterpret_cast<u32x2*>(out + idx) = pack4(v);
}

template <int BM, int BN, bool TRA, bool TRB, int EPI>
int launch_cfg(const GemmArgs& a, int splits, hipStream_t st) {
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    dim3 grid(tiles, splits, 1);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, TRA, TRB, EPI>), grid, dim3(256), 0, st, a);
    UH_LAUNCH_CHECK();
    return 0;
}

// Tile choice.  cfg: 0 = 128x128, 1 = 128x64 (BM x BN), 2 = 64x128, 3 = 64x64.
template <bool TRA, bool TRB, int EPI>
int launch_gemm(const GemmArgs& a, int cfg, int splits, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_cfg<128, 128, TRA, TRB, EPI>(a, splits, st);
        case 1: return launch_cfg<128, 64, TRA, TRB, EPI>(a, splits, st);
        case 2: return launch_cfg<64, 128, TRA, TRB, EPI>(a, splits, st);
        default: return launch_cfg<64, 64, TRA, TRB, EPI>(a, splits, st);
    }
}

int g_force_cfg = -1;      // test / tuning hook (uniter_gemm_debug_force)
int g_force_splits = -1;
int g_num_cus = 256;

// Pick the tile so that the grid fills the chip: prefer 128x128 when it already gives >= ~2 blocks per CU
// worth of work, otherwise shrink the dimension that keeps alignment.  trm: M must be a multiple of BM
// (transposed M-side operand has no column guard).
int pick_cfg(int M, int N, bool trm) {
    if (g_force_cfg >= 0) return g_force_cfg;
    auto ok = [&](int bm, int bn) { return (N % bn == 0) && (!trm || M % bm == 0); };
    auto blocks = [&](int bm, int bn) { return (int64_t)((M + bm - 1) / bm) * (N / bn); };
    const int64_t want = (int64_t)g_num_cus * 3 / 2;
    if (ok(128, 128) && blocks(128, 128) >= want) return 0;
    if (ok(128, 64) && blocks(128, 64) >= want) return 1;
    if (ok(64, 128) && blocks(64, 128) >= want) return 2;
    if (ok(64, 64) && blocks(64, 64) >= want) return 3;
    // not enough tiles to fill the chip at any size: take the smallest legal tile (most blocks)
    if (ok(64, 64)) return 3;
    if (ok(64, 128)) return 2;
    if (ok(128, 64)) return 1;
    return 0;
}

}  // namespace

namespace uh {

void gemm_debug_force(int cfg, int splits) { g_force_cfg = cfg; g_force_splits = splits; }
void gemm_set_num_cus(int n) { if (n > 0) g_num_cus = n; }

static int check_common(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) { uh_set_error("gemm: non-positive dimension"); return -1; }
    if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) { uh_set_error("gemm: dimension too large"); return -1; }
    return 0;
}

int gemm_fwd(int epi, const void* x, const void* w, const void* bias, const void* resid, void* y, void* y2,
             int64_t M, int64_t N, int64_t K, const DropoutCfg& drop, hipStream_t st) {
    if (check_common(M, N, K)) return -1;
    if (N % 64 != 0 || K % 8 != 0) { uh_set_error("gemm_fwd: need N %% 64 == 0 and K %% 8 == 0 (N=%lld K=%lld)", (long long)N, (long long)K); return -1; }
    GemmArgs a{};
    a.R = (const bf16_t*)x; a.ldr = K;
    a.Cc = (const bf16_t*)w; a.ldcc = K;
    a.C = (bf16_t*)y; a.C2 = (bf16_t*)y2; a.ldc = N;
    a.bias = (const bf16_t*)bias;
    a.aux = (const bf16_t*)resid; a.ldaux = N;
    a.partial = nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.k_per_split = (int)((K + 63) / 64 * 64);
    a.accumulate = 0;
    a.drop = drop;
    const int cfg = pick_cfg((int)M, (int)N, false);
    switch (epi) {
        case EPI_BIAS: return launch_gemm<false, false, EPI_BIAS>(a, cfg, 1, st);
        case EPI_BIAS_GELU: return launch_gemm<false, false, EPI_BIAS_GELU>(a, cfg, 1, st);
        case EPI_BIAS_DROP_RES: return launch_gemm<false, false, EPI_BIAS_DROP_RES>(a, cfg, 1, st);
        default: uh_set_error("gemm_fwd: bad epilogue"); return -1;
    }
}

// dx[M,K] = dy[M,N] * w[N,K]  -> output dims (M, K), contraction N
int gemm_dgrad(int epi, const void* dy, const void* w, const void* aux, void* dx,
               int64_t M, int64_t N, int64_t K, hipStream_t st) {
    if (check_common(M, N, K)) return -1;
    if (K % 64 != 0 || N % 8 != 0) { uh_set_error("gemm_dgrad: need K %% 64 == 0 and N %% 8 == 0"); return -1; }
    GemmArgs a{};
    a.R = (const bf16_t*)dy; a.ldr = N;
    a.Cc = (const bf16_t*)w; a.ldcc = K;          // stored [contraction = N][out cols = K]
    a.C = (bf16_t*)dx; a.C2 = nullptr; a.ldc = K;
    a.bias = nullptr;
    a.aux = (const bf16_t*)aux; a.ldaux = K;
    a.partial = nullptr;
    a.M = (int)M; a.N = (int)K; a.K = (int)N;
    a.k_per_split = (int)((N + 63) / 64 * 64);
    a.accumulate = 0;
    a.drop = make_dropout(0.f, 0, 0);
    const int cfg = pick_cfg((int)M, (int)K, false);
    if (epi == EPI_RES) return launch_gemm<false, true, EPI_RES>(a, cfg, 1, st);
    if (epi == EPI_GELU_BWD) return launch_gemm<false, true, EPI_GELU_BWD>(a, cfg, 1, st);
    uh_set_error("gemm_dgrad: bad epilogue");
    return -1;
}

static int wgrad_splits(int64_t M, int64_t N, int64_t K, int cfg) {
    if (g_force_splits > 0) return g_force_splits;
    const int bm = (cfg == 0 || cfg == 1) ? 128 : 64, bn = (cfg == 0 || cfg == 2) ? 128 : 64;
    const int64_t tiles = (N / bm) * (K / bn);
    int64_t ktiles = (M + 63) / 64;
    int s = 1;
    while (tiles * s < g_num_cus && s * 2 <= ktiles / 4 && s < 16) s *= 2;
    return s;
}

size_t gemm_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    // worst case over tile choices: 16 splits
    return (size_t)16 * (size_t)N * (size_t)K * sizeof(float);
}

// dw[N,K] (+)= dy[M,N]^T x[M,K] -> output dims (N, K), contraction M
int gemm_wgrad(const void* dy, const void* x, void* dw, int64_t M, int64_t N, int64_t K, int accumulate,
               void* workspace, size_t ws_bytes, hipStream_t st) {
    if (check_common(M, N, K)) return -1;
    if (N % 64 != 0 || K % 64 != 0) { uh_set_error("gemm_wgrad: need N %% 64 == 0 and K %% 64 == 0 (N=%lld K=%lld)", (long long)N, (long long)K); return -1; }
    GemmArgs a{};
    a.R = (const bf16_t*)dy; a.ldr = N;           // stored [contraction = M][out rows = N]
    a.Cc = (const bf16_t*)x; a.ldcc = K;          // stored [contraction = M][out cols = K]
    a.C = (bf16_t*)dw; a.C2 = nullptr; a.ldc = K;
    a.bias = nullptr; a.aux = nullptr; a.ldaux = 0;
    a.M = (int)N; a.N = (int)K; a.K = (int)M;
    a.accumulate = accumulate;
    a.drop = make_dropout(0.f, 0, 0);
    const int cfg = pick_cfg((int)N, (int)K, true);
    int splits = wgrad_splits(M, N, K, cfg);
    while (splits > 1 && (size_t)splits * N * K * sizeof(float) > ws_bytes) splits >>= 1;
    const int64_t ktiles = (M + 63) / 64;
    a.k_per_split = (int)(((ktiles + splits - 1) / splits) * 64);
    a.partial = splits > 1 ? (float*)workspace : nullptr;
    int rc = launch_gemm<true, true, EPI_WGRAD>(a, cfg, splits, st);
    if (rc) return rc;
    if (splits > 1) {
        const int64_t mn = N * K;
        const int64_t nblk = (mn / 4 + 255) / 256;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, st,
                           (const float*)workspace, (bf16_t*)dw, mn, splits, accumulate);
        UH_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace uh
