// embed.hip — text / image embedding kernels and the compaction gather (all HBM-bound, tiny).
//
// Reference: UniterTextEmbeddings.forward model/model.py:232-245, UniterImageEmbeddings.forward
// model/model.py:261-272, UniterModel._compute_img_txt_embeddings model/model.py:321-334 and the
// additive mask model/model.py:342-345.  The LayerNorms (+dropout) of those blocks are layernorm.hip,
// the 2048->H image projection is gemm.hip; this file holds the gathers, the K=7 position projection,
// the elementwise glue and the deterministic scatter-adds of their backward passes.
#include "common.cuh"
#include "kernels.h"

namespace {

// ---- text: z = word[id] + pos[pid] + type[tt] -------------------------------------------------------
__global__ __launch_bounds__(256) void embed_txt_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pids,
                                                            const int64_t* __restrict__ tids, const bf16_t* __restrict__ word,
                                                            const bf16_t* __restrict__ pos, const bf16_t* __restrict__ type,
                                                            bf16_t* __restrict__ z, int n_rows, int Lt, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t id = ids[row], pid = pids[row % Lt], tid = tids ? tids[row] : 0;
    const bf16_t* w = word + id * H;
    const bf16_t* p = pos + pid * H;
    const bf16_t* t = type + tid * H;
    for (int ch = lane; ch < (H >> 2); ch += 64) {
        float a[4], b[4], c[4];
        unpack4(*reinterpret_cast<const u32x2*>(w + ch * 4), a);
        unpack4(*reinterpret_cast<const u32x2*>(p + ch * 4), b);
        unpack4(*reinterpret_cast<const u32x2*>(t + ch * 4), c);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = a[e] + b[e] + c[e];
        *reinterpret_cast<u32x2*>(z + (int64_t)row * H + ch * 4) = pack4(a);
    }
}

// Deterministic scatter-add: wave `p` owns table row key[p] iff p is the first position holding that key;
// it then sums (fp32) the dz rows of every position with the same key and adds them to the table row.
// keys: n_keys entries; the dz rows of key position k are rows {k + r*row_stride, r < reps} (reps > 1 for the
// position table whose ids are shared by the whole batch).
template <int NC>
__device__ __forceinline__ void scatter_rows_body(const int64_t* keys, int n_keys, int reps, int row_stride,
                                                  const bf16_t* __restrict__ dz, bf16_t* __restrict__ table,
                                                  int H, int64_t padding_idx) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_keys) return;
    const int64_t key = keys[p];
    if (key == padding_idx) return;
    // first-occurrence test
    // first-occurrence test; four keys per lane and step so that the LDS latency is paid once per 256 keys
    bool dup = false;
    for (int q0 = 0; q0 < p; q0 += 256) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + 64 * j + lane;
            dup |= (q < p) && (keys[q] == key);
        }
    }
    if (__any(dup)) return;
    bf16_t* trow = table + key * H;
    const int nch = H >> 2;
    const int ch0 = blockIdx.y * NC * 64 + lane;
    float acc[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;
    // Sum, in position order (deterministic), the dz rows of every position that holds this key.  Rows are fetched eight
    // at a time before they are added: a frequent token ([CLS], [SEP], [MASK]: tens to hundreds of positions per batch)
    // would otherwise cost its owner wave one dependent memory round trip per occurrence.
    constexpr int RB = 8;                                // rows in flight per batch
    // The loads are unconditional (absent rows / columns are clamped to a valid address and discarded afterwards): with
    // predicated loads the compiler keeps a wait between them and the batch degenerates into RB dependent round trips.
    auto add4 = [&](const int64_t (&rows)[RB]) {
        u32x2 raw[RB][NC];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int64_t r = rows[k] >= 0 ? rows[k] : 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = ch0 + 64 * c;
                raw[k][c] = *reinterpret_cast<const u32x2*>(dz + r * H + (ch < nch ? ch : nch - 1) * 4);
            }
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const bool live = rows[k] >= 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = ch0 + 64 * c;
                float v[4];
                unpack4(raw[k][c], v);
                if (live && ch < nch) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] += v[e];
                }
            }
        }
    };
    // matching rows are queued (in position order) and fetched RB at a time, across scan steps: the occurrences of a
    // frequent token are spread over the whole key list, a flush per 64-key mask would again be one round trip each
    int64_t pend[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) pend[k] = -1;
    int npend = 0;
    auto push = [&](int64_t row) {
#pragma unroll
        for (int k = 0; k < RB; ++k)
            if (k == npend) pend[k] = row;
        if (++npend == RB) {
            add4(pend);
#pragma unroll
            for (int k = 0; k < RB; ++k) pend[k] = -1;
            npend = 0;
        }
    };
    for (int q0 = p; q0 < n_keys; q0 += 256) {           // the key list is scanned once, 256 keys per step
        bool hit[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + 64 * j + lane;
            hit[j] = (q < n_keys) && (keys[q] == key);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned long long mask = __ballot(hit[j]);
            const int base = q0 + 64 * j;
            while (mask) {
                const int pos = base + (__ffsll((long long)mask) - 1);
                mask &= mask - 1;
                for (int r = 0; r < reps; ++r) push((int64_t)pos + (int64_t)r * row_stride);
            }
        }
    }
    if (npend > 0) add4(pend);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = ch0 + 64 * c;
        if (ch < nch) {
            float o[4];
            unpack4(*reinterpret_cast<const u32x2*>(trow + ch * 4), o);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += acc[c][e];
            *reinterpret_cast<u32x2*>(trow + ch * 4) = pack4(o);
        }
    }
}

// grid (ceil(n_keys / 4), column groups): a wave covers NC chunks of 256 columns starting at blockIdx.y * NC * 256.
// The key list is staged in LDS first (one burst of coalesced loads) and the body is instantiated on the LDS pointer
// itself: its two scans are ~n_keys/64 dependent reads per wave, which through global (or generic "flat") addressing cost
// 46 us for the 1920 word ids of a 32 x 60 batch.
template <int NC>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const int64_t* __restrict__ keys_g, int n_keys, int reps, int row_stride,
                                                           const bf16_t* __restrict__ dz, bf16_t* __restrict__ table,
                                                           int H, int64_t padding_idx, int use_lds) {
    extern __shared__ int64_t skeys[];
    if (use_lds) {
        for (int q = threadIdx.x; q < n_keys; q += 256) skeys[q] = keys_g[q];
        __syncthreads();
        scatter_rows_body<NC>(skeys, n_keys, reps, row_stride, dz, table, H, padding_idx);
    } else {
        scatter_rows_body<NC>(keys_g, n_keys, reps, row_stride, dz, table, H, padding_idx);
    }
}

// column sums of the rows of a[rows][N] whose filter value equals `match`
// (filter_i64 / filter_u8: at most one non-null; both null -> `all_match` decides).  partial [gridDim.y][N]
__global__ __launch_bounds__(256) void colsum_filtered_kernel(const bf16_t* __restrict__ a, const int64_t* __restrict__ f64,
                                                              const uint8_t* __restrict__ f8, int64_t match, int all_match,
                                                              float* __restrict__ partial, int rows, int N) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int row = blockIdx.y * 4 + wid; row < rows; row += gridDim.y * 4) {
        bool take;
        if (f64) take = (f64[row] == match);
        else if (f8) take = ((int64_t)(f8[row] != 0) == match);
        else take = all_match != 0;
        if (take && col < N) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(a + (int64_t)row * N + col), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wid][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int gc = blockIdx.x * 512 + c;
        if (gc < N) partial[(int64_t)blockIdx.y * N + gc] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}
// out[c] += sum_b partial[b][c]: one block = 64 columns x 16 groups of partial rows, LDS tree over the groups
__global__ __launch_bounds__(1024) void add_partials_kernel(const float* __restrict__ partial, int nb, int N, bf16_t* __restrict__ out) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (c < N)
        for (int b = grp; b < nb; b += 16) s += partial[(int64_t)b * N + c];
    red[grp][cx] = s;
    __syncthreads();
    if (grp == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][cx];
        out[c] = f2bf(bf2f(out[c]) + t);
    }
}

// ---- image ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void img_prep_kernel(const void* __restrict__ feat, int is_fp32, const uint8_t* __restrict__ masks,
                                                       const bf16_t* __restrict__ mask_row, bf16_t* __restrict__ out,
                                                       int64_t rows, int D) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= rows * D) return;
    const int64_t row = idx / D;
    const int col = (int)(idx % D);
    float v[4];
    if (is_fp32) {
        const f32x4 q = *reinterpret_cast<const f32x4*>((const float*)feat + idx);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
        unpack4(*reinterpret_cast<const u32x2*>((const bf16_t*)feat + idx), v);
    }
    if (masks != nullptr && masks[row] != 0) {
        float m[4];
        unpack4(*reinterpret_cast<const u32x2*>(mask_row + col), m);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += m[e];
    }
    *reinterpret_cast<u32x2*>(out + idx) = pack4(v);
}

__device__ __forceinline__ float load_feat(const void* p, int is_fp32, int64_t i) {
    return is_fp32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}

// out[r][h] = sum_k pf[r][k] * w[h][k] + b[h],  k < 7
__global__ __launch_bounds__(256) void pos_linear_fwd_kernel(const void* __restrict__ pf, int is_fp32, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                             int rows, int H) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    float wv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) wv[k] = bf2f(w[h * 7 + k]);
    const float bv = bias ? bf2f(bias[h]) : 0.f;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        float s = bv;
#pragma unroll
        for (int k = 0; k < 7; ++k) s += load_feat(pf, is_fp32, (int64_t)r * 7 + k) * wv[k];
        out[(int64_t)r * H + h] = f2bf(s);
    }
}
// partial[rb][8][H]: k<7 -> dw[h][k], k=7 -> db[h]
__global__ __launch_bounds__(256) void pos_linear_bwd_kernel(const void* __restrict__ pf, int is_fp32, const bf16_t* __restrict__ d,
                                                             float* __restrict__ partial, int rows, int H) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const float dv = bf2f(d[(int64_t)r * H + h]);
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[k] += dv * load_feat(pf, is_fp32, (int64_t)r * 7 + k);
        acc[7] += dv;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[((int64_t)blockIdx.y * 8 + k) * H + h] = acc[k];
}
__global__ __launch_bounds__(1024) void pos_linear_finalize_kernel(const float* __restrict__ partial, int nb, int H,
                                                                   bf16_t* __restrict__ dw, bf16_t* __restrict__ db) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (idx < 8 * H)
        for (int b = grp; b < nb; b += 16) s += partial[(int64_t)b * 8 * H + idx];      // partial[b][k][h], idx = k*H + h
    red[grp][cx] = s;
    __syncthreads();
    if (grp == 0 && idx < 8 * H) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][cx];
        const int k = idx / H, h = idx % H;
        if (k < 7) { if (dw) dw[h * 7 + k] = f2bf(bf2f(dw[h * 7 + k]) + t); }
        else if (db) db[h] = f2bf(bf2f(db[h]) + t);
    }
}

__global__ __launch_bounds__(256) void img_combine_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                          const int64_t* __restrict__ tids, const bf16_t* __restrict__ type,
                                                          bf16_t* __restrict__ z, int64_t rows, int H) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= rows * H) return;
    const int64_t row = idx / H;
    const int col = (int)(idx % H);
    const int64_t tid = tids ? tids[row] : 1;
    float x[4], y[4], t[4];
    unpack4(*reinterpret_cast<const u32x2*>(a + idx), x);
    unpack4(*reinterpret_cast<const u32x2*>(b + idx), y);
    unpack4(*reinterpret_cast<const u32x2*>(type + tid * H + col), t);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = x[e] + y[e] + t[e];
    *reinterpret_cast<u32x2*>(z + idx) = pack4(x);
}

// ---- gather / scatter ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_fwd_kernel(const bf16_t* __restrict__ txt, const bf16_t* __restrict__ img,
                                                         const int64_t* __restrict__ gi, bf16_t* __restrict__ out,
                                                         int B, int Lt, int Li, int Lout, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * Lout) return;
    const int b = row / Lout;
    const int64_t s = gi[row];
    const bf16_t* src = (s < Lt) ? txt + ((int64_t)b * Lt + s) * H : img + ((int64_t)b * Li + (s - Lt)) * H;
    for (int ch = lane; ch < (H >> 2); ch += 64)
        *reinterpret_cast<u32x2*>(out + (int64_t)row * H + ch * 4) = *reinterpret_cast<const u32x2*>(src + ch * 4);
}
// one wave per source row (b, s), s in [0, Lt+Li): sum of dout rows j with gi[b][j] == s (overwrite)
__global__ __launch_bounds__(256) void gather_bwd_kernel(const bf16_t* __restrict__ dout, const int64_t* __restrict__ gi,
                                                         bf16_t* __restrict__ dtxt, bf16_t* __restrict__ dimg,
                                                         int B, int Lt, int Li, int Lout, int H) {
    const int lane = threadIdx.x & 63;
    const int srow = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Ls = Lt + Li;
    if (srow >= B * Ls) return;
    const int b = srow / Ls, s = srow % Ls;
    bf16_t* dst = (s < Lt) ? dtxt + ((int64_t)b * Lt + s) * H : dimg + ((int64_t)b * Li + (s - Lt)) * H;
    const int64_t* g = gi + (int64_t)b * Lout;
    for (int c0 = 0; c0 < (H >> 2); c0 += 64) {
        const int ch = c0 + lane;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < Lout; j0 += 64) {
            const int j = j0 + lane;
            unsigned long long mask = __ballot(j < Lout && g[j] == s);
            while (mask) {
                const int bit = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if (ch < (H >> 2)) {
                    float v[4];
                    unpack4(*reinterpret_cast<const u32x2*>(dout + ((int64_t)b * Lout + j0 + bit) * H + ch * 4), v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += v[e];
                }
            }
        }
        if (ch < (H >> 2)) *reinterpret_cast<u32x2*>(dst + ch * 4) = pack4(acc);
    }
}

__global__ __launch_bounds__(256) void mask_bias_kernel(const int64_t* __restrict__ m, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (1.0f - (float)m[i]) * -10000.0f;
}

int filtered_blocks(int64_t rows, int64_t N) {
    const int64_t strips = (N + 511) / 512;
    int64_t nb = 512 / strips;
    const int64_t maxb = (rows + 15) / 16;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

#include "../../include/uniter_hip.h"

extern "C" {

int uniter_embed_txt_fwd(const int64_t* ids, const int64_t* position_ids, const int64_t* type_ids,
                         const void* word, const void* pos, const void* type, void* z,
                         int64_t B, int64_t Lt, int64_t H, int64_t vocab, int64_t max_pos,
                         int64_t n_types, void* stream) {
    UH_CHECK_ARG(ids && position_ids && word && pos && type && z, "null pointer");
    UH_CHECK_ARG(B > 0 && Lt > 0 && H > 0 && H % 4 == 0, "bad shape (H %% 4 == 0 required)");
    (void)vocab; (void)max_pos; (void)n_types;
    const int n = (int)(B * Lt);
    hipLaunchKernelGGL(embed_txt_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, position_ids, type_ids,
                       (const bf16_t*)word, (const bf16_t*)pos, (const bf16_t*)type, (bf16_t*)z, n, (int)Lt, (int)H);
    UH_LAUNCH_CHECK();
    return 0;
}

size_t uniter_embed_ws_bytes(int64_t rows, int64_t N) {
    size_t a = (size_t)filtered_blocks(rows, N) * (size_t)N * sizeof(float);
    size_t b = (size_t)64 * 8 * (size_t)N * sizeof(float);   // pos_linear partials (<= 64 row blocks)
    return a > b ? a : b;
}

int uniter_embed_type_bwd(const void* dz, const int64_t* type_ids, void* dtype_table,
                          int64_t rows, int64_t H, int64_t n_types, int default_type,
                          void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(dz && dtype_table && workspace, "null pointer");
    UH_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0 && n_types > 0, "bad shape (H %% 8 == 0 required)");
    const int nb = filtered_blocks(rows, H);
    UH_CHECK_ARG(workspace_bytes >= (size_t)nb * H * sizeof(float), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    for (int64_t k = 0; k < n_types; ++k) {
        if (type_ids == nullptr && k != default_type) continue;
        hipLaunchKernelGGL(colsum_filtered_kernel, dim3((unsigned)((H + 511) / 512), nb), dim3(256), 0, st, (const bf16_t*)dz,
                           type_ids, (const uint8_t*)nullptr, k, 1, (float*)workspace, (int)rows, (int)H);
        UH_LAUNCH_CHECK();
        hipLaunchKernelGGL(add_partials_kernel, dim3((unsigned)((H + 63) / 64)), dim3(1024), 0, st, (const float*)workspace, nb,
                           (int)H, (bf16_t*)dtype_table + k * H);
        UH_LAUNCH_CHECK();
    }
    return 0;
}

int uniter_embed_txt_bwd(const int64_t* ids, const int64_t* position_ids, const int64_t* type_ids,
                         const void* dz, void* dword, void* dpos, void* dtype,
                         int64_t B, int64_t Lt, int64_t H, int64_t vocab, int64_t max_pos,
                         int64_t n_types, void* stream) {
    UH_CHECK_ARG(ids && position_ids && dz, "null pointer");
    UH_CHECK_ARG(B > 0 && Lt > 0 && H > 0 && H % 4 == 0, "bad shape");
    (void)vocab; (void)max_pos; (void)n_types; (void)type_ids; (void)dtype;
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)(B * Lt);
    if (dword) {
        // nn.Embedding(vocab, H, padding_idx=0) (model/model.py:220-221): row 0 receives no gradient
        // one wave per token covers the whole row (<= 4 chunks of 256 columns; wider rows add column groups)
        const int nch256 = (int)((H / 4 + 63) / 64);
        const int kuse = (size_t)n * 8 <= 60 * 1024 ? 1 : 0;          // key list in LDS when it fits the default 64 KiB
        const size_t klds = kuse ? (size_t)n * 8 : 0;
        if (nch256 <= 4) {
            hipLaunchKernelGGL(scatter_rows_kernel<4>, dim3((n + 3) / 4, 1), dim3(256), klds, st, ids, n, 1, 0, (const bf16_t*)dz,
                               (bf16_t*)dword, (int)H, (int64_t)0, kuse);
        } else {
            hipLaunchKernelGGL(scatter_rows_kernel<4>, dim3((n + 3) / 4, (nch256 + 3) / 4), dim3(256), klds, st, ids, n, 1, 0,
                               (const bf16_t*)dz, (bf16_t*)dword, (int)H, (int64_t)0, kuse);
        }
        UH_LAUNCH_CHECK();
    }
    if (dpos) {
        // few keys, B rows each: spread the columns over the grid instead (one 256-column chunk per wave)
        hipLaunchKernelGGL(scatter_rows_kernel<1>, dim3(((int)Lt + 3) / 4, (unsigned)((H / 4 + 63) / 64)), dim3(256), (size_t)Lt * 8, st,
                           position_ids, (int)Lt, (int)B, (int)Lt, (const bf16_t*)dz, (bf16_t*)dpos, (int)H, (int64_t)-1, 1);
        UH_LAUNCH_CHECK();
    }
    return 0;
}

int uniter_embed_img_prep(const void* img_feat, int feat_is_fp32, const uint8_t* img_masks,
                          const void* mask_row, void* f_out, int64_t rows, int64_t D, void* stream) {
    UH_CHECK_ARG(img_feat && f_out, "null pointer");
    UH_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0, "bad shape (D %% 4 == 0 required)");
    UH_CHECK_ARG(img_masks == nullptr || mask_row != nullptr, "img_masks given without mask_row");
    const int64_t n4 = rows * D / 4;
    hipLaunchKernelGGL(img_prep_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img_feat,
                       feat_is_fp32, img_masks, (const bf16_t*)mask_row, (bf16_t*)f_out, rows, (int)D);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_embed_pos_linear_fwd(const void* pos_feat, int feat_is_fp32, const void* wpos,
                                const void* bpos, void* out, int64_t rows, int64_t H, void* stream) {
    UH_CHECK_ARG(pos_feat && wpos && out, "null pointer");
    UH_CHECK_ARG(rows > 0 && H > 0, "bad shape");
    int rb = (int)(rows < 256 ? rows : 256);
    hipLaunchKernelGGL(pos_linear_fwd_kernel, dim3((unsigned)((H + 255) / 256), rb), dim3(256), 0, (hipStream_t)stream, pos_feat,
                       feat_is_fp32, (const bf16_t*)wpos, (const bf16_t*)bpos, (bf16_t*)out, (int)rows, (int)H);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_embed_pos_linear_bwd(const void* pos_feat, int feat_is_fp32, const void* d,
                                void* dwpos, void* dbpos, int64_t rows, int64_t H,
                                void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(pos_feat && d && workspace, "null pointer");
    UH_CHECK_ARG(rows > 0 && H > 0, "bad shape");
    int rb = (int)(rows < 64 ? rows : 64);
    UH_CHECK_ARG(workspace_bytes >= (size_t)rb * 8 * H * sizeof(float), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pos_linear_bwd_kernel, dim3((unsigned)((H + 255) / 256), rb), dim3(256), 0, st, pos_feat, feat_is_fp32,
                       (const bf16_t*)d, (float*)workspace, (int)rows, (int)H);
    UH_LAUNCH_CHECK();
    hipLaunchKernelGGL(pos_linear_finalize_kernel, dim3((unsigned)((8 * H + 63) / 64)), dim3(1024), 0, st, (const float*)workspace,
                       rb, (int)H, (bf16_t*)dwpos, (bf16_t*)dbpos);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_embed_img_combine_fwd(const void* a, const void* b, const int64_t* type_ids, const void* type,
                                 void* z, int64_t rows, int64_t H, int64_t n_types, void* stream) {
    UH_CHECK_ARG(a && b && type && z, "null pointer");
    UH_CHECK_ARG(rows > 0 && H > 0 && H % 4 == 0 && n_types >= 2, "bad shape (needs >= 2 token types)");
    const int64_t n4 = rows * H / 4;
    hipLaunchKernelGGL(img_combine_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, type_ids, (const bf16_t*)type, (bf16_t*)z, rows, (int)H);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_embed_mask_bwd(const void* df, const uint8_t* img_masks, void* dmask_row,
                          int64_t rows, int64_t D, void* workspace, size_t workspace_bytes, void* stream) {
    UH_CHECK_ARG(df && img_masks && dmask_row && workspace, "null pointer");
    UH_CHECK_ARG(rows > 0 && D > 0 && D % 8 == 0, "bad shape");
    const int nb = filtered_blocks(rows, D);
    UH_CHECK_ARG(workspace_bytes >= (size_t)nb * D * sizeof(float), "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_filtered_kernel, dim3((unsigned)((D + 511) / 512), nb), dim3(256), 0, st, (const bf16_t*)df,
                       (const int64_t*)nullptr, img_masks, (int64_t)1, 0, (float*)workspace, (int)rows, (int)D);
    UH_LAUNCH_CHECK();
    hipLaunchKernelGGL(add_partials_kernel, dim3((unsigned)((D + 63) / 64)), dim3(1024), 0, st, (const float*)workspace, nb, (int)D,
                       (bf16_t*)dmask_row);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_embed_gather_fwd(const void* txt, const void* img, const int64_t* gather_index, void* out,
                            int64_t B, int64_t Lt, int64_t Li, int64_t Lout, int64_t H, void* stream) {
    UH_CHECK_ARG(txt && img && gather_index && out, "null pointer");
    UH_CHECK_ARG(B > 0 && Lt > 0 && Li > 0 && Lout > 0 && H % 4 == 0, "bad shape");
    const int n = (int)(B * Lout);
    hipLaunchKernelGGL(gather_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)txt, (const bf16_t*)img,
                       gather_index, (bf16_t*)out, (int)B, (int)Lt, (int)Li, (int)Lout, (int)H);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_embed_gather_bwd(const void* dout, const int64_t* gather_index, void* dtxt, void* dimg,
                            int64_t B, int64_t Lt, int64_t Li, int64_t Lout, int64_t H, void* stream) {
    UH_CHECK_ARG(dout && gather_index && dtxt && dimg, "null pointer");
    UH_CHECK_ARG(B > 0 && Lt > 0 && Li > 0 && Lout > 0 && H % 4 == 0, "bad shape");
    const int n = (int)(B * (Lt + Li));
    hipLaunchKernelGGL(gather_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, gather_index,
                       (bf16_t*)dtxt, (bf16_t*)dimg, (int)B, (int)Lt, (int)Li, (int)Lout, (int)H);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_mask_bias(const int64_t* attn_masks, float* mask_bias, int64_t n, void* stream) {
    UH_CHECK_ARG(attn_masks && mask_bias && n > 0, "null pointer / empty");
    hipLaunchKernelGGL(mask_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, attn_masks, mask_bias, n);
    UH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
