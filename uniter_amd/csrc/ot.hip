// ot.hip — word-region alignment loss: IPOT approximation of the optimal-transport distance between the text and the
// image part of every joint sequence (reference model/ot.py:11-85, wired at model/pretrain.py:166-188).
//
// The reference spends ~8 tiny PyTorch kernels per IPOT iteration x 50 iterations per ITM step (launch-latency bound,
// SURVEY.md §8 f-1).  Here one workgroup owns one example: the cosine cost [tl x il] is one MFMA pass over the bf16
// encoder rows (read through the `ot_scatter` indirection, no un-compaction copy), exp(-C/beta) and the transport plan
// live in LDS for all 50 iterations, and the distance trace(C T) falls out at the end.  Backward consumes the saved plan.
//
// Layouts: seq [B, L, H] bf16 (compact [txt_i ; img_i ; pad] rows), scatter [B, L] int64 = destination slot of each row
// (text slots 0..tl-1, image slots tl..tl+il-1, anything else is ignored), pads uint8 (1 = padded slot),
// plan [B, il, tl] fp32 = T of model/ot.py (n-major, as the reference holds it), dist [B] fp32.
#include "common.cuh"
#include "kernels.h"
#include "../../include/uniter_hip.h"

namespace {

constexpr int OT_THREADS = 1024;       // 16 waves: the 50 dependent iterations are latency-bound, so phases are kept short
constexpr int OT_WAVES = OT_THREADS / 64;
constexpr float OT_EPS = 1e-5f;          // F.normalize(eps=1e-5), model/ot.py:17-18

struct OtArgs {
    const bf16_t* seq;
    const int64_t* scatter;
    const uint8_t* txt_pad;
    const uint8_t* img_pad;
    float* dist;
    float* plan;
    const float* gdist;
    bf16_t* dseq;
    int B, L, H, tl, il;
    float inv_beta;
    int iterations, k;
};

// LDS carve-up shared by both kernels: inverse maps, inverse norms, then kernel-specific arrays
__device__ __forceinline__ void build_inverse(const OtArgs& p, int b, int* inv_t, int* inv_i) {
    for (int s = threadIdx.x; s < p.tl; s += OT_THREADS) inv_t[s] = -1;
    for (int s = threadIdx.x; s < p.il; s += OT_THREADS) inv_i[s] = -1;
    __syncthreads();
    for (int j = threadIdx.x; j < p.L; j += OT_THREADS) {
        const int64_t d = p.scatter[(int64_t)b * p.L + j];
        if (d >= 0 && d < p.tl) inv_t[d] = j;
        else if (d >= p.tl && d < p.tl + p.il) inv_i[d - p.tl] = j;
    }
    __syncthreads();
}

// 1 / max(||row||, eps) per slot (0 for an empty slot: its vector is all zeros)
__device__ __forceinline__ void inverse_norms(const OtArgs& p, int b, const int* inv, int n_slots, float* rinv) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int s = wid; s < n_slots; s += OT_THREADS / 64) {
        const int j = inv[s];
        float ss = 0.f;
        if (j >= 0) {
            const bf16_t* row = p.seq + ((int64_t)b * p.L + j) * p.H;
            for (int c = lane * 8; c < p.H; c += 64 * 8) {
                float v[8];
                unpack8(*reinterpret_cast<const u32x4*>(row + c), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) rinv[s] = (j >= 0) ? 1.f / fmaxf(sqrtf(ss), OT_EPS) : 0.f;
    }
}

__global__ __launch_bounds__(OT_THREADS) void ot_fwd_kernel(const OtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int tl = p.tl, il = p.il;
    int* inv_t = reinterpret_cast<int*>(smem);
    int* inv_i = inv_t + tl;
    float* rx = reinterpret_cast<float*>(inv_i + il);
    float* ry = rx + tl;
    float* delta = ry + il;                  // [il]
    float* part = delta + il;                // [OT_WAVES][tl] column-sum partials of the sigma update
    float* red = part + OT_WAVES * tl;       // [OT_WAVES] block reduction
    float* A = red + OT_WAVES;               // [il][tl]  exp(-C/beta), 0 at padded pairs
    float* T = A + il * tl;                  // [il][tl]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

    build_inverse(p, b, inv_t, inv_i);
    inverse_norms(p, b, inv_t, tl, rx);
    inverse_norms(p, b, inv_i, il, ry);
    __syncthreads();

    const uint8_t* tpad = p.txt_pad + (int64_t)b * tl;
    const uint8_t* ipad = p.img_pad + (int64_t)b * il;
    float* plan = p.plan + (int64_t)b * il * tl;

    // ---- cosine cost: 16x16 MFMA tiles of X (text rows) x Y^T (image rows), operands straight from global ----
    // A operand: lane holds row (l & 15), k = 8*(l >> 4) .. +7 ; B operand: lane holds column (l & 15), same k ;
    // D: lane holds column (l & 15), rows 4*(l >> 4) .. +3.
    {
        const int tiles_m = (tl + 15) >> 4, tiles_n = (il + 15) >> 4;
        const int g = lane >> 4, i = lane & 15;
        for (int t = wid; t < tiles_m * tiles_n; t += OT_WAVES) {
            const int m0 = (t / tiles_n) * 16, n0 = (t % tiles_n) * 16;
            const int mr = m0 + i, nr = n0 + i;
            const int jm = (mr < tl) ? inv_t[mr] : -1;
            const int jn = (nr < il) ? inv_i[nr] : -1;
            const bf16_t* xrow = p.seq + ((int64_t)b * p.L + (jm >= 0 ? jm : 0)) * p.H + 8 * g;
            const bf16_t* yrow = p.seq + ((int64_t)b * p.L + (jn >= 0 ? jn : 0)) * p.H + 8 * g;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16x8 zero = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            const int ksteps = p.H >> 5;                    // full 32-wide steps; a 8/16/24-wide tail follows
#pragma unroll 4
            for (int ks = 0; ks < ksteps; ++ks) {
                bf16x8 fa = *reinterpret_cast<const bf16x8*>(xrow + ks * 32);
                bf16x8 fb = *reinterpret_cast<const bf16x8*>(yrow + ks * 32);
                if (jm < 0) fa = zero;
                if (jn < 0) fb = zero;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
            }
            if ((p.H & 31) != 0) {
                const int kk = ksteps * 32 + 8 * g;
                bf16x8 fa = zero, fb = zero;
                if (jm >= 0 && kk < p.H) fa = *reinterpret_cast<const bf16x8*>(xrow + ksteps * 32);
                if (jn >= 0 && kk < p.H) fb = *reinterpret_cast<const bf16x8*>(yrow + ksteps * 32);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
            }
            const int n = n0 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 4 * g + r;
                if (m < tl && n < il) {
                    const bool masked = tpad[m] != 0 || ipad[n] != 0;
                    const float c = masked ? 0.f : 1.f - acc[r] * rx[m] * ry[n];       // model/ot.py:19-21, :76
                    A[n * tl + m] = masked ? 0.f : __expf(-c * p.inv_beta);            // :42, :48
                    T[n * tl + m] = masked ? 0.f : 1.f;                                // :41, :47
                    plan[n * tl + m] = c;              // the plan buffer carries C until the end of the kernel
                }
            }
        }
    }
    // lengths (model/ot.py:78-81)
    float xl = 0.f, yl = 0.f;
    for (int m = lane; m < tl; m += 64) xl += tpad[m] ? 0.f : 1.f;
    for (int n = lane; n < il; n += 64) yl += ipad[n] ? 0.f : 1.f;
    xl = wave_sum(xl);
    yl = wave_sum(yl);
    // this lane's text columns: m = lane and lane + 64 (tl <= 128)
    const int mA = lane, mB = lane + 64;
    const bool hasA = mA < tl, hasB = mB < tl;
    const float xmA = (hasA && tpad[mA]) ? 1e4f : 0.f, xmB = (hasB && tpad[mB]) ? 1e4f : 0.f;
    float sgA = (hasA && !tpad[mA]) ? 1.f / xl : 0.f;       // sigma_0 = 1 / x_len, 0 at padded slots (:39-40, :45)
    float sgB = (hasB && !tpad[mB]) ? 1.f / xl : 0.f;
    __syncthreads();

    // ---- IPOT iterations (model/ot.py:58-65), Q = A .* T never materialised.  Every lane keeps sigma of its own two
    // columns in registers.  Phase 1 (wave per image row n): apply the previous iteration's T <- delta Q sigma, then
    // delta[n] = 1 / (y_len * sum_m Q[n][m] sigma[m] + 1e4 y_pad[n]).  Phase 2 (wave = group of rows): partial column
    // sums of delta[n] Q[n][m]; after the barrier every lane adds the OT_WAVES partials of its columns:
    // sigma[m] = 1 / (x_len * sum_n delta[n] Q[n][m] + 1e4 x_pad[m]).  The inner k-loop (k > 1) repeats both phases on
    // the same Q, i.e. without the T update. ----
    const int total = p.iterations * p.k;
    for (int step = 0; step <= total; ++step) {
        const bool apply_T = step > 0 && (step % p.k) == 0;          // an outer iteration just finished
        for (int n = wid; n < il; n += OT_WAVES) {
            float* Tn = T + n * tl;
            const float* An = A + n * tl;
            float s = 0.f;
            const float dn = delta[n];
            if (hasA) {
                float t = Tn[mA];
                if (apply_T) { t = dn * An[mA] * t * sgA; Tn[mA] = t; }
                s += An[mA] * t * sgA;
            }
            if (hasB) {
                float t = Tn[mB];
                if (apply_T) { t = dn * An[mB] * t * sgB; Tn[mB] = t; }
                s += An[mB] * t * sgB;
            }
            if (step < total) {
                s = wave_sum(s);
                if (lane == 0) delta[n] = 1.f / (yl * s + (ipad[n] ? 1e4f : 0.f));
            }
        }
        if (step == total) break;
        __syncthreads();
        float cA = 0.f, cB = 0.f;
        for (int n = wid; n < il; n += OT_WAVES) {
            const float dn = delta[n];
            if (hasA) cA += dn * A[n * tl + mA] * T[n * tl + mA];
            if (hasB) cB += dn * A[n * tl + mB] * T[n * tl + mB];
        }
        if (hasA) part[wid * tl + mA] = cA;
        if (hasB) part[wid * tl + mB] = cB;
        __syncthreads();
        float tA = 0.f, tB = 0.f;
#pragma unroll
        for (int w = 0; w < OT_WAVES; ++w) {
            if (hasA) tA += part[w * tl + mA];
            if (hasB) tB += part[w * tl + mB];
        }
        sgA = hasA ? 1.f / (xl * tA + xmA) : 0.f;
        sgB = hasB ? 1.f / (xl * tB + xmB) : 0.f;
        // (the barrier above already orders this phase's reads of delta / T before the next phase 1's writes, and the
        //  next writes to `part` come after the next barrier, by which time every lane has finished reading it)
    }
    __syncthreads();
    // ---- dist = trace(C T) = sum_{m,n} C[m][n] T[n][m]  (model/ot.py:84) ; plan <- T (masked, :66) ----
    float s = 0.f;
    for (int e = threadIdx.x; e < il * tl; e += OT_THREADS) {
        const float c = plan[e];
        const float t = (A[e] == 0.f) ? 0.f : T[e];
        s += c * t;
        plan[e] = t;
    }
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < OT_WAVES; ++w) t += red[w];
        p.dist[b] = t;
    }
}

// d dist / d seq: dC[m][n] = g * T[n][m]; C = 1 - xh . yh with xh = x / max(||x||, eps):
//   d xh_m = -sum_n dC[m][n] yh_n ;  d x_m = rinv_m * (d xh_m - xh_m (xh_m . d xh_m))   (||x|| >= eps)
// One wave per compact row j; rows whose destination is neither a text nor an image slot get zeros.
__global__ __launch_bounds__(OT_THREADS) void ot_bwd_kernel(const OtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int tl = p.tl, il = p.il;
    int* inv_t = reinterpret_cast<int*>(smem);
    int* inv_i = inv_t + tl;
    float* rx = reinterpret_cast<float*>(inv_i + il);
    float* ry = rx + tl;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

    build_inverse(p, b, inv_t, inv_i);
    inverse_norms(p, b, inv_t, tl, rx);
    inverse_norms(p, b, inv_i, il, ry);
    __syncthreads();

    const uint8_t* tpad = p.txt_pad + (int64_t)b * tl;
    const uint8_t* ipad = p.img_pad + (int64_t)b * il;
    const float* plan = p.plan + (int64_t)b * il * tl;
    const float g = p.gdist[b];
    const bf16_t* base = p.seq + (int64_t)b * p.L * p.H;
    constexpr int NC = 4;                                   // 4 chunks of 256 columns: H <= 1024
    for (int j = wid; j < p.L; j += OT_WAVES) {
        const int64_t d = p.scatter[(int64_t)b * p.L + j];
        bf16_t* out = p.dseq + ((int64_t)b * p.L + j) * p.H;
        const bool is_txt = d >= 0 && d < tl && inv_t[d] == j && tpad[d] == 0;
        const bool is_img = d >= tl && d < tl + il && inv_i[d - tl] == j && ipad[d - tl] == 0;
        float acc[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;
        if (is_txt || is_img) {
            const int self = is_txt ? (int)d : (int)(d - tl);
            const int n_other = is_txt ? il : tl;
            const int* inv_o = is_txt ? inv_i : inv_t;
            const float* r_o = is_txt ? ry : rx;
            // lanes fetch the partner weights w_o = -g T[n][m] / max(||partner||, eps) and partner rows once (o = lane,
            // lane + 64), the loop then broadcasts them: no dependent global load sits in the accumulation loop
            float wv[2];
            int jv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int o = lane + 64 * h;
                float w = 0.f;
                int jo = 0;
                if (o < n_other) {
                    const float t = is_txt ? plan[o * tl + self] : plan[self * tl + o];     // T[n][m]
                    const int q = inv_o[o];
                    if (q >= 0) { w = -g * t * r_o[o]; jo = q; }
                }
                wv[h] = w;
                jv[h] = jo;
            }
            for (int o0 = 0; o0 < n_other; o0 += 4) {
                float w4[4];
                const bf16_t* r4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int o = o0 + u;                   // < 128; slots past n_other carry weight 0 and row 0
                    const float wa = __shfl(wv[0], o & 63, WAVE), wb = __shfl(wv[1], o & 63, WAVE);
                    const int ja = __shfl(jv[0], o & 63, WAVE), jb = __shfl(jv[1], o & 63, WAVE);
                    w4[u] = (o < 64) ? wa : wb;
                    r4[u] = base + (int64_t)((o < 64) ? ja : jb) * p.H;
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int col = (c * 64 + lane) * 4;
                    if (col < p.H) {
                        u32x2 raw[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const u32x2*>(r4[u] + col);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float v[4];
                            unpack4(raw[u], v);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[c][e] += w4[u] * v[e];
                        }
                    }
                }
            }
            // through the normalisation of this row
            const float rs = is_txt ? rx[self] : ry[self];
            const bf16_t* srow = base + (int64_t)j * p.H;
            float xh[NC][4];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = (c * 64 + lane) * 4;
                if (col < p.H) {
                    float v[4];
                    unpack4(*reinterpret_cast<const u32x2*>(srow + col), v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { xh[c][e] = v[e] * rs; dot += xh[c][e] * acc[c][e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xh[c][e] = 0.f;
                }
            }
            dot = wave_sum(dot);
            const bool clamped = rs >= 1.f / OT_EPS;        // ||x|| < eps: xh = x / eps, a plain scaling
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][e] = rs * (acc[c][e] - (clamped ? 0.f : xh[c][e] * dot));
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = (c * 64 + lane) * 4;
            if (col < p.H) *reinterpret_cast<u32x2*>(out + col) = pack4(acc[c]);
        }
    }
}

size_t ot_lds_bytes(int tl, int il, bool fwd) {
    size_t n = (size_t)(tl + il) * 4 * 2;                    // inverse maps + inverse norms
    if (fwd) n += ((size_t)il + OT_WAVES * (size_t)tl + OT_WAVES + 2 * (size_t)il * tl) * 4;
    return (n + 15) / 16 * 16;
}

int check(const OtArgs& a) {
    if (a.B <= 0 || a.L <= 0 || a.H <= 0 || a.tl <= 0 || a.il <= 0) { uh_set_error("ot: non-positive dimension"); return -1; }
    if (a.H % 8 != 0 || a.H > 1024) { uh_set_error("ot: need H %% 8 == 0 and H <= 1024 (H=%d)", a.H); return -1; }
    if (a.tl > 128) { uh_set_error("ot: at most 128 text slots (tl=%d)", a.tl); return -1; }
    if (ot_lds_bytes(a.tl, a.il, true) > 160 * 1024) { uh_set_error("ot: tl x il = %d x %d does not fit the 160 KiB LDS", a.tl, a.il); return -1; }
    return 0;
}

}  // namespace

extern "C" {

int uniter_ot_fwd(const void* seq, const int64_t* scatter, const uint8_t* txt_pad, const uint8_t* img_pad,
                  float* dist, float* plan, int64_t B, int64_t L, int64_t H, int64_t tl, int64_t il,
                  float beta, int32_t iterations, int32_t k, void* stream) {
    UH_CHECK_ARG(seq && scatter && txt_pad && img_pad && dist && plan, "null pointer");
    UH_CHECK_ARG(beta > 0.f && iterations >= 0 && k >= 1, "bad IPOT parameters");
    OtArgs a{};
    a.seq = (const bf16_t*)seq; a.scatter = scatter; a.txt_pad = txt_pad; a.img_pad = img_pad;
    a.dist = dist; a.plan = plan;
    a.B = (int)B; a.L = (int)L; a.H = (int)H; a.tl = (int)tl; a.il = (int)il;
    a.inv_beta = 1.f / beta; a.iterations = iterations; a.k = k;
    if (check(a)) return -1;
    const size_t lds = ot_lds_bytes(a.tl, a.il, true);
    if (lds > 64 * 1024)
        UH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ot_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ot_fwd_kernel, dim3((unsigned)B), dim3(OT_THREADS), lds, (hipStream_t)stream, a);
    UH_LAUNCH_CHECK();
    return 0;
}

int uniter_ot_bwd(const void* seq, const int64_t* scatter, const uint8_t* txt_pad, const uint8_t* img_pad,
                  const float* plan, const float* gdist, void* dseq,
                  int64_t B, int64_t L, int64_t H, int64_t tl, int64_t il, void* stream) {
    UH_CHECK_ARG(seq && scatter && txt_pad && img_pad && plan && gdist && dseq, "null pointer");
    OtArgs a{};
    a.seq = (const bf16_t*)seq; a.scatter = scatter; a.txt_pad = txt_pad; a.img_pad = img_pad;
    a.plan = const_cast<float*>(plan); a.gdist = gdist; a.dseq = (bf16_t*)dseq;
    a.B = (int)B; a.L = (int)L; a.H = (int)H; a.tl = (int)tl; a.il = (int)il;
    if (check(a)) return -1;
    const size_t lds = ot_lds_bytes(a.tl, a.il, false);
    hipLaunchKernelGGL(ot_bwd_kernel, dim3((unsigned)B), dim3(OT_THREADS), lds, (hipStream_t)stream, a);
    UH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
