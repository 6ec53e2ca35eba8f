// row_tail.cuh — LayerNorm as the tail of the GEMM that produces its input.
//
// Reference: model/layer.py:111-115,152-156 — `LayerNorm(dropout(dense(h)) + input)`: the reference's apex kernel is a launch of
// its own, and so it was here (ln_fwd_kernel after the out-proj / FFN2 GEMM, ln_bwd_rows_kernel after the FFN1 / QKV data
// gradient): 48 launches per step of 5-7 us each, bounded by latency, not by bytes.  A LayerNorm row needs the whole output
// row, i.e. the results of all column tiles of its row block — so it cannot go into a tile's epilogue, but it can go BEHIND the
// tiles of the row block:
//   1. every tile stores its output write-through (sc1), waits for the acknowledgement and adds 1 to the counters of the
//      32-row units it covers (the flags of the overlapped chains, common.cuh);
//   2. it then waits until those counters show all `tiles_n` column tiles of the row block (all of them are resident: the
//      launcher only enables the tail when the whole grid fits on the chip at once; a wait that does not complete in 50 ms sets
//      *status and falls through — a test failure, never a hang);
//   3. the tiles of the row block split its rows among themselves (tile tn takes rows tn, tn + tiles_n, ... of the block) and
//      normalise them: plain loads of the full rows (no cache of the chip can hold an older version: nobody read them since the
//      kernel began), the row arithmetic of layernorm_fwd.cuh, outputs for the next kernel.
// The separate launch, its ~1.5 us queue gap and its ramp disappear; what is added to the GEMM is one counter round trip and one
// row round trip with every workgroup of the chip taking part.
#pragma once
#include "common.cuh"
#include "layernorm_fwd.cuh"

namespace {

struct RowTail {
    uint32_t* count;          // one counter per 32-row unit: zero at the start of the encoder call, monotonic within it
    uint32_t* status;         // set to 1 by a wait that timed out
    uint32_t expect;          // what the units of a finished row block hold after this launch
    int kind;                 // 0 = none, 1 = LayerNorm forward, 2 = LayerNorm backward (row half)
    int local;                // 1 = the column tiles of a row block all run on one XCD (2-D tile mapping with 8 XCD rows): their L2 is the
                              //     meeting point, the output needs no write-through
    const bf16_t* gamma;
    const bf16_t* beta;       // kind 1
    const bf16_t* z;          // kind 2: the LayerNorm's saved input
    bf16_t* out;              // kind 1: y          kind 2: dz
    bf16_t* out2;             // kind 2: dd = dz under the dense branch's dropout mask (nullptr: not wanted)
    float* mean;              // kind 1: written    kind 2: read
    float* rstd;
    float eps;
    DropoutCfg drop;          // kind 2
};

// All NCW compute waves of the workgroup (t < NCW * 64), after their last store of C rows [m0, m0 + nrows).  RB: rows a wave keeps
// in flight at once (register budget of the kernel it is inlined into).
// C: this launch's output [M][H] (leading dimension H), tn / tiles_n: this tile's column index / column tiles per row block.
template <int NCW, int KIND, int RB>
__device__ __forceinline__ void row_tail_run(const RowTail& rt, const bf16_t* __restrict__ C, const int H, const int m0, const int nrows,
                                             const int tn, const int tiles_n, const int t) {
    const int lane = t & 63, wid = t >> 6;
    // (1) my rows have left the chip's caches; tell the row block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int u0 = m0 >> 5;
    const int nu = ((m0 + nrows + 31) >> 5) - u0;
    if (t < nu) __hip_atomic_fetch_add(rt.count + u0 + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (2) wait for the other column tiles
    if (t < 64) {
        bool ok = lane >= nu;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 0;; ++spins) {
            if (!ok) ok = __hip_atomic_load(rt.count + u0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= rt.expect;
            if (__all(ok)) break;
            if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) {
                if (lane == 0 && rt.status != nullptr) __hip_atomic_store(rt.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_s_barrier();
    // (3) rows tn + tiles_n * (wid + NCW * j) of the block are this wave's
    const int step = tiles_n * NCW;
    const int first = tn + tiles_n * wid;
    const int nc = (H + 255) >> 8;
    for (int r = first; r < nrows; r += step * RB) {
        const int left = (nrows - r + step - 1) / step;
        const int cnt = left < RB ? left : RB;
        if constexpr (KIND == 1) {
            if (nc <= 3) ln_fwd_rows_batch<3, RB, true>(C, rt.gamma, rt.beta, rt.out, rt.mean, rt.rstd, m0 + r, step, cnt, H, rt.eps, lane);
            else         ln_fwd_rows_batch<4, RB, true>(C, rt.gamma, rt.beta, rt.out, rt.mean, rt.rstd, m0 + r, step, cnt, H, rt.eps, lane);
        } else {
            const bool use_drop = rt.drop.p > 0.f;
            const int nch = H >> 2;
            if (nc <= 3) {
                float gv[3][4];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ch = lane + 64 * c;
                    if (ch < nch) unpack4(*reinterpret_cast<const u32x2*>(rt.gamma + ch * 4), gv[c]);
                    else gv[c][0] = gv[c][1] = gv[c][2] = gv[c][3] = 0.f;
                }
                ln_bwd_rows_batch<3, RB, true>(C, nullptr, rt.z, rt.mean, rt.rstd, gv, rt.out, rt.out2, m0 + r, step, cnt, H, use_drop, false,
                                         rt.drop, lane, false);
            } else {
                float gv[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ch = lane + 64 * c;
                    if (ch < nch) unpack4(*reinterpret_cast<const u32x2*>(rt.gamma + ch * 4), gv[c]);
                    else gv[c][0] = gv[c][1] = gv[c][2] = gv[c][3] = 0.f;
                }
                ln_bwd_rows_batch<4, RB, true>(C, nullptr, rt.z, rt.mean, rt.rstd, gv, rt.out, rt.out2, m0 + r, step, cnt, H, use_drop, false,
                                         rt.drop, lane, false);
            }
        }
    }
}

}  // namespace
