// Feasibility probe for a persistent per-XCD layer kernel (EXPERIMENTS.md section 8): measures, on one MI355X,
//   E1  the latency of a barrier among the workgroups that share one XCD (L2-resident counter, L1-bypassing poll),
//   E2  the observed blockIdx -> XCD / CU placement of a 256-workgroup, one-per-CU grid,
//   E3  whether a tile written with plain stores by one CU is read correctly by another CU of the SAME XCD
//       (decided at run time from HW_REG_XCC_ID, never from blockIdx) through each load flavour, with the
//       reader's L1 deliberately holding the previous round's lines; and, for contrast, by a CU of ANOTHER XCD.
// Every spin is bounded; a timeout word makes all workgroups leave.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tests/native/xcd_probe.cpp -o probe_bin/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static constexpr int kThreads = 512;
static constexpr int kBufWords = 4096;          // 16 KB per workgroup
static constexpr unsigned kSpinLimit = 1u << 18;

struct Ctl {
    unsigned joined[8];
    unsigned total;
    unsigned timeout;
    unsigned pad[22];
    unsigned bar[8 * 32];                       // one 128-byte line per XCD
};

struct Rec { unsigned xcc, hwid, role, nx; unsigned long long t_bar, t_all; unsigned errs[8]; };
static constexpr int kMaxRounds = 256;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ unsigned hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld_sc0sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void atom_add(unsigned* p, unsigned x, bool sc1) {
    if (sc1) asm volatile("global_atomic_add %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
    else     asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ unsigned atom_add_ret(unsigned* p, unsigned x, bool sc1) {
    unsigned v;
    if (sc1) asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(x) : "memory");
    else     asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(x) : "memory");
    return v;
}

// barrier among the n workgroups of one XCD: monotonic counter, target = arrivals so far
// atom: bit0 = sc1 on the arriving atomic; poll: 0 sc1 load, 1 sc0 sc1 load, 2 returning atomic add of 0
__device__ bool xbar(unsigned* cnt, unsigned target, Ctl* c, int atom, int poll, unsigned* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        atom_add(cnt, 1u, atom & 1);
        unsigned ok = 1;
        for (unsigned spins = 0;; ++spins) {
            unsigned v = poll == 0 ? ld_sc1(cnt) : poll == 1 ? ld_sc0sc1(cnt) : atom_add_ret(cnt, 0u, atom & 1);
            if ((int)(v - target) >= 0) break;
            if (spins > kSpinLimit || ((spins & 255u) == 255u && ld_sc0sc1(&c->timeout))) {
                __hip_atomic_store(&c->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

__device__ __forceinline__ unsigned pattern(unsigned round, unsigned owner, unsigned idx) {
    return round * 0x9E3779B1u + (owner << 14) + idx;
}

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int FLAVOUR>
__device__ __forceinline__ u4 load16(const u4* p) {
    u4 v;
    if (FLAVOUR == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FLAVOUR == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FLAVOUR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FLAVOUR == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (FLAVOUR == 4) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// rounds of: warm L1 with the partner's old tile -> write own tile -> barrier -> read the partner's tile -> barrier
//   store_sc1: write-through stores; cross: partner sits on the next XCD instead of the same one
__global__ __launch_bounds__(kThreads) void probe_kernel(Ctl* c, Rec* rec, unsigned* bufs, unsigned* round_errs, int rounds, int bar_rounds,
                                                         int atom, int poll, int store_sc1, int cross, int prewarm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_role, s_nx, s_flag;
    const unsigned xcc = xcc_id() & 7u;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_role = atom_add_ret(&c->joined[xcc], 1u, true);
        atom_add(&c->total, 1u, true);
        unsigned ok = 1;
        for (unsigned spins = 0;; ++spins) {                       // everyone resident?
            if (ld_sc0sc1(&c->total) == gridDim.x) break;
            if (spins > kSpinLimit) { __hip_atomic_store(&c->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        s_nx = ld_sc0sc1(&c->joined[xcc]);
        s_flag = ok;
    }
    __syncthreads();
    Rec& r = rec[blockIdx.x];
    const unsigned role = s_role, nx = s_nx;
    if (tid == 0) { r.xcc = xcc; r.hwid = hw_id(); r.role = role; r.nx = nx; }
    if (!s_flag) return;
    unsigned* cnt = &c->bar[xcc * 32];
    unsigned target = 0;

    // E1: barrier latency alone
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < bar_rounds; ++i) {
        target += nx;
        if (!xbar(cnt, target, c, atom, poll, &s_flag)) return;
    }
    unsigned long long t1 = wall_clock64();
    if (tid == 0) r.t_bar = t1 - t0;

    // E3: hand-off through each load flavour
    const unsigned me = xcc * 64 + role;                                          // buffer slot
    unsigned pxcc = cross ? (xcc + 1) & 7u : xcc;
    unsigned prole = cross ? role : (role + 1) % nx;
    if (cross) { unsigned pn = ld_sc0sc1(&c->joined[pxcc]); if (prole >= pn) prole = pn - 1; }
    const unsigned partner = pxcc * 64 + prole;
    u4* mine = reinterpret_cast<u4*>(bufs + (size_t)me * kBufWords);
    const u4* theirs = reinterpret_cast<const u4*>(bufs + (size_t)partner * kBufWords);
    unsigned errs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned sink = 0;
    unsigned* lds = reinterpret_cast<unsigned*>(smem);
    for (int rd = 1; rd <= rounds; ++rd) {
        const int flavour = rd % 7;
        for (int k = 0; k < 2 * prewarm; ++k) {                                   // L1 <- the partner's previous tile
            u4 v = load16<0>(theirs + tid + k * kThreads);
            sink ^= v.x ^ v.w;
        }
        for (int k = 0; k < 2; ++k) {
            const unsigned idx = (tid + k * kThreads) * 4;
            u4 v = {pattern(rd, me, idx), pattern(rd, me, idx + 1), pattern(rd, me, idx + 2), pattern(rd, me, idx + 3)};
            u4* q = mine + tid + k * kThreads;
            if (store_sc1 == 1)      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
            else if (store_sc1 == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
            else if (store_sc1 == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
            else if (store_sc1 == 4) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
            else                     asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
        }
        if (store_sc1 == 5) {                                                      // plain stores + this XCD's L2 written back
            asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
        target += nx;
        if (!xbar(cnt, target, c, atom, poll, &s_flag)) return;
        if (cross) {                                                               // cross-XCD needs a chip-wide meeting: crude one
            if (tid == 0) {
                atom_add(&c->pad[0], 1u, true);
                for (unsigned spins = 0; spins < kSpinLimit; ++spins) { if ((int)(ld_sc0sc1(&c->pad[0]) - gridDim.x * (2 * rd - 1)) >= 0) break; __builtin_amdgcn_s_sleep(2); }
            }
            __syncthreads();
        }
        unsigned rbad = 0;
        for (int k = 0; k < 2; ++k) {
            const unsigned idx = (tid + k * kThreads) * 4;
            const u4* p = theirs + tid + k * kThreads;
            u4 v;
            if (flavour == 0) v = load16<0>(p);
            else if (flavour == 1) v = load16<1>(p);
            else if (flavour == 2) v = load16<2>(p);
            else if (flavour == 3) v = load16<3>(p);
            else if (flavour == 4) v = load16<4>(p);
            else {                                                                  // 5: LDS-DMA plain, 6: LDS-DMA sc1
                // wave-uniform LDS base + lane*16 is implied by the instruction; every wave owns 1 KB per k
                unsigned* dst = lds + (k * kThreads + (tid & ~63)) * 4;
                if (flavour == 5) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                else              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)dst, 16, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_waitcnt(0);
                const unsigned* q = lds + (k * kThreads + tid) * 4;
                v = u4{q[0], q[1], q[2], q[3]};
            }
            const unsigned bad = (v.x != pattern(rd, partner, idx)) + (v.y != pattern(rd, partner, idx + 1)) +
                                 (v.z != pattern(rd, partner, idx + 2)) + (v.w != pattern(rd, partner, idx + 3));
            errs[flavour] += bad;
            rbad += bad;
            if (bad && (flavour == 1 || flavour == 6) && blockIdx.x < 2) {
                unsigned slot = atomicAdd(&round_errs[0], 1u);
                if (slot < 48) { unsigned* d = round_errs + kMaxRounds + slot * 8; d[0] = rd; d[1] = blockIdx.x; d[2] = idx; d[3] = bad;
                                 d[4] = v.x - pattern(rd, partner, idx); d[5] = v.y - pattern(rd, partner, idx + 1); d[6] = v.z - pattern(rd, partner, idx + 2); d[7] = v.w - pattern(rd, partner, idx + 3); }
            }
        }
        if (rbad && rd < kMaxRounds) atomicAdd(&round_errs[rd], rbad);
        target += nx;
        if (!xbar(cnt, target, c, atom, poll, &s_flag)) return;
        if (cross) {
            if (tid == 0) {
                atom_add(&c->pad[0], 1u, true);
                for (unsigned spins = 0; spins < kSpinLimit; ++spins) { if ((int)(ld_sc0sc1(&c->pad[0]) - gridDim.x * (2 * rd)) >= 0) break; __builtin_amdgcn_s_sleep(2); }
            }
            __syncthreads();
        }
    }
    unsigned long long t2 = wall_clock64();
    for (int f = 0; f < 7; ++f) {
        unsigned e = errs[f];
        for (int o = 32; o; o >>= 1) e += __shfl_xor(e, o);
        if ((tid & 63) == 0 && e) atomicAdd(&r.errs[f], e);
    }
    if (tid == 0) { r.t_all = t2 - t1; if (sink == 0x12345678u) r.errs[7] = sink; }
}

int main(int argc, char** argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 256;
    int rounds = argc > 2 ? atoi(argv[2]) : 70;
    int bar_rounds = argc > 3 ? atoi(argv[3]) : 400;
    int prewarm = argc > 4 ? atoi(argv[4]) : 1;
    Ctl* c; Rec* rec; unsigned* bufs; unsigned* round_errs;
    CK(hipMalloc(&round_errs, sizeof(unsigned) * (kMaxRounds + 48 * 8)));
    CK(hipMalloc(&c, sizeof(Ctl)));
    CK(hipMalloc(&rec, sizeof(Rec) * grid));
    CK(hipMalloc(&bufs, sizeof(unsigned) * kBufWords * 8 * 64));
    const int lds_bytes = 96 * 1024;                                               // one workgroup per CU
    CK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    int clk_khz = 100000;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    printf("wall clock %d kHz, grid %d x %d threads, %d hand-off rounds, %d barrier rounds\n", clk_khz, grid, kThreads, rounds, bar_rounds);
    const char* st[6] = {"plain", "sc1", "sc0", "sc0 sc1", "nt", "plain+wbl2"};
    const char* fl[7] = {"plain", "sc1", "sc0 sc1", "sc0", "nt", "lds-dma", "lds-dma sc1"};
    bool placement_done = false;
    for (int cross = 0; cross < 2; ++cross)
    for (int store_sc1 = 0; store_sc1 < 6; ++store_sc1)
    for (int atom = 0; atom < 2; ++atom)
    for (int poll = 0; poll < 3; ++poll) {
        if ((cross || store_sc1) && (atom != 1 || poll != 0)) continue;            // the data matrix needs one barrier form only
        CK(hipMemset(c, 0, sizeof(Ctl)));
        CK(hipMemset(rec, 0, sizeof(Rec) * grid));
        CK(hipMemset(round_errs, 0, sizeof(unsigned) * (kMaxRounds + 48 * 8)));
        CK(hipMemset(bufs, 0, sizeof(unsigned) * kBufWords * 8 * 64));
        hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(kThreads), lds_bytes, 0, c, rec, bufs, round_errs, rounds, bar_rounds, atom, poll, store_sc1, cross, prewarm);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        Ctl hc; std::vector<Rec> hr(grid);
        CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), rec, sizeof(Rec) * grid, hipMemcpyDeviceToHost));
        if (!placement_done) {
            placement_done = true;
            int mism = 0; std::vector<unsigned> cus;
            for (int b = 0; b < grid; ++b) { mism += (hr[b].xcc != (unsigned)(b & 7)); cus.push_back((hr[b].xcc << 16) | (hr[b].hwid & 0xff00u)); }
            std::sort(cus.begin(), cus.end());
            int distinct = (int)(std::unique(cus.begin(), cus.end()) - cus.begin());
            printf("E2 placement: joined per XCD = %u %u %u %u %u %u %u %u | blocks with xcc != blockIdx%%8: %d | distinct (xcc, se/sh/cu) = %d of %d\n",
                   hc.joined[0], hc.joined[1], hc.joined[2], hc.joined[3], hc.joined[4], hc.joined[5], hc.joined[6], hc.joined[7], mism, distinct, grid);
        }
        unsigned long long tb = 0, ta = 0; unsigned long long errs[7] = {0};
        for (int b = 0; b < grid; ++b) { tb = std::max(tb, hr[b].t_bar); ta = std::max(ta, hr[b].t_all); for (int f = 0; f < 7; ++f) errs[f] += hr[b].errs[f]; }
        const double us_per_tick = 1e3 / clk_khz;
        printf("cross=%d store=%s atomic%s poll=%s timeout=%u | E1 barrier %.3f us | hand-off round (2 barriers + 16 KB each way) %.3f us\n",
               cross, st[store_sc1], atom ? " sc1" : "", poll == 0 ? "sc1-load" : poll == 1 ? "sc0sc1-load" : "atomic", hc.timeout,
               bar_rounds ? tb * us_per_tick / bar_rounds : 0.0, rounds ? ta * us_per_tick / rounds : 0.0);
        if (atom == 1 && poll == 0) {
            const int per = rounds / 7;
            printf("   E3 wrong words of %lld per flavour:", (long long)per * grid * 4096);
            for (int f = 0; f < 7; ++f) printf("  %s=%llu", fl[f], errs[f]);
            printf("\n");
            std::vector<unsigned> re(kMaxRounds + 48 * 8);
            CK(hipMemcpy(re.data(), round_errs, sizeof(unsigned) * (kMaxRounds + 48 * 8), hipMemcpyDeviceToHost));
            for (unsigned i = 0; i < std::min(re[0], 48u); ++i) { unsigned* d = re.data() + kMaxRounds + i * 8; printf("      wrong: round %u block %u word %u (%u bad) got-expected = %08x %08x %08x %08x  [one round back = %08x]\n", d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], 0u - 0x9E3779B1u); }
            printf("   wrong words per round (L1-bypassing flavours only):");
            for (int rd = 1; rd <= rounds && rd < kMaxRounds; ++rd) { int f = rd % 7; if (f == 1 || f == 2 || f == 4 || f == 6) printf(" %u", re[rd]); }
            int bad_wg = 0; for (int b = 0; b < grid; ++b) bad_wg += (hr[b].errs[1] + hr[b].errs[2] + hr[b].errs[4] + hr[b].errs[6]) != 0;
            printf("\n   workgroups that ever read a wrong word through an L1-bypassing flavour: %d of %d\n", bad_wg, grid);
        }
        fflush(stdout);
    }
    return 0;
}
