// l2_feed_probe.cpp — how many bytes per clock can ONE CU pull out of its XCD's L2 when every CU does the same?
// (round 5: the K loops of the chain GEMMs run at ~27-36 B/clk per CU whatever the tile; is that the L2, the vector-memory path of
//  a CU, or the LDS-DMA form of the load?)
// Every workgroup streams the same L2-resident region (default 2 MiB, re-read `reps` times) in 1 KiB wave pieces, as a GEMM loader
// does, in one of three forms:
//   reg   global_load_dwordx4 -> VGPR (scalar origin + 32-bit lane offset), DEPTH loads in flight per wave, results xor-ed
//   dma   global_load_lds_dwordx4 -> an LDS ring (scalar origin + lane offset, M0 = ring slot), DEPTH pieces in flight per wave
//   mix   half of the waves each way
// for 4 / 8 / 16 waves per workgroup, one workgroup per CU (and two for the small ones), grid = 256.  Prints GB/s per CU, B/clk per
// CU at the measured clock (s_memtime / s_memrealtime), chip TB/s.
// build: hipcc --offload-arch=gfx950 -O3 -o aux_bin/l2_feed_probe tests/native/l2_feed_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned voff, const void* origin, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(origin), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ u32x4 ld16(unsigned voff, const void* origin) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(origin) : "memory");
    return v;
}

// MODE 0 reg, 1 dma, 2 mix (even waves reg, odd waves dma).  Each wave walks pieces (1 KiB) w, w + nw, ... of the region.
template <int MODE, int DEPTH>
__global__ void feed(const char* __restrict__ base, const unsigned region_bytes, const int reps, unsigned* sink, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nw = (int)blockDim.x >> 6;
    const unsigned pieces = region_bytes >> 10;
    // stagger the workgroups over the region so that the L2 channels see different lines at the same time
    const unsigned start = ((unsigned)blockIdx.x * 37u) % pieces;
    const unsigned voff = (unsigned)lane * 16u;
    typedef __attribute__((address_space(3))) char lds_char_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem + (unsigned)wid * (DEPTH * 1024u);
    const bool use_dma = MODE == 1 || (MODE == 2 && (wid & 1));
    u32x4 acc = {0u, 0u, 0u, 0u};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned p = (start + (unsigned)wid) % pieces;
    const unsigned total = (pieces / (unsigned)nw) * (unsigned)reps;      // pieces this wave fetches
    if (use_dma) {
        for (unsigned k = 0; k < total; k += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                dma16(voff, base + (size_t)p * 1024u, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + d * 1024u)));
                p += (unsigned)nw; if (p >= pieces) p -= pieces;
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");      // half of the ring stays in flight
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        u32x4 r[DEPTH];
        for (unsigned k = 0; k < total; k += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                r[d] = ld16(voff, base + (size_t)p * 1024u);
                p += (unsigned)nw; if (p >= pieces) p -= pieces;
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < DEPTH / 2; ++d) acc ^= r[d];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = DEPTH / 2; d < DEPTH; ++d) acc ^= r[d];
        }
    }
    __syncthreads();
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc[2] ^ acc[3];
    if (use_dma && smem[threadIdx.x] == 0x7f && smem[threadIdx.x + 1] == 0x3c && lane == 63) sink[1] = 1;
}

template <int MODE, int DEPTH>
static int run(const char* name, const char* d_base, unsigned region, int waves, int wgs, unsigned* sink, unsigned long long* d_cyc) {
    const int reps = 24;
    const size_t lds = (size_t)waves * DEPTH * 1024;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&feed<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(wgs), dim3(waves * 64), lds, 0, d_base, region, reps, sink, d_cyc);
    CHK(hipEventRecord(e0, 0));
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(wgs), dim3(waves * 64), lds, 0, d_base, region, reps, sink, d_cyc);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(wgs);
    CHK(hipMemcpy(h.data(), d_cyc, wgs * 8, hipMemcpyDeviceToHost));
    double cs = 0;
    for (auto c : h) cs += (double)c;
    cs /= wgs;
    const unsigned pieces = region >> 10;
    const double bytes_per_wg = (double)(pieces / (unsigned)waves) * waves * reps * 1024.0;
    const double us = ms * 1000.0 / iters;
    printf("  %-4s depth %2d  %2d waves x %3d workgroups : %8.1f us/launch  %6.1f GB/s per workgroup  %5.1f B/clk per workgroup (in-kernel cycles)  %6.2f TB/s chip\n",
           name, DEPTH, waves, wgs, us, bytes_per_wg / us * 1e-3, bytes_per_wg / cs, bytes_per_wg * wgs / us * 1e-6);
    return 0;
}

// MODE 3 ("frag"): MFMA-fragment-shaped register loads — lane (g = lane >> 4, i = lane & 15) fetches the 16 bytes of k chunk g (+ 4 for
// the second instruction) of row i of a 16-row block of a row-major [rows][row_bytes] matrix: 16 rows x 64 contiguous bytes per
// instruction, two instructions per 16 x 128-byte block (what a compute wave would issue to take an operand straight from global
// memory instead of through LDS).  Blocks are walked along the row (k) first.
template <int DEPTH>
__global__ void feed_frag(const char* __restrict__ base, const unsigned region_bytes, const unsigned row_bytes, const int reps, unsigned* sink,
                          unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nw = (int)blockDim.x >> 6;
    const unsigned kblocks = row_bytes / 128u;                    // 128-byte blocks per row
    const unsigned rblocks = region_bytes / (row_bytes * 16u);    // 16-row blocks
    const unsigned blocks = kblocks * rblocks;
    const unsigned voff = (unsigned)(lane & 15) * row_bytes + (unsigned)(lane >> 4) * 16u;
    u32x4 acc = {0u, 0u, 0u, 0u};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned b = (((unsigned)blockIdx.x * 37u) + (unsigned)wid) % blocks;
    const unsigned total = (blocks / (unsigned)nw) * (unsigned)reps;          // blocks this wave fetches (2 instructions each)
    u32x4 r[DEPTH];
    for (unsigned k = 0; k < total; k += DEPTH / 2) {
#pragma unroll
        for (int d = 0; d < DEPTH / 2; ++d) {
            const unsigned rb = b / kblocks, kb = b - rb * kblocks;
            const char* o = base + (size_t)rb * 16u * row_bytes + (size_t)kb * 128u;
            r[2 * d] = ld16(voff, o);
            r[2 * d + 1] = ld16(voff, o + 64);
            b += (unsigned)nw; if (b >= blocks) b -= blocks;
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < DEPTH / 2; ++d) acc ^= r[d];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = DEPTH / 2; d < DEPTH; ++d) acc ^= r[d];
    }
    __syncthreads();
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc[2] ^ acc[3];
}

template <int DEPTH>
static int run_frag(const char* d_base, unsigned region, unsigned row_bytes, int waves, int wgs, unsigned* sink, unsigned long long* d_cyc) {
    const int reps = 24;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((feed_frag<DEPTH>), dim3(wgs), dim3(waves * 64), 0, 0, d_base, region, row_bytes, reps, sink, d_cyc);
    CHK(hipEventRecord(e0, 0));
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((feed_frag<DEPTH>), dim3(wgs), dim3(waves * 64), 0, 0, d_base, region, row_bytes, reps, sink, d_cyc);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(wgs);
    CHK(hipMemcpy(h.data(), d_cyc, wgs * 8, hipMemcpyDeviceToHost));
    double cs = 0;
    for (auto c : h) cs += (double)c;
    cs /= wgs;
    const unsigned blocks = (row_bytes / 128u) * (region / (row_bytes * 16u));
    const double bytes_per_wg = (double)(blocks / (unsigned)waves) * waves * reps * 2048.0;
    const double us = ms * 1000.0 / iters;
    printf("  frag depth %2d  row %5u B  %2d waves x %3d workgroups : %8.1f us/launch  %6.1f GB/s per workgroup  %5.1f B/clk per workgroup  %6.2f TB/s chip\n",
           DEPTH, row_bytes, waves, wgs, us, bytes_per_wg / us * 1e-3, bytes_per_wg / cs, bytes_per_wg * wgs / us * 1e-6);
    return 0;
}

int main(int argc, char** argv) {
    const unsigned region = (argc > 1 ? (unsigned)atoi(argv[1]) : 2048u) * 1024u;     // KiB
    char* d_base; unsigned* sink; unsigned long long* d_cyc;
    CHK(hipMalloc(&d_base, region)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&d_cyc, 4096 * 8));
    CHK(hipMemset(d_base, 0x3c, region)); CHK(hipMemset(sink, 0, 64));
    printf("region %u KiB (every workgroup streams all of it, 24 times, in 1 KiB wave pieces)\n", region >> 10);
    for (int waves : {4, 8, 16}) {
        for (int wgs : {256, 64}) {
            if (run<0, 8>("reg", d_base, region, waves, wgs, sink, d_cyc)) return 1;
            if (run<0, 16>("reg", d_base, region, waves, wgs, sink, d_cyc)) return 1;
            if (run<1, 8>("dma", d_base, region, waves, wgs, sink, d_cyc)) return 1;
            if (waves <= 8 && run<1, 16>("dma", d_base, region, waves, wgs, sink, d_cyc)) return 1;
            if (run<2, 8>("mix", d_base, region, waves, wgs, sink, d_cyc)) return 1;
        }
    }
    // fragment-shaped register loads (rows of 1 536 B = K 768 and 6 144 B = K 3072)
    for (unsigned rowb : {1536u, 6144u})
        for (int waves : {4, 8}) {
            if (run_frag<8>(d_base, region, rowb, waves, 256, sink, d_cyc)) return 1;
            if (run_frag<16>(d_base, region, rowb, waves, 256, sink, d_cyc)) return 1;
        }
    if (run_frag<16>(d_base, region, 1536u, 4, 64, sink, d_cyc)) return 1;
    // two workgroups per CU
    if (run<0, 8>("reg", d_base, region, 8, 512, sink, d_cyc)) return 1;
    if (run<1, 8>("dma", d_base, region, 8, 512, sink, d_cyc)) return 1;
    if (run<1, 8>("dma", d_base, region, 4, 512, sink, d_cyc)) return 1;
    return 0;
}
