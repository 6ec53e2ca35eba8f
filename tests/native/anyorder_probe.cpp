// Feasibility probe for overlapped kernel chains (EXPERIMENTS.md section 10): can dependent kernels of ONE stream be dispatched
// without the barrier between them (hipExtLaunchKernel + hipExtAnyOrderLaunch) and synchronise through row-block flags
// instead — and which store / load flavours make the hand-off coherent across XCDs without a kernel boundary's cache
// write-back / invalidate?  Measures, on one MI355X:
//   T1  time per kernel of a chain of K dependent kernels (256 workgroups, each ~W us of skewed busy work, reads the 16 KB
//       row block its producer wrote on ANOTHER XCD, checks every word, writes its own), launched (a) normally, (b) any-order
//       with flags, for store flavours {plain + release fence, sc1 write-through, plain without fence} x load flavours
//       {plain, plain after an acquire fence, sc1} and buffer rings of 2 (aggressively recycled) or K (fresh) buffers;
//       wrong words are counted (a stale read shows as an old generation number);
//   T2  bandwidth of L2-resident re-reads through plain vs sc1 loads (what sc1 operand loads would cost a GEMM).
// Every spin is bounded by a wall-clock limit and a global timeout word: a broken assumption ends in a report, not a hang.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tests/native/anyorder_probe.cpp -o aux_bin/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static constexpr int kThreads = 256;
static constexpr int kWords = 4096;            // 16 KB per row block
static constexpr int kBlocks = 256;

struct P {
    const unsigned* in;        // previous generation (null for the first kernel)
    unsigned* out;
    const unsigned* wait;      // producer's per-row-block counters (null: no flag wait — normal launch order protects us)
    unsigned* sig;             // this kernel's per-row-block counters
    unsigned* err;             // [2]: wrong words, timeouts
    unsigned* timeout;
    unsigned expect;           // generation number the input must hold
    int shift;                 // row block = (blockIdx + shift) % kBlocks: consecutive kernels use different shifts, so a
                               // row block is produced and consumed on different XCDs (blockIdx % 8 picks the XCD)
    int store_mode;            // 0 plain + release fence before the signal, 1 sc1 stores, 2 plain, no fence
    int load_mode;             // 0 plain, 1 plain after an acquire fence, 2 sc1
    int work_us;               // busy work per workgroup, skewed by blockIdx
    unsigned long long* stamps; // [2]: earliest workgroup start, latest workgroup end of this kernel (100 MHz wall clock)
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(kThreads) void step_kernel(const P p) {
    const int rb = ((int)blockIdx.x + p.shift) % kBlocks;
    __shared__ unsigned s_ok;
    if (threadIdx.x == 0 && p.stamps) atomicMin(p.stamps, wall_clock64());
    if (threadIdx.x == 0) {
        unsigned ok = 1;
        if (p.wait != nullptr) {
            const unsigned long long t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                const unsigned v = __hip_atomic_load(p.wait + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= 1u) break;
                if ((spins & 63u) == 63u) {
                    if (wall_clock64() - t0 > 20000000ull /* 0.2 s at 100 MHz */ ||
                        __hip_atomic_load(p.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(p.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicAdd(p.err + 1, 1u);
                        ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (p.load_mode == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);      // agent scope by default on device: buffer_inv sc1
        }
        s_ok = ok;
    }
    __syncthreads();
    unsigned bad = 0;
    unsigned vals[kWords / kThreads];
    if (p.in != nullptr && s_ok) {
        const unsigned* src = p.in + (size_t)rb * kWords;
#pragma unroll
        for (int i = 0; i < kWords / kThreads; ++i) {
            const unsigned v = p.load_mode == 2 ? ld_sc1(src + i * kThreads + threadIdx.x) : src[i * kThreads + threadIdx.x];
            vals[i] = v;
            bad += (v != p.expect);
        }
    } else {
#pragma unroll
        for (int i = 0; i < kWords / kThreads; ++i) vals[i] = p.expect;
    }
    // skewed busy work: 1 + (blockIdx % 4) / 4 of work_us
    {
        const unsigned long long t0 = wall_clock64();
        const unsigned long long dur = (unsigned long long)p.work_us * 100ull * (4 + (blockIdx.x & 3)) / 4;
        float acc = (float)vals[0];
        while (wall_clock64() - t0 < dur) {
#pragma unroll
            for (int j = 0; j < 64; ++j) acc = acc * 1.0001f + 0.5f;
        }
        if (acc == 123.456f) vals[0] += 1;           // keep the loop
    }
    unsigned* dst = p.out + (size_t)rb * kWords;
#pragma unroll
    for (int i = 0; i < kWords / kThreads; ++i) {
        const unsigned v = p.expect + 1u;          // (an error does not propagate: every generation is checked on its own)
        if (p.store_mode == 1) st_sc1(dst + i * kThreads + threadIdx.x, v);
        else dst[i * kThreads + threadIdx.x] = v;
    }
    if (bad) atomicAdd(p.err, bad);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (p.store_mode == 0) __hip_atomic_fetch_add(p.sig + rb, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // buffer_wbl2 sc1 first
        else __hip_atomic_fetch_add(p.sig + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.stamps) atomicMax(p.stamps + 1, wall_clock64());
    }
}

// T2: every workgroup re-reads the same `bytes` region `reps` times
__global__ __launch_bounds__(256) void reread_kernel(const uint4* src, size_t n16, int reps, int sc1, unsigned* sink) {
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i + 768 < n16; i += 1024) {
            uint4 v0, v1, v2, v3;
            if (sc1) {
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                             "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                             : "v"(src + i), "v"(src + i + 256), "v"(src + i + 512), "v"(src + i + 768) : "memory");
            } else {
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                             "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                             : "v"(src + i), "v"(src + i + 256), "v"(src + i + 512), "v"(src + i + 768) : "memory");
            }
            acc += v0.x ^ v1.w ^ v2.y ^ v3.z;
        }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 120;
    const int work_us = argc > 2 ? atoi(argv[2]) : 6;
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *bufs, *cnt, *err, *timeout;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, (size_t)(K + 1) * 16));
    std::vector<unsigned long long> hst((size_t)(K + 1) * 2);
    const size_t bufWords = (size_t)kBlocks * kWords;
    CK(hipMalloc(&bufs, bufWords * 4 * (size_t)(K + 1)));
    CK(hipMalloc(&cnt, (size_t)(K + 1) * kBlocks * 4));
    CK(hipMalloc(&err, 16));
    CK(hipMalloc(&timeout, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    auto run = [&](bool anyorder, int store_mode, int load_mode, int NB, const char* label) {
        CK(hipMemsetAsync(bufs, 0xff, bufWords * 4 * (size_t)(K + 1), st));
        CK(hipMemsetAsync(err, 0, 16, st));
        CK(hipMemsetAsync(timeout, 0, 4, st));
        float best = 1e30f;
        double host_us = 0;
        unsigned herr[2] = {0, 0};
        for (int trial = 0; trial < 3; ++trial) {
            CK(hipMemsetAsync(cnt, 0, (size_t)(K + 1) * kBlocks * 4, st));
            for (int k = 0; k <= K; ++k) { hst[2 * k] = ~0ull; hst[2 * k + 1] = 0; }
            CK(hipMemcpyAsync(stamps, hst.data(), (size_t)(K + 1) * 16, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            const auto h0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < K; ++k) {
                P p{};
                p.in = k == 0 ? nullptr : bufs + (size_t)((k - 1) % NB) * bufWords;
                p.out = bufs + (size_t)(k % NB) * bufWords;
                p.wait = (anyorder && k > 0) ? cnt + (size_t)(k - 1) * kBlocks : nullptr;
                p.sig = cnt + (size_t)k * kBlocks;
                p.err = err;
                p.timeout = timeout;
                p.expect = (unsigned)(trial * 1000 + k);
                p.shift = (k * 3) % kBlocks;
                p.store_mode = store_mode;
                p.load_mode = load_mode;
                p.work_us = work_us;
                p.stamps = stamps + 2 * k;
                if (anyorder && k > 0) {
                    void* args[] = {(void*)&p};
                    CK(hipExtLaunchKernel((const void*)step_kernel, dim3(kBlocks), dim3(kThreads), args, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch));
                } else {
                    hipLaunchKernelGGL(step_kernel, dim3(kBlocks), dim3(kThreads), 0, st, p);
                }
            }
            CK(hipEventRecord(e1, st));
            const auto h1 = std::chrono::steady_clock::now();
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; host_us = std::chrono::duration<double, std::micro>(h1 - h0).count(); }
        }
        CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hst.data(), stamps, (size_t)(K + 1) * 16, hipMemcpyDeviceToHost));
        // of the last trial: how long a kernel lives (first workgroup start -> last workgroup end) and how far the next kernel's
        // first workgroup starts before (negative gap) or after (positive) this kernel's last workgroup ends
        double live = 0, gap = 0, period = 0;
        for (int k = 10; k < K - 1; ++k) {
            live += (double)(hst[2 * k + 1] - hst[2 * k]) * 0.01;
            gap += ((double)hst[2 * (k + 1)] - (double)hst[2 * k + 1]) * 0.01;
            period += ((double)hst[2 * (k + 1)] - (double)hst[2 * k]) * 0.01;
        }
        const int nk = K - 11;
        printf("%-58s %7.2f us/kernel (host issue %5.2f)  live %5.2f  next-start minus end %+6.2f  start-to-start %5.2f   wrong %u  timeouts %u\n",
               label, best * 1000.f / K, host_us / K, live / nk, gap / nk, period / nk, herr[0], herr[1]);
        fflush(stdout);
        return herr[1] == 0;
    };

    printf("T1: chain of %d dependent kernels, 256 workgroups, busy work %d us x (1 .. 1.75)\n", K, work_us);
    run(false, 2, 0, K, "normal launches, plain stores / loads, fresh buffers");
    run(false, 1, 0, K, "normal launches, sc1 stores / plain loads, fresh buffers");
    run(true, 1, 0, K, "any-order + flags: sc1 stores / plain loads, fresh buffers");
    run(true, 1, 0, 2, "any-order + flags: sc1 stores / plain loads, 2 buffers");
    run(true, 0, 0, K, "any-order + flags: plain+release / plain loads, fresh");
    run(true, 1, 1, K, "any-order + flags: sc1 stores / acquire+plain, fresh");
t2:
    {
        printf("T2: 256 workgroups re-read one L2-sized region 16 times\n");
        unsigned* sink;
        CK(hipMalloc(&sink, 4));
        for (size_t kb : {256, 1024, 2048}) {
            const size_t n16 = kb * 1024 / 16;
            for (int sc1 = 0; sc1 < 2; ++sc1) {
                hipLaunchKernelGGL(reread_kernel, dim3(256), dim3(256), 0, st, (const uint4*)bufs, n16, 2, sc1, sink);
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(reread_kernel, dim3(256), dim3(256), 0, st, (const uint4*)bufs, n16, 16, sc1, sink);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  %4zu KB  %-5s loads: %8.1f us  -> %7.1f GB/s aggregate\n", kb, sc1 ? "sc1" : "plain", ms * 1000.f,
                       256.0 * 16 * kb * 1024 / (ms * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
