// l2_carry_probe.cpp — does an XCD's L2 keep what a kernel wrote for the NEXT kernel on the same stream?
// (EXPERIMENTS.md section 10.7 item 1: a row-block -> XCD affinity across the kernel chain only pays if it does.)
// Kernel W: workgroup b (256 of them; hardware block b runs on XCD b % 8) writes slice b (64 KiB) with the default store policy
// , write-through (sc1) or non-temporal.  Kernel R: workgroup b reads slice (b + shift) % 256 and stamps the wall clock around the read.
// shift 0: the reader sits where the writer sat; shift 8: another CU of the same XCD; shift 1: the neighbouring XCD.
// build: hipcc --offload-arch=gfx950 -O3 -o aux_bin/l2_carry_probe tests/native/l2_carry_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gmem_u32x4;
constexpr int SLICE = 64 * 1024, WGS = 256, THREADS = 256;

template <int POLICY>
__global__ __launch_bounds__(THREADS) void writer(char* buf, unsigned tag) {
    char* s = buf + (size_t)blockIdx.x * SLICE;
    for (int i = threadIdx.x * 16; i < SLICE; i += THREADS * 16) {
        u32x4 v = {tag, (unsigned)i, blockIdx.x, 1u};
        if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((gmem_u32x4*)(s + i)), "v"(v) : "memory");
        else if (POLICY == 2) __builtin_nontemporal_store(v, (gmem_u32x4*)(s + i));
        else *(u32x4*)(s + i) = v;
    }
}
__global__ __launch_bounds__(THREADS) void reader(const char* buf, int shift, unsigned tag, unsigned long long* stamps, unsigned* bad) {
    const int src = ((int)blockIdx.x + shift) % WGS;
    const char* s = buf + (size_t)src * SLICE;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    u32x4 acc = {0u, 0u, 0u, 0u};
    unsigned wrong = 0;
#pragma unroll
    for (int k = 0; k < SLICE / (THREADS * 16); ++k) {
        const u32x4 v = *(const u32x4*)(s + (k * THREADS + threadIdx.x) * 16);
        wrong += (v[0] != tag) || (v[2] != (unsigned)src);
        acc += v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
    if (wrong || acc[3] == 0xffffffffu) atomicAdd(bad, wrong);
}

int main() {
    char* buf; unsigned long long* st; unsigned* bad;
    CHK(hipMalloc(&buf, (size_t)WGS * SLICE)); CHK(hipMalloc(&st, WGS * 8)); CHK(hipMalloc(&bad, 4));
    CHK(hipMemset(bad, 0, 4));
    std::vector<unsigned long long> h(WGS);
    unsigned tag = 100;
    for (int policy = 0; policy < 3; ++policy)
        for (int shift : {0, 8, 1, 0, 8, 1}) {
            double sum = 0; std::vector<double> med;
            for (int rep = 0; rep < 20; ++rep) {
                ++tag;
                if (policy == 0) hipLaunchKernelGGL(writer<0>, dim3(WGS), dim3(THREADS), 0, 0, buf, tag);
                else if (policy == 1) hipLaunchKernelGGL(writer<1>, dim3(WGS), dim3(THREADS), 0, 0, buf, tag);
                else hipLaunchKernelGGL(writer<2>, dim3(WGS), dim3(THREADS), 0, 0, buf, tag);
                hipLaunchKernelGGL(reader, dim3(WGS), dim3(THREADS), 0, 0, (const char*)buf, shift, tag, st, bad);
                CHK(hipDeviceSynchronize());
                CHK(hipMemcpy(h.data(), st, WGS * 8, hipMemcpyDeviceToHost));
                std::vector<unsigned long long> v(h); std::sort(v.begin(), v.end());
                med.push_back((double)v[WGS / 2] * 0.01);
            }
            std::sort(med.begin(), med.end());
            (void)sum;
            printf("writer stores %-13s reader shift %d (%s): median time of a workgroup's 64 KiB read %.2f us\n", policy == 0 ? "default" : (policy == 1 ? "write-through" : "non-temporal"), shift,
                   shift == 0 ? "same CU slot" : (shift % 8 == 0 ? "same XCD" : "other XCD"), med[med.size() / 2]);
        }
    unsigned hb = 0; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("wrong words read: %u\n", hb);
    return 0;
}
