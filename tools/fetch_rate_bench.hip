// Per-CU operand-fetch rate: global_load_lds (LDS-DMA) vs global_load_dwordx4 -> VGPR -> ds_write_b128, data L2/MALL resident.
// Each workgroup (256 threads) streams its own 64 x K bf16 panel (row pitch K*2 B) in K-steps of 64 columns (8 KiB per step),
// like one operand of the 64x64 decode GEMM tile.  Build: hipcc -O3 --offload-arch=gfx950 tools/fetch_rate_bench.hip -o tools/_bin/fetch_rate_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef SHARE
#define SHARE 4   // distinct panels: 4 x 786 KB = 3 MB, L2-resident on every XCD
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// rows: 2 x 64 rows per workgroup (an "A" and a "W" panel), STAGES-deep ring, 16 KiB per stage
template <int STAGES>
__global__ __launch_bounds__(256) void glds_kernel(const uint16_t *src, int K, int nk, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t *base = src + (size_t)(blockIdx.x % SHARE) * 128 * K;
    const uint16_t *p[4];
    for (int i = 0; i < 4; i++) p[i] = base + (size_t)((wave + 4 * i) * 8 + (lane >> 3)) * K + (lane & 7) * 8;
    auto issue = [&](int kt, int slot) {
        char *sb = smem + slot * 16384;
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p[i] + (size_t)kt * 64),
                                             (__attribute__((address_space(3))) void *)(sb + (wave + 4 * i) * 1024), 16, 0, 0);
    };
    for (int s = 0; s < STAGES - 1; s++) if (s < nk) issue(s, s);
    uint32_t acc = 0;
    int slot = 0;
    for (int kt = 0; kt < nk; kt++) {
        const int ahead = nk - 1 - kt;
        const int inflight = ahead < STAGES - 2 ? ahead : STAGES - 2;
        if (inflight >= 2) wait_vmcnt<8>(); else if (inflight == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < nk) { int ns = slot + STAGES - 1; ns = ns >= STAGES ? ns - STAGES : ns; issue(kt + STAGES - 1, ns); }
        acc ^= *reinterpret_cast<const uint32_t *>(smem + slot * 16384 + tid * 16);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// register-staged: U K-steps in flight in VGPRs (4 x 16 B per thread per step), then ds_write_b128
template <int U>
__global__ __launch_bounds__(256) void reg_kernel(const uint16_t *src, int K, int nk, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t *base = src + (size_t)(blockIdx.x % SHARE) * 128 * K;
    const uint16_t *p[4];
    for (int i = 0; i < 4; i++) p[i] = base + (size_t)((wave + 4 * i) * 8 + (lane >> 3)) * K + (lane & 7) * 8;
    u32x4 r[U][4];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) r[u][i] = u < nk ? *reinterpret_cast<const u32x4 *>(p[i] + (size_t)u * 64) : u32x4{0, 0, 0, 0};
    uint32_t acc = 0;
    for (int kt = 0; kt < nk; kt += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            char *sb = smem + ((kt + u) & 1) * 16384;
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<u32x4 *>(sb + (wave + 4 * i) * 1024 + lane * 16) = r[u][i];
            const int nxt = kt + u + U;
#pragma unroll
            for (int i = 0; i < 4; i++) if (nxt < nk) r[u][i] = *reinterpret_cast<const u32x4 *>(p[i] + (size_t)nxt * 64);
            __syncthreads();
            acc ^= *reinterpret_cast<const uint32_t *>(sb + tid * 16);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const int K = 3072, nk = K / 64;
    for (int blocks : {192, 256, 512, 768}) {
        uint16_t *src; uint32_t *sink;
        const size_t bytes = (size_t)blocks * 128 * K * 2;
        CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&sink, 16));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto run = [&](int which) {
            switch (which) {
            case 0: hipLaunchKernelGGL(glds_kernel<2>, dim3(blocks), dim3(256), 2 * 16384, 0, src, K, nk, sink); break;
            case 1: hipLaunchKernelGGL(glds_kernel<4>, dim3(blocks), dim3(256), 4 * 16384, 0, src, K, nk, sink); break;
            case 2: hipLaunchKernelGGL(reg_kernel<2>, dim3(blocks), dim3(256), 2 * 16384, 0, src, K, nk, sink); break;
            case 3: hipLaunchKernelGGL(reg_kernel<4>, dim3(blocks), dim3(256), 2 * 16384, 0, src, K, nk, sink); break;
            }
        };
        const char *names[4] = {"glds S2", "glds S4", "reg U2", "reg U4"};
        for (int w = 0; w < 4; w++) {
            for (int i = 0; i < 5; i++) run(w);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 50; i++) run(w);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 50;
            const int active = blocks < 256 ? blocks : 256;
            printf("blocks %4d  %-8s %7.2f us  %6.2f TB/s aggregate  %5.1f GB/s per active CU (%4.1f B/clk @2.4GHz)\n", blocks, names[w], us,
                   bytes / us / 1e6, bytes / us / 1e3 / active, bytes / us / 1e3 / active / 2.4);
        }
        CK(hipFree(src)); CK(hipFree(sink));
    }
    return 0;
}
