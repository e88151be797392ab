// tools/hbm_read_bench.hip — what HBM gives a ~115 MB read-only launch (the decode-attention KV sweep), cold (buffers cycle through 2.4+ GB):
//   contiguous vs the KV-cache shape (2048 chunks of T x 1536 B at a Tmax x 1536 B stride), by workgroup count and loads in flight.
// Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_read_bench.hip -o tools/_bin/hbm_read_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each workgroup reads `chunk_bytes` starting at base + blockIdx.x * stride_bytes; U 16-byte loads per thread in flight
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const char *base, size_t stride_bytes, int chunk_bytes, uint32_t *sink) {
    const char *p = base + (size_t)blockIdx.x * stride_bytes;
    u32x4 acc = {0, 0, 0, 0};
    const int n16 = chunk_bytes / 16;
    for (int i = threadIdx.x; i < n16; i += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int j = i + 256 * u; v[u] = *reinterpret_cast<const u32x4 *>(p + (size_t)(j < n16 ? j : n16 - 1) * 16); }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

int main() {
    const size_t layer_bytes = (size_t)2 * 1024 * 128 * 1536;      // K + V of one layer, Tmax = 128: 402 MB
    const int NL = 8;
    char *buf; uint32_t *sink;
    CK(hipMalloc(&buf, layer_bytes * NL)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, layer_bytes * NL));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char *name; int wgs; size_t stride; int chunk; };
    const int T = 37;
    std::vector<Case> cases = {
        {"contiguous 116 MB, 2048 WGs x 56.8 KB", 2048, (size_t)T * 1536, T * 1536},
        {"contiguous 116 MB, 1024 WGs x 113.7 KB", 1024, (size_t)2 * T * 1536, 2 * T * 1536},
        {"contiguous 116 MB, 4096 WGs x 28.4 KB", 4096, (size_t)T * 768, T * 768},
        {"KV shape: 2048 chunks x 56.8 KB at 196.6 KB stride", 2048, (size_t)128 * 1536, T * 1536},
        {"KV shape, 8192 WGs x 14.2 KB (4 per chunk)", 0, 0, 0},
    };
    for (auto &c : cases) {
        if (!c.wgs) continue;
        for (int U : {2, 4, 8}) {
            auto launch = [&](int L) {
                const char *b = buf + layer_bytes * L;
                if (U == 2) hipLaunchKernelGGL(read_kernel<2>, dim3(c.wgs), dim3(256), 0, st, b, c.stride, c.chunk, sink);
                else if (U == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(c.wgs), dim3(256), 0, st, b, c.stride, c.chunk, sink);
                else hipLaunchKernelGGL(read_kernel<8>, dim3(c.wgs), dim3(256), 0, st, b, c.stride, c.chunk, sink);
            };
            for (int it = 0; it < 16; it++) launch(it % NL);
            CK(hipEventRecord(e0, st));
            const int iters = 96;
            for (int it = 0; it < iters; it++) launch(it % NL);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)c.wgs * c.chunk;
            printf("%-52s U=%d  %7.2f us  %6.0f GB/s\n", c.name, U, ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e9);
        }
    }
    // the same KV-shaped read when the bytes were touched just before (same buffer every launch: served by the 256 MB memory-side cache)
    // and when 1 / 2 other layers' worth of bytes (402 / 804 MB) went through in between
    for (int gap : {0, 1, 2}) {
        auto launch = [&](int L) {
            hipLaunchKernelGGL(read_kernel<4>, dim3(2048), dim3(256), 0, st, buf + layer_bytes * L, (size_t)128 * 1536, T * 1536, sink);
        };
        const int iters = 64;
        float tot = 0.f;
        for (int it = 0; it < iters + 4; it++) {
            for (int g2 = 1; g2 <= gap; g2++)          // evict: stream other layers' bytes (whole 402 MB blocks)
                hipLaunchKernelGGL(read_kernel<8>, dim3(4096), dim3(256), 0, st, buf + layer_bytes * g2, layer_bytes / 4096, (int)(layer_bytes / 4096), sink);
            launch(0);                                  // "prefetch"
            CK(hipEventRecord(e0, st));
            launch(0);                                  // the read that would be the attention sweep
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 4) tot += ms;
        }
        printf("KV shape re-read right after a touch, %d x 402 MB streamed before      %7.2f us  %6.0f GB/s\n", gap, tot * 1e3 / iters,
               (double)2048 * T * 1536 / (tot * 1e-3 / iters) / 1e9);
    }
    // one large contiguous read for reference: 1.6 GB
    {
        const size_t big = layer_bytes * 4;
        for (int it = 0; it < 3; it++) hipLaunchKernelGGL(read_kernel<8>, dim3(16384), dim3(256), 0, st, buf, big / 16384, (int)(big / 16384), sink);
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < 10; it++) hipLaunchKernelGGL(read_kernel<8>, dim3(16384), dim3(256), 0, st, buf + (it & 1) * big, big / 16384, (int)(big / 16384), sink);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s       %7.2f us  %6.0f GB/s\n", "bulk contiguous 1.6 GB read", ms * 1e3 / 10, (double)big / (ms * 1e-3 / 10) / 1e9);
    }
    return 0;
}
