// tools/kv_layout_bench.hip — does the KV-cache layout matter for the decode-attention sweep?  Pure reads with the attention kernel's
// wave -> (env, head) mapping and load shape (8 rows x 128 B per wave instruction, U = 4 instructions per batch):
//   A  token-major (the product layout): row t of env b = 1536 B, head h at +128 h        -> a workgroup (4 heads) reads 512 B pieces at a 1536 B stride
//   B  head-group-major: [env][3 groups][Tmax][4 heads x 128 B]                              -> a workgroup reads ONE contiguous L x 512 B block
//   C  head-major: [env][12 heads][Tmax][128 B]                                              -> a wave reads one contiguous L x 128 B block
// Build: hipcc -O3 --offload-arch=gfx950 tools/kv_layout_bench.hip -o tools/_bin/kv_layout_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT>
__global__ __launch_bounds__(256) void sweep(const char *k, const char *v, int B, int Tmax, int L, uint32_t *sink) {
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave_id >= B * 12) return;
    const int b = wave_id / 12, h = wave_id - b * 12, rr = lane >> 3, cc = lane & 7;
    u32x4 acc = {0, 0, 0, 0};
    for (int t0 = 0; t0 < L; t0 += 32) {
        u32x4 kr[4], vr[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int t = t0 + u * 8 + rr;
            t = t < L ? t : L - 1;
            size_t off;
            if (LAYOUT == 0) off = ((size_t)b * Tmax + t) * 1536 + h * 128 + cc * 16;
            else if (LAYOUT == 1) off = (((size_t)b * 3 + h / 4) * Tmax + t) * 512 + (h & 3) * 128 + cc * 16;
            else off = (((size_t)b * 12 + h) * Tmax + t) * 128 + cc * 16;
            kr[u] = *reinterpret_cast<const u32x4 *>(k + off);
            vr[u] = *reinterpret_cast<const u32x4 *>(v + off);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc ^= kr[u] ^ vr[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

int main() {
    const int B = 1024, Tmax = 128, NL = 8;
    const size_t layer = (size_t)B * Tmax * 1536;          // K (or V) of one layer
    char *buf; uint32_t *sink;
    CK(hipMalloc(&buf, layer * 2 * NL)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, layer * 2 * NL));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int L : {37, 64, 96}) {
        for (int layout = 0; layout < 3; layout++) {
            auto launch = [&](int l) {
                const char *k = buf + layer * 2 * l, *v = k + layer;
                if (layout == 0) hipLaunchKernelGGL(sweep<0>, dim3(B * 3), dim3(256), 0, st, k, v, B, Tmax, L, sink);
                else if (layout == 1) hipLaunchKernelGGL(sweep<1>, dim3(B * 3), dim3(256), 0, st, k, v, B, Tmax, L, sink);
                else hipLaunchKernelGGL(sweep<2>, dim3(B * 3), dim3(256), 0, st, k, v, B, Tmax, L, sink);
            };
            for (int it = 0; it < 16; it++) launch(it % NL);
            CK(hipEventRecord(e0, st));
            const int iters = 96;
            for (int it = 0; it < iters; it++) launch(it % NL);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)B * L * 1536 * 2;
            printf("L=%3d layout %c  %7.2f us  %6.0f GB/s\n", L, "ABC"[layout], ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e9);
        }
    }
    // occupancy sensitivity of layout A at L = 37: dynamic LDS caps the resident workgroups (= waves per SIMD: 4-wave workgroups)
    for (int wgs : {4, 5, 6, 7, 8, 10, 12}) {
        const size_t lds = (size_t)160 * 1024 / wgs / 1024 * 1024 - 1024;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        auto launch = [&](int l) {
            const char *k = buf + layer * 2 * l, *v = k + layer;
            hipLaunchKernelGGL(sweep<0>, dim3(B * 3), dim3(256), lds, st, k, v, B, Tmax, 37, sink);
        };
        for (int it = 0; it < 16; it++) launch(it % NL);
        CK(hipEventRecord(e0, st));
        const int iters = 96;
        for (int it = 0; it < iters; it++) launch(it % NL);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("layout A, L=37, <= %2d waves per SIMD  %7.2f us  %6.0f GB/s\n", wgs, ms * 1e3 / iters, (double)B * 37 * 1536 * 2 / (ms * 1e-3 / iters) / 1e9);
    }
    return 0;
}
