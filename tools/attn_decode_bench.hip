// Stand-alone micro-benchmark for the decode-attention access pattern (C = 1, token-major bf16 KV cache).
// Build:  hipcc -O3 --offload-arch=gfx950 tools/attn_decode_bench.hip -o gpurun_out/attn_decode_bench
// Variants are selected by number; every variant computes the same result (checked against variant 0).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}

// ---- variant 1: pure streaming read of the same bytes (upper bound for this access pattern)
__global__ __launch_bounds__(256) void stream_kernel(const uint16_t *kcache, const uint16_t *vcache, const int32_t *len, uint16_t *out,
                                                     int B, int H, int Tmax, int d) {
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave_id >= B * H) return;
    const int b = wave_id / H, h = wave_id - b * H;
    const int T = len[b] + 1;
    const int rr = lane >> 3, cc = lane & 7;
    const uint16_t *kc = kcache + (size_t)b * Tmax * d + (size_t)h * 64, *vc = vcache + (size_t)b * Tmax * d + (size_t)h * 64;
    u32x4 acc = {0, 0, 0, 0};
    for (int t = rr; t < T; t += 8) {
        acc ^= *reinterpret_cast<const u32x4 *>(kc + (size_t)t * d + cc * 8);
        acc ^= *reinterpret_cast<const u32x4 *>(vc + (size_t)t * d + cc * 8);
    }
    for (int o = 8; o < 64; o <<= 1) {
        acc.x ^= __shfl_xor(acc.x, o); acc.y ^= __shfl_xor(acc.y, o); acc.z ^= __shfl_xor(acc.z, o); acc.w ^= __shfl_xor(acc.w, o);
    }
    if (rr == 0) *reinterpret_cast<u32x4 *>(out + (size_t)b * d + (size_t)h * 64 + cc * 8) = acc;
}

// ---- generic decode attention: WPB waves per block, U key blocks in flight, PRED = predicate instead of clamping
template <int WPB, int U, bool PRED>
__global__ __launch_bounds__(WPB * 64) void attn_kernel(const uint16_t *__restrict__ qkv, uint16_t *__restrict__ kcache,
                                                        uint16_t *__restrict__ vcache, const int32_t *__restrict__ len,
                                                        uint16_t *__restrict__ out, int B, int H, int Tmax, int d) {
    const int wave_id = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave_id >= B * H) return;
    const int b = wave_id / H, h = wave_id - b * H;
    const int L0 = len[b];
    const int T = L0 + 1;
    const int rr = lane >> 3, cc = lane & 7;
    const size_t ld = (size_t)3 * d;
    uint16_t *kc = kcache + (size_t)b * Tmax * d + (size_t)h * 64;
    uint16_t *vc = vcache + (size_t)b * Tmax * d + (size_t)h * 64;
    const uint16_t *qbase = qkv + (size_t)b * ld + (size_t)h * 64;
    if (rr == 0 && L0 < Tmax) {
        *reinterpret_cast<u32x4 *>(kc + (size_t)L0 * d + cc * 8) = *reinterpret_cast<const u32x4 *>(qbase + d + cc * 8);
        *reinterpret_cast<u32x4 *>(vc + (size_t)L0 * d + cc * 8) = *reinterpret_cast<const u32x4 *>(qbase + 2 * d + cc * 8);
    }
    float q[8];
    {
        const u32x4 qv = *reinterpret_cast<const u32x4 *>(qbase + cc * 8);
#pragma unroll
        for (int k = 0; k < 4; k++) { q[2 * k] = bf2f((uint16_t)(qv[k] & 0xffff)) * 0.125f; q[2 * k + 1] = bf2f((uint16_t)(qv[k] >> 16)) * 0.125f; }
    }
    float m = -1e30f, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = 0.f;
    for (int t0 = 0; t0 < T; t0 += 8 * U) {
        u32x4 kraw[U], vraw[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * 8 + rr;
            if (PRED) {
                kraw[u] = u32x4{0, 0, 0, 0}; vraw[u] = u32x4{0, 0, 0, 0};
                if (t < T) {
                    const uint16_t *kp = t < L0 ? kc + (size_t)t * d + cc * 8 : qbase + d + cc * 8;
                    const uint16_t *vp = t < L0 ? vc + (size_t)t * d + cc * 8 : qbase + 2 * d + cc * 8;
                    kraw[u] = *reinterpret_cast<const u32x4 *>(kp);
                    vraw[u] = *reinterpret_cast<const u32x4 *>(vp);
                }
            } else {
                const int tc = t < T ? t : T - 1;
                const uint16_t *kp = tc < L0 ? kc + (size_t)tc * d + cc * 8 : qbase + d + cc * 8;
                const uint16_t *vp = tc < L0 ? vc + (size_t)tc * d + cc * 8 : qbase + 2 * d + cc * 8;
                kraw[u] = *reinterpret_cast<const u32x4 *>(kp);
                vraw[u] = *reinterpret_cast<const u32x4 *>(vp);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * 8 + rr;
            const bool ok = t < T;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                s = fmaf(q[2 * k], bf2f((uint16_t)(kraw[u][k] & 0xffff)), s);
                s = fmaf(q[2 * k + 1], bf2f((uint16_t)(kraw[u][k] >> 16)), s);
            }
            s += dpp<0xB1>(s); s += dpp<0x4E>(s); s += dpp<0x141>(s);
            const float m_new = ok ? fmaxf(m, s) : m;
            const float alpha = __expf(m - m_new);
            const float p = ok ? __expf(s - m_new) : 0.f;
            l = l * alpha + p;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                o[2 * k] = fmaf(p, bf2f((uint16_t)(vraw[u][k] & 0xffff)), o[2 * k] * alpha);
                o[2 * k + 1] = fmaf(p, bf2f((uint16_t)(vraw[u][k] >> 16)), o[2 * k + 1] * alpha);
            }
            m = m_new;
        }
    }
    float mm = m;
    mm = fmaxf(mm, __shfl_xor(mm, 8)); mm = fmaxf(mm, __shfl_xor(mm, 16)); mm = fmaxf(mm, __shfl_xor(mm, 32));
    const float w = __expf(m - mm);
    float lt = l * w;
    lt += __shfl_xor(lt, 8); lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
    const float inv = 1.f / lt;
    float r8[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float a = o[e] * w;
        a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        r8[e] = a * inv;
    }
    if (rr == 0) {
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; k++) pk[k] = (uint32_t)f2bf(r8[2 * k]) | ((uint32_t)f2bf(r8[2 * k + 1]) << 16);
        *reinterpret_cast<u32x4 *>(out + (size_t)b * d + (size_t)h * 64 + cc * 8) = pk;
    }
}

// ---- two-pass variant: scores for ALL keys first (one pass, all K loads in flight), single softmax, then V pass.
// Lane layout as above; keys beyond 8*U*... handled by the loop.  Softmax without online rescaling: per-lane running max is
// replaced by a wave-wide max after the K pass (scores kept in registers: up to U_MAX blocks).
template <int WPB, int NB>   // NB = max key blocks (8 keys each) held in registers: T <= 8*NB
__global__ __launch_bounds__(WPB * 64) void attn2p_kernel(const uint16_t *__restrict__ qkv, uint16_t *__restrict__ kcache,
                                                          uint16_t *__restrict__ vcache, const int32_t *__restrict__ len,
                                                          uint16_t *__restrict__ out, int B, int H, int Tmax, int d) {
    const int wave_id = blockIdx.x * WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave_id >= B * H) return;
    const int b = wave_id / H, h = wave_id - b * H;
    const int L0 = len[b];
    const int T = L0 + 1;
    const int rr = lane >> 3, cc = lane & 7;
    const size_t ld = (size_t)3 * d;
    uint16_t *kc = kcache + (size_t)b * Tmax * d + (size_t)h * 64;
    uint16_t *vc = vcache + (size_t)b * Tmax * d + (size_t)h * 64;
    const uint16_t *qbase = qkv + (size_t)b * ld + (size_t)h * 64;
    if (rr == 0 && L0 < Tmax) {
        *reinterpret_cast<u32x4 *>(kc + (size_t)L0 * d + cc * 8) = *reinterpret_cast<const u32x4 *>(qbase + d + cc * 8);
        *reinterpret_cast<u32x4 *>(vc + (size_t)L0 * d + cc * 8) = *reinterpret_cast<const u32x4 *>(qbase + 2 * d + cc * 8);
    }
    u32x4 kraw[NB], vraw[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int t = u * 8 + rr;
        kraw[u] = u32x4{0, 0, 0, 0}; vraw[u] = u32x4{0, 0, 0, 0};
        if (u * 8 < T) {   // wave-uniform guard, lane clamp inside
            const int tc = t < T ? t : T - 1;
            const uint16_t *kp = tc < L0 ? kc + (size_t)tc * d + cc * 8 : qbase + d + cc * 8;
            const uint16_t *vp = tc < L0 ? vc + (size_t)tc * d + cc * 8 : qbase + 2 * d + cc * 8;
            kraw[u] = *reinterpret_cast<const u32x4 *>(kp);
            vraw[u] = *reinterpret_cast<const u32x4 *>(vp);
        }
    }
    float q[8];
    {
        const u32x4 qv = *reinterpret_cast<const u32x4 *>(qbase + cc * 8);
#pragma unroll
        for (int k = 0; k < 4; k++) { q[2 * k] = bf2f((uint16_t)(qv[k] & 0xffff)) * 0.125f; q[2 * k + 1] = bf2f((uint16_t)(qv[k] >> 16)) * 0.125f; }
    }
    float s[NB];
    float m = -1e30f;
#pragma unroll
    for (int u = 0; u < NB; u++) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            a = fmaf(q[2 * k], bf2f((uint16_t)(kraw[u][k] & 0xffff)), a);
            a = fmaf(q[2 * k + 1], bf2f((uint16_t)(kraw[u][k] >> 16)), a);
        }
        a += dpp<0xB1>(a); a += dpp<0x4E>(a); a += dpp<0x141>(a);
        s[u] = (u * 8 + rr < T) ? a : -1e30f;
        m = fmaxf(m, s[u]);
    }
    m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = 0.f;
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const float p = (u * 8 + rr < T) ? __expf(s[u] - m) : 0.f;
        l += p;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[2 * k] = fmaf(p, bf2f((uint16_t)(vraw[u][k] & 0xffff)), o[2 * k]);
            o[2 * k + 1] = fmaf(p, bf2f((uint16_t)(vraw[u][k] >> 16)), o[2 * k + 1]);
        }
    }
    l += __shfl_xor(l, 8); l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    float r8[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float a = o[e];
        a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        r8[e] = a * inv;
    }
    if (rr == 0) {
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; k++) pk[k] = (uint32_t)f2bf(r8[2 * k]) | ((uint32_t)f2bf(r8[2 * k + 1]) << 16);
        *reinterpret_cast<u32x4 *>(out + (size_t)b * d + (size_t)h * 64 + cc * 8) = pk;
    }
}

int main(int argc, char **argv) {
    const int B = 1024, H = 12, d = 768, Tmax = 96, NL = 12;
    const size_t per = (size_t)B * Tmax * d;
    uint16_t *kc, *vc, *qkv, *out, *ref;
    int32_t *len;
    CK(hipMalloc(&kc, per * NL * 2)); CK(hipMalloc(&vc, per * NL * 2));
    CK(hipMalloc(&qkv, (size_t)B * 3 * d * 2)); CK(hipMalloc(&out, (size_t)B * d * 2)); CK(hipMalloc(&ref, (size_t)B * d * 2));
    CK(hipMalloc(&len, B * 4));
    {
        std::vector<uint16_t> h(per);
        uint32_t s = 12345;
        for (int l = 0; l < 2; l++) {
            for (size_t i = 0; i < per; i++) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3c00 + ((s >> 16) & 0x3ff) + ((s >> 30) << 15)); }
            for (int L = 0; L < NL; L++) CK(hipMemcpy((l ? vc : kc) + per * L, h.data(), per * 2, hipMemcpyHostToDevice));
        }
        std::vector<uint16_t> hq((size_t)B * 3 * d);
        for (auto &x : hq) { s = s * 1664525u + 1013904223u; x = (uint16_t)(0x3c00 + ((s >> 16) & 0x3ff) + ((s >> 30) << 15)); }
        CK(hipMemcpy(qkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int dist = 0; dist < 3; dist++) {
        std::vector<int32_t> hl(B);
        uint32_t s = 777;
        double tot = 0;
        for (auto &x : hl) {
            s = s * 1664525u + 1013904223u;
            x = dist == 0 ? 20 + (int)((s >> 16) % 31) : dist == 1 ? 35 : 60 + (int)((s >> 16) % 12);
            tot += x + 1;
        }
        CK(hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice));
        const double bytes = tot * H * 256.0 + (double)B * d * 2 * 4;   // K+V rows + q + out + appended rows
        printf("len dist %d: mean T %.1f, %.1f MB per launch\n", dist, tot / B, bytes / 1e6);
        for (int v = 0; v < 12; v++) {
            auto launch = [&](int L) {
                uint16_t *k = kc + per * L, *vv = vc + per * L;
                const int nw = B * H;
                switch (v) {
                case 0: hipLaunchKernelGGL((attn_kernel<4, 4, false>), dim3(nw / 4), dim3(256), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 1: hipLaunchKernelGGL(stream_kernel, dim3(nw / 4), dim3(256), 0, st, k, vv, len, out, B, H, Tmax, d); break;
                case 2: hipLaunchKernelGGL((attn_kernel<4, 4, true>), dim3(nw / 4), dim3(256), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 3: hipLaunchKernelGGL((attn_kernel<4, 8, true>), dim3(nw / 4), dim3(256), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 4: hipLaunchKernelGGL((attn_kernel<12, 4, true>), dim3(nw / 12), dim3(768), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 5: hipLaunchKernelGGL((attn_kernel<2, 4, true>), dim3(nw / 2), dim3(128), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 6: hipLaunchKernelGGL((attn_kernel<1, 4, true>), dim3(nw), dim3(64), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 7: hipLaunchKernelGGL((attn2p_kernel<4, 12>), dim3(nw / 4), dim3(256), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 8: hipLaunchKernelGGL((attn2p_kernel<12, 12>), dim3(nw / 12), dim3(768), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 9: hipLaunchKernelGGL((attn2p_kernel<1, 12>), dim3(nw), dim3(64), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 10: hipLaunchKernelGGL((attn_kernel<4, 2, true>), dim3(nw / 4), dim3(256), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                case 11: hipLaunchKernelGGL((attn_kernel<6, 4, true>), dim3(nw / 6), dim3(384), 0, st, qkv, k, vv, len, out, B, H, Tmax, d); break;
                }
            };
            launch(0);
            CK(hipStreamSynchronize(st));
            if (v == 0) CK(hipMemcpy(ref, out, (size_t)B * d * 2, hipMemcpyDeviceToDevice));
            int bad = 0;
            if (v != 1) {
                std::vector<uint16_t> a((size_t)B * d), r((size_t)B * d);
                CK(hipMemcpy(a.data(), out, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), ref, r.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < a.size(); i++) { int df = (int)a[i] - (int)r[i]; if (df < -1 || df > 1) bad++; }
            }
            for (int it = 0; it < 24; it++) launch(it % NL);
            CK(hipEventRecord(e0, st));
            const int iters = 120;
            for (int it = 0; it < iters; it++) launch(it % NL);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  v%-2d %7.2f us  %6.0f GB/s  mismatches %d\n", v, ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e9, bad);
        }
    }
    return 0;
}
