// gemm_bf16.h — bf16 MFMA GEMM for gfx950 with fused epilogues (device templates + launcher).
//
//   C[M,N] = A[M,K] . W[N,K]^T + bias[N]        A, W bf16 (K contiguous), fp32 accumulate
//
// Tiling (CDNA4, 64-wide waves): 256 threads = 4 waves in a 2x2 arrangement over a BM x BN
// output tile, BK = 64.  Each wave owns (BM/2) x (BN/2) outputs as 16x16 MFMA fragments
// (v_mfma_f32_16x16x32_bf16, 2 k-substeps per BK).  Operands are swapped — the WEIGHT fragment
// is the MFMA "A" operand and the activation fragment the "B" operand — so every lane ends up
// with 4 CONSECUTIVE output columns of one row (C/D map: row=(lane>>4)*4+r, col=lane&15): bias,
// activation, residual add and the store are then 8/16-byte vector operations per lane.
//
// HBM -> registers -> LDS staging (16 B per lane, full 128 B lines per 8 lanes), LDS double
// buffered with one barrier per K-step, 16-byte chunks XOR-swizzled by (row & 7) so the
// ds_read_b128 fragment reads are at most 2-way bank conflicted (guide §6 G4 / T2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace lmrl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

enum GemmEpi {
    EPI_BF16 = 0,        // C bf16 = acc + bias
    EPI_GELU_BF16 = 1,   // C bf16 = gelu_new(acc + bias)
    EPI_RESID_F32 = 2,   // C f32 += acc + bias   (residual stream, in place)
    EPI_F32 = 3,         // C f32 = acc + bias
    EPI_RELU_BF16 = 4,   // C bf16 = relu(acc + bias)
};

__device__ __forceinline__ uint16_t f32_to_bf16_rn(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

__device__ __forceinline__ float gelu_new(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  — GPT-2 "gelu_new"
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.f + tanhf(u));
}

struct GemmArgs {
    const uint16_t *A;   // [M][lda] bf16
    const uint16_t *W;   // [N][K] bf16 (N a multiple of BN, zero padded)
    const float *bias;   // [N] or null
    void *C;             // [M][ldc] bf16 or f32 by epilogue
    int M, N, K, lda, ldc;
    int n_store;         // columns >= n_store are not stored (logical N, <= N)
};

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int BK = 64;
    constexpr int FM = BM / 32;   // 16-row activation fragments per wave
    constexpr int FN = BN / 32;   // 16-row weight fragments per wave
    constexpr int A_CH = BM * 8 / 256;   // 16-byte chunks per thread per tile
    constexpr int W_CH = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [2][BM*128] A then [2][BN*128] W
    char *sA = smem;
    char *sW = smem + 2 * BM * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // staging coordinates: chunk id = tid + i*256 -> row = id>>3, c = id&7
    u32x4 ra[A_CH], rw[W_CH];
    const int nk = g.K / BK;

// staging helpers are macros (not lambdas) so the chunk registers stay in VGPRs instead of scratch
#define LMRL_GEMM_LOAD_TILES(KT)                                                                                   \
    do {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < A_CH; i_++) {                                                      \
            const int id_ = tid + i_ * 256, row_ = id_ >> 3, c_ = id_ & 7;                                         \
            int m_ = m0 + row_;                                                                                    \
            m_ = m_ < g.M ? m_ : g.M - 1;                                                                          \
            ra[i_] = *reinterpret_cast<const u32x4 *>(g.A + (size_t)m_ * g.lda + (size_t)(KT) * BK + c_ * 8);      \
        }                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < W_CH; i_++) {                                                      \
            const int id_ = tid + i_ * 256, row_ = id_ >> 3, c_ = id_ & 7;                                         \
            rw[i_] = *reinterpret_cast<const u32x4 *>(g.W + (size_t)(n0 + row_) * g.K + (size_t)(KT) * BK + c_ * 8); \
        }                                                                                                          \
    } while (0)
#define LMRL_GEMM_STORE_TILES(BUF)                                                                                 \
    do {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < A_CH; i_++) {                                                      \
            const int id_ = tid + i_ * 256, row_ = id_ >> 3, c_ = id_ & 7;                                         \
            *reinterpret_cast<u32x4 *>(sA + (BUF) * BM * 128 + row_ * 128 + ((c_ ^ (row_ & 7)) << 4)) = ra[i_];    \
        }                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < W_CH; i_++) {                                                      \
            const int id_ = tid + i_ * 256, row_ = id_ >> 3, c_ = id_ & 7;                                         \
            *reinterpret_cast<u32x4 *>(sW + (BUF) * BN * 128 + row_ * 128 + ((c_ ^ (row_ & 7)) << 4)) = rw[i_];    \
        }                                                                                                          \
    } while (0)

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; i++)
#pragma unroll
        for (int j = 0; j < FM; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    LMRL_GEMM_LOAD_TILES(0);
    LMRL_GEMM_STORE_TILES(0);
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) LMRL_GEMM_LOAD_TILES(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 fw[FN], fa[FM];
            const int c = kk * 4 + lq;
#pragma unroll
            for (int i = 0; i < FN; i++) {
                const int row = wn * (BN / 2) + i * 16 + lr;
                fw[i] = *reinterpret_cast<const bf16x8 *>(sW + buf * BN * 128 + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FM; j++) {
                const int row = wm * (BM / 2) + j * 16 + lr;
                fa[j] = *reinterpret_cast<const bf16x8 *>(sA + buf * BM * 128 + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FN; i++)
#pragma unroll
                for (int j = 0; j < FM; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) LMRL_GEMM_STORE_TILES(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds C[m][n..n+3], n = n0 + wn*BN/2 + i*16 + lq*4, m = m0 + wm*BM/2 + j*16 + lr
#pragma unroll
    for (int i = 0; i < FN; i++) {
        const int n = n0 + wn * (BN / 2) + i * 16 + lq * 4;
        if (n >= g.n_store) continue;
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g.bias) b4 = *reinterpret_cast<const f32x4 *>(g.bias + n);
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int m = m0 + wm * (BM / 2) + j * 16 + lr;
            if (m >= g.M) continue;
            f32x4 v = acc[i][j] + b4;
            if (EPI == EPI_GELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gelu_new(v[r]);
            }
            if (EPI == EPI_RELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            }
            if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16) {
                uint2 o;
                o.x = (uint32_t)f32_to_bf16_rn(v[0]) | ((uint32_t)f32_to_bf16_rn(v[1]) << 16);
                o.y = (uint32_t)f32_to_bf16_rn(v[2]) | ((uint32_t)f32_to_bf16_rn(v[3]) << 16);
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(g.C) + (size_t)m * g.ldc + n) = o;
            } else if (EPI == EPI_RESID_F32) {
                f32x4 *p = reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n);
                *p = *p + v;
            } else {
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n) = v;
            }
        }
    }
}

template <int BM, int BN, int EPI>
inline hipError_t gemm_launch_cfg(const GemmArgs &g, hipStream_t s) {
    const size_t shmem = 2 * (BM + BN) * 128;
    const int tiles = ((g.M + BM - 1) / BM) * (g.N / BN);
    static bool attr_set = false;
    if (!attr_set && shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_bf16_kernel<BM, BN, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    {
        ProfScope ps(BM == 128 ? PROF_GEMM_128x128 : (BN == 128 ? PROF_GEMM_64x128 : PROF_GEMM_64x64), s,
                     2.0 * (double)g.M * (double)g.N * (double)g.K);
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, EPI>), dim3(tiles), dim3(256), shmem, s, g);
    }
    return hipGetLastError();
}

// Tile choice: keep >= ~1 workgroup per CU (256 CUs) when the problem allows it.
template <int EPI>
inline hipError_t gemm_launch(const GemmArgs &g, hipStream_t s) {
    const long t128 = (long)((g.M + 127) / 128) * (g.N / 128);
    if (g.N % 128 == 0 && t128 >= 192) return gemm_launch_cfg<128, 128, EPI>(g, s);
    if (g.N % 128 == 0 && (long)((g.M + 63) / 64) * (g.N / 128) >= 192) return gemm_launch_cfg<64, 128, EPI>(g, s);
    return gemm_launch_cfg<64, 64, EPI>(g, s);
}

}  // namespace lmrl
