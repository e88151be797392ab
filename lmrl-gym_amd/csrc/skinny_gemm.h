// skinny_gemm.h — the decode layer's Dense products for SMALL lock-step batches (M <= 16 rows: configs[0]'s 8-env Maze run, any policy driven with a
// handful of envs).
//
// At M = 8 the tile kernels of gemm_bf16.h / gemm8_bf16.h are pure latency chains: a 64 x 64 tile walks 12 (K = 768) or 48 (K = 3072) dependent
// K-steps of ~0.33-0.6 us each for microseconds of MFMA work — 7-8 us and 15.7 us per launch (profiles/r06_maze_b8_kernel_stats.csv), i.e. 0.5 ms
// per generated token.  Here the M rows are ONE MFMA row block (v_mfma_f32_16x16x32_bf16, rows >= M are clamped copies that are never stored) and
// the parallelism comes from N and K instead:
//   * a 1024-thread workgroup = 16 waves = 2 column blocks of 16 x 8 slices of K; grid = N / 32 workgroups (24 ... 96 for GPT-2-small);
//   * a wave loads its whole K-slice of both operands straight into MFMA-layout registers (16 B per lane and step: a W row segment of 64 B per
//     4 lanes, an activation row segment likewise) — EVERY load is issued before the first MFMA, so a launch pays one memory latency, not one per
//     K-step; 3 (K = 768) or 12 (K = 3072) steps per wave;
//   * the 8 slices of a column block meet in LDS (16 KB) and are added in slice order by the block's first wave (fixed order: bit-reproducible),
//     which applies the epilogue.
// Epilogues: the four LayerNorm-folded forms the decode layer uses (gemm_bf16.h: EPI_BF16_LN, EPI_GELU_BF16_LN, EPI_RESID_F32_STATS, EPI_RESID_F32),
// same formulas, same (mu, rstd) association order (ln_row_moments); a stats slot covers 32 columns = this workgroup's two column blocks.
// Not bit-identical to the tile kernels (K is summed in 8 slices): a per-SESSION variant (LMRL_FWD_SKINNY), never mixed within a session.
#pragma once
#include "gemm_bf16.h"

namespace lmrl {

constexpr int kSkNB = 2, kSkKS = 8, kSkThreads = kSkNB * kSkKS * 64;

template <int EPI, int STEPS>   // STEPS = K / (8 slices * 32): 32-wide MFMA steps per wave
__global__ __launch_bounds__(kSkThreads) void skinny_gemm_kernel(GemmArgs g) {
    constexpr bool LN_IN = (EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN);
    constexpr bool RESID = (EPI == EPI_RESID_F32_STATS || EPI == EPI_RESID_F32);
    __shared__ f32x4 part[kSkNB][kSkKS][64];
    __shared__ float2 slot_part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = wave & (kSkNB - 1), ks = wave / kSkNB;
    const int lr = lane & 15, lq = lane >> 4;
    const int Mr = g.m_dev ? min(*g.m_dev, g.M) : g.M;
    if (Mr <= 0) return;
    const int n0 = (int)blockIdx.x * (16 * kSkNB) + nb * 16;
    const int ldw = g.ldw > 0 ? g.ldw : g.K;
    const int kbase = ks * (STEPS * 32) + lq * 8;
    const int mrow = lr < Mr ? lr : Mr - 1;
    const uint16_t *ap = g.A + (size_t)mrow * g.lda + kbase;
    const uint16_t *wp = g.W + (size_t)(n0 + lr) * ldw + kbase;
    bf16x8 fa[STEPS], fw[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) {       // the whole K-slice of both operands: every load in flight before the first MFMA
        fw[s] = *reinterpret_cast<const bf16x8 *>(wp + s * 32);
        fa[s] = *reinterpret_cast<const bf16x8 *>(ap + s * 32);
    }
    // epilogue operands of the reducing waves, requested behind the operand loads
    float mu = 0.f, rs = 1.f;
    f32x4 xres = f32x4{0.f, 0.f, 0.f, 0.f}, b4 = f32x4{0.f, 0.f, 0.f, 0.f}, c4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n = n0 + lq * 4;              // this lane's 4 output columns (rows = lr)
    if (ks == 0) {
        if (g.bias) b4 = *reinterpret_cast<const f32x4 *>(g.bias + n);
        if (LN_IN) {
            c4 = *reinterpret_cast<const f32x4 *>(g.colsum + n);
            // (mu, rstd) of row lr from the producer's slots, in ln_row_moments' association order: four quarters of nslots / 8 slot pairs each
            const int h4 = g.nslots / 2, per = h4 / 4;
            const f32x4 *sp = reinterpret_cast<const f32x4 *>(g.stats + (size_t)mrow * g.nslots);
            float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < 4; q++)
                for (int k = 0; k < per; k++) { const f32x4 p = sp[q * per + k]; a[q] += p[0] + p[2]; b[q] += p[1] + p[3]; }
            const float s1 = (a[0] + a[1]) + (a[2] + a[3]), s2 = (b[0] + b[1]) + (b[2] + b[3]);
            mu = s1 * g.inv_d;
            rs = rsqrtf(fmaxf(s2 * g.inv_d - mu * mu, 0.f) + g.eps);
        }
        if (RESID) {
            const float *src = (EPI == EPI_RESID_F32 && g.resid) ? g.resid + (size_t)mrow * g.ldr : reinterpret_cast<const float *>(g.C) + (size_t)mrow * g.ldc;
            xres = *reinterpret_cast<const f32x4 *>(src + (n < g.n_store ? n : 0));
        }
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s], fa[s], acc, 0, 0, 0);
    part[nb][ks][lane] = acc;
    __syncthreads();
    if (EPI != EPI_RESID_F32_STATS && ks != 0) return;
    f32x4 sum = part[nb][0][lane];
#pragma unroll
    for (int k = 1; k < kSkKS; k++) sum = sum + part[nb][k][lane];      // slice order: one association order per output
    const bool ok = lr < Mr && n < g.n_store;
    if (EPI == EPI_RESID_F32_STATS) {
        // x += acc + bias ; xb = bf16(x) ; the row's slot of this 32-column group gets (sum x, sum x^2): block 0's 16 columns + block 1's, in that order
        float s1 = 0.f, s2 = 0.f;
        if (ks == 0) {
            const f32x4 v = xres + (sum + b4);
            s1 = ok ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
            s2 = ok ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (ok) {
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)lr * g.ldc + n) = v;
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(g.xb + (size_t)lr * g.ldc + n) = o;
            }
            if (nb == 1 && lq == 0) slot_part[lr] = make_float2(s1, s2);
        }
        __syncthreads();
        if (ks == 0 && nb == 0 && lq == 0 && lr < Mr) {
            const float2 o = slot_part[lr];
            g.stats[(size_t)lr * g.nslots + blockIdx.x] = make_float2(s1 + o.x, s2 + o.y);       // slot = 32-column group = this workgroup
        }
        return;
    }
    if (!ok) return;
    f32x4 v = LN_IN ? (sum - c4 * mu) * rs + b4 : sum + b4;
    if (EPI == EPI_GELU_BF16_LN) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = gelu_new(v[r]);
    }
    if (LN_IN) {
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(g.C) + (size_t)lr * g.ldc + n) = o;
    } else {      // EPI_RESID_F32
        *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)lr * g.ldc + n) = xres + v;
    }
}

inline bool skinny_ok(const GemmArgs &g) {
    const int steps = g.K / (kSkKS * 32);
    return g.M >= 1 && g.M <= 16 && g.N % (16 * kSkNB) == 0 && g.K % (kSkKS * 32) == 0 && (steps == 3 || steps == 4 || steps == 5 || steps == 12) &&
           g.lda % 8 == 0 && (g.ldw == 0 || g.ldw % 8 == 0);
}

template <int EPI>
inline hipError_t skinny_launch(const GemmArgs &g, hipStream_t s) {
    static_assert(EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN || EPI == EPI_RESID_F32_STATS || EPI == EPI_RESID_F32, "decode-layer epilogues only");
    const dim3 grid(g.N / (16 * kSkNB)), block(kSkThreads);
    switch (g.K / (kSkKS * 32)) {
        case 3: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 3>), grid, block, 0, s, g); break;
        case 4: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 4>), grid, block, 0, s, g); break;
        case 5: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 5>), grid, block, 0, s, g); break;
        case 12: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 12>), grid, block, 0, s, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lmrl
