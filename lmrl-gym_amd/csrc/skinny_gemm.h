// skinny_gemm.h — the decode layer's Dense products for SMALL lock-step batches (M <= 16 rows: configs[0]'s 8-env Maze run, any policy driven with a
// handful of envs).
//
// At M = 8 the tile kernels of gemm_bf16.h / gemm8_bf16.h are pure latency chains: a 64 x 64 tile walks 12 (K = 768) or 48 (K = 3072) dependent
// K-steps of ~0.33-0.6 us each for microseconds of MFMA work — 7-8 us and 15.7 us per launch (profiles/r06_maze_b8_kernel_stats.csv), i.e. 0.5 ms
// per generated token.  Here the M rows are ONE MFMA row block (v_mfma_f32_16x16x32_bf16, rows >= M are clamped copies that are never stored) and
// the parallelism comes from N and K instead:
//   * a workgroup = one column block of 16 x 8 slices of K = 8 waves; grid = N / 16 workgroups (48 ... 192 for GPT-2-small) — as many CUs as possible
//     pull: a CU ingests ~55 GB/s, and the first version (two column blocks per workgroup, 24 workgroups for N = 768) left fc2 at 12 us;
//   * a wave loads its whole K-slice of both operands straight into MFMA-layout registers (16 B per lane and step: a W row segment of 64 B per
//     4 lanes, an activation row segment likewise) — EVERY load is issued before the first MFMA, so a launch pays one memory latency, not one per
//     K-step; 3 (K = 768) or 12 (K = 3072) steps per wave;
//   * the 8 slices of a column block meet in LDS (16 KB) and are added in slice order by the block's first wave (fixed order: bit-reproducible),
//     which applies the epilogue.
// Epilogues: the four LayerNorm-folded forms the decode layer uses (gemm_bf16.h: EPI_BF16_LN, EPI_GELU_BF16_LN, EPI_RESID_F32_STATS, EPI_RESID_F32),
// same formulas — except where the row moments come from: a stats slot of the tile kernels covers 32 columns (two of these workgroups), so the
// skinny residual producers write NO slots and the skinny consumers take (mu, rstd) from the fp32 residual rows themselves (GemmArgs::resid: every
// wave adds its K-slice's columns of the <= 16 rows behind its operand loads, the slices meet in LDS with the accumulators).
// Not bit-identical to the tile kernels (K is summed in 8 slices): a per-SESSION variant (LMRL_FWD_SKINNY), never mixed within a session.
#pragma once
#include "gemm_bf16.h"

namespace lmrl {

constexpr int kSkKS = 8;

// NB: column blocks per workgroup (1: as many pulling CUs as possible; 2 kept for the A/B)
template <int EPI, int STEPS, int kSkNB>   // STEPS = K / (8 slices * 32): 32-wide MFMA steps per wave
__global__ __launch_bounds__(kSkNB * kSkKS * 64) void skinny_gemm_kernel(GemmArgs g) {
    constexpr bool LN_IN = (EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN);
    constexpr bool RESID = (EPI == EPI_RESID_F32_STATS || EPI == EPI_RESID_F32);
    __shared__ f32x4 part[kSkNB][kSkKS][64];
    __shared__ float2 mom_part[kSkKS][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = wave % kSkNB, ks = wave / kSkNB;
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
    // LN consumers: this wave's share of the rows' moments — row lr, the 8 columns [kbase + 32 s, + 8) of every step of its K-slice (K = d_model)
    float ms1 = 0.f, ms2 = 0.f;
    if (LN_IN) {
        const float *xp = g.resid + (size_t)mrow * g.ldr + kbase;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xp + s * 32), x1 = *reinterpret_cast<const f32x4 *>(xp + s * 32 + 4);
            ms1 += ((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3]));
            ms2 += ((x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3])) + ((x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3]));
        }
        ms1 += __shfl_xor(ms1, 16); ms2 += __shfl_xor(ms2, 16);
        ms1 += __shfl_xor(ms1, 32); ms2 += __shfl_xor(ms2, 32);      // row lr over this wave's K-slice
    }
    // epilogue operands of the reducing waves, requested behind the operand loads
    float mu = 0.f, rs = 1.f;
    f32x4 xres = f32x4{0.f, 0.f, 0.f, 0.f}, b4 = f32x4{0.f, 0.f, 0.f, 0.f}, c4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n = n0 + lq * 4;              // this lane's 4 output columns (rows = lr)
    if (ks == 0) {
        if (g.bias) b4 = *reinterpret_cast<const f32x4 *>(g.bias + n);
        if (LN_IN) c4 = *reinterpret_cast<const f32x4 *>(g.colsum + n);
        if (RESID) {
            const float *src = (EPI == EPI_RESID_F32 && g.resid) ? g.resid + (size_t)mrow * g.ldr : reinterpret_cast<const float *>(g.C) + (size_t)mrow * g.ldc;
            xres = *reinterpret_cast<const f32x4 *>(src + (n < g.n_store ? n : 0));
        }
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s], fa[s], acc, 0, 0, 0);
    part[nb][ks][lane] = acc;
    if (LN_IN && lq == 0) mom_part[ks][lr] = make_float2(ms1, ms2);
    __syncthreads();
    if (ks != 0) return;
    f32x4 sum = part[nb][0][lane];
#pragma unroll
    for (int k = 1; k < kSkKS; k++) sum = sum + part[nb][k][lane];      // slice order: one association order per output
    if (LN_IN) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < kSkKS; k++) { const float2 p = mom_part[k][lr]; s1 += p.x; s2 += p.y; }
        mu = s1 * g.inv_d;
        rs = rsqrtf(fmaxf(s2 * g.inv_d - mu * mu, 0.f) + g.eps);
    }
    const bool ok = lr < Mr && n < g.n_store;
    if (EPI == EPI_RESID_F32_STATS) {
        // x += acc + bias ; xb = bf16(x) (the next product's operand).  No stats slot: see the header
        if (ok) {
            const f32x4 v = xres + (sum + b4);
            *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)lr * g.ldc + n) = v;
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2 *>(g.xb + (size_t)lr * g.ldc + n) = o;
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
    return g.M >= 1 && g.M <= 16 && g.N % 16 == 0 && g.K % (kSkKS * 32) == 0 && (steps == 3 || steps == 4 || steps == 5 || steps == 12 || steps == 16 || steps == 20) &&
           g.lda % 8 == 0 && (g.ldw == 0 || g.ldw % 8 == 0);
}

template <int EPI>
inline hipError_t skinny_launch(const GemmArgs &g, hipStream_t s) {
    static_assert(EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN || EPI == EPI_RESID_F32_STATS || EPI == EPI_RESID_F32, "decode-layer epilogues only");
    constexpr int NB = 1;
    const dim3 grid(g.N / (16 * NB)), block(NB * kSkKS * 64);
    switch (g.K / (kSkKS * 32)) {
        case 3: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 3, NB>), grid, block, 0, s, g); break;
        case 4: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 4, NB>), grid, block, 0, s, g); break;
        case 5: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 5, NB>), grid, block, 0, s, g); break;
        case 12: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 12, NB>), grid, block, 0, s, g); break;
        case 16: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 16, NB>), grid, block, 0, s, g); break;      // (8 waves per workgroup: up to 256 VGPRs per lane)
        case 20: hipLaunchKernelGGL((skinny_gemm_kernel<EPI, 20, NB>), grid, block, 0, s, g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lmrl
