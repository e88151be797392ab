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

#ifdef LMRL_G8_PROBE   // tools/gemm8_bench.hip only: per-workgroup s_memtime stamps (entry, stage 0 landed, K loop done, epilogue done)
__device__ unsigned long long *g8_probe = nullptr;
#define LMRL_G8_STAMP(I) do { if (g8_probe && threadIdx.x == 0) g8_probe[(size_t)blockIdx.x * 4 + (I)] = __builtin_readcyclecounter(); } while (0)
#else
#define LMRL_G8_STAMP(I) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

enum GemmEpi {
    EPI_BF16 = 0,        // C bf16 = acc + bias
    EPI_GELU_BF16 = 1,   // C bf16 = gelu_new(acc + bias)
    EPI_RESID_F32 = 2,   // C f32 += acc + bias   (residual stream, in place)
    EPI_F32 = 3,         // C f32 = acc + bias
    EPI_RELU_BF16 = 4,   // C bf16 = relu(acc + bias)
    // ---- LayerNorm folded into the neighbouring GEMMs (no stand-alone LN launch between them):
    // producer: the residual GEMM also writes a bf16 copy of the updated row and per-(row, 32-column group) partial
    //           sums (sum x, sum x^2) into fixed slots (deterministic: one writer per slot, fixed read order);
    // consumer: runs on the RAW bf16 residual copy with gamma pre-multiplied into W (W' = W . diag(gamma)) and applies
    //           LN(x) W^T = rstd * (x W'^T - mu * colsum(W')) + (bias + W beta) in the epilogue.
    EPI_RESID_F32_STATS = 5,
    EPI_BF16_LN = 6,
    EPI_GELU_BF16_LN = 7,
    EPI_BF16_LN_KV = 8,  // EPI_BF16_LN + the K / V columns of every row also appended to the KV cache (GemmArgs::kv_*)
    // ---- train step, bf16-matmul mode (gemm8_bf16.h tiles only): epilogues that write the NEXT kernel's bf16 operand themselves
    EPI_F32_GELU_BF16 = 9,   // C f32 = acc + bias (the pre-activation the backward reads) AND xb bf16 [M][ldxb] = gelu_new(C)
    EPI_BF16_HEADS = 10,     // the c_attn product straight into the flash kernels' per-head matrices: column n = which * d + h * 64 + e of row
                             // m = b * T + t -> C + which * hd_plane + ((b * H + h) * Tp + t) * 64 + e, bf16, q columns (which = 0) times 1/8
    EPI_GELU_BWD_BF16 = 11,  // C bf16 = acc * gelu_new'(resid[m][n]): d(pre-activation) as the bf16 operand of the c_fc backward products
    EPI_BF16_CE = 12,        // vocabulary heads: C bf16 = acc + bias, AND from the registers: per (row, wave-column-slab) partial
                             // (max, sum exp) of the logits AS STORED (bf16-rounded) -> stats[m][slot] (log-sum-exp without a pass over the
                             // logits, consistent with the backward's softmax on the stored logits) and the fp32 logit of column
                             // ce_targets[m] -> ce_tgt_logit[m] (Q(s, a) / the token's logit: take_along_axis)
    // ---- fp32-accurate rollout mode "bf16x3" (gpt2_f32_engine.py): the c_fc product writes the NEXT product's three-term split operand itself
    EPI_GELU_SPLIT3 = 13,    // C bf16 [M][ldc >= 3 N] = [hi | lo | hi](gelu_new(acc + bias)), hi = bf16(y) RNE, lo = bf16(y - hi): lmrl_split3_bf16's
                             // layout — no fp32 [M][d_ff] pre-activation written, re-read by a gelu + split pass and written again
};

// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32); the integer bit trick it replaces
// cost ~6 VALU per value, which made the bf16 epilogues VALU-bound (profiles/r02_gemm8_bench.txt)
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const hw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16_rn(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// the split operand of 4 consecutive columns n .. n + 3 of row `row16` (pitch 3 N: [hi | lo | hi]) — EPI_GELU_SPLIT3
__device__ __forceinline__ void store_split3_x4(uint16_t *row16, int N, int n, f32x4 v) {
    const uint32_t h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
    const uint32_t l0 = pack_bf16x2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
    const uint32_t l1 = pack_bf16x2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
    *reinterpret_cast<uint2 *>(row16 + n) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(row16 + N + n) = make_uint2(l0, l1);
    *reinterpret_cast<uint2 *>(row16 + 2 * N + n) = make_uint2(h0, h1);
}

__device__ __forceinline__ float gelu_new(float x) {
    // GPT-2 "gelu_new": 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)  ==  x * sigmoid(2u) = x / (1 + e^(-2u)).
    // 4 plain VALU + v_exp_f32 + v_rcp_f32: the fc-GEMM epilogue applies it 25 M times per prefill launch.
    const float t = x * fmaf(0.044715f * x, x, 1.0f);                                  // x + 0.044715 x^3
    const float e = __builtin_amdgcn_exp2f(t * (-2.0f * 0.7978845608028654f * 1.4426950408889634f));   // e^(-2u)
    return x * __builtin_amdgcn_rcpf(1.f + e);
}

// d gelu_new(x) / dx with the same sigmoid form: g = x s, s = sigmoid(2u), u = c (x + a x^3)  ->  g' = s + x s (1 - s) 2c (1 + 3a x^2)
__device__ __forceinline__ float gelu_new_grad(float x) {
    const float c2 = 2.0f * 0.7978845608028654f;
    const float t = x * fmaf(0.044715f * x, x, 1.0f);
    const float e = __builtin_amdgcn_exp2f(t * (-c2 * 1.4426950408889634f));
    const float s = __builtin_amdgcn_rcpf(1.f + e);
    return fmaf(x * s * (1.f - s), c2 * fmaf(3.f * 0.044715f * x, x, 1.0f), s);
}

// Per-row LayerNorm moments from the producer's (sum x, sum x^2) slots: ONE association order everywhere — four partial sums of nslots/8
// consecutive float4 (= 2 slots each), combined as (p0 + p1) + (p2 + p3) — so every kernel / tile shape derives bit-identical (mu, rstd).
// TPR threads cooperate on a row (1, 2 or 4 consecutive lanes); `part` holds the row's nslots/2 (sum, sum^2) float2 partials.
template <int TPR>
__device__ __forceinline__ float2 ln_row_moments(const float2 *part_row, int h4, int q, float inv_d, float eps) {
    const int per = h4 / 4;                     // float2 entries per quarter
    float s1 = 0.f, s2 = 0.f;
    if (TPR == 4) {
        for (int k = 0; k < per; k++) { const float2 p = part_row[q * per + k]; s1 += p.x; s2 += p.y; }
        s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
        s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
    } else if (TPR == 2) {
        float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
        for (int k = 0; k < per; k++) { const float2 p = part_row[(2 * q) * per + k]; a1 += p.x; a2 += p.y; }
        for (int k = 0; k < per; k++) { const float2 p = part_row[(2 * q + 1) * per + k]; b1 += p.x; b2 += p.y; }
        s1 = a1 + b1; s2 = a2 + b2;
        s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
    } else {
        float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
        for (int qq = 0; qq < 4; qq++)
            for (int k = 0; k < per; k++) { const float2 p = part_row[qq * per + k]; a[qq] += p.x; b[qq] += p.y; }
        s1 = (a[0] + a[1]) + (a[2] + a[3]); s2 = (b[0] + b[1]) + (b[2] + b[3]);
    }
    const float mu = s1 * inv_d;
    return make_float2(mu, rsqrtf(fmaxf(s2 * inv_d - mu * mu, 0.f) + eps));
}

struct GemmArgs {
    const uint16_t *A;   // [M][lda] bf16
    const uint16_t *W;   // [N][K] bf16 (N a multiple of BN, zero padded)
    const float *bias;   // [N] or null
    void *C;             // [M][ldc] bf16 or f32 by epilogue
    int M, N, K, lda, ldc;
    int n_store;         // columns >= n_store are not stored (logical N, <= N)
    // LayerNorm fusion operands (EPI_RESID_F32_STATS / EPI_*_LN only)
    float2 *stats;        // [M][nslots] (sum x, sum x^2) partials of the residual row
    uint16_t *xb;         // producer: bf16 copy of the updated residual stream, [M][ldc]
    const float *colsum;  // consumer: c[n] = sum_k W'[n][k]
    int nslots;           // consumer: slots to add per row ; producer: row pitch of `stats`
    float inv_d, eps;     // consumer: 1/d_model, LN epsilon
    const int *m_dev;     // optional DEVICE row count (<= M): ragged batches — tiles beyond it exit, `M` then only sizes the grid
    int ldw;              // row pitch of W in elements (0: dense, = K).  Operands whose natural pitch is a large power of two (the
                          // transposed train-step operands, K = B*T) are padded by the caller: every row of a tile would otherwise
                          // start in the same HBM channel
    // EPI_BF16_LN_KV only (the decode qkv GEMM under LMRL_FWD_KV_FROM_GEMM): also append the new token's K / V row to the KV cache — row m belongs to env b = kv_rowmap ?
    // kv_rowmap[m] >> 5 : m; columns [kv_d, 2 kv_d) go to kv_k + (b * kv_tmax + kv_len[b]) * kv_d, columns [2 kv_d, 3 kv_d) to kv_v + the same
    // offset (envs with kv_cnt[b] <= 0 or a full cache are skipped) — so that the decode attention kernel does no stores into the stream it reads
    uint16_t *kv_k, *kv_v;
    const int *kv_len, *kv_cnt, *kv_rowmap;
    int kv_tmax, kv_d;
    // EPI_RESID_F32 only: the residual operand when it is NOT the output buffer — C = resid + acc + bias (null: in place, C += acc + bias).
    // The train step keeps a block's input, middle and output residual streams as separate tensors (its LayerNorm backward reads them).
    const float *resid;
    int ldr;
    const int *ce_targets;    // EPI_BF16_CE: [M] column whose fp32 logit goes to ce_tgt_logit (or null)
    float *ce_tgt_logit;      // EPI_BF16_CE: [M]
    int ldxb;                 // EPI_F32_GELU_BF16: row pitch of xb (elements)
    int hd_T, hd_Tp, hd_H;    // EPI_BF16_HEADS: tokens per sequence, its padding to 64, heads
    long hd_plane;            // EPI_BF16_HEADS: elements between the q, k and v matrices
    int pre_bf16;             // EPI_F32_GELU_BF16 / EPI_GELU_BWD_BF16: the pre-activation (C / resid) is stored as bf16 with row pitch ldc / ldr ELEMENTS
};

// cache row (b * tmax + len[b]) the K/V columns of GEMM row m are appended to, or -1.  Loads are unconditional on clamped indices (selects, no
// divergent branches around them).
__device__ __forceinline__ long kv_append_row(const GemmArgs &g, int m, int Mr) {
    const int mc = m < Mr ? m : Mr - 1;
    const int b = g.kv_rowmap ? (g.kv_rowmap[mc] >> 5) : mc;
    const int len = g.kv_len[b];
    const int cnt = g.kv_cnt ? g.kv_cnt[b] : 1;
    const bool ok = (m < Mr) & (cnt > 0) & (len < g.kv_tmax);
    return ok ? (long)b * g.kv_tmax + len : -1l;
}

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
            rw[i_] = *reinterpret_cast<const u32x4 *>(g.W + (size_t)(n0 + row_) * (g.ldw > 0 ? g.ldw : g.K) + (size_t)(KT) * BK + c_ * 8); \
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
            if (EPI == EPI_GELU_BF16 || EPI == EPI_GELU_SPLIT3) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gelu_new(v[r]);
            }
            if (EPI == EPI_RELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            }
            if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(g.C) + (size_t)m * g.ldc + n) = o;
            } else if (EPI == EPI_GELU_SPLIT3) {
                store_split3_x4(reinterpret_cast<uint16_t *>(g.C) + (size_t)m * g.ldc, g.N, n, v);
            } else if (EPI == EPI_RESID_F32) {
                f32x4 *p = reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n);
                const f32x4 r = g.resid ? *reinterpret_cast<const f32x4 *>(g.resid + (size_t)m * g.ldr + n) : *p;
                *p = r + v;
            } else {
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 main loop: HBM -> LDS directly with global_load_lds (16 B per lane, no staging VGPRs, no ds_write pass), an
// S-stage LDS ring and COUNTED vmcnt waits so that S-2 whole K-tiles stay in flight across the (raw) barrier.
// LDS image per stage is lane-linear per wave instruction (1 KiB = 8 rows x 128 B), so the XOR swizzle is applied
// to the per-lane SOURCE address: LDS slot s of row r holds global chunk s ^ (r & 7); fragment reads use the same
// involution (guide §5.4 rule 21).  All LDS lives in the single dynamic array (guide §5 "three .s-level traps").
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XCD-aware workgroup -> tile mapping.  MI355X dispatches workgroup id to XCD (id % 8) and each XCD has a private 4 MiB
// L2, so with a naive id -> (tile_m, tile_n) map every XCD pulls ALL of A and W through its own L2 (8x fabric traffic,
// which dominates the ~10 us decode GEMMs).  Here the 8 XCDs form a gm x gn grid: XCD (i, j) owns the i-th group of
// m-tiles and the j-th group of n-tiles, walking its tiles n-panel by n-panel.  Fabric traffic becomes A*gn + W*gm;
// the launcher picks gm to minimise it.  Placement is observed behaviour, not a contract: a different placement only
// changes speed.  Workgroups whose tile falls outside the problem exit immediately.
struct XcdMap { int tiles_m, tiles_n, gm, tmg, tng; };

inline XcdMap make_xcd_map(int tiles_m, int tiles_n, double a_bytes, double w_bytes) {
    XcdMap x; x.tiles_m = tiles_m; x.tiles_n = tiles_n;
    int best = 1; double best_cost = 1e300;
    for (int gm = 1; gm <= 8; gm *= 2) {
        if (gm > tiles_m && gm > 1) break;
        const double cost = a_bytes * (8 / gm) + w_bytes * gm;
        if (cost < best_cost) { best_cost = cost; best = gm; }
    }
    x.gm = best;
    x.tmg = (tiles_m + best - 1) / best;
    x.tng = (tiles_n + (8 / best) - 1) / (8 / best);
    return x;
}
inline int xcd_grid(const XcdMap &x) { return 8 * x.tmg * x.tng; }

__device__ __forceinline__ bool xcd_tile(const XcdMap &x, int id, int &tile_m, int &tile_n) {
    const int xcd = id & 7, q = id >> 3;
    const int gn = 8 / x.gm;
    const int gi = xcd / gn, gj = xcd - gi * gn;
    tile_m = (q % x.tmg) * x.gm + gi;      // m-tiles dealt round-robin to the gm groups: a ragged (short) M thins every group equally
    tile_n = gj * x.tng + (q / x.tmg);
    return tile_m < x.tiles_m && tile_n < x.tiles_n;
}

// Shared main loop: acc[i][j] += W-fragment i (rows n0 + wn*BN/2 + 16 i ..) x A-fragment j (rows m0 + wm*BM/2 + 16 j ..).
// Starts with a barrier so that it can be called repeatedly on the same LDS ring (fused multi-operand kernels).
struct NoExtraLoads { __device__ __forceinline__ void operator()() const {} };

// EXTRA / extra(): `EXTRA` additional per-lane global loads that the caller issues through `extra()` right AFTER the ring
// prologue; vmcnt retires loads in order, so the wait for the first K-tile allows them to stay in flight and their
// latency hides behind the K loop (they are complete by the second wait; with nk == 1 the caller's own use waits).
template <int BM, int BN, int STAGES, int EXTRA = 0, class Extra = NoExtraLoads>
__device__ __forceinline__ void glds_mainloop(const uint16_t *__restrict__ A, int lda, const uint16_t *__restrict__ W, int ldw, int K, int M,
                                              int m0, int n0, char *smem, f32x4 (&acc)[BN / 32][BM / 32], Extra extra = Extra()) {
    constexpr int BK = 64;
    constexpr int FM = BM / 32, FN = BN / 32;
    constexpr int LA = BM / 32, LW = BN / 32;          // glds instructions per wave per stage (1 KiB segments / 4 waves)
    constexpr int L = LA + LW;
    constexpr int STAGE = (BM + BN) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = K / BK;
    const int lrow = lane >> 3;                 // row within the 8-row segment == (row & 7)
    const int src_c = (lane & 7) ^ lrow;        // source chunk for LDS slot (lane & 7)

    const uint16_t *ap[LA];
    const uint16_t *wp[LW];
#pragma unroll
    for (int i = 0; i < LA; i++) {
        int m = m0 + (wave + 4 * i) * 8 + lrow;
        m = m < M ? m : M - 1;
        ap[i] = A + (size_t)m * lda + src_c * 8;
    }
#pragma unroll
    for (int i = 0; i < LW; i++) wp[i] = W + (size_t)(n0 + (wave + 4 * i) * 8 + lrow) * ldw + src_c * 8;

#define LMRL_GLDS_ISSUE(KT, SLOT)                                                                                     \
    do {                                                                                                              \
        char *sb_ = smem + (SLOT) * STAGE;                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < LA; i_++)                                                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ap[i_] + (size_t)(KT) * BK), \
                                             (__attribute__((address_space(3))) void *)(sb_ + (wave + 4 * i_) * 1024), 16, 0, 0); \
        _Pragma("unroll") for (int i_ = 0; i_ < LW; i_++)                                                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wp[i_] + (size_t)(KT) * BK), \
                                             (__attribute__((address_space(3))) void *)(sb_ + BM * 128 + (wave + 4 * i_) * 1024), 16, 0, 0); \
    } while (0)

    __builtin_amdgcn_s_barrier();   // every wave is done reading the ring from a previous call
#pragma unroll
    for (int s = 0; s < STAGES - 1; s++)
        if (s < nk) LMRL_GLDS_ISSUE(s, s);
    {
        // compiler barriers: the hook's loads must be issued after the prologue and before the first counted wait — the
        // wait for K-tile 0 is vmcnt(... + EXTRA), which is only correct for exactly this issue order.  A hook that drains
        // everything itself (EXTRA == 0, ends with vmcnt(0): the LayerNorm moments) may use the ring slot STAGES-1, which
        // the prologue leaves free until the loop's first barrier.
        asm volatile("" ::: "memory");
        extra();
        asm volatile("" ::: "memory");
    }

    const int lr = lane & 15, lq = lane >> 4;
    int slot = 0;
    for (int kt = 0; kt < nk; kt++) {
        // stage kt must have landed; up to STAGES-2 later stages may stay in flight (plus, before the first K-tile, the
        // caller's EXTRA loads, which were issued after every prologue stage)
        const int ahead = nk - 1 - kt;
        const int inflight = ahead < STAGES - 2 ? ahead : STAGES - 2;
        if (EXTRA > 0 && kt == 0) {
            if (inflight >= 4 && STAGES >= 6) wait_vmcnt<4 * L + EXTRA>();
            else if (inflight == 3 && STAGES >= 5) wait_vmcnt<3 * L + EXTRA>();
            else if (inflight == 2 && STAGES >= 4) wait_vmcnt<2 * L + EXTRA>();
            else if (inflight == 1 && STAGES >= 3) wait_vmcnt<L + EXTRA>();
            else wait_vmcnt<EXTRA>();
        } else if (inflight >= 4 && STAGES >= 6) wait_vmcnt<4 * L>();
        else if (inflight == 3 && STAGES >= 5) wait_vmcnt<3 * L>();
        else if (inflight == 2 && STAGES >= 4) wait_vmcnt<2 * L>();
        else if (inflight == 1 && STAGES >= 3) wait_vmcnt<L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt == 0) LMRL_G8_STAMP(1);
        if (kt + STAGES - 1 < nk) {
            int nslot = slot + STAGES - 1;
            nslot = nslot >= STAGES ? nslot - STAGES : nslot;
            LMRL_GLDS_ISSUE(kt + STAGES - 1, nslot);
        }
        const char *sA = smem + slot * STAGE;
        const char *sW = sA + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 fw[FN], fa[FM];
            const int c = kk * 4 + lq;
#pragma unroll
            for (int i = 0; i < FN; i++) {
                const int row = wn * (BN / 2) + i * 16 + lr;
                fw[i] = *reinterpret_cast<const bf16x8 *>(sW + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FM; j++) {
                const int row = wm * (BM / 2) + j * 16 + lr;
                fa[j] = *reinterpret_cast<const bf16x8 *>(sA + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FN; i++)
#pragma unroll
                for (int j = 0; j < FM; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
#undef LMRL_GLDS_ISSUE
}

template <int BM, int BN, int STAGES, int EPI, int NQ = 0>
__global__ __launch_bounds__(256) void gemm_bf16_glds_kernel(GemmArgs g, XcdMap xm) {
    constexpr int FM = BM / 32, FN = BN / 32;
    constexpr bool LN_IN = (EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN || EPI == EPI_BF16_LN_KV);
    constexpr bool RESID = (EPI == EPI_RESID_F32_STATS || EPI == EPI_RESID_F32);
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    if (!xcd_tile(xm, blockIdx.x, tile_m, tile_n)) return;   // workgroup-uniform
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int lr = lane & 15, lq = lane >> 4;
    const int Mr = g.m_dev ? *g.m_dev : g.M;          // rows actually present (workgroup-uniform scalar load)
    if (m0 >= Mr) return;
    LMRL_G8_STAMP(0);
    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; i++)
#pragma unroll
        for (int j = 0; j < FM; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float ln_mu[FM], ln_rs[FM];          // LN_IN: this lane's rows' (mu, rstd)
    long kvrow[FM];                      // EPI_BF16_LN_KV: cache rows this lane's K / V columns are appended to (looked up in the prologue)
    f32x4 xres[FN][FM];                  // RESID: this lane's slice of the residual stream, prefetched under the K loop
    if (LN_IN) {
        // The tile's BM x nslots (sum, sum^2) slots are one contiguous region of `stats`.  Right after the ring prologue is in flight the
        // 256 threads fetch it coalesced, wait for everything, reduce it through the ring slot the prologue left free ((mu, rstd) per row,
        // fixed association order: ln_row_moments) and keep their own rows' moments in registers: the K loop itself then carries no
        // extra registers or waits, and nothing is left to do between the loop and the epilogue.
        constexpr int NL = (NQ * BM) / 64;           // float4 loads per thread: BM*nslots*8 B / (256*16 B)
        constexpr int TPR = 256 / BM >= 4 ? 4 : (256 / BM >= 2 ? 2 : 1);
        static_assert(BM * 4 * NQ * 8 + BM * 8 <= STAGE, "LN scratch must fit one ring slot");
        auto head = [&]() {
            if constexpr (EPI == EPI_BF16_LN_KV) {
                if (g.kv_k) {
#pragma unroll
                    for (int j = 0; j < FM; j++) kvrow[j] = kv_append_row(g, m0 + wm * (BM / 2) + j * 16 + lr, Mr);
                }
            }
            const int h4 = g.nslots / 2;                         // float4 per row
            const f32x4 *sp = reinterpret_cast<const f32x4 *>(g.stats + (size_t)m0 * g.nslots);
            const int lim = (Mr - m0 < BM ? Mr - m0 : BM) * h4;
            f32x4 st[NL > 0 ? NL : 1];
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int idx = (int)threadIdx.x + 256 * k;
                st[k] = sp[idx < lim ? idx : lim - 1];
            }
            wait_vmcnt<0>();
            float2 *part = reinterpret_cast<float2 *>(smem + (STAGES - 1) * STAGE);     // [BM * h4] partial (sum, sum^2)
            float2 *murs = part + BM * 4 * NQ;                                          // [BM] (mu, rstd)
#pragma unroll
            for (int k = 0; k < NL; k++) part[threadIdx.x + 256 * k] = make_float2(st[k][0] + st[k][2], st[k][1] + st[k][3]);   // rows >= M: unused
            __syncthreads();
            {
                const int row = (int)threadIdx.x / TPR, q = (int)threadIdx.x % TPR;
                if (row < BM) {
                    const float2 mr = ln_row_moments<TPR>(part + row * h4, h4, q, g.inv_d, g.eps);
                    if (q == 0) murs[row] = mr;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < FM; j++) {
                const float2 p2 = murs[wm * (BM / 2) + j * 16 + lr];
                ln_mu[j] = p2.x; ln_rs[j] = p2.y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is handed to the ring at the loop's first barrier
        };
        glds_mainloop<BM, BN, STAGES, 0>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, head);
    } else if (RESID) {
        // read-modify-write epilogue: fetch this lane's residual values NOW (FN*FM float4 loads behind the ring prologue, in-order
        // vmcnt: the counted waits of the loop leave them in flight) so that the epilogue does not start with a dependent HBM round trip
        auto prefetch = [&]() {
#pragma unroll
            for (int i = 0; i < FN; i++)
#pragma unroll
                for (int j = 0; j < FM; j++) {
                    int m = m0 + wm * (BM / 2) + j * 16 + lr;
                    m = m < Mr ? m : Mr - 1;
                    int n = n0 + wn * (BN / 2) + i * 16 + lq * 4;
                    n = n < g.n_store ? n : 0;
                    xres[i][j] = (EPI == EPI_RESID_F32 && g.resid) ? *reinterpret_cast<const f32x4 *>(g.resid + (size_t)m * g.ldr + n)
                                                                   : *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(g.C) + (size_t)m * g.ldc + n);
                }
        };
        glds_mainloop<BM, BN, STAGES, FN * FM>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, prefetch);
    } else {
        glds_mainloop<BM, BN, STAGES>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc);
    }
    LMRL_G8_STAMP(2);

    if (EPI == EPI_RESID_F32_STATS) {
        // x += acc + bias ; xb = bf16(x) ; slot (tile_n, wn) of the row gets (sum x, sum x^2) over this wave's BN/2 columns
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int m = m0 + wm * (BM / 2) + j * 16 + lr;
            const bool row_ok = m < Mr;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < FN; i++) {
                const int n = n0 + wn * (BN / 2) + i * 16 + lq * 4;
                if (n >= g.n_store || !row_ok) continue;
                f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (g.bias) b4 = *reinterpret_cast<const f32x4 *>(g.bias + n);
                const f32x4 v = xres[i][j] + (acc[i][j] + b4);
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n) = v;
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(g.xb + (size_t)m * g.ldc + n) = o;
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (lq == 0 && row_ok) g.stats[(size_t)m * g.nslots + tile_n * 2 + wn] = make_float2(s1, s2);
        }
        LMRL_G8_STAMP(3);
        return;
    }

#pragma unroll
    for (int i = 0; i < FN; i++) {
        const int n = n0 + wn * (BN / 2) + i * 16 + lq * 4;
        if (n >= g.n_store) continue;
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g.bias) b4 = *reinterpret_cast<const f32x4 *>(g.bias + n);
        f32x4 c4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (LN_IN) c4 = *reinterpret_cast<const f32x4 *>(g.colsum + n);
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int rl = wm * (BM / 2) + j * 16 + lr;
            const int m = m0 + rl;
            if (m >= Mr) continue;
            f32x4 v;
            if (LN_IN) {
                v = (acc[i][j] - c4 * ln_mu[j]) * ln_rs[j] + b4;
            } else {
                v = acc[i][j] + b4;
            }
            if (EPI == EPI_GELU_BF16 || EPI == EPI_GELU_BF16_LN || EPI == EPI_GELU_SPLIT3) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gelu_new(v[r]);
            }
            if (EPI == EPI_RELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
            }
            if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16 || LN_IN) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(g.C) + (size_t)m * g.ldc + n) = o;
                if constexpr (EPI == EPI_BF16_LN_KV) {
                    if (g.kv_k && n >= g.kv_d && kvrow[j] >= 0) {
                        const bool isv = n >= 2 * g.kv_d;
                        uint16_t *dst = (isv ? g.kv_v : g.kv_k) + kvrow[j] * g.kv_d + (n - (isv ? 2 : 1) * g.kv_d);
                        *reinterpret_cast<uint2 *>(dst) = o;
                    }
                }
            } else if (EPI == EPI_GELU_SPLIT3) {
                store_split3_x4(reinterpret_cast<uint16_t *>(g.C) + (size_t)m * g.ldc, g.N, n, v);
            } else if (EPI == EPI_RESID_F32) {
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n) = xres[i][j] + v;
            } else {
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n) = v;
            }
        }
    }
#ifdef LMRL_G8_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LMRL_G8_STAMP(3);
#endif
}

extern int g_gemm_variant;   // test/bench hook: 0 = auto (v2), 1 = v1 register-staged kernels

template <int BM, int BN, int STAGES, int EPI, int NQ = 0>
inline hipError_t gemm_launch_glds(const GemmArgs &g, hipStream_t s) {
    const size_t shmem = (size_t)STAGES * (BM + BN) * 128;
    const XcdMap xm = make_xcd_map((g.M + BM - 1) / BM, g.N / BN, 2.0 * g.M * g.K, 2.0 * g.N * g.K);
    const int tiles = xcd_grid(xm);
    static bool attr_set = false;
    if (!attr_set && shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_bf16_glds_kernel<BM, BN, STAGES, EPI, NQ>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    {
        ProfScope ps(BM == 128 && BN == 128 ? PROF_GEMM_128x128 : (BM * BN == 128 * 64 ? PROF_GEMM_64x128 : PROF_GEMM_64x64), s,
                     2.0 * (double)g.M * (double)g.N * (double)g.K);
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<BM, BN, STAGES, EPI, NQ>), dim3(tiles), dim3(256), shmem, s, g, xm);
    }
    return hipGetLastError();
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

}  // namespace lmrl
