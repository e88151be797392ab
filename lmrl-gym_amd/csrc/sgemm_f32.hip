// sgemm_f32.hip — exact-fp32 batched GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32) for the TRAIN path.
//
// The reference trains in float32 (`jnp.float32` params / activations in LLM_RL/algorithms/{ppo,ilql}/gpt2/interface.py
// and the heads), and BASELINE.json asks for ILQL train-step outputs within 1e-4 relative: the forward/backward GEMMs of
// the train step therefore run on the f32-input MFMA (bit-for-bit an fmaf chain, 157 TFLOP/s peak) instead of bf16.
//
//   C[b] = alpha * op(A[b]) . op(B[b]) + beta * C[b] (+ bias[n])      row-major, arbitrary M, N, K, two-level batches
//   op(A) is M x K:  transA = 0 -> A[m*lda + k]   transA = 1 -> A[k*lda + m]
//   op(B) is K x N:  transB = 0 -> B[k*ldb + n]   transB = 1 -> B[n*ldb + k]
//
// The f32 MFMA takes ONE float per lane per operand (A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]), so both
// LDS tiles are stored k-major and every transpose combination is just a different global->LDS copy; fragment reads are
// conflict-free ds_read_b32.  64 x 64 output tile, 4 waves (one 32x32 MFMA tile each), BK = 16, register prefetch of the
// next K-step.  At 64 cycles per MFMA the matrix pipe, not LDS, is the limiter.
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4v;

struct SgemmArgs {
    const float *A, *B;
    float *C;
    const float *bias;
    int M, N, K, lda, ldb, ldc;
    long sAo, sAi, sBo, sBi, sCo, sCi;   // batch strides (outer, inner) in elements
    int nb_inner;
    float alpha, beta;
    int k_total;   // split-K launches: the whole K (slice z = blockIdx.z covers [z * K, min(k_total, (z + 1) * K)); 0: K is the extent
};

constexpr int SG_BM = 64, SG_BN = 64, SG_BK = 16, SG_LD = 68;   // LDS row stride (floats): 68 % 32 = 4 -> spread banks

// Load a [BK x 64] k-major tile of op(X) for K-step k0 into registers (4 floats per thread per operand)
// thread t handles 4 consecutive elements along the CONTIGUOUS global dimension.
template <bool TRANS_K_CONTIG>   // true: memory is [row64][k] (k contiguous) ; false: memory is [k][row64] (row contiguous)
__device__ __forceinline__ void sg_load(const float *__restrict__ X, int ld, int row0, int nrows, int k0, int K, int t, float (&r)[4]) {
    if (TRANS_K_CONTIG) {
        const int row = row0 + (t >> 2), k = k0 + (t & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = (row < nrows && k + i < K) ? X[(size_t)row * ld + k + i] : 0.f;
    } else {
        const int k = k0 + (t >> 4), row = row0 + (t & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = (k < K && row + i < nrows) ? X[(size_t)k * ld + row + i] : 0.f;
    }
}
template <bool TRANS_K_CONTIG>
__device__ __forceinline__ void sg_store(float *__restrict__ S, int t, const float (&r)[4]) {
    if (TRANS_K_CONTIG) {
        const int row = t >> 2, k = (t & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) S[(k + i) * SG_LD + row] = r[i];
    } else {
        const int k = t >> 4, row = (t & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) S[k * SG_LD + row + i] = r[i];
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_f32_kernel(SgemmArgs g) {
    __shared__ float sA[2][SG_BK * SG_LD];
    __shared__ float sB[2][SG_BK * SG_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bo = blockIdx.z / g.nb_inner, bi = blockIdx.z - bo * g.nb_inner;
    const float *A = g.A + bo * g.sAo + bi * g.sAi;
    const float *B = g.B + bo * g.sBo + bi * g.sBi;
    float *C = g.C + bo * g.sCo + bi * g.sCi;
    const int m0 = blockIdx.y * SG_BM, n0 = blockIdx.x * SG_BN;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    float ra[4], rb[4];
    // op(A): rows = m. transA=0 -> memory [m][k] (k contiguous) ; transA=1 -> memory [k][m]
    sg_load<!TA>(A, g.lda, m0, g.M, 0, g.K, t, ra);
    // op(B): "rows" = n. transB=1 -> memory [n][k] (k contiguous) ; transB=0 -> memory [k][n]
    sg_load<TB>(B, g.ldb, n0, g.N, 0, g.K, t, rb);
    sg_store<!TA>(sA[0], t, ra);
    sg_store<TB>(sB[0], t, rb);
    __syncthreads();
    const int nk = (g.K + SG_BK - 1) / SG_BK;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            sg_load<!TA>(A, g.lda, m0, g.M, (kt + 1) * SG_BK, g.K, t, ra);
            sg_load<TB>(B, g.ldb, n0, g.N, (kt + 1) * SG_BK, g.K, t, rb);
        }
        const float *pa = sA[buf] + (lane >> 5) * SG_LD + wm * 32 + (lane & 31);
        const float *pb = sB[buf] + (lane >> 5) * SG_LD + wn * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < SG_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk * SG_LD], pb[kk * SG_LD], acc, 0, 0, 0);
        if (kt + 1 < nk) {
            sg_store<!TA>(sA[buf ^ 1], t, ra);
            sg_store<TB>(sB[buf ^ 1], t, rb);
        }
        __syncthreads();
    }
    // C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) {
                float *p = C + (size_t)row * g.ldc + col;
                float v = g.alpha * acc[r] + bv;
                if (g.beta != 0.f) v += g.beta * *p;
                *p = v;
            }
        }
    }
}

// ---- 128 x 128 tile variant for the large GEMMs (Dense / LM head / Q heads and their gradients).
// With 64 x 64 tiles every K-step pulls 8 KiB through L2 for 131 kFLOP (16 FLOP/B): at the f32 MFMA rate that is ~10 TB/s
// of L2 -> LDS traffic chip-wide, so the small tile is L2-bound.  Here each wave owns 2 x 2 MFMA tiles (64 x 64 outputs,
// 64 accumulator registers): 32 FLOP/B, every A/B fragment read from LDS feeds two MFMAs.  Same k-major LDS image and the
// same four transpose forms; global loads are lane-consecutive along the contiguous dimension (arbitrary ld / alignment:
// the head matrices have ld = 50258).
constexpr int SG2_BM = 128, SG2_BN = 128, SG2_BK = 16, SG2_LD = 132;

template <bool K_CONTIG>   // true: memory [row][k] ; false: memory [k][row]
__device__ __forceinline__ void sg2_load(const float *__restrict__ X, int ld, int row0, int nrows, int k0, int K, int t, float (&r)[8]) {
    if (K_CONTIG) {
        const int k = k0 + (t & 15);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = row0 + (t >> 4) + 16 * i;
            r[i] = (row < nrows && k < K) ? X[(size_t)row * ld + k] : 0.f;
        }
    } else {
        const int k = k0 + (t >> 4);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = row0 + (t & 15) + 16 * i;
            r[i] = (k < K && row < nrows) ? X[(size_t)k * ld + row] : 0.f;
        }
    }
}
template <bool K_CONTIG>
__device__ __forceinline__ void sg2_store(float *__restrict__ S, int t, const float (&r)[8]) {
    if (K_CONTIG) {
        const int k = t & 15;
#pragma unroll
        for (int i = 0; i < 8; i++) S[k * SG2_LD + (t >> 4) + 16 * i] = r[i];
    } else {
        const int k = t >> 4;
#pragma unroll
        for (int i = 0; i < 8; i++) S[k * SG2_LD + (t & 15) + 16 * i] = r[i];
    }
}

// 16-byte fast path for a full, aligned tile (workgroup-uniform choice per operand): the same k-major LDS image, fetched as
// two float4 per thread instead of eight guarded scalar loads.
template <bool K_CONTIG>
__device__ __forceinline__ void sg2_load_vec(const float *__restrict__ X, int ld, int row0, int k0, int t, float (&r)[8]) {
    const float *p = K_CONTIG ? X + (size_t)(row0 + (t >> 1)) * ld + k0 + (t & 1) * 8      // [row][k]: row t>>1, k (t&1)*8 .. +7
                              : X + (size_t)(k0 + (t >> 4)) * ld + row0 + (t & 15) * 8;     // [k][row]: k t>>4, rows (t&15)*8 .. +7
    const f32x4v v0 = *reinterpret_cast<const f32x4v *>(p), v1 = *reinterpret_cast<const f32x4v *>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) { r[i] = v0[i]; r[4 + i] = v1[i]; }
}
template <bool K_CONTIG>
__device__ __forceinline__ void sg2_store_vec(float *__restrict__ S, int t, const float (&r)[8]) {
    if (K_CONTIG) {
        const int row = t >> 1, k = (t & 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) S[(k + i) * SG2_LD + row] = r[i];
    } else {
        float *p = S + (t >> 4) * SG2_LD + (t & 15) * 8;
        *reinterpret_cast<f32x4v *>(p) = f32x4v{r[0], r[1], r[2], r[3]};
        *reinterpret_cast<f32x4v *>(p + 4) = f32x4v{r[4], r[5], r[6], r[7]};
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_f32_128_kernel(SgemmArgs g) {
    __shared__ float sA[2][SG2_BK * SG2_LD];
    __shared__ float sB[2][SG2_BK * SG2_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bo = blockIdx.z / g.nb_inner, bi = blockIdx.z - bo * g.nb_inner;
    const float *A = g.A + bo * g.sAo + bi * g.sAi;
    const float *B = g.B + bo * g.sBo + bi * g.sBi;
    float *C = g.C + bo * g.sCo + bi * g.sCi;
    const int m0 = blockIdx.y * SG2_BM, n0 = blockIdx.x * SG2_BN;
    if (g.k_total > 0) g.K = min(g.K, g.k_total - (int)blockIdx.z * g.K);      // the last split-K slice may be shorter

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    // Global -> register -> LDS staging with the loads TWO K-steps ahead of their use: a K-step is 32 MFMAs = 2048 cycles (~0.85 us), less than a
    // loaded HBM / L2 round trip, so with the loads one step ahead a wave sat on vmcnt(0) before its LDS stores (SQ_VALU_MFMA_BUSY 0.72 - 0.77 of
    // the CU-busy cycles at 3 waves per SIMD: tools/pmc_mfma_sgemm.sh).  rn* = the step after next (in flight), rc* = the next step (landed).
    float rca[8], rcb[8], rna[8], rnb[8];
    // full, 16-byte-aligned tiles take the float4 path (workgroup-uniform per operand and K-step)
    // (a K that is not a multiple of the 16-wide step — the vocabulary, 50 257 — keeps the 16-byte path on every full step: only the last,
    //  partial step takes the guarded 4-byte loads; before, such products ran guarded throughout: 71 vs 113 TFLOP/s on the heads' dX)
    const bool va0 = m0 + SG2_BM <= g.M && (g.lda & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool vb0 = n0 + SG2_BN <= g.N && (g.ldb & 3) == 0 && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const int nk = (g.K + SG2_BK - 1) / SG2_BK;
#define SG2_LOAD(RA, RB, KT)                                                                                                         \
    do {                                                                                                                             \
        const bool full_ = ((KT) + 1) * SG2_BK <= g.K;                                                                               \
        if (va0 && full_) sg2_load_vec<!TA>(A, g.lda, m0, (KT) * SG2_BK, t, RA); else sg2_load<!TA>(A, g.lda, m0, g.M, (KT) * SG2_BK, g.K, t, RA); \
        if (vb0 && full_) sg2_load_vec<TB>(B, g.ldb, n0, (KT) * SG2_BK, t, RB); else sg2_load<TB>(B, g.ldb, n0, g.N, (KT) * SG2_BK, g.K, t, RB);   \
    } while (0)
#define SG2_STORE(RA, RB, KT, BUF)                                                                                                   \
    do {                                                                                                                             \
        const bool full_ = ((KT) + 1) * SG2_BK <= g.K;                                                                               \
        if (va0 && full_) sg2_store_vec<!TA>(sA[BUF], t, RA); else sg2_store<!TA>(sA[BUF], t, RA);                                   \
        if (vb0 && full_) sg2_store_vec<TB>(sB[BUF], t, RB); else sg2_store<TB>(sB[BUF], t, RB);                                     \
    } while (0)
    SG2_LOAD(rca, rcb, 0);
    SG2_STORE(rca, rcb, 0, 0);
    if (nk > 1) SG2_LOAD(rca, rcb, 1);              // in flight across the whole of step 0
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 2 < nk) SG2_LOAD(rna, rnb, kt + 2);
        const float *pa = sA[buf] + (lane >> 5) * SG2_LD + wm * 64 + (lane & 31);
        const float *pb = sB[buf] + (lane >> 5) * SG2_LD + wn * 64 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < SG2_BK; kk += 2) {
            const float a0 = pa[kk * SG2_LD], a1 = pa[kk * SG2_LD + 32];
            const float b0 = pb[kk * SG2_LD], b1 = pb[kk * SG2_LD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) SG2_STORE(rca, rcb, kt + 1, buf ^ 1);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) { rca[i] = rna[i]; rcb[i] = rnb[i]; }
    }
#undef SG2_LOAD
#undef SG2_STORE
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int col = n0 + wn * 64 + b * 32 + (lane & 31);
        if (col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; a++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    float *p = C + (size_t)row * g.ldc + col;
                    float v = g.alpha * acc[a][b][r] + bv;
                    if (g.beta != 0.f) v += g.beta * *p;
                    *p = v;
                }
            }
        }
    }
}

// split-K second pass: C = alpha * sum_s ws[s] + beta * C + bias, partials added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int S, int M, int N, float *__restrict__ C, int ldc,
                                                            float alpha, float beta, const float *__restrict__ bias) {
    const size_t MN = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int sidx = 0; sidx < S; sidx++) a += ws[(size_t)sidx * MN + i];
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        float v = alpha * a + (bias ? bias[c] : 0.f);
        float *p = C + (size_t)r * ldc + c;
        if (beta != 0.f) v += beta * *p;
        *p = v;
    }
}

}  // namespace lmrl

using namespace lmrl;

// process-wide split-K workspace (train path only; grown on demand, never inside a graph capture)
static float *g_splitk_ws = nullptr;
static size_t g_splitk_ws_floats = 0;

extern "C" void lmrl_sgemm_set_variant(int v);
static int g_sgemm_variant = 0;   // 0: auto, 1: always the 64 x 64 tile kernel (A/B hook)
void lmrl_sgemm_set_variant(int v) { g_sgemm_variant = v; }


extern "C" int lmrl_sgemm(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a_d, int lda, long sa_outer,
                          long sa_inner, const float *b_d, int ldb, long sb_outer, long sb_inner, float beta, float *c_d, int ldc,
                          long sc_outer, long sc_inner, int nb_outer, int nb_inner, const float *bias_d, void *stream) {
    LMRL_REQUIRE(a_d && b_d && c_d && m > 0 && n > 0 && k > 0 && nb_outer > 0 && nb_inner > 0, "lmrl_sgemm: bad argument");
    SgemmArgs g{a_d, b_d, c_d, bias_d, m, n, k, lda, ldb, ldc, sa_outer, sa_inner, sb_outer, sb_inner, sc_outer, sc_inner,
                nb_inner, alpha, beta};
    hipStream_t s = as_stream(stream);
    // Weight-gradient shapes (dW = X^T dY: small M x N, K = all tokens of the batch) have fewer 128 x 128 tiles than CUs:
    // split K across workgroups into a workspace and add the partials in a fixed order (no atomics -> deterministic).
    // The fp32 ROLLOUT's decode products (M = one row per env, ~1024; gpt2_f32_engine.py) have the same problem at a short K: the output
    // projections (N = d_model) are 48 tiles of 128 x 128 on 256 CUs — K = 768 / 3072 split 3 / 5 ways puts a workgroup on (almost) every CU
    // (41 -> ~17 us and 164 -> ~40 us by the cost model below); slices down to 128 long are allowed there.
    const long tiles0 = (long)ceil_div(n, SG2_BN) * ceil_div(m, SG2_BM);
    const bool short_k_few_tiles = m <= 2048 && k >= 512 && tiles0 <= 160;
    if (g_sgemm_variant == 0 && m >= 128 && n >= 128 && nb_outer * nb_inner == 1 && (k >= 4096 || short_k_few_tiles)) {
        const long tiles = tiles0;
        const int min_slice = k >= 4096 ? 1024 : 128;
        // S slices: tiles * S workgroups run in rounds of the 256 CUs and a round costs one slice (K / S), so time ~ ceil(tiles S / 256) / S — S = 4
        // on 144 tiles is 2.25 rounds = 3 (0.75 of a full-K tile time, the measured 85 vs 113 TFLOP/s), S = 7 is 3.94 = 4 (0.57) — plus the
        // partials' round trip through HBM (2 S m n floats at ~5 TB/s against ~0.5 TFLOP/s per CU).  Slices of unequal length (the last one shorter).
        // (also above 256 tiles: 384 tiles — the vocabulary heads' dX, K = 50 257 — are 1.5 rounds = 2 unsplit, 3 rounds of half the K when split in two)
        int S = 1, kc = k;
        if (tiles < 1024) {
            double best = 1e300;
            const double t_tile = 2.0 * SG2_BM * SG2_BN * (double)k / 0.5e12, t_part = 8.0 * (double)m * n / 5e12;
            for (int c = 1; c <= 16; c++) {
                const int per = ceil_div(ceil_div(k, SG2_BK), c) * SG2_BK;      // slice length, a multiple of the K-step
                const int slices = ceil_div(k, per);
                if (c > 1 && (per < min_slice || slices < 2)) continue;
                const double cost = (double)ceil_div((int)(tiles * slices), 256) * t_tile * per / k + (slices > 1 ? slices * t_part : 0.0);
                if (cost < best - 1e-12) { best = cost; S = slices; kc = per; }
            }
        }
        if (S > 1) {
            const size_t need = (size_t)S * m * n;
            if (need > g_splitk_ws_floats) {
                if (g_splitk_ws) { LMRL_CHECK_HIP(hipStreamSynchronize(s)); LMRL_CHECK_HIP(hipFree(g_splitk_ws)); g_splitk_ws = nullptr; g_splitk_ws_floats = 0; }
                LMRL_CHECK_HIP(hipMalloc(&g_splitk_ws, need * sizeof(float)));
                g_splitk_ws_floats = need;
            }
            SgemmArgs gs{a_d, b_d, g_splitk_ws, nullptr, m, n, kc, lda, ldb, n,
                         trans_a ? (long)kc * lda : (long)kc, 0, trans_b ? (long)kc : (long)kc * ldb, 0, (long)m * n, 0, 1, 1.f, 0.f, k};
            dim3 grid2(ceil_div(n, SG2_BN), ceil_div(m, SG2_BM), S);
            if (!trans_a && !trans_b) hipLaunchKernelGGL((sgemm_f32_128_kernel<false, false>), grid2, dim3(256), 0, s, gs);
            else if (!trans_a && trans_b) hipLaunchKernelGGL((sgemm_f32_128_kernel<false, true>), grid2, dim3(256), 0, s, gs);
            else if (trans_a && !trans_b) hipLaunchKernelGGL((sgemm_f32_128_kernel<true, false>), grid2, dim3(256), 0, s, gs);
            else hipLaunchKernelGGL((sgemm_f32_128_kernel<true, true>), grid2, dim3(256), 0, s, gs);
            LMRL_CHECK_LAUNCH();
            const size_t mn = (size_t)m * n;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 1023) / 1024 < 2048 ? (mn + 1023) / 1024 : 2048)), dim3(256), 0, s,
                               g_splitk_ws, S, m, n, c_d, ldc, alpha, beta, bias_d);
            LMRL_CHECK_LAUNCH();
            return LMRL_OK;
        }
    }
    if (g_sgemm_variant != 1 && m >= 128 && n >= 128) {   // large GEMMs: 128 x 128 tiles (2 x 2 MFMA tiles per wave)
        dim3 grid2(ceil_div(n, SG2_BN), ceil_div(m, SG2_BM), nb_outer * nb_inner);
        if (!trans_a && !trans_b) hipLaunchKernelGGL((sgemm_f32_128_kernel<false, false>), grid2, dim3(256), 0, s, g);
        else if (!trans_a && trans_b) hipLaunchKernelGGL((sgemm_f32_128_kernel<false, true>), grid2, dim3(256), 0, s, g);
        else if (trans_a && !trans_b) hipLaunchKernelGGL((sgemm_f32_128_kernel<true, false>), grid2, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((sgemm_f32_128_kernel<true, true>), grid2, dim3(256), 0, s, g);
        LMRL_CHECK_LAUNCH();
        return LMRL_OK;
    }
    dim3 grid(ceil_div(n, SG_BN), ceil_div(m, SG_BM), nb_outer * nb_inner);
    if (!trans_a && !trans_b) hipLaunchKernelGGL((sgemm_f32_kernel<false, false>), grid, dim3(256), 0, s, g);
    else if (!trans_a && trans_b) hipLaunchKernelGGL((sgemm_f32_kernel<false, true>), grid, dim3(256), 0, s, g);
    else if (trans_a && !trans_b) hipLaunchKernelGGL((sgemm_f32_kernel<true, false>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((sgemm_f32_kernel<true, true>), grid, dim3(256), 0, s, g);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
