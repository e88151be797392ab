// gemm8_bf16.h — large-tile bf16 MFMA GEMM for gfx950 (prefill GEMMs, LM head): 8 waves per workgroup, register-double-buffered
// fragments, one barrier per K-step placed MID-step.
//
//   C[M,N] = A[M,K] . W[N,K]^T (+ the fused epilogues of gemm_bf16.h)
//
// Why a second kernel: the 4-wave 64x64 / 128x64 tiles of gemm_bf16.h move (BM+BN)*128 B through L2 -> LDS -> VGPR per K-step for
// BM*BN*128 flop; at ~135 GB/s of L2 bandwidth per CU and 256 B/clk of LDS bandwidth that caps them well below the MFMA rate
// (128x64 ~1.5 PF L2 ceiling, and the LDS array is busier than the MFMA pipe).  A 256x256 (or 256x128 / 128x128) tile
// with 8 waves owning 128x64 (64x64) each quadruples the flop per staged byte.
//
// Pipeline (per workgroup; S LDS slots of (BM+BN)*128 B, filled by global_load_lds, XOR swizzle on the source address exactly as in
// gemm_bf16.h):
//   prologue   issue stages 0..S-1 ; wait stage 0 (counted vmcnt) ; barrier ; read fragments F0 = (stage 0, k-half 0)
//   step t     read F1 = (t, k-half 1)                      <- LDS reads of the second half fly under ...
//              MFMA(F0)                                     <- ... the MFMAs of the first half
//              wait stage t+1 (counted) ; lgkmcnt(0) ; barrier     [mid-step: every wave now holds stage t entirely in registers]
//              issue stage t+S into slot t % S              <- the slot just vacated; S-1 stages stay in flight
//              read F0' = (t+1, k-half 0)                   <- fly under ...
//              MFMA(F1)                                     <- ... the MFMAs of the second half
// so the matrix pipe always has a block of independent MFMAs to issue while the next fragments and the next stages are in flight,
// with ONE barrier per K-step.  Hazards: a slot is re-filled only after a barrier that every wave reached with lgkmcnt(0) (all its
// reads of that slot retired); a stage is read only after the issuing waves' counted vmcnt AND the barrier behind it.
#pragma once
#include "gemm_bf16.h"

namespace lmrl {


struct G8NoHook { __device__ __forceinline__ void operator()() const {} };

// acc[i][j] += W-fragment i (rows n0 + wn*TN + 16 i ..) x A-fragment j (rows m0 + wm*TM + 16 j ..) over K.  Starts with a barrier so that it
// can be called repeatedly on the same LDS ring (fused multi-operand kernels).  `head()` runs right after the ring prologue has been
// issued: either plain global loads whose results are first used after the loop (they fly under it), or a block that ends with
// vmcnt(0) (the LayerNorm moments, which need LDS scratch OUTSIDE the ring: every slot is being filled).
// KM ("K-major in memory", the train step's weight-gradient products dW = x^T . dy on the operands as their producers left them): A is
// [K][lda >= M] and W is [K][ldw >= N] — row kk holds x[kk][:] / dy[kk][:] — instead of [M][K] / [N][K].  A stage is then 64 kk-rows of BM (BN)
// columns; one wave-wide global_load_lds covers 4 rows x 256 B (BM = BN = 128), and the MFMA operand "8 consecutive kk of column m" is gathered
// by two ds_read_b64_tr_b16 (each hands lane (i, g) column i of a [4 kk][16 col] block whose sixteen 8-byte row pieces the group's lanes
// address: tools/tr_probe.hip).  The 32-byte column chunks of row r sit at chunk index c ^ key(r), key = (r & 3) | ((r >> 3) & 1) << 2, applied
// on the SOURCE address of the lane-linear load: the 32 lanes of a half-wave then read 8 distinct chunks = all 64 banks once.  No transposed
// copy of x or dy is ever written (98 transpose launches, 2.6 ms of a 37.6 ms ILQL step: profiles/r03_ilql_bf16_step_kernel_stats_fused_epilogues.csv).
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 g8_tr_frag(const char *p) {
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t *)(p));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t *)(p + 1024));   // 4 rows of 256 B further
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int BM, int BN, int WM, int WN, int STAGES, class Head = G8NoHook, bool KM = false>
__device__ __forceinline__ void g8_mainloop(const uint16_t *__restrict__ A, int lda, const uint16_t *__restrict__ W, int ldw, int K, int Mr, int m0,
                                            int n0,
                                            char *smem, f32x4 (&acc)[BN / WN / 16][BM / WM / 16], Head head = Head()) {
    constexpr int NW = WM * WN, BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN;            // per-wave output tile
    constexpr int FM = TM / 16, FN = TN / 16;            // 16-row activation / weight fragments per wave
    constexpr int LA = BM / 8 / NW, LW = BN / 8 / NW;    // global_load_lds instructions per wave per stage (1 KiB = 8 rows each)
    constexpr int L = LA + LW;
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && TM % 16 == 0 && TN % 16 == 0, "tile / wave layout");
    static_assert(STAGES >= 2 && STAGES <= 6, "2 to 6 LDS slots");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 15, lq = lane >> 4;
    const int nk = K / BK;
    const int lrow = lane >> 3, src_c = (lane & 7) ^ lrow;

    const uint16_t *ap[LA];
    const uint16_t *wp[LW];
    if constexpr (KM) {
        static_assert(BM == 128 && BN == 128, "K-major operands: 256-byte tile rows (128 x 128 tiles)");
#pragma unroll
        for (int i = 0; i < LA; i++) {
            const int r = 4 * (wave + NW * i) + (lane >> 4), key = (r & 3) | (((r >> 3) & 1) << 2);
            ap[i] = A + (size_t)r * lda + m0 + ((((lane & 15) >> 1) ^ key) << 4) + (lane & 1) * 8;
        }
#pragma unroll
        for (int i = 0; i < LW; i++) {
            const int r = 4 * (wave + NW * i) + (lane >> 4), key = (r & 3) | (((r >> 3) & 1) << 2);
            wp[i] = W + (size_t)r * ldw + n0 + ((((lane & 15) >> 1) ^ key) << 4) + (lane & 1) * 8;
        }
    } else {
#pragma unroll
        for (int i = 0; i < LA; i++) {
            int m = m0 + (wave + NW * i) * 8 + lrow;
            m = m < Mr ? m : Mr - 1;
            ap[i] = A + (size_t)m * lda + src_c * 8;
        }
#pragma unroll
        for (int i = 0; i < LW; i++) wp[i] = W + (size_t)(n0 + (wave + NW * i) * 8 + lrow) * ldw + src_c * 8;
    }
    const size_t kstep_a = KM ? (size_t)BK * lda : (size_t)BK, kstep_w = KM ? (size_t)BK * ldw : (size_t)BK;   // elements per K-step
    const int km_key = (lr >> 2) | ((lq & 1) << 2), km_row = 8 * lq + (lr >> 2), km_piece = (lr & 3) * 8;            // KM fragment gather

#ifndef LMRL_G8_W_AUX
#define LMRL_G8_W_AUX 0   /* cache policy bits of the weight-stream loads (tools: -DLMRL_G8_W_AUX=2 = nt) */
#endif
#define LMRL_G8_ISSUE(KT, SLOT)                                                                                       \
    do {                                                                                                              \
        char *sb_ = smem + (SLOT) * STAGE;                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < LA; i_++)                                                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ap[i_] + (size_t)(KT) * kstep_a), \
                                             (__attribute__((address_space(3))) void *)(sb_ + (wave + NW * i_) * 1024), 16, 0, 0); \
        _Pragma("unroll") for (int i_ = 0; i_ < LW; i_++)                                                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wp[i_] + (size_t)(KT) * kstep_w), \
                                             (__attribute__((address_space(3))) void *)(sb_ + BM * 128 + (wave + NW * i_) * 1024), 16, 0, LMRL_G8_W_AUX); \
    } while (0)
#define LMRL_G8_READ(FW, FA, SLOT, KK)                                                                                \
    do {                                                                                                              \
        const char *sA_ = smem + (SLOT) * STAGE;                                                                      \
        const char *sW_ = sA_ + BM * 128;                                                                             \
        if constexpr (KM) {                                                                                           \
            const int ro_ = ((KK) * 32 + km_row) * 256 + km_piece;                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++)                                                         \
                FW[i_] = g8_tr_frag(sW_ + ro_ + (((wn * FN + i_) ^ km_key) << 5));                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++)                                                         \
                FA[j_] = g8_tr_frag(sA_ + ro_ + (((wm * FM + j_) ^ km_key) << 5));                                    \
            break;                                                                                                    \
        }                                                                                                             \
        const int c_ = (KK) * 4 + lq;                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++) {                                                           \
            const int row_ = wn * TN + i_ * 16 + lr;                                                                  \
            FW[i_] = *reinterpret_cast<const bf16x8 *>(sW_ + row_ * 128 + ((c_ ^ (row_ & 7)) << 4));                  \
        }                                                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++) {                                                           \
            const int row_ = wm * TM + j_ * 16 + lr;                                                                  \
            FA[j_] = *reinterpret_cast<const bf16x8 *>(sA_ + row_ * 128 + ((c_ ^ (row_ & 7)) << 4));                  \
        }                                                                                                             \
    } while (0)
#if defined(LMRL_G8_ABLATE) && (LMRL_G8_ABLATE & 1)   /* tools/gemm8_bench.hip only: K loop without the MFMAs (fragments kept alive) */
#define LMRL_G8_MFMA(FW, FA)                                                                                          \
    do {                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++) asm volatile("" ::"v"(FW[i_]));                             \
        _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++) asm volatile("" ::"v"(FA[j_]));                             \
    } while (0)
#else
#define LMRL_G8_MFMA(FW, FA)                                                                                          \
    do {                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++)                                                             \
            _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++)                                                         \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FW[i_], FA[j_], acc[i_][j_], 0, 0, 0);          \
    } while (0)
#endif

    // ---- prologue
    __builtin_amdgcn_s_barrier();   // every wave is done reading the ring from a previous call
#pragma unroll
    for (int s = 0; s < STAGES; s++)
        if (s < nk) LMRL_G8_ISSUE(s, s);
    asm volatile("" ::: "memory");
    head();
    asm volatile("" ::: "memory");
    {
        const int inflight = (nk - 1 < STAGES - 1) ? nk - 1 : STAGES - 1;     // stages that may stay in flight behind stage 0
        if (inflight >= 5 && STAGES >= 6) wait_vmcnt<5 * L>();
        else if (inflight == 4 && STAGES >= 5) wait_vmcnt<4 * L>();
        else if (inflight == 3 && STAGES >= 4) wait_vmcnt<3 * L>();
        else if (inflight == 2 && STAGES >= 3) wait_vmcnt<2 * L>();
        else if (inflight == 1) wait_vmcnt<L>();
        else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    LMRL_G8_STAMP(1);
    bf16x8 fw0[FN], fa0[FM], fw1[FN], fa1[FM];
    LMRL_G8_READ(fw0, fa0, 0, 0);
    int slot = 0;
    for (int t = 0; t < nk; t++) {
        LMRL_G8_READ(fw1, fa1, slot, 1);
        LMRL_G8_MFMA(fw0, fa0);
        const int nslot = slot + 1 == STAGES ? 0 : slot + 1;
        if (t + 1 < nk) {
            const int behind = nk - 2 - t;                                    // stages issued after stage t+1
            const int inflight = behind < STAGES - 2 ? behind : STAGES - 2;
            if (inflight >= 4 && STAGES >= 6) wait_vmcnt<4 * L>();
            else if (inflight == 3 && STAGES >= 5) wait_vmcnt<3 * L>();
            else if (inflight == 2 && STAGES >= 4) wait_vmcnt<2 * L>();
            else if (inflight == 1 && STAGES >= 3) wait_vmcnt<L>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#if defined(LMRL_G8_ABLATE) && (LMRL_G8_ABLATE & 2)   /* K loop without the in-loop fetches */
#else
            if (t + STAGES < nk) LMRL_G8_ISSUE(t + STAGES, slot);
#endif
            LMRL_G8_READ(fw0, fa0, nslot, 0);
        }
        LMRL_G8_MFMA(fw1, fa1);
        slot = nslot;
    }
}

// Two K-steps per barrier on a 4-slot ring (slots {0,1} and {2,3} alternate): for GEMMs with ONE resident workgroup per CU (decode, M ~ 1024:
// 144-192 tiles of 128x128) the K loop is a dependent chain — fragment reads -> lgkmcnt(0) -> barrier -> next reads — that costs ~0.33 us per
// K-step before any MFMA or fetch (DESIGN.md 6b ablation); pairing the steps halves the number of barriers and drains, and keeps two whole
// stages (64 KB) in flight across each pair.  128 KB of LDS: affordable exactly because nothing else shares the CU.  K / 64 must be even.
template <int BM, int BN, int WM, int WN, class Head = G8NoHook>
__device__ __forceinline__ void g8_mainloop_pair(const uint16_t *__restrict__ A, int lda, const uint16_t *__restrict__ W, int ldw, int K, int Mr,
                                                 int m0, int n0, char *smem, f32x4 (&acc)[BN / WN / 16][BM / WM / 16], Head head = Head()) {
    constexpr int NW = WM * WN, BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int LA = BM / 8 / NW, LW = BN / 8 / NW;
    constexpr int L = LA + LW;
    constexpr int STAGE = (BM + BN) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 15, lq = lane >> 4;
    const int np = K / BK / 2;                              // pairs of K-steps
    const int lrow = lane >> 3, src_c = (lane & 7) ^ lrow;
    constexpr bool KM = false;                              // (the shared ISSUE / READ macros: K-major operands run on g8_mainloop only)
    constexpr size_t kstep_a = BK, kstep_w = BK;
    constexpr int km_key = 0, km_row = 0, km_piece = 0;
    const uint16_t *ap[LA];
    const uint16_t *wp[LW];
#pragma unroll
    for (int i = 0; i < LA; i++) {
        int m = m0 + (wave + NW * i) * 8 + lrow;
        m = m < Mr ? m : Mr - 1;
        ap[i] = A + (size_t)m * lda + src_c * 8;
    }
#pragma unroll
    for (int i = 0; i < LW; i++) wp[i] = W + (size_t)(n0 + (wave + NW * i) * 8 + lrow) * ldw + src_c * 8;

    __builtin_amdgcn_s_barrier();
    LMRL_G8_ISSUE(0, 0);
    LMRL_G8_ISSUE(1, 1);
    if (np > 1) { LMRL_G8_ISSUE(2, 2); LMRL_G8_ISSUE(3, 3); }
    asm volatile("" ::: "memory");
    head();
    asm volatile("" ::: "memory");
    if (np > 1) wait_vmcnt<2 * L>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    LMRL_G8_STAMP(1);
    bf16x8 fw0[FN], fa0[FM], fw1[FN], fa1[FM];
    LMRL_G8_READ(fw0, fa0, 0, 0);
    int a = 0;
    for (int u = 0; u < np; u++) {
        const int b = a + 1, an = a ^ 2;
        LMRL_G8_READ(fw1, fa1, a, 1);
        LMRL_G8_MFMA(fw0, fa0);
        LMRL_G8_READ(fw0, fa0, b, 0);
        LMRL_G8_MFMA(fw1, fa1);
        LMRL_G8_READ(fw1, fa1, b, 1);
        LMRL_G8_MFMA(fw0, fa0);
        if (u + 1 < np) {
            wait_vmcnt<0>();                                // the next pair (issued one pair ago) has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // ... for every wave, and every wave holds this pair in registers
            if (u + 2 < np) { LMRL_G8_ISSUE(2 * u + 4, a); LMRL_G8_ISSUE(2 * u + 5, b); }
            LMRL_G8_READ(fw0, fa0, an, 0);
        }
        LMRL_G8_MFMA(fw1, fa1);
        a = an;
    }
}
#undef LMRL_G8_ISSUE
#undef LMRL_G8_READ
#undef LMRL_G8_MFMA

// ---- WD ("weights direct"), round 5: for the decode-sized products (M ~ 1024: 144 - 192 tiles, ONE workgroup per CU) the K loop is bound by
// what a CU ingests through global_load_lds (~56 GB/s per CU): both operands of a K-step, (BM + BN) * 128 B, pass through that one
// path.  The weight operand does not need LDS at all: the MFMA fragment of W that lane (lr, lq) of wave (wm, wn) feeds — row n0 + wn TN + 16 i + lr,
// 8 consecutive k at 32 kk + 8 lq — is 16 contiguous bytes of the K-major weight matrix, i.e. exactly one global_load_dwordx4 into the VGPRs the
// MFMA reads.  Here only the ACTIVATION tile rides the LDS ring (BM * 128 B per stage: half the DMA bytes, half the LDS), the weight fragments of
// K-step t + STAGES are requested straight into a register ring right after step t's MFMAs released them, and both streams share the counted
// vmcnt waits (per K index: LA DMA instructions, then 2 FN register loads; everything retires in order).  The WM waves that share a weight row
// block fetch the same lines — the first miss fills the CU's L1, the others hit it.  K / 64 must be a multiple of STAGES (the register ring is
// indexed statically: the loop is unrolled STAGES-fold).  Same products in the same order as g8_mainloop: bit-identical results.
// The register loads are ordinary (compiler-visible) loads: the compiler's own s_waitcnt pass then guards every fragment's first use.  For that
// pass to arrive at the SAME counted waits as the hand-placed ones (and not at vmcnt(0) at the loop's back edge) the steady-state loop body is
// branch-free and issues its VMEM instructions in exactly the order the prologue does (compiler barriers pin the order of the register loads
// relative to the LDS-DMA builtins); the last STAGES K-steps — which issue nothing — are peeled.  (Inline-asm loads, which the compiler cannot see
// as asynchronous, were tried first: the register allocator is then free to re-use or copy a destination register while its load is in flight.)
__device__ __forceinline__ bf16x8 g8_load_frag(const uint16_t *p) { return *reinterpret_cast<const bf16x8 *>(p); }

template <int BM, int BN, int WM, int WN, int STAGES, class Head = G8NoHook>
__device__ __forceinline__ void g8_mainloop_wd(const uint16_t *__restrict__ A, int lda, const uint16_t *__restrict__ W, int ldw, int K, int Mr, int m0,
                                               int n0, char *smem, f32x4 (&acc)[BN / WN / 16][BM / WM / 16], Head head = Head()) {
    constexpr int NW = WM * WN, BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int LA = BM / 8 / NW;                      // global_load_lds instructions per wave per stage (activation tile only)
    constexpr int L = LA + 2 * FN;                       // VMEM instructions per wave per K index
    constexpr int STAGE = BM * 128;
    static_assert(BM % (8 * NW) == 0 && TM % 16 == 0 && TN % 16 == 0, "tile / wave layout");
    static_assert(STAGES >= 2 && STAGES <= 4, "2 to 4 ring slots (4 FN VGPRs each for the weight ring)");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 15, lq = lane >> 4;
    const int nk = K / BK;
    const int lrow = lane >> 3, src_c = (lane & 7) ^ lrow;
    const uint16_t *ap[LA];
#pragma unroll
    for (int i = 0; i < LA; i++) {
        int m = m0 + (wave + NW * i) * 8 + lrow;
        m = m < Mr ? m : Mr - 1;
        ap[i] = A + (size_t)m * lda + src_c * 8;
    }
    const uint16_t *wq[FN];
#pragma unroll
    for (int i = 0; i < FN; i++) wq[i] = W + (size_t)(n0 + wn * TN + i * 16 + lr) * ldw + lq * 8;
    bf16x8 wr[STAGES][2][FN];                            // weight fragments of K indices t .. t + STAGES - 1 (ring slot = index % STAGES)

#define LMRL_WD_ISSUE(KT, R)                                                                                          \
    do {                                                                                                              \
        char *sb_ = smem + (R) * STAGE;                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < LA; i_++)                                                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ap[i_] + (size_t)(KT) * BK), \
                                             (__attribute__((address_space(3))) void *)(sb_ + (wave + NW * i_) * 1024), 16, 0, 0); \
        asm volatile("" ::: "memory");                                                                                \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 2; kk_++)                                                           \
            _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++) wr[R][kk_][i_] = g8_load_frag(wq[i_] + (size_t)(KT) * BK + kk_ * 32); \
        asm volatile("" ::: "memory");                                                                                \
    } while (0)
#define LMRL_WD_READ_A(FA, SLOT, KK)                                                                                  \
    do {                                                                                                              \
        const char *sA_ = smem + (SLOT) * STAGE;                                                                      \
        const int c_ = (KK) * 4 + lq;                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++) {                                                           \
            const int row_ = wm * TM + j_ * 16 + lr;                                                                  \
            FA[j_] = *reinterpret_cast<const bf16x8 *>(sA_ + row_ * 128 + ((c_ ^ (row_ & 7)) << 4));                  \
        }                                                                                                             \
    } while (0)
#define LMRL_WD_MFMA(R, KK, FA)                                                                                       \
    do {                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++)                                                             \
            _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++)                                                         \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[R][KK][i_], FA[j_], acc[i_][j_], 0, 0, 0);   \
    } while (0)

    __builtin_amdgcn_s_barrier();   // every wave is done reading the ring from a previous call
#pragma unroll
    for (int s = 0; s < STAGES; s++) LMRL_WD_ISSUE(s, s);              // (K / 64 is a multiple of STAGES: every slot has its first index)
    head();
    asm volatile("" ::: "memory");
    wait_vmcnt<(STAGES - 1) * L>();
    __builtin_amdgcn_s_barrier();
    LMRL_G8_STAMP(1);
    bf16x8 fa0[FM], fa1[FM];
    LMRL_WD_READ_A(fa0, 0, 0);
    // one K-step on ring slot R: activation fragments of its second half, MFMAs of the first half, [counted wait for index t + 1 ; barrier],
    // DMA + register loads of index t + STAGES (ISSUE_), activation fragments of the next step's first half, MFMAs of the second half
#define LMRL_WD_STEP(R, WAITN, ISSUE_, MORE_)                                                                         \
    do {                                                                                                              \
        constexpr int RN_ = (R) + 1 == STAGES ? 0 : (R) + 1;                                                          \
        LMRL_WD_READ_A(fa1, R, 1);                                                                                    \
        LMRL_WD_MFMA(R, 0, fa0);                                                                                      \
        if (MORE_) {                                                                                                  \
            wait_vmcnt<(WAITN)>();                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
            __builtin_amdgcn_s_barrier();                                                                             \
        }                                                                                                             \
        if (MORE_) LMRL_WD_READ_A(fa0, RN_, 0);                               /* flies under the MFMAs below */       \
        LMRL_WD_MFMA(R, 1, fa1);                                                                                      \
        if (ISSUE_) LMRL_WD_ISSUE(t0 + (R) + STAGES, R);                      /* slot R: its LDS tile and its registers are free now */ \
    } while (0)
    int t0 = 0;
    for (; t0 + STAGES < nk; t0 += STAGES) {             // steady state: every step issues index t + STAGES; (STAGES - 2) newer indices behind t + 1
        LMRL_WD_STEP(0, (STAGES - 2) * L, true, true);
        LMRL_WD_STEP(1, (STAGES - 2) * L, true, true);
        if constexpr (STAGES >= 3) LMRL_WD_STEP(2, (STAGES - 2) * L, true, true);
        if constexpr (STAGES >= 4) LMRL_WD_STEP(3, (STAGES - 2) * L, true, true);
    }
    // the last STAGES steps: nothing left to issue, one index fewer in flight per step
    LMRL_WD_STEP(0, (STAGES - 2) * L, false, true);
    if constexpr (STAGES == 2) { LMRL_WD_STEP(1, 0, false, false); }
    if constexpr (STAGES == 3) { LMRL_WD_STEP(1, 0, false, true); LMRL_WD_STEP(2, 0, false, false); }
    if constexpr (STAGES == 4) { LMRL_WD_STEP(1, L, false, true); LMRL_WD_STEP(2, 0, false, true); LMRL_WD_STEP(3, 0, false, false); }
#undef LMRL_WD_STEP
#undef LMRL_WD_MFMA
#undef LMRL_WD_READ_A
#undef LMRL_WD_ISSUE
}

// ---- persistent form of g8_mainloop for a SEQUENCE of tiles on a 2-slot ring (K / 64 even): the ring never drains between tiles.  On entry the
// first two stages of THIS tile are already in flight or landed — issued by g8_stream_prime (first tile of the workgroup) or by the previous
// tile's call — and on exit the first two stages of the NEXT tile (if any) are in flight, so that its operands stream into LDS under this tile's
// epilogue instead of behind a cold ring prologue (a 128 x 128 tile with K = 768 is 12 K-steps behind ~1.2 us of fill latency).  Every wait is
// vmcnt(0): with two slots nothing else may be outstanding at a wait anyway, and the epilogue's own global accesses then never enter a counted wait.
// The epilogue must not touch the ring (it is being filled) and must use raw s_barrier (a __syncthreads carries vmcnt(0): it would wait for the DMA).
template <int BM, int BN, int WM, int WN>
struct G8Stream {
    static constexpr int NW = WM * WN, BK = 64, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static constexpr int LA = BM / 8 / NW, LW = BN / 8 / NW, STAGE = (BM + BN) * 128;
    const uint16_t *A, *W;       // wave-uniform operand bases (SGPRs); a tile is (m0, n0), also wave-uniform: per-lane state is recomputed per issue (a handful of
    int lda, ldw, Mr;            // VALU) instead of being held in VGPRs across the K loop and the epilogue — the kernel must stay within 128 VGPRs (2 workgroups / CU)

    // stage kt of tile (m0, n0) -> ring slot `slot`.  Operands below 4 GiB (32-bit byte offsets).
    __device__ __forceinline__ void issue(int m0, int n0, int kt, int slot, char *smem) const {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int lrow = lane >> 3, src_c = (lane & 7) ^ lrow;
        char *sb = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < LA; i++) {
            int m = m0 + (wave + NW * i) * 8 + lrow;
            m = m < Mr ? m : Mr - 1;
            const uint32_t off = ((uint32_t)m * (uint32_t)lda + (uint32_t)(src_c * 8 + kt * BK)) * 2u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(A) + off),
                                             (__attribute__((address_space(3))) void *)(sb + (wave + NW * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LW; i++) {
            const uint32_t off = ((uint32_t)(n0 + (wave + NW * i) * 8 + lrow) * (uint32_t)ldw + (uint32_t)(src_c * 8 + kt * BK)) * 2u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(W) + off),
                                             (__attribute__((address_space(3))) void *)(sb + BM * 128 + (wave + NW * i) * 1024), 16, 0, 0);
        }
    }
};

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void g8_stream_tile(const G8Stream<BM, BN, WM, WN> &st, int m0, int n0, bool has_next, int m0n, int n0n, int K, char *smem,
                                               f32x4 (&acc)[BN / WN / 16][BM / WM / 16]) {
    typedef G8Stream<BM, BN, WM, WN> S;
    constexpr int TM = S::TM, TN = S::TN, FM = S::FM, FN = S::FN, STAGE = S::STAGE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 15, lq = lane >> 4;
    const int nk = K / 64;
#define LMRL_G8S_READ(FW, FA, SLOT, KK)                                                                               \
    do {                                                                                                              \
        const char *sA_ = smem + (SLOT) * STAGE;                                                                      \
        const char *sW_ = sA_ + BM * 128;                                                                             \
        const int c_ = (KK) * 4 + lq;                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++) {                                                           \
            const int row_ = wn * TN + i_ * 16 + lr;                                                                  \
            FW[i_] = *reinterpret_cast<const bf16x8 *>(sW_ + row_ * 128 + ((c_ ^ (row_ & 7)) << 4));                  \
        }                                                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++) {                                                           \
            const int row_ = wm * TM + j_ * 16 + lr;                                                                  \
            FA[j_] = *reinterpret_cast<const bf16x8 *>(sA_ + row_ * 128 + ((c_ ^ (row_ & 7)) << 4));                  \
        }                                                                                                             \
    } while (0)
#define LMRL_G8S_MFMA(FW, FA)                                                                                         \
    do {                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < FN; i_++)                                                             \
            _Pragma("unroll") for (int j_ = 0; j_ < FM; j_++)                                                         \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FW[i_], FA[j_], acc[i_][j_], 0, 0, 0);          \
    } while (0)
    wait_vmcnt<0>();                         // this tile's stages 0 and 1 (issued one tile ago, or by the caller for the first tile) have landed for this wave ...
    __builtin_amdgcn_s_barrier();            // ... and for every wave
    bf16x8 fw0[FN], fa0[FM], fw1[FN], fa1[FM];
    LMRL_G8S_READ(fw0, fa0, 0, 0);
    int slot = 0;
    for (int t = 0; t < nk; t++) {
        LMRL_G8S_READ(fw1, fa1, slot, 1);
        LMRL_G8S_MFMA(fw0, fa0);
        const int nslot = slot ^ 1;
        if (t + 1 < nk) {
            wait_vmcnt<0>();                                           // stage t + 1 has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // every wave holds stage t in registers: its slot is free
            if (t + 2 < nk) st.issue(m0, n0, t + 2, slot, smem);
            else if (has_next) st.issue(m0n, n0n, 0, slot, smem);      // t = nk - 2: the next tile's stage 0 (slot 0: nk is even)
            LMRL_G8S_READ(fw0, fa0, nslot, 0);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (has_next) st.issue(m0n, n0n, 1, slot, smem);           // t = nk - 1: the next tile's stage 1 (slot 1)
        }
        LMRL_G8S_MFMA(fw1, fa1);
        slot = nslot;
    }
#undef LMRL_G8S_READ
#undef LMRL_G8S_MFMA
}

// SPLITK (fp32-output epilogue only; the weight-gradient products of the train step: K = B*T = 16 k .. 32 k against an output of a few dozen
// tiles): the grid holds kv_tmax = S copies of the tile grid; copy z accumulates K-steps [z * kv_d, min(K, (z + 1) * kv_d)) and stores its
// partial tile to C + z * M * ldc (splitk_reduce_kernel adds the copies in a fixed order).  The two ints alias the kv_* fields, which only
// EPI_BF16_LN_KV reads: no other instantiation's argument block changes.
template <int BM, int BN, int WM, int WN, int STAGES, int EPI, int NQ = 0, bool PAIR = false, bool SPLITK = false, bool KM = false, bool WD = false>
__global__ __launch_bounds__(WM *WN * 64) void gemm8_kernel(GemmArgs g, XcdMap xm) {
    static_assert(!PAIR || STAGES == 4, "the paired K loop runs on a 4-slot ring");
    static_assert(!WD || (!PAIR && !SPLITK && !KM), "weights-direct: the plain K loop only");
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / WM, TN = BN / WN;            // per-wave output tile
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr bool LN_IN = (EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN || EPI == EPI_BF16_LN_KV);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    int tile_m, tile_n;
    int wg_id = blockIdx.x;
    if constexpr (SPLITK) {
        static_assert(EPI == EPI_F32, "split-K partials are plain fp32 tiles");
        const int per = gridDim.x / g.kv_tmax, z = wg_id / per;      // per is a multiple of 8: id % 8 (the XCD) is unchanged
        wg_id -= z * per;
        const int k0 = z * g.kv_d;
        if constexpr (KM) { g.A += (size_t)k0 * g.lda; g.W += (size_t)k0 * (g.ldw > 0 ? g.ldw : g.N); }
        else { g.A += k0; g.W += k0; }
        g.K = min(g.kv_d, g.K - k0);
        g.C = reinterpret_cast<float *>(g.C) + (size_t)z * g.M * g.ldc;
    }
    if (!xcd_tile(xm, wg_id, tile_m, tile_n)) return;   // workgroup-uniform
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int Mr = g.m_dev ? *g.m_dev : g.M;
    if (m0 >= Mr) return;
    LMRL_G8_STAMP(0);
    const int lr = lane & 15, lq = lane >> 4;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; i++)
#pragma unroll
        for (int j = 0; j < FM; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr bool RESID = (EPI == EPI_RESID_F32_STATS || EPI == EPI_RESID_F32);
    constexpr int STAGE = WD ? BM * 128 : (BM + BN) * 128;
    float ln_mu[FM], ln_rs[FM];          // LN_IN: this lane's rows' (mu, rstd)
    f32x4 xres[FN][FM];                  // RESID: this lane's slice of the residual stream, prefetched under the K loop
    long kvrow[FM];                      // EPI_BF16_LN_KV: cache rows this lane's K / V columns are appended to (looked up in the prologue)
    if (LN_IN) {
        // as in gemm_bf16_glds_kernel: fetch the tile's BM x nslots moment slots behind the ring prologue, wait for everything, reduce to
        // (mu, rstd) per row in LDS scratch behind the ring (same association order: ln_row_moments) and keep this lane's rows in registers
        constexpr int NLQ = NQ > 0 ? (NQ * BM * 4 + NT - 1) / NT : 1;      // float4 loads per thread: BM * nslots * 8 B / (NT * 16 B)
        constexpr int TPR = NT / BM >= 4 ? 4 : (NT / BM >= 2 ? 2 : 1);
        auto head = [&]() {
            if constexpr (EPI == EPI_BF16_LN_KV) {
                if (g.kv_k) {
#pragma unroll
                    for (int j = 0; j < FM; j++) kvrow[j] = kv_append_row(g, m0 + wm * TM + j * 16 + lr, Mr);
                }
            }
            const int h4 = g.nslots / 2;
            const f32x4 *sp = reinterpret_cast<const f32x4 *>(g.stats + (size_t)m0 * g.nslots);
            const int lim = (Mr - m0 < BM ? Mr - m0 : BM) * h4;
            f32x4 st[NLQ];
#pragma unroll
            for (int k = 0; k < NLQ; k++) {
                const int idx = tid + NT * k;
                st[k] = sp[idx < lim ? idx : lim - 1];
            }
            wait_vmcnt<0>();
            float2 *part = reinterpret_cast<float2 *>(smem + STAGES * STAGE);    // [BM * h4] partial (sum, sum^2), behind the ring
            float2 *murs = part + BM * 4 * NQ;                                   // [BM] (mu, rstd)
#pragma unroll
            for (int k = 0; k < NLQ; k++)
                if (tid + NT * k < BM * h4) part[tid + NT * k] = make_float2(st[k][0] + st[k][2], st[k][1] + st[k][3]);   // rows >= M: unused
            __syncthreads();
            {
                const int row = tid / TPR, q = tid % TPR;
                if (row < BM) {
                    const float2 mr = ln_row_moments<TPR>(part + row * h4, h4, q, g.inv_d, g.eps);
                    if (q == 0) murs[row] = mr;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < FM; j++) {
                const float2 p2 = murs[wm * TM + j * 16 + lr];
                ln_mu[j] = p2.x; ln_rs[j] = p2.y;
            }
        };
        if constexpr (WD) g8_mainloop_wd<BM, BN, WM, WN, STAGES>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, head);
        else if constexpr (PAIR) g8_mainloop_pair<BM, BN, WM, WN>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, head);
        else g8_mainloop<BM, BN, WM, WN, STAGES>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, head);
    } else if (RESID) {
        auto prefetch = [&]() {
#pragma unroll
            for (int i = 0; i < FN; i++)
#pragma unroll
                for (int j = 0; j < FM; j++) {
                    int m = m0 + wm * TM + j * 16 + lr;
                    m = m < Mr ? m : Mr - 1;
                    int n = n0 + wn * TN + i * 16 + lq * 4;
                    n = n < g.n_store ? n : 0;
                    xres[i][j] = (EPI == EPI_RESID_F32 && g.resid) ? *reinterpret_cast<const f32x4 *>(g.resid + (size_t)m * g.ldr + n)
                                                                   : *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(g.C) + (size_t)m * g.ldc + n);
                }
        };
        if constexpr (WD) g8_mainloop_wd<BM, BN, WM, WN, STAGES>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, prefetch);
        else if constexpr (PAIR) g8_mainloop_pair<BM, BN, WM, WN>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, prefetch);
        else g8_mainloop<BM, BN, WM, WN, STAGES>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc, prefetch);
    } else {
        if constexpr (WD) g8_mainloop_wd<BM, BN, WM, WN, STAGES>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc);
        else
        if constexpr (KM) g8_mainloop<BM, BN, WM, WN, STAGES, G8NoHook, true>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.N, g.K, Mr, m0, n0, smem, acc);
        else if constexpr (PAIR) g8_mainloop_pair<BM, BN, WM, WN>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc);
        else g8_mainloop<BM, BN, WM, WN, STAGES>(g.A, g.lda, g.W, g.ldw > 0 ? g.ldw : g.K, g.K, Mr, m0, n0, smem, acc);
    }
    LMRL_G8_STAMP(2);

    // ---- epilogues (same arithmetic as gemm_bf16_glds_kernel; a lane holds C[m][n..n+3], n = n0 + wn*TN + 16 i + 4 lq, m = m0 + wm*TM + 16 j + lr)
    if constexpr (EPI == EPI_RESID_F32_STATS) {
        // x += acc + bias ; xb = bf16(x) ; the wave's TN columns are TN/32 stat slots of the row: slot s holds (sum x, sum x^2) of columns
        // [32 s, 32 s + 32).  A fragment i covers 16 columns: slot = (n0 + wn*TN + 16 i) / 32, two fragments per slot.
        static_assert(TN % 32 == 0, "stat slots are 32 columns wide");
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int m = m0 + wm * TM + j * 16 + lr;
            const bool row_ok = m < Mr;
#pragma unroll
            for (int i2 = 0; i2 < FN; i2 += 2) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = i2; i < i2 + 2; i++) {
                    const int n = n0 + wn * TN + i * 16 + lq * 4;
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
                const int sl = (n0 + wn * TN + i2 * 16) >> 5;
                if (lq == 0 && row_ok && sl < g.nslots) g.stats[(size_t)m * g.nslots + sl] = make_float2(s1, s2);
            }
        }
        LMRL_G8_STAMP(3);
        return;
    }

    if constexpr (EPI == EPI_F32_GELU_BF16) {
        // train forward, c_fc: the fp32 pre-activation (the gelu backward reads it) and the bf16 gelu output (the c_proj operand and, transposed,
        // the operand of its dW product) from one accumulator pass — no stand-alone gelu launch re-reading [M][N] floats
        static_assert(FN % 2 == 0, "fragment pairs");
#pragma unroll
        for (int i = 0; i < FN; i += 2) {
            const int nA = n0 + wn * TN + i * 16 + lq * 4, nB = nA + 16;
            f32x4 bA = f32x4{0.f, 0.f, 0.f, 0.f}, bB = bA;
            if (g.bias) { bA = *reinterpret_cast<const f32x4 *>(g.bias + nA); bB = *reinterpret_cast<const f32x4 *>(g.bias + nB); }
            const int n_st = n0 + wn * TN + (i + (lq & 1)) * 16 + (lq >> 1) * 8;
#pragma unroll
            for (int j = 0; j < FM; j++) {
                const int m = m0 + wm * TM + j * 16 + lr;
                f32x4 vA = acc[i][j] + bA, vB = acc[i + 1][j] + bB;
                if (g.C && g.pre_bf16) {      // the pre-activation rounded to bf16 (all the gelu backward reads of it): half the bytes, same 16-byte stores as the gelu output
                    auto p0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(vA[0], vA[1]), pack_bf16x2(vB[0], vB[1]), false, false);
                    auto p1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(vA[2], vA[3]), pack_bf16x2(vB[2], vB[3]), false, false);
                    if (m < Mr && n_st < g.n_store) *reinterpret_cast<u32x4 *>(reinterpret_cast<uint16_t *>(g.C) + (size_t)m * g.ldc + n_st) = u32x4{p0[0], p1[0], p0[1], p1[1]};
                } else if (g.C) {  // null: a forward nobody differentiates (the ILQL target network) — the pre-activation is not stored
                    if (m < Mr && nA < g.n_store) *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + nA) = vA;
                    if (m < Mr && nB < g.n_store) *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + nB) = vB;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) { vA[r] = gelu_new(vA[r]); vB[r] = gelu_new(vB[r]); }
                uint32_t a0 = pack_bf16x2(vA[0], vA[1]), a1 = pack_bf16x2(vA[2], vA[3]);
                uint32_t b0 = pack_bf16x2(vB[0], vB[1]), b1 = pack_bf16x2(vB[2], vB[3]);
                auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                if (m < Mr && n_st < g.n_store) *reinterpret_cast<u32x4 *>(g.xb + (size_t)m * g.ldxb + n_st) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LMRL_G8_STAMP(3);
        return;
    }

    if constexpr (EPI == EPI_BF16_CE) {
        // log-sum-exp partials of the logits AS STORED (their bf16 rounding, widened back to fp32: the reference's bf16 mode rounds the logits once
        // and evaluates lse and softmax on those), and the target column's logit from the fp32 accumulators (Q(s, a) of the TD terms keeps its
        // fp32 value).  lse and the backward's exp(stored logit - lse) then describe the SAME numbers: softmax rows sum to 1 whatever |logit| is
        // (with lse taken from the unrounded accumulators every p was off by exp(|x| 2^-9): 10 - 30 % at |x| ~ 50 - 100).  A lane holds,
        // for each of its FM rows, 4 columns of each of the FN fragments; the row's other columns of this wave's TN-wide slab sit in the lanes lq ^ 1,
        // lq ^ 2 (xor 16 / 32).  One (max, sum exp) per (row, slab): slot = tile_n * WN + wn, pitch g.nslots.  Columns >= n_store are padding.
        const int slot = tile_n * WN + wn;
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int m = m0 + wm * TM + j * 16 + lr;
            const int mc = m < Mr ? m : Mr - 1;
            const int tgt = g.ce_targets ? g.ce_targets[mc] : -1;
            float vmax = -INFINITY;
            f32x4 v[FN];
#pragma unroll
            for (int i = 0; i < FN; i++) {
                const int n = n0 + wn * TN + i * 16 + lq * 4;
                f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (g.bias) b4 = *reinterpret_cast<const f32x4 *>(g.bias + n);
                const f32x4 full = acc[i][j] + b4;
                acc[i][j] = full;                                   // the stored logits below: bf16 of these values
                const uint32_t p0 = pack_bf16x2(full[0], full[1]), p1 = pack_bf16x2(full[2], full[3]);
                v[i] = f32x4{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (n + e < g.n_store) vmax = fmaxf(vmax, v[i][e]);
                    if (n + e == tgt && m < Mr) g.ce_tgt_logit[m] = full[e];
                }
            }
            vmax = fmaxf(vmax, __shfl_xor(vmax, 16));
            vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < FN; i++) {
                const int n = n0 + wn * TN + i * 16 + lq * 4;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (n + e < g.n_store) sum += __expf(v[i][e] - vmax);
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            if (lq == 0 && m < Mr) g.stats[(size_t)m * g.nslots + slot] = make_float2(vmax, sum);
        }
        if (!g.C) {       // inference (token log-probabilities only): nobody reads the logits themselves
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }

    constexpr bool BF16_OUT = (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16 || LN_IN || EPI == EPI_BF16_HEADS || EPI == EPI_GELU_BWD_BF16 ||
                               EPI == EPI_BF16_CE);
    static_assert(!(EPI == EPI_BF16_HEADS || EPI == EPI_GELU_BWD_BF16 || EPI == EPI_BF16_CE) || FN % 2 == 0, "the train-step bf16 epilogues store fragment pairs");
    if (BF16_OUT && (FN % 2 == 0)) {
        // bf16 outputs, fragment pairs (i, i+1): a lane holds columns [16i + 4lq, +4) of both; v_permlane16_swap (odd 16-lane rows of the
        // first operand <-> even rows of the second) regroups them so that every lane owns 8 CONSECUTIVE columns (16 B) of one fragment:
        // lanes with lq even keep fragment i (columns 8(lq/2) ..), lanes with lq odd take fragment i+1 -> one dwordx4 store per pair and
        // 64 contiguous bytes per row per instruction instead of two dwordx2 stores of 32 contiguous bytes.
#pragma unroll
        for (int i = 0; i < FN; i += 2) {
            const int nA = n0 + wn * TN + i * 16 + lq * 4, nB = nA + 16;
            f32x4 bA = f32x4{0.f, 0.f, 0.f, 0.f}, bB = bA, cA = bA, cB = bA;
            if (g.bias && EPI != EPI_BF16_CE) { bA = *reinterpret_cast<const f32x4 *>(g.bias + nA); bB = *reinterpret_cast<const f32x4 *>(g.bias + nB); }
            if (LN_IN) { cA = *reinterpret_cast<const f32x4 *>(g.colsum + nA); cB = *reinterpret_cast<const f32x4 *>(g.colsum + nB); }
            const int n_st = n0 + wn * TN + (i + (lq & 1)) * 16 + (lq >> 1) * 8;      // first of this lane's 8 columns after the regrouping
            // EPI_BF16_HEADS: both fragments lie in one 32-column group, i.e. in one head of one of q / k / v
            const int hd_d = g.hd_H * 64;
            const float hd_scale = (EPI == EPI_BF16_HEADS && nA < hd_d) ? 0.125f : 1.f;
            long hd_col = 0;
            if constexpr (EPI == EPI_BF16_HEADS) {
                const int which = n_st / hd_d, c = n_st - which * hd_d;
                hd_col = (long)which * g.hd_plane + (long)(c >> 6) * g.hd_Tp * 64 + (c & 63);
            }
            f32x4 fA[FM], fB[FM];          // EPI_GELU_BWD_BF16: the pre-activations under this lane's 2 x 4 columns, all rows of the fragment pair
            if constexpr (EPI == EPI_GELU_BWD_BF16) {
#pragma unroll
                for (int j = 0; j < FM; j++) {
                    int m = m0 + wm * TM + j * 16 + lr;
                    m = m < Mr ? m : Mr - 1;
                    if (g.pre_bf16) {      // bf16 pre-activations (row pitch ldr elements): 8 bytes per fragment
                        const uint16_t *pr = reinterpret_cast<const uint16_t *>(g.resid) + (size_t)m * g.ldr;
                        const uint2 ua = *reinterpret_cast<const uint2 *>(pr + (nA < g.n_store ? nA : 0)), ub = *reinterpret_cast<const uint2 *>(pr + (nB < g.n_store ? nB : 0));
                        fA[j] = f32x4{__uint_as_float(ua.x << 16), __uint_as_float(ua.x & 0xffff0000u), __uint_as_float(ua.y << 16), __uint_as_float(ua.y & 0xffff0000u)};
                        fB[j] = f32x4{__uint_as_float(ub.x << 16), __uint_as_float(ub.x & 0xffff0000u), __uint_as_float(ub.y << 16), __uint_as_float(ub.y & 0xffff0000u)};
                    } else {
                        fA[j] = *reinterpret_cast<const f32x4 *>(g.resid + (size_t)m * g.ldr + (nA < g.n_store ? nA : 0));
                        fB[j] = *reinterpret_cast<const f32x4 *>(g.resid + (size_t)m * g.ldr + (nB < g.n_store ? nB : 0));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < FM; j++) {
                const int m = m0 + wm * TM + j * 16 + lr;
                f32x4 vA, vB;
                if (LN_IN) { vA = (acc[i][j] - cA * ln_mu[j]) * ln_rs[j] + bA; vB = (acc[i + 1][j] - cB * ln_mu[j]) * ln_rs[j] + bB; }
                else { vA = acc[i][j] + bA; vB = acc[i + 1][j] + bB; }
                if constexpr (EPI == EPI_BF16_HEADS) { vA = vA * hd_scale; vB = vB * hd_scale; }
                if constexpr (EPI == EPI_GELU_BWD_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; r++) { vA[r] *= gelu_new_grad(fA[j][r]); vB[r] *= gelu_new_grad(fB[j][r]); }
                }
                if (EPI == EPI_GELU_BF16 || EPI == EPI_GELU_BF16_LN) {
#pragma unroll
                    for (int r = 0; r < 4; r++) { vA[r] = gelu_new(vA[r]); vB[r] = gelu_new(vB[r]); }
                }
                if (EPI == EPI_RELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; r++) { vA[r] = fmaxf(vA[r], 0.f); vB[r] = fmaxf(vB[r], 0.f); }
                }
                uint32_t a0 = pack_bf16x2(vA[0], vA[1]), a1 = pack_bf16x2(vA[2], vA[3]);
                uint32_t b0 = pack_bf16x2(vB[0], vB[1]), b1 = pack_bf16x2(vB[2], vB[3]);
                // after the swaps: lq = 0: (A.r0, A.r1) ; lq = 1: (B.r0, B.r1) ; lq = 2: (A.r2, A.r3) ; lq = 3: (B.r2, B.r3)   (r = source lane row)
                auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                if constexpr (EPI == EPI_BF16_HEADS) {
                    if (m < Mr && n_st < g.n_store) {
                        const int b = m / g.hd_T, t = m - b * g.hd_T;
                        *reinterpret_cast<u32x4 *>(reinterpret_cast<uint16_t *>(g.C) + hd_col + ((long)b * g.hd_H * g.hd_Tp + t) * 64) =
                            u32x4{s0[0], s1[0], s0[1], s1[1]};
                    }
                } else if (m < Mr && n_st < g.n_store)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<uint16_t *>(g.C) + (size_t)m * g.ldc + n_st) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                if constexpr (EPI == EPI_BF16_LN_KV) {
                    if (g.kv_k && n_st >= g.kv_d && n_st < g.n_store && kvrow[j] >= 0) {
                        const bool isv = n_st >= 2 * g.kv_d;
                        uint16_t *dst = (isv ? g.kv_v : g.kv_k) + kvrow[j] * g.kv_d + (n_st - (isv ? 2 : 1) * g.kv_d);
                        *reinterpret_cast<u32x4 *>(dst) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LMRL_G8_STAMP(3);
        return;
    }
#pragma unroll
    for (int i = 0; i < FN; i++) {
        const int n = n0 + wn * TN + i * 16 + lq * 4;
        if (n >= g.n_store) continue;
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g.bias) b4 = *reinterpret_cast<const f32x4 *>(g.bias + n);
        f32x4 c4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (LN_IN) c4 = *reinterpret_cast<const f32x4 *>(g.colsum + n);
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int m = m0 + wm * TM + j * 16 + lr;
            if (m >= Mr) continue;
            f32x4 v;
            if (LN_IN) v = (acc[i][j] - c4 * ln_mu[j]) * ln_rs[j] + b4;
            else v = acc[i][j] + b4;
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
                        *reinterpret_cast<uint2 *>((isv ? g.kv_v : g.kv_k) + kvrow[j] * g.kv_d + (n - (isv ? 2 : 1) * g.kv_d)) = o;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LMRL_G8_STAMP(3);
}

template <int BM, int BN, int WM, int WN, int STAGES, int EPI, int NQ = 0, bool PAIR = false, bool WD = false>
inline hipError_t gemm8_launch(const GemmArgs &g, hipStream_t s) {
    constexpr bool LN_IN = (EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN || EPI == EPI_BF16_LN_KV);
    constexpr size_t shmem = (size_t)STAGES * (WD ? BM : BM + BN) * 128 + (LN_IN ? (size_t)BM * 4 * NQ * 8 + BM * 8 : 0);   // ring (+ LayerNorm-moment scratch)
    static_assert(shmem <= 160 * 1024, "LDS ring exceeds 160 KiB");
    const XcdMap xm = make_xcd_map((g.M + BM - 1) / BM, g.N / BN, 2.0 * g.M * g.K, 2.0 * g.N * g.K);
    const int tiles = xcd_grid(xm);
    static bool attr_set = false;
    if (!attr_set && shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm8_kernel<BM, BN, WM, WN, STAGES, EPI, NQ, PAIR, false, false, WD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    {
        ProfScope ps(PROF_GEMM_128x128, s, 2.0 * (double)g.M * (double)g.N * (double)g.K);
        hipLaunchKernelGGL((gemm8_kernel<BM, BN, WM, WN, STAGES, EPI, NQ, PAIR, false, false, WD>), dim3(tiles), dim3(WM * WN * 64), shmem, s, g, xm);
    }
    return hipGetLastError();
}

// Split-K launch of the fp32-output kernel: S copies of the tile grid, partials to ws [S][M][ldc = N] (see gemm8_kernel<.., SPLITK>).
template <int BM, int BN, int WM, int WN, int STAGES, bool KM = false>
inline hipError_t gemm8_launch_splitk(GemmArgs g, float *ws, int S, int kchunk, hipStream_t s) {
    constexpr size_t shmem = (size_t)STAGES * (BM + BN) * 128;
    const XcdMap xm = make_xcd_map((g.M + BM - 1) / BM, g.N / BN, 2.0 * g.M * g.K / S, 2.0 * g.N * g.K / S);
    const int tiles = xcd_grid(xm);
    g.C = ws; g.ldc = g.N; g.n_store = g.N; g.bias = nullptr; g.m_dev = nullptr;
    g.kv_tmax = S; g.kv_d = kchunk;
    static bool attr_set = false;
    if (!attr_set && shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm8_kernel<BM, BN, WM, WN, STAGES, EPI_F32, 0, false, true, KM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm8_kernel<BM, BN, WM, WN, STAGES, EPI_F32, 0, false, true, KM>), dim3(tiles * S), dim3(WM * WN * 64), shmem, s, g, xm);
    return hipGetLastError();
}

}  // namespace lmrl
