// sampler.hip — fused LM-head GEMM + token sampling for gfx950.
//
// Replaces the per-token sampling step of the reference's generation loop (HF-Flax `_sample` reached via
// GPT2PPOPolicy.act, LLM_RL/algorithms/ppo/gpt2/interface.py:527-535): logits[:, -1] -> temperature ->
// (top-k) -> jax.random.categorical, and the ILQL variant whose logits are
//     pi_beta_logits + beta * min(q1_logits, q2_logits)
// (LLM_RL/algorithms/value_rl_base/gpt2/generation.py:97-119).
//
// jax.random.categorical(logits) IS argmax(logits + Gumbel noise), so sampling needs no normalisation:
// the LM-head GEMM tile epilogue draws the noise (Philox4x32-7, counter = (row, column/4, step, epoch)), keeps a
// per-row running (max, sum-exp, best perturbed score, its column, its logit) and only ~10 floats per
// (row, 128-column tile) ever reach HBM instead of the [B, 50257] fp32 logits (206 MB / token at B=1024).
// A second tiny kernel merges the per-tile partials into (token, log-prob).
//
// The random stream is OUR OWN documented counter scheme, not JAX threefry: the reference's key schedule
// lives in un-vendored third-party code so bit-matching its samples is "parity unpinned";
// greedy decoding (temperature == 0) is deterministic and is what parity tests pin.
#include "../../include/lmrl_amd.h"
#include "common.h"
#include <atomic>
#include "gemm_dispatch.h"
#include "threefry.h"

namespace lmrl {

// ---- Philox4x32-R (Salmon et al., SC'11), the counter-based generator also used by cuRAND/rocRAND.  The sampler uses
// R = 7 rounds: the smallest round count the paper reports as passing BigCrush; its 32x32->64 multiplies are quarter-rate
// on CDNA, so rounds are what the Gumbel epilogue pays for (10 -> 7 rounds = -12 us per 1024 x 50257 sampling step).
constexpr int kPhiloxRounds = 7;
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                       uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < kPhiloxRounds; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// u in (0,1): ((x >> 9) + 0.5) * 2^-23 — 23 random bits so that "+ 0.5" is exact in fp32 (with 24 bits the top value
// rounds to u == 1.0 and the Gumbel becomes +inf once per ~16.7 M draws, i.e. several times per 1024 x 50257 sampling
// step).  Gumbel(0,1) = -log(-log u), range [-2.8, 16.6].
__device__ __forceinline__ float gumbel_from_bits(uint32_t x) {
    const float u = ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f;
    // raw v_log_f32 (log2): u in [2^-24, 1) and -ln u in [6e-8, 16.7] are normal numbers, so the denormal pre-scaling that __logf
    // wraps around the instruction (compare + select + ldexp per call) is dead weight here
    const float e = -0.6931471805599453f * __builtin_amdgcn_logf(u);          // -ln u  ~ Exp(1)
    return -0.6931471805599453f * __builtin_amdgcn_logf(e);                   // -ln(-ln u)
}

// Gumbel(0,1) word of the LMRL_RNG_JAX stream: element i of jax.random.gumbel(key, (n,)) — accurate logf (the fast log of the Philox
// path is a 1-ulp-class approximation; this mode exists to match another implementation's draws, not to be fast)
__device__ __forceinline__ float gumbel_jax(uint32_t k0, uint32_t k1, uint32_t i, uint32_t n) {
    return -logf(-logf(jax_uniform_open(jax_random_word(k0, k1, i, n))));
}

// order-preserving unsigned key of a float (radix selects, candidate lists): a < b  <=>  key(a) < key(b); key 0 is below every float's key
__device__ __forceinline__ uint32_t f32_order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

struct SampleParams {
    float inv_temperature;   // 1/T ; greedy when `greedy` != 0
    float temperature;       // T (LMRL_RNG_JAX divides, as the TemperatureLogitsWarper does)
    int rng;                 // LMRL_RNG_PHILOX / LMRL_RNG_JAX
    uint32_t jax_n;          // LMRL_RNG_JAX: total words of the noise array = rows * vocab (the shape of the logits handed to categorical)
    int greedy;
    uint32_t seed_lo, seed_hi, step;
    const uint32_t *epoch;   // optional device word used as the 4th Philox counter word (lets a captured hipGraph draw fresh noise per replay)
    float steer_strength;    // added to the logit of steer_tok[row] (synthetic-workload hook, see bench.py)
    float beta;              // ILQL: logits = pi + beta*min(q1,q2) ; 0 with no q operands
    int vocab;               // logical vocabulary (columns >= vocab are padding -> -inf)
    unsigned long long *tile_mass;   // fused top-p WITHOUT top-k: [M][tiles_n] probability mass of every 128-column tile relative to the tile's own maximum (else null)
};

// 32.32 fixed-point probability mass of a logit relative to `ref` (>= the logit): the unit of every top-p sum — integers, so that LDS / shuffle sums
// come out the same in any order and in every kernel that forms them
__device__ __forceinline__ unsigned long long mass_fixed(float v, float ref, float inv_t) {
    const float e = __expf((v - ref) * inv_t);          // in (0, 1]
    return (unsigned long long)((double)e * 4294967296.0);
}
// The nucleus total when top-k is off, TILE-WISE (so that the fused path can form it without the row's logits): per 128-column tile t the integer sum
// S_t of mass_fixed(. , tile maximum), rescaled to the row maximum in one double product; total = sum_t tile_mass_rescaled(S_t, max_t, row max).
// Within 1e-5 (relative) of the sum of the per-column masses relative to the row maximum, and the same integer in every kernel.
__device__ __forceinline__ unsigned long long tile_mass_rescaled(unsigned long long s_t, float tmax, float rmax, float inv_t) {
    return (unsigned long long)((double)s_t * (double)__expf((tmax - rmax) * inv_t));
}

constexpr int kPartialFloats = 6;   // {max, sumexp, best_score, best_col, best_z, pad}

// LM head:   z[m][n] = (h[m] . wte[n])   (+ ILQL perturbation, + steer), then the sampling epilogue on z / T.
// One 128 x 128 tile per 8-wave workgroup (2 x 4 waves, 64 x 32 outputs each) on the large-tile main loop of gemm8_bf16.h, two
// workgroups per CU; up to three GEMM passes over the same output tile (pi, q1, q2) combined in registers.  The four column-quarter
// waves of a row merge their partials through LDS so exactly one partial per (row, 128-column tile) reaches HBM.
constexpr int kLmBM = 128, kLmBN = 128, kLmWM = 2, kLmWN = 4, kLmStages = 2;
// The Q-head forms (two or three products per tile, three accumulator sets: 173 VGPRs) fit ONE workgroup per CU whatever the ring — so they take a
// 4-slot ring (128 KB of the CU's 160 KB LDS): three stages in flight behind the one being consumed instead of one (round 6)
#ifdef LMRL_LM_Q_STAGES
constexpr int kLmQStages = LMRL_LM_Q_STAGES;
#else
constexpr int kLmQStages = 4;
#endif

struct RowPartial { float pmax, psum, best, best_z; int best_col; };

template <bool WANT_LP = true>
__device__ __forceinline__ void merge_partial(RowPartial &a, float om, float os, float ob, float oz, int oc) {
    if (WANT_LP) {      // log-sum-exp pieces only when the caller asked for the sampled token's log-probability
        const float nm = fmaxf(a.pmax, om);
        const float e1 = (a.pmax == -INFINITY) ? 0.f : __expf(a.pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
        a.psum = a.psum * e1 + os * e2;
        a.pmax = nm;
    }
    if (ob > a.best || (ob == a.best && oc < a.best_col)) { a.best = ob; a.best_col = oc; a.best_z = oz; }
}

// ---- sampling epilogue of one 128 x 128 tile (shared by the one-tile-per-workgroup kernel and the persistent one).  `st_pre[j]`: the steered column of
// this lane's row j (or -1), loaded by the caller; `xch`: LDS hand-off area [WM][TM rows][WN - 1][5] floats; RAW: raw s_barrier + lgkmcnt(0) instead of
// __syncthreads (whose release carries a vmcnt(0): the persistent kernel has the NEXT tile's LDS-DMA in flight here and must not wait for it).
template <bool RAW>
__device__ __forceinline__ void lm_barrier() {
    if (RAW) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    else __syncthreads();
}
template <int NOPS, bool WANT_LP, bool JAX, bool RAW>
__device__ __forceinline__ void lm_sample_epilogue(f32x4 (&z)[kLmBN / kLmWN / 16][kLmBM / kLmWM / 16], f32x4 (&qmin)[kLmBN / kLmWN / 16][kLmBM / kLmWM / 16],
                                                   const int (&st_pre)[kLmBM / kLmWM / 16], int m0, int n0, int tile_n, int tiles_n, int M,
                                                   float *__restrict__ partials, float *__restrict__ logits_out, int ldo, const SampleParams &sp, float *xch) {
    constexpr int BM = kLmBM, BN = kLmBN, WM = kLmWM, WN = kLmWN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4;
    // ---- sampling epilogue: every wave first reduces ALL its rows (FM partials per lane, lane groups merged by shuffles), then ONE
    // exchange through LDS merges the WN column quarters — two barriers per tile instead of two per 16-row fragment
    const uint32_t epoch = sp.epoch ? *sp.epoch : 0u;
    RowPartial rps[FM];
#pragma unroll
    for (int j = 0; j < FM; j++) {
        const int m = m0 + wm * TM + j * 16 + lr;
        const int st = st_pre[j];
        RowPartial rp{-INFINITY, 0.f, -INFINITY, 0.f, 0x7fffffff};
        float vv[FN][4];
#pragma unroll
        for (int i = 0; i < FN; i++) {
            const int n = n0 + wn * TN + i * 16 + lq * 4;
            uint32_t rnd[4] = {0, 0, 0, 0};
            if (!JAX && !sp.greedy) philox4x32_10((uint32_t)m, (uint32_t)(n >> 2), sp.step, epoch, sp.seed_lo, sp.seed_hi, rnd);
            float zz[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = z[i][j][r];
                if (NOPS > 1) v += sp.beta * qmin[i][j][r];   // generation.py:112-117
                if (n + r == st) v += sp.steer_strength;
                zz[r] = v;
            }
            if (logits_out && m < M) *reinterpret_cast<f32x4 *>(logits_out + (size_t)m * ldo + n) = f32x4{zz[0], zz[1], zz[2], zz[3]};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool colok = (n + r) < sp.vocab;
                float v, sc;
                if (JAX) {      // jax.random.categorical(key, logits / T): key = (seed_hi, seed_lo), word index = row * V + column
                    v = colok ? zz[r] / sp.temperature : -INFINITY;
                    sc = (sp.greedy || !colok || m >= M) ? v : v + gumbel_jax(sp.seed_hi, sp.seed_lo, (uint32_t)m * (uint32_t)sp.vocab + (uint32_t)(n + r), sp.jax_n);
                } else {
                    v = colok ? zz[r] * sp.inv_temperature : -INFINITY;
                    sc = sp.greedy ? v : v + gumbel_from_bits(rnd[r]);
                }
                const bool take = sc > rp.best;                 // a padding column scores -inf and never wins; selects, not branches
                rp.best = take ? sc : rp.best; rp.best_col = take ? n + r : rp.best_col; rp.best_z = take ? v : rp.best_z;
                vv[i][r] = v;
                if (WANT_LP) rp.pmax = fmaxf(rp.pmax, v);
            }
        }
        // this lane's 4*FN logits: one exp each against the lane maximum (padding columns are -inf -> exp = 0)
        if (WANT_LP && rp.pmax > -INFINITY) {
#pragma unroll
            for (int i = 0; i < FN; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) rp.psum += __expf(vv[i][r] - rp.pmax);
        }
        // merge the 4 lane groups (lq) that hold the same row: xor 16, 32
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1)
            merge_partial<WANT_LP>(rp, __shfl_xor(rp.pmax, o), __shfl_xor(rp.psum, o), __shfl_xor(rp.best, o), __shfl_xor(rp.best_z, o),
                                   __shfl_xor(rp.best_col, o));
        rps[j] = rp;
    }
    lm_barrier<RAW>();                                 // (one-tile kernel: the LDS ring is free and doubles as the hand-off area)
    if (wn > 0 && lq == 0) {
#pragma unroll
        for (int j = 0; j < FM; j++) {
            float *sl = xch + ((((size_t)wm * TM + j * 16 + lr) * (WN - 1)) + (wn - 1)) * 5;
            sl[0] = rps[j].pmax; sl[1] = rps[j].psum; sl[2] = rps[j].best; sl[3] = rps[j].best_z; sl[4] = __int_as_float(rps[j].best_col);
        }
    }
    lm_barrier<RAW>();
    if (wn == 0 && lq == 0) {
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int m = m0 + wm * TM + j * 16 + lr;
            const float *slot = xch + (((size_t)wm * TM + j * 16 + lr) * (WN - 1)) * 5;
            RowPartial rp = rps[j];
#pragma unroll
            for (int w = 0; w < WN - 1; w++)            // fixed order: column quarters 1, 2, 3
                merge_partial<WANT_LP>(rp, slot[w * 5 + 0], slot[w * 5 + 1], slot[w * 5 + 2], slot[w * 5 + 3], __float_as_int(slot[w * 5 + 4]));
            if (m < M && partials) {        // (partials == NULL: the fused top-k path's re-run of a flagged row block, which only wants logits_out)
                float *p = partials + ((size_t)m * tiles_n + tile_n) * kPartialFloats;
                p[0] = rp.pmax; p[1] = rp.psum; p[2] = rp.best; p[3] = __int_as_float(rp.best_col); p[4] = rp.best_z; p[5] = 0.f;
            }
        }
    }
}

// ---- fused top-k (round 5): candidate epilogue.  With top_k <= kTopCMaxK the sampled token can only be one of the row's k largest logits, and a
// 128-column tile holds eight or more of those only in rare rows: the epilogue keeps, per (row, tile), the tile's kTopC = 8 largest logits (order keys)
// with their columns — 64 bytes instead of 512 bytes of logits, and no noise at all: the Gumbel draw happens in the reduce kernel, for the k kept
// columns only — and topc_reduce_sample_kernel selects the row's k-th largest among the 8 x tiles candidates.  That selection is EXACT whenever no
// tile's 8th candidate reaches the k-th largest (everything a tile did not report lies at or below its 8th candidate); a row where one does is
// flagged and re-done from materialised logits (lmrl_lm_head_sample: re-run of the flagged 128-row blocks + the register-row kernel on the listed
// rows), so the result is the materialised path's, bit for bit, in every case.
// Selection in two steps, on VALUES only (v_max_u32 / v_min_u32 networks on order keys, no column payload): (1) the tile's 8th largest key per row —
// per lane a 19-comparator sort of its 8 keys, bitonic top-8 merges across the 4 lanes and (through LDS) the 4 waves that share a row; (2) every lane
// stores its keys at or above that threshold into the row's record, at slots the merging lane laid out (keys above the threshold first, in column-quarter
// order, then the keys equal to it; no slot-counter atomics).  Ties beyond 8 slots are dropped: they equal the record's minimum, which the reduce
// kernel's check treats as "possibly hidden".
constexpr int kTopC = 8, kCandWords = 2 * kTopC, kTopCMaxK = 256, kFbHeader = 16, kFbMaxBlocks = 64;      // record: 8 x {key, tile-local column} = 64 B
constexpr int kTopcCompactCap = 1024;             // reduce kernel: candidates at or above the pre-filter floor kept in LDS per row (more: the row is handed back)
constexpr int kTopcLdsBytes = (kLmBM * (kLmWN - 1) * kTopC + 2 * kLmBM) * 4;                                // merge lists [BM][WN-1][8] + {threshold, slot ranges} [BM][2]

#define LMRL_CE(x, y) do { const uint32_t mx_ = (x) > (y) ? (x) : (y); (y) = (x) > (y) ? (y) : (x); (x) = mx_; } while (0)
__device__ __forceinline__ void sort8_desc(uint32_t (&a)[8]) {          // Batcher's odd-even merge sort, 19 comparators
    LMRL_CE(a[0], a[1]); LMRL_CE(a[2], a[3]); LMRL_CE(a[4], a[5]); LMRL_CE(a[6], a[7]);
    LMRL_CE(a[0], a[2]); LMRL_CE(a[1], a[3]); LMRL_CE(a[4], a[6]); LMRL_CE(a[5], a[7]);
    LMRL_CE(a[1], a[2]); LMRL_CE(a[5], a[6]);
    LMRL_CE(a[0], a[4]); LMRL_CE(a[1], a[5]); LMRL_CE(a[2], a[6]); LMRL_CE(a[3], a[7]);
    LMRL_CE(a[2], a[4]); LMRL_CE(a[3], a[5]);
    LMRL_CE(a[1], a[2]); LMRL_CE(a[3], a[4]); LMRL_CE(a[5], a[6]);
}
// a bitonic sequence -> sorted descending (x, y sorted descending: max(x[i], y[7 - i]) is a bitonic sequence holding the 8 largest of the union)
__device__ __forceinline__ void bitonic8_desc(uint32_t (&a)[8]) {
    LMRL_CE(a[0], a[4]); LMRL_CE(a[1], a[5]); LMRL_CE(a[2], a[6]); LMRL_CE(a[3], a[7]);
    LMRL_CE(a[0], a[2]); LMRL_CE(a[1], a[3]); LMRL_CE(a[4], a[6]); LMRL_CE(a[5], a[7]);
    LMRL_CE(a[0], a[1]); LMRL_CE(a[2], a[3]); LMRL_CE(a[4], a[5]); LMRL_CE(a[6], a[7]);
}
#undef LMRL_CE

// cand: [M][tiles_n][8] {key, column} (a row's records are contiguous: the reduce kernel reads them coalesced; a tile scatters 128 x 64 B).  `lds`: kTopcLdsBytes.
template <bool RAW, bool MASS>      // MASS: also the tile's probability mass per row (top-p without top-k)
__device__ __forceinline__ void lm_topc_epilogue(f32x4 (&z)[kLmBN / kLmWN / 16][kLmBM / kLmWM / 16], const int (&st_pre)[kLmBM / kLmWM / 16], int m0, int n0,
                                                 int tile_n, int tiles_n, int M, uint32_t *__restrict__ cand, const SampleParams &sp, char *lds) {
    constexpr int BM = kLmBM, BN = kLmBN, WM = kLmWM, WN = kLmWN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static_assert(FN * 4 == 8 && FM == 4 && WN == 4, "lm_topc_epilogue: 8 keys per lane and row, 4 rows per lane, 4 column-quarter waves");
    uint32_t *lists = reinterpret_cast<uint32_t *>(lds);                // [BM][WN - 1][8]
    uint32_t *thr_l = lists + BM * (WN - 1) * kTopC;                    // [BM] {threshold key, slot range ends of the 4 column quarters (4 bytes)}
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4;
    if (!RAW) __syncthreads();                         // one-tile kernel: the LDS ring doubles as the hand-off area — every wave is done reading it
    uint32_t keys[FM][8], my8[8];                      // my8 (column-quarter 0 waves): the list of row j = lq, the row this lane merges across waves below
#pragma unroll
    for (int e = 0; e < 8; e++) my8[e] = 0u;
    // keys of this lane's 4 x 8 logits.  Wave-uniform fast path: no padding column in the tile and no steered column of any of the wave's rows in it
    bool st_here = false;
#pragma unroll
    for (int j = 0; j < FM; j++) st_here |= (unsigned)(st_pre[j] - n0) < (unsigned)BN;
    if (n0 + BN > sp.vocab || __ballot(st_here) != 0ull) {
#pragma unroll
        for (int j = 0; j < FM; j++)
#pragma unroll
            for (int i = 0; i < FN; i++) {
                const int n = n0 + wn * TN + i * 16 + lq * 4;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = z[i][j][r];
                    if (n + r == st_pre[j]) v += sp.steer_strength;
                    keys[j][i * 4 + r] = (n + r) < sp.vocab ? f32_order_key(v) : 0u;      // padding columns: the empty key
                }
            }
    } else {
#pragma unroll
        for (int j = 0; j < FM; j++)
#pragma unroll
            for (int i = 0; i < FN; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) keys[j][i * 4 + r] = f32_order_key(z[i][j][r]);
    }
#pragma unroll
    for (int j = 0; j < FM; j++) {
        uint32_t a[8];
#pragma unroll
        for (int e = 0; e < 8; e++) a[e] = keys[j][e];
        sort8_desc(a);
        // the 4 lane groups (lq) that hold the same row, all VALU: v_permlane16_swap(v, v) hands BOTH lanes of a row pair {even row's v, odd row's v},
        // v_permlane32_swap(v, v) {lower half's v, upper half's v} — the two lists of a merge arrive in the same registers on both sides, no selects
        {
            uint32_t x[8], y[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const auto sw = __builtin_amdgcn_permlane16_swap(a[e], a[e], false, false);
                x[e] = sw[0]; y[e] = sw[1];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) a[e] = x[e] > y[7 - e] ? x[e] : y[7 - e];
            bitonic8_desc(a);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const auto sw = __builtin_amdgcn_permlane32_swap(a[e], a[e], false, false);
                x[e] = sw[0]; y[e] = sw[1];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) a[e] = x[e] > y[7 - e] ? x[e] : y[7 - e];
            bitonic8_desc(a);
        }
        if (wn > 0) {
            if (lq == 0) {
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                u32x4_t *dst = reinterpret_cast<u32x4_t *>(lists + ((size_t)(wm * TM + j * 16 + lr) * (WN - 1) + (wn - 1)) * kTopC);
                dst[0] = u32x4_t{a[0], a[1], a[2], a[3]};
                dst[1] = u32x4_t{a[4], a[5], a[6], a[7]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) my8[e] = lq == j ? a[e] : my8[e];
        }
        __builtin_amdgcn_sched_barrier(0);             // one row at a time: interleaving the four rows' networks only adds register pressure
    }
    lm_barrier<RAW>();
    unsigned long long *cand64 = reinterpret_cast<unsigned long long *>(cand);
    if (wn == 0) {                                     // lane group lq merges row j = lq of this wave's 64 rows with the three other column quarters
        const int row = wm * TM + lq * 16 + lr;
        uint32_t lw[WN][8];                            // the four quarters' lists (quarter 0: this wave's own)
#pragma unroll
        for (int e = 0; e < 8; e++) lw[0][e] = my8[e];
#pragma unroll
        for (int w = 1; w < WN; w++) {
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t *src = reinterpret_cast<const u32x4_t *>(lists + ((size_t)row * (WN - 1) + (w - 1)) * kTopC);
            const u32x4_t b0 = src[0], b1 = src[1];
            lw[w][0] = b0[0]; lw[w][1] = b0[1]; lw[w][2] = b0[2]; lw[w][3] = b0[3]; lw[w][4] = b1[0]; lw[w][5] = b1[1]; lw[w][6] = b1[2]; lw[w][7] = b1[3];
#pragma unroll
            for (int e = 0; e < 8; e++) my8[e] = my8[e] > lw[w][7 - e] ? my8[e] : lw[w][7 - e];
            bitonic8_desc(my8);
        }
        // thr: the tile's 8th largest key of this row (>= 1: an empty key never hits).  Slot ranges instead of a slot counter (LDS return atomics cost this
        // step 18 us per launch): per column quarter, how many of its keys lie ABOVE thr (at most 7 in all: they get the first slots, in quarter order) and
        // how many EQUAL it (the slots after those, clipped at 8: a dropped key equals the record's minimum) — each quarter's list holds every key of its
        // quarter above thr, so the counts are exact
        const uint32_t thr = my8[7] > 1u ? my8[7] : 1u;
        uint32_t g[WN], q[WN];
#pragma unroll
        for (int w = 0; w < WN; w++) {
            g[w] = 0; q[w] = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) { g[w] += lw[w][e] > thr ? 1u : 0u; q[w] += lw[w][e] == thr ? 1u : 0u; }
        }
        const uint32_t g1 = g[0], g2 = g1 + g[1], g3 = g2 + g[2], gt = g3 + g[3];
        auto c8 = [](uint32_t x) { return x < 8u ? x : 8u; };
        const uint32_t e0 = gt, e1 = c8(e0 + q[0]), e2 = c8(e1 + q[1]), e3 = c8(e2 + q[2]), total = c8(e3 + q[3]);
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2_t *>(thr_l + 2 * row) = u32x2_t{thr, (g1 << 4) | (g2 << 8) | (g3 << 12) | (e0 << 16) | (e1 << 20) | (e2 << 24) | (e3 << 28)};
        if (MASS) lists[(size_t)row * (WN - 1) * kTopC] = my8[0];      // the tile's largest key of this row, into the row's own (now read) list area
        if (total < (uint32_t)kTopC && m0 + row < M)   // fewer than 8 real columns (the vocabulary's last tile): empty slots
            for (uint32_t sl = total; sl < (uint32_t)kTopC; sl++) cand64[((size_t)(m0 + row) * tiles_n + tile_n) * kTopC + sl] = 0ull;
    }
    lm_barrier<RAW>();
#pragma unroll
    for (int j = 0; j < FM; j++) {
        const int row = wm * TM + j * 16 + lr;
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t tb = *reinterpret_cast<const u32x2_t *>(thr_l + 2 * row);
        const uint32_t thr = tb[0];
        uint32_t hg = 0, he = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) { hg |= keys[j][e] > thr ? (1u << e) : 0u; he |= keys[j][e] == thr ? (1u << e) : 0u; }
        // slots of the 4 lanes that share the row, in lane-group order: both counts through the two swaps (even / odd row of the pair, lower / upper half)
        const uint32_t c = (uint32_t)__popc(hg) | ((uint32_t)__popc(he) << 8);
        const auto s16 = __builtin_amdgcn_permlane16_swap(c, c, false, false);
        const uint32_t pair = s16[0] + s16[1];
        const auto s32 = __builtin_amdgcn_permlane32_swap(pair, pair, false, false);
        const uint32_t off = ((lq & 2) ? s32[0] : 0u) + ((lq & 1) ? s16[0] : 0u);
        uint32_t sg = ((tb[1] >> (4 * wn)) & (wn == 0 ? 0u : 15u)) + (off & 255u), se = ((tb[1] >> (16 + 4 * wn)) & 15u) + (off >> 8);
        uint32_t hit = hg | he;
        if (hit && m0 + row < M) {                     // (1 / 16 of the keys: a loop over the set bits, typically one or two trips per wave)
            unsigned long long *dst = cand64 + ((size_t)(m0 + row) * tiles_n + tile_n) * kTopC;
            while (hit) {
                const int e = __ffs((int)hit) - 1;
                hit &= hit - 1u;
                uint32_t k = keys[j][0];
#pragma unroll
                for (int qq = 1; qq < 8; qq++) k = e == qq ? keys[j][qq] : k;
                const bool above = (hg >> e) & 1u;
                const uint32_t slot = above ? sg : se;
                sg += above ? 1u : 0u; se += above ? 0u : 1u;
                if (slot < (uint32_t)kTopC) dst[slot] = (unsigned long long)k | ((unsigned long long)(uint32_t)(wn * TN + (e >> 2) * 16 + lq * 4 + (e & 3)) << 32);
            }
        }
    }
    if (MASS) {
        // top-p without top-k: the tile's probability mass per row, relative to the tile's own maximum (tile_mass_rescaled) — integer sums over the lane's
        // 8 columns, the 4 lanes of the row, the 4 column-quarter waves (through the row's list area: 12 eight-byte words, word 0 holds the maximum)
        unsigned long long *mx = reinterpret_cast<unsigned long long *>(lists);
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int row = wm * TM + j * 16 + lr;
            const uint32_t tk = lists[(size_t)row * (WN - 1) * kTopC];
            const float tmax = f32_from_order_key(tk);
            unsigned long long sm = 0ull;
            if (tk != 0u && tmax != -INFINITY) {
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (keys[j][e] != 0u) sm += mass_fixed(f32_from_order_key(keys[j][e]), tmax, sp.inv_temperature);
            }
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            if (lq == 0) mx[(size_t)row * 12 + 1 + wn] = sm;
        }
        lm_barrier<RAW>();
        if (wn == 0) {
            const int row = wm * TM + lq * 16 + lr;
            if (m0 + row < M) sp.tile_mass[(size_t)(m0 + row) * tiles_n + tile_n] = (mx[(size_t)row * 12 + 1] + mx[(size_t)row * 12 + 2]) + (mx[(size_t)row * 12 + 3] + mx[(size_t)row * 12 + 4]);
        }
    }
}

// FLAGGED (the fused top-k path's hand-back): grid = the vocabulary's tiles; a workgroup walks the 128-row blocks and computes only those whose flag is set
template <int NOPS, bool WANT_LP, bool JAX = false, int TOPC = 0, bool FLAGGED = false>      // TOPC: 1 = candidate epilogue, 2 = candidates + tile masses
__global__ __launch_bounds__(kLmWM *kLmWN * 64) void lm_head_sample_kernel(const uint16_t *__restrict__ A0, const uint16_t *__restrict__ W0,
                                                             const uint16_t *__restrict__ A1, const uint16_t *__restrict__ W1,
                                                             const float *__restrict__ bias1,
                                                             const uint16_t *__restrict__ A2, const uint16_t *__restrict__ W2,
                                                             const float *__restrict__ bias2,
                                                             const int32_t *__restrict__ steer_tok,
                                                             float *__restrict__ partials,   // [M][tiles_n][kPartialFloats]
                                                             float *__restrict__ logits_out, // optional [M][ldo] f32
                                                             int M, int N, int K, int ldo, SampleParams sp, XcdMap xm,
                                                             const int32_t *__restrict__ block_flag,   // optional: run only the 128-row blocks whose flag is set
                                                             int32_t *__restrict__ fb) {               // TOPC: header of the flagged-row list, cleared here
    constexpr int BM = kLmBM, BN = kLmBN, WM = kLmWM, WN = kLmWN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = N / BN;
    if (TOPC && blockIdx.x == 0 && tid < kFbHeader + kFbMaxBlocks) fb[tid] = 0;     // (the reduce kernel of THIS step fills it, after this kernel)
    int tile_m = 0, tile_n = (int)blockIdx.x, tm_end = (M + BM - 1) / BM;
    if (!FLAGGED) {
        if (!xcd_tile(xm, blockIdx.x, tile_m, tile_n)) return;   // workgroup-uniform
        tm_end = tile_m + 1;
    }
    for (; tile_m < tm_end; tile_m++) {
    if (FLAGGED) {
        if (!block_flag[tile_m]) continue;
        __syncthreads();                               // (the previous block's epilogue is done with the LDS)
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int lr = lane & 15, lq = lane >> 4;

    f32x4 z[FN][FM];     // combined logits
    f32x4 qmin[FN][FM];  // running min(q1, q2) for the ILQL form

#pragma unroll
    for (int op = 0; op < NOPS; op++) {
        const uint16_t *A = op == 0 ? A0 : (op == 1 ? A1 : A2);
        const uint16_t *W = op == 0 ? W0 : (op == 1 ? W1 : W2);
        f32x4 acc[FN][FM];
#pragma unroll
        for (int i = 0; i < FN; i++)
#pragma unroll
            for (int j = 0; j < FM; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef LMRL_LM_Q_PAIR
        if (NOPS > 1 && (K / 64) % 2 == 0) g8_mainloop_pair<BM, BN, WM, WN>(A, K, W, K, K, M, m0, n0, smem, acc);
        else
#endif
        g8_mainloop<BM, BN, WM, WN, (NOPS > 1 ? kLmQStages : kLmStages)>(A, K, W, K, K, M, m0, n0, smem, acc);
#pragma unroll
        for (int i = 0; i < FN; i++) {
            const int n = n0 + wn * TN + i * 16 + lq * 4;
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (op == 1 && bias1) b4 = *reinterpret_cast<const f32x4 *>(bias1 + n);
            if (op == 2 && bias2) b4 = *reinterpret_cast<const f32x4 *>(bias2 + n);
#pragma unroll
            for (int j = 0; j < FM; j++) {
                if (op == 0) z[i][j] = acc[i][j];
                else if (op == 1) qmin[i][j] = acc[i][j] + b4;
                else {
                    const f32x4 q2 = acc[i][j] + b4;
#pragma unroll
                    for (int r = 0; r < 4; r++) qmin[i][j][r] = fminf(qmin[i][j][r], q2[r]);
                }
            }
        }
    }

    int st_pre[FM];
#pragma unroll
    for (int j = 0; j < FM; j++) {
        const int m = m0 + wm * TM + j * 16 + lr;
        st_pre[j] = (steer_tok && m < M) ? steer_tok[m] : -1;
    }
    if (TOPC) {
        // ILQL value policy (generation.py:112-117): the candidates are taken AFTER logits = pi_beta + beta * min(q1, q2) is formed — the same
        // expression (one fma per logit) as lm_sample_epilogue's, so a hand-back's materialised logits equal the values ranked here bit for bit
        if (NOPS > 1) {
#pragma unroll
            for (int i = 0; i < FN; i++)
#pragma unroll
                for (int j = 0; j < FM; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) z[i][j][r] += sp.beta * qmin[i][j][r];
        }
        lm_topc_epilogue<false, TOPC == 2>(z, st_pre, m0, n0, tile_n, tiles_n, M, reinterpret_cast<uint32_t *>(partials), sp, smem);
    }
    else lm_sample_epilogue<NOPS, WANT_LP, JAX, false>(z, qmin, st_pre, m0, n0, tile_n, tiles_n, M, partials, logits_out, ldo, sp, reinterpret_cast<float *>(smem));
    }
}

// ---- persistent form of the policy-only LM head (NOPS = 1, Philox / greedy): the first `n_persist` workgroups (two per CU) each walk `rounds` tile ids of
// the XCD map — b, b + n_persist, ... (the stride is a multiple of 8: a workgroup stays on its XCD's vocabulary slice and on ONE m-tile) — with the LDS ring
// running ahead into the next tile while the current tile's sampling epilogue executes (g8_stream_tile): a 128 x 128 tile with K = 768 is only 12 K-steps, and
// in the one-tile-per-workgroup kernel every tile starts behind a cold ring prologue.  The ids beyond rounds * n_persist (128 of 3200 at the bench shape) are
// one-tile workgroups at the END of the grid: the dispatcher starts them as the persistent ones retire, so the remainder spreads over the whole chip instead
// of giving a quarter of the workgroups a seventh tile.  (A dynamic work list — per-XCD heads pulled with returning atomics, stealing across XCDs — was built
// and measured: 168 us against 138 us for this static form, back to back: the pulls' round trips under a streaming load cost more than the balance buys.)
// Same arithmetic per tile, same association orders: bit-identical partials to the one-tile kernel (tests/test_gpu_timed_path.py).
// LDS: 2 x 32 KB ring + 7.7 KB hand-off area = 2 workgroups per CU.
template <bool WANT_LP, int TOPC = 0>
__global__ __launch_bounds__(kLmWM *kLmWN * 64, 4) void lm_head_sample_persist_kernel(const uint16_t *__restrict__ A0, const uint16_t *__restrict__ W0,
                                                                                      const int32_t *__restrict__ steer_tok, float *__restrict__ partials,
                                                                                      float *__restrict__ logits_out, int M, int N, int K, int ldo,
                                                                                      SampleParams sp, XcdMap xm, int n_persist, int rounds, int32_t *__restrict__ fb) {
    if (TOPC && blockIdx.x == 0 && threadIdx.x < kFbHeader + kFbMaxBlocks) fb[threadIdx.x] = 0;      // flagged-row list header (filled by this step's reduce kernel)
    constexpr int BM = kLmBM, BN = kLmBN, WM = kLmWM, WN = kLmWN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    typedef G8Stream<BM, BN, WM, WN> Stream;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xch = reinterpret_cast<float *>(smem + 2 * Stream::STAGE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN;
    const int lr = lane & 15;
    const int tiles_n = N / BN;
    // this workgroup's ids: [first, first + stride, ...), `count` of them (workgroup-uniform)
    const bool tail = (int)blockIdx.x >= n_persist;
    const int stride = n_persist, count = tail ? 1 : rounds;
    const int first = tail ? rounds * n_persist + ((int)blockIdx.x - n_persist) : (int)blockIdx.x;
    int k = 0, tile_m = 0, tile_n = 0;
    while (k < count && !xcd_tile(xm, first + k * stride, tile_m, tile_n)) k++;       // surplus ids of the XCD map fall outside the problem
    if (k >= count) return;
    const Stream st{A0, W0, K, K, M};
    st.issue(tile_m * BM, tile_n * BN, 0, 0, smem);
    st.issue(tile_m * BM, tile_n * BN, 1, 1, smem);
    while (true) {
        const int m0 = tile_m * BM, n0 = tile_n * BN, this_tile_n = tile_n;
        int nk_ = k + 1, ntm = 0, ntn = 0;
        while (nk_ < count && !xcd_tile(xm, first + nk_ * stride, ntm, ntn)) nk_++;
        const bool has_next = nk_ < count;
        int st_pre[FM];                                                // issued before the K loop: retired by its first wait, never behind the next tile's DMA
#pragma unroll
        for (int j = 0; j < FM; j++) {
            const int m = m0 + wm * TM + j * 16 + lr;
            st_pre[j] = (steer_tok && m < M) ? steer_tok[m] : -1;
        }
        f32x4 z[FN][FM];
#pragma unroll
        for (int i = 0; i < FN; i++)
#pragma unroll
            for (int j = 0; j < FM; j++) z[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        g8_stream_tile<BM, BN, WM, WN>(st, m0, n0, has_next, ntm * BM, ntn * BN, K, smem, z);
        if (TOPC) lm_topc_epilogue<true, TOPC == 2>(z, st_pre, m0, n0, this_tile_n, tiles_n, M, reinterpret_cast<uint32_t *>(partials), sp, reinterpret_cast<char *>(xch));
        else lm_sample_epilogue<1, WANT_LP, false, true>(z, z, st_pre, m0, n0, this_tile_n, tiles_n, M, partials, logits_out, ldo, sp, xch);
        if (!has_next) break;
        k = nk_; tile_m = ntm; tile_n = ntn;
    }
}

// merge partials of one row: token = argmax best_score ; logprob = best_z - logsumexp(z)
__global__ __launch_bounds__(256) void sample_reduce_kernel(const float *__restrict__ partials, const uint8_t *__restrict__ active,
                                                            int32_t *__restrict__ token, float *__restrict__ logprob,
                                                            int M, int np, int pad_token) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    if (active && !active[m]) {
        if (lane == 0) { token[m] = pad_token; if (logprob) logprob[m] = 0.f; }
        return;
    }
    float pmax = -INFINITY, psum = 0.f, best = -INFINITY, best_z = 0.f;
    int best_col = 0x7fffffff;
    for (int t = lane; t < np; t += 64) {
        const float *p = partials + ((size_t)m * np + t) * kPartialFloats;
        const float om = p[0], os = p[1], ob = p[2], oz = p[4];
        const int oc = __float_as_int(p[3]);
        const float nm = fmaxf(pmax, om);
        const float e1 = (pmax == -INFINITY) ? 0.f : __expf(pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
        psum = psum * e1 + os * e2; pmax = nm;
        if (ob > best || (ob == best && oc < best_col)) { best = ob; best_col = oc; best_z = oz; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float om = __shfl_xor(pmax, o), os = __shfl_xor(psum, o), ob = __shfl_xor(best, o), oz = __shfl_xor(best_z, o);
        const int oc = __shfl_xor(best_col, o);
        const float nm = fmaxf(pmax, om);
        const float e1 = (pmax == -INFINITY) ? 0.f : __expf(pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
        psum = psum * e1 + os * e2; pmax = nm;
        if (ob > best || (ob == best && oc < best_col)) { best = ob; best_col = oc; best_z = oz; }
    }
    if (lane == 0) {
        token[m] = best_col;
        if (logprob) logprob[m] = best_z - (pmax + __logf(psum));
    }
}

// ---- top-k sampling over materialised logits: one workgroup per row.
// Radix select (4 passes of 8 bits over the order-preserving uint key) finds the k-th largest value; tokens with
// logit >= that value are kept (ties kept, as HF's TopKLogitsWarper `scores < kth` removal rule) and sampled with
// the same Philox/Gumbel stream as the fused path.
__device__ __forceinline__ void topk_sample_row(int m, const float *__restrict__ logits, int ld, int vocab, int top_k, float top_p,
                                                const uint8_t *__restrict__ active, int32_t *__restrict__ token,
                                                float *__restrict__ logprob, const SampleParams &sp, int pad_token) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (active && !active[m]) {
        if (tid == 0) { token[m] = pad_token; if (logprob) logprob[m] = 0.f; }
        return;
    }
    const float *row = logits + (size_t)m * ld;
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_prefix, sel_remaining;
    __shared__ float red_f[4][4];
    __shared__ int red_i[4];
    uint32_t prefix = 0, remaining = (uint32_t)(top_k < vocab ? top_k : vocab);
    if (top_k <= 0) remaining = (uint32_t)vocab;
    const bool topk_on = top_k > 0 && top_k < vocab;        // block-uniform: without top-k the threshold below is 0 and the four select passes are skipped
    for (int pass = 0; pass < (topk_on ? 4 : 0); pass++) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (int n = tid; n < vocab; n += 256) {
            const uint32_t key = f32_order_key(row[n]);
            const bool match = pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8));
            if (match) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t rem = remaining, b = 255;
            for (;; b--) {
                if (hist[b] >= rem || b == 0) break;
                rem -= hist[b];
            }
            sel_prefix = prefix | (b << shift);
            sel_remaining = rem;
        }
        __syncthreads();
        prefix = sel_prefix; remaining = sel_remaining;
        __syncthreads();
    }
    uint32_t thr_key = prefix;   // key of the k-th largest logit (0 = keep everything when top_k is off)
    if (top_k <= 0 || top_k >= vocab) thr_key = 0;

    // ---- top-p (nucleus) over what top-k kept, HF TopPLogitsWarper: keep the descending-sorted prefix whose cumulative
    // probability reaches top_p, i.e. token K is kept iff mass(keys > K) < top_p * Z.  A second radix select, this time on
    // probability MASS per bucket; masses are 32.32 fixed-point integers so that the LDS atomics add in any order to the
    // same sums (bit-reproducible).  Tokens tied with the crossing token are all kept.
    if (top_p > 0.f && top_p < 1.f && !sp.greedy) {
        __shared__ unsigned long long mhist[256];
        __shared__ unsigned long long sel_above, sel_total;
        __shared__ float s_rowmax[4];
        float rmax = -INFINITY;
        for (int n = tid; n < vocab; n += 256) {
            const float raw = row[n];
            if (f32_order_key(raw) >= thr_key) rmax = fmaxf(rmax, raw);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o));
        if (lane == 0) s_rowmax[wave] = rmax;
        __syncthreads();
        rmax = fmaxf(fmaxf(s_rowmax[0], s_rowmax[1]), fmaxf(s_rowmax[2], s_rowmax[3]));
        // without top-k the total is formed TILE-WISE (tile_mass_rescaled: what the fused path can compute without the row); a wave per 128-column tile
        __shared__ unsigned long long tile_total;
        if (!topk_on) {
            if (tid == 0) tile_total = 0ull;
            __syncthreads();
            unsigned long long tt = 0ull;
            for (int t0 = wave * 128; t0 < vocab; t0 += 4 * 128) {
                const int na = t0 + lane, nb = na + 64;
                const float ra = na < vocab ? row[na] : -INFINITY, rb = nb < vocab ? row[nb] : -INFINITY;
                float tmax = fmaxf(ra, rb);
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
                unsigned long long sm = 0ull;
                if (tmax != -INFINITY) sm = (na < vocab ? mass_fixed(ra, tmax, sp.inv_temperature) : 0ull) + (nb < vocab ? mass_fixed(rb, tmax, sp.inv_temperature) : 0ull);
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) sm += __shfl_xor(sm, o);
                if (tmax != -INFINITY) tt += tile_mass_rescaled(sm, tmax, rmax, sp.inv_temperature);
            }
            if (lane == 0) atomicAdd(&tile_total, tt);
            __syncthreads();
        }
        uint32_t pprefix = 0;
        unsigned long long above = 0, target = 0;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            mhist[tid] = 0ull;
            __syncthreads();
            for (int n = tid; n < vocab; n += 256) {
                const float raw = row[n];
                const uint32_t key = f32_order_key(raw);
                if (key < thr_key) continue;
                const bool match = pass == 0 || (key >> (shift + 8)) == (pprefix >> (shift + 8));
                if (match) {
                    const float e = __expf((raw - rmax) * sp.inv_temperature);          // in (0, 1]
                    atomicAdd(&mhist[(key >> shift) & 255u], (unsigned long long)((double)e * 4294967296.0));
                }
            }
            __syncthreads();
            if (tid == 0) {
                if (pass == 0) {
                    unsigned long long tot = 0;
                    for (int b = 0; b < 256; b++) tot += mhist[b];
                    if (!topk_on) tot = tile_total;
                    sel_total = (unsigned long long)((double)top_p * (double)tot);
                    if (sel_total == 0) sel_total = 1;
                }
                const unsigned long long tgt = sel_total;
                unsigned long long ab = pass == 0 ? 0ull : sel_above;
                uint32_t b = 255;
                for (;; b--) {
                    if (ab + mhist[b] >= tgt || b == 0) break;      // the crossing token lives in bucket b
                    ab += mhist[b];
                }
                sel_above = ab;
                sel_prefix = pprefix | (b << shift);
            }
            __syncthreads();
            pprefix = sel_prefix; above = sel_above; target = sel_total;
            __syncthreads();
        }
        (void)above; (void)target;
        if (pprefix > thr_key) thr_key = pprefix;
    }
    const uint32_t epoch = sp.epoch ? *sp.epoch : 0u;
    float pmax = -INFINITY, psum = 0.f, best = -INFINITY, best_z = 0.f;
    int best_col = 0x7fffffff;
    for (int n4 = tid * 4; n4 < vocab; n4 += 256 * 4) {
        uint32_t rnd[4] = {0, 0, 0, 0};
        const bool jax = sp.rng == LMRL_RNG_JAX;
        if (!sp.greedy && !jax) philox4x32_10((uint32_t)m, (uint32_t)(n4 >> 2), sp.step, epoch, sp.seed_lo, sp.seed_hi, rnd);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int n = n4 + r;
            if (n >= vocab) continue;
            const float raw = row[n];
            if (f32_order_key(raw) < thr_key) continue;
            const float v = jax ? raw / sp.temperature : raw * sp.inv_temperature;
            const float sc = sp.greedy ? v : v + (jax ? gumbel_jax(sp.seed_hi, sp.seed_lo, (uint32_t)m * (uint32_t)vocab + (uint32_t)n, sp.jax_n)
                                                      : gumbel_from_bits(rnd[r]));
            if (sc > best || (sc == best && n < best_col)) { best = sc; best_col = n; best_z = v; }
            const float nm = fmaxf(pmax, v);
            psum = psum * ((pmax == -INFINITY) ? 0.f : __expf(pmax - nm)) + __expf(v - nm);
            pmax = nm;
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float om = __shfl_xor(pmax, o), os = __shfl_xor(psum, o), ob = __shfl_xor(best, o), oz = __shfl_xor(best_z, o);
        const int oc = __shfl_xor(best_col, o);
        const float nm = fmaxf(pmax, om);
        const float e1 = (pmax == -INFINITY) ? 0.f : __expf(pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
        psum = psum * e1 + os * e2; pmax = nm;
        if (ob > best || (ob == best && oc < best_col)) { best = ob; best_col = oc; best_z = oz; }
    }
    if (lane == 0) { red_f[wave][0] = pmax; red_f[wave][1] = psum; red_f[wave][2] = best; red_f[wave][3] = best_z; red_i[wave] = best_col; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) {
            const float om = red_f[w][0], os = red_f[w][1], ob = red_f[w][2], oz = red_f[w][3];
            const int oc = red_i[w];
            const float nm = fmaxf(pmax, om);
            const float e1 = (pmax == -INFINITY) ? 0.f : __expf(pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
            psum = psum * e1 + os * e2; pmax = nm;
            if (ob > best || (ob == best && oc < best_col)) { best = ob; best_col = oc; best_z = oz; }
        }
        token[m] = best_col;
        if (logprob) logprob[m] = best_z - (pmax + __logf(psum));
    }
}

// row_list (fused top-k: only the rows its reduce kernel flagged): a small grid walks the list
__global__ __launch_bounds__(256) void topk_sample_kernel(const float *__restrict__ logits, int ld, int vocab, int top_k, float top_p,
                                                          const uint8_t *__restrict__ active, int32_t *__restrict__ token,
                                                          float *__restrict__ logprob, SampleParams sp, int pad_token,
                                                          const int32_t *__restrict__ row_list, const int32_t *__restrict__ row_count) {
    if (!row_list) { topk_sample_row((int)blockIdx.x, logits, ld, vocab, top_k, top_p, active, token, logprob, sp, pad_token); return; }
    const int cnt = *row_count;
    for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
        topk_sample_row(row_list[i], logits, ld, vocab, top_k, top_p, active, token, logprob, sp, pad_token);
        __syncthreads();
    }
}

// ---- the same selection with the ROW IN REGISTERS (round 5): topk_sample_kernel above walks its 200 KB row up to ten times (4 radix passes for
// top-k, the row maximum + 4 mass passes for top-p, the sampling pass) with 4-byte loads — 410 us per sampled token at 1024 x 50 257, 15 ms of a
// 62 ms warper episode.  Here one 1024-thread workgroup reads the row ONCE (16-byte loads, NV float4 per thread: chunk c = tid + 1024 j holds columns
// 4 c .. 4 c + 3) and every pass runs on registers.  Same keys, same histograms (integer / 32.32 fixed-point: order independent), same thresholds, same
// Philox counter per 4-column chunk, same arg-max tie rule: the sampled token is bit-identical to the kernel above; the log-probability's
// sum runs in a different order (fp32 rounding).
// Scan of a 256-bucket histogram from the top by ONE wave (lane l owns buckets 4 l .. 4 l + 3): the largest bucket b whose suffix sum
// S(b) = sum_{b' >= b} h[b'] reaches `target` (b = 0 if none does), and `above` = S(b + 1) + base.  What thread 0 of topk_sample_kernel finds with a
// serial loop of up to 256 dependent LDS reads (x 8 - 12 passes per row).  Integer sums: identical results.
template <class T>
__device__ __forceinline__ void wave_scan_from_top(const T *h, T target, T base, int lane, uint32_t &b_out, T &above_out) {
    T hv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hv[k] = h[4 * lane + k];
    const T mine = hv[0] + hv[1] + hv[2] + hv[3];
    T suf = mine;                                            // -> inclusive suffix sum over lanes l' >= lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = __shfl_down(suf, d);
        if (lane + d < 64) suf += o;
    }
    T run = suf - mine + base;                               // S(4 lane + 4) + base: everything above this lane's buckets
    int found = -1;
    T above = run;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        if (found < 0) {
            if (run + hv[k] >= target) { found = 4 * lane + k; above = run; }
            else run += hv[k];
        }
    }
    const unsigned long long bal = __ballot(found >= 0);
    if (bal) {
        const int src = 63 - __builtin_clzll(bal);           // the highest lane that holds a crossing bucket
        b_out = (uint32_t)__shfl(found, src);
        above_out = __shfl(above, src);
    } else {                                                 // nothing reaches the target: bucket 0, everything above it
        b_out = 0u;
        const T s0 = __shfl(run, 0);                         // lane 0 ran through all of its buckets: run = S(0) + base
        const T h00 = __shfl(hv[0], 0);
        above_out = s0 - h00;
    }
}

template <int NV>
__device__ __forceinline__ void topk_sample_reg_row(int m, const float *__restrict__ logits, int ld, int vocab, int top_k, float top_p,
                                                    const uint8_t *__restrict__ active, int32_t *__restrict__ token,
                                                    float *__restrict__ logprob, const SampleParams &sp, int pad_token) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (active && !active[m]) {
        if (tid == 0) { token[m] = pad_token; if (logprob) logprob[m] = 0.f; }
        return;
    }
    const float *row = logits + (size_t)m * ld;
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long mhist[256];
    __shared__ uint32_t sel_prefix, sel_remaining;
    __shared__ unsigned long long sel_above, sel_total;
    __shared__ float red_f[16][4];
    __shared__ int red_i[16];
    // the row as ORDER KEYS (what every selection pass compares; the logit itself is one xor away: key_to_f32) — holding both the values and the
    // keys the passes keep re-deriving is what pushed the kernel over its 128 VGPRs
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t v[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int n4 = 4 * (tid + 1024 * j);
        const f32x4 x = n4 < ld ? *reinterpret_cast<const f32x4 *>(row + n4) : f32x4{0.f, 0.f, 0.f, 0.f};      // (ld a multiple of 4: whole chunks)
        v[j] = u32x4_t{f32_order_key(x[0]), f32_order_key(x[1]), f32_order_key(x[2]), f32_order_key(x[3])};
        asm volatile("" : "+v"(v[j]));                      // the keys are the stored form: do not re-derive them from a second copy
    }
    auto key_to_f32 = [](uint32_t k) { return f32_from_order_key(k); };
    // every element this thread holds: n its column, raw its logit (columns >= vocab skipped)
    // (a scheduling barrier per chunk: without it the 13 - 16 unrolled chunk bodies are interleaved and the kernel spills 135 VGPRs at 128)
#define LMRL_TK_FOREACH(BODY)                                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < NV; j_++) {                                                        \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                     \
            const int n = 4 * (tid + 1024 * j_) + r_;                                                          \
            if (n < vocab) { const uint32_t key = v[j_][r_]; BODY }                                            \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    uint32_t prefix = 0, remaining = (uint32_t)(top_k < vocab ? top_k : vocab);
    if (top_k <= 0) remaining = (uint32_t)vocab;
    const bool topk_on = top_k > 0 && top_k < vocab;
    // Pre-filter (top_k <= 1024): the k-th largest of the 1024 per-thread maxima is a LOWER bound of the row's k-th largest value (those are k
    // distinct elements at or above it), so only elements at or above it can be selected.  The four radix passes then histogram a few multiples of
    // k candidates instead of all 50 k elements — whose top byte falls into three or four buckets, i.e. 50 k LDS atomics on the same few addresses
    // (the first pass alone cost ~20 us per workgroup).  The threshold found is the same: every element >= the true k-th largest is a candidate.
    uint32_t floor_key = 0;
    if (topk_on && top_k <= 1024) {
        uint32_t tkey = 0;                                   // this thread's largest key (0: it holds no valid column)
        LMRL_TK_FOREACH({ tkey = key > tkey ? key : tkey; })
        uint32_t fp = 0, frem = (uint32_t)top_k;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            if (pass == 0 || (tkey >> (shift + 8)) == (fp >> (shift + 8))) atomicAdd(&hist[(tkey >> shift) & 255u], 1u);
            __syncthreads();
            if (wave == 0) {
                uint32_t b, above;
                wave_scan_from_top<uint32_t>(hist, frem, 0u, lane, b, above);
                if (lane == 0) { sel_prefix = fp | (b << shift); sel_remaining = frem - above; }
            }
            __syncthreads();
            fp = sel_prefix; frem = sel_remaining;
            __syncthreads();
        }
        floor_key = fp;
    }
    for (int pass = 0; pass < (topk_on ? 4 : 0); pass++) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        LMRL_TK_FOREACH({
            const bool match = key >= floor_key && (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8)));
            if (match) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        })
        __syncthreads();
        if (wave == 0) {
            uint32_t b, above;
            wave_scan_from_top<uint32_t>(hist, remaining, 0u, lane, b, above);
            if (lane == 0) { sel_prefix = prefix | (b << shift); sel_remaining = remaining - above; }
        }
        __syncthreads();
        prefix = sel_prefix; remaining = sel_remaining;
        __syncthreads();
    }
    uint32_t thr_key = prefix;
    if (top_k <= 0 || top_k >= vocab) thr_key = 0;
    if (top_p > 0.f && top_p < 1.f && !sp.greedy) {
        float rmax = -INFINITY;
        LMRL_TK_FOREACH({ if (key >= thr_key) rmax = fmaxf(rmax, key_to_f32(key)); })
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o));
        if (lane == 0) red_f[wave][0] = rmax;
        __syncthreads();
        rmax = red_f[0][0];
#pragma unroll
        for (int w = 1; w < 16; w++) rmax = fmaxf(rmax, red_f[w][0]);
        __syncthreads();
        // without top-k: the tile-wise total (tile_mass_rescaled) — chunk c = tid + 1024 j holds 4 columns, so a 128-column tile is the 32 lanes of a half-wave
        __shared__ unsigned long long tile_total;
        if (!topk_on) {
            if (tid == 0) tile_total = 0ull;
            __syncthreads();
            unsigned long long tt = 0ull;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const int n4 = 4 * (tid + 1024 * j);
                uint32_t tk = 0u;
#pragma unroll
                for (int r = 0; r < 4; r++) tk = (n4 + r < vocab && v[j][r] > tk) ? v[j][r] : tk;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { const uint32_t ok = (uint32_t)__shfl_xor((int)tk, o); tk = ok > tk ? ok : tk; }
                const float tmax = key_to_f32(tk);
                const bool live = tk != 0u && tmax != -INFINITY;
                unsigned long long sm = 0ull;
                if (live) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (n4 + r < vocab) sm += mass_fixed(key_to_f32(v[j][r]), tmax, sp.inv_temperature);
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) sm += __shfl_xor(sm, o);
                if (live && (lane & 31) == 0) tt += tile_mass_rescaled(sm, tmax, rmax, sp.inv_temperature);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) tt += __shfl_xor(tt, o);
            if (lane == 0) atomicAdd(&tile_total, tt);
            __syncthreads();
        }
        uint32_t pprefix = 0;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) mhist[tid] = 0ull;
            __syncthreads();
            LMRL_TK_FOREACH({
                if (key >= thr_key) {
                    const bool match = pass == 0 || (key >> (shift + 8)) == (pprefix >> (shift + 8));
                    if (match) {
                        const float e = __expf((key_to_f32(key) - rmax) * sp.inv_temperature);
                        atomicAdd(&mhist[(key >> shift) & 255u], (unsigned long long)((double)e * 4294967296.0));
                    }
                }
            })
            __syncthreads();
            if (wave == 0) {
                unsigned long long tgt = sel_total;
                if (pass == 0) {
                    unsigned long long tot = mhist[4 * lane] + mhist[4 * lane + 1] + mhist[4 * lane + 2] + mhist[4 * lane + 3];
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
                    if (!topk_on) tot = tile_total;
                    tgt = (unsigned long long)((double)top_p * (double)tot);
                    if (tgt == 0) tgt = 1;
                }
                uint32_t b;
                unsigned long long ab;
                wave_scan_from_top<unsigned long long>(mhist, tgt, pass == 0 ? 0ull : sel_above, lane, b, ab);      // the crossing token lives in bucket b
                if (lane == 0) { sel_total = tgt; sel_above = ab; sel_prefix = pprefix | (b << shift); }
            }
            __syncthreads();
            pprefix = sel_prefix;
            __syncthreads();
        }
        if (pprefix > thr_key) thr_key = pprefix;
    }
    const uint32_t epoch = sp.epoch ? *sp.epoch : 0u;
    float pmax = -INFINITY, psum = 0.f, best = -INFINITY, best_z = 0.f;
    int best_col = 0x7fffffff;
    const bool jax = sp.rng == LMRL_RNG_JAX;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int c4 = tid + 1024 * j, n4 = 4 * c4;
        if (n4 >= vocab) continue;
        // (with top-k on, a handful of the row's 12.5 k chunks hold a kept column: the others need neither noise nor a softmax term)
        if (v[j][0] < thr_key && v[j][1] < thr_key && v[j][2] < thr_key && v[j][3] < thr_key) continue;
        uint32_t rnd[4] = {0, 0, 0, 0};
        if (!sp.greedy && !jax) philox4x32_10((uint32_t)m, (uint32_t)c4, sp.step, epoch, sp.seed_lo, sp.seed_hi, rnd);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int n = n4 + r;
            if (n >= vocab) continue;
            if (v[j][r] < thr_key) continue;
            const float raw = key_to_f32(v[j][r]);
            const float vv = jax ? raw / sp.temperature : raw * sp.inv_temperature;
            const float sc = sp.greedy ? vv : vv + (jax ? gumbel_jax(sp.seed_hi, sp.seed_lo, (uint32_t)m * (uint32_t)vocab + (uint32_t)n, sp.jax_n)
                                                        : gumbel_from_bits(rnd[r]));
            if (sc > best || (sc == best && n < best_col)) { best = sc; best_col = n; best_z = vv; }
            const float nm = fmaxf(pmax, vv);
            psum = psum * ((pmax == -INFINITY) ? 0.f : __expf(pmax - nm)) + __expf(vv - nm);
            pmax = nm;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef LMRL_TK_FOREACH
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float om = __shfl_xor(pmax, o), os = __shfl_xor(psum, o), ob = __shfl_xor(best, o), oz = __shfl_xor(best_z, o);
        const int oc = __shfl_xor(best_col, o);
        const float nm = fmaxf(pmax, om);
        const float e1 = (pmax == -INFINITY) ? 0.f : __expf(pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
        psum = psum * e1 + os * e2; pmax = nm;
        if (ob > best || (ob == best && oc < best_col)) { best = ob; best_col = oc; best_z = oz; }
    }
    if (lane == 0) { red_f[wave][0] = pmax; red_f[wave][1] = psum; red_f[wave][2] = best; red_f[wave][3] = best_z; red_i[wave] = best_col; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) {
            const float om = red_f[w][0], os = red_f[w][1], ob = red_f[w][2], oz = red_f[w][3];
            const int oc = red_i[w];
            const float nm = fmaxf(pmax, om);
            const float e1 = (pmax == -INFINITY) ? 0.f : __expf(pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
            psum = psum * e1 + os * e2; pmax = nm;
            if (ob > best || (ob == best && oc < best_col)) { best = ob; best_col = oc; best_z = oz; }
        }
        token[m] = best_col;
        if (logprob) logprob[m] = best_z - (pmax + __logf(psum));
    }
}

template <int NV>
__global__ __launch_bounds__(1024) void topk_sample_reg_kernel(const float *__restrict__ logits, int ld, int vocab, int top_k, float top_p,
                                                               const uint8_t *__restrict__ active, int32_t *__restrict__ token,
                                                               float *__restrict__ logprob, SampleParams sp, int pad_token,
                                                               const int32_t *__restrict__ row_list, const int32_t *__restrict__ row_count) {
    if (!row_list) { topk_sample_reg_row<NV>((int)blockIdx.x, logits, ld, vocab, top_k, top_p, active, token, logprob, sp, pad_token); return; }
    const int cnt = *row_count;
    for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
        topk_sample_reg_row<NV>(row_list[i], logits, ld, vocab, top_k, top_p, active, token, logprob, sp, pad_token);
        __syncthreads();
    }
}

// ---- fused top-k, second half: the row's k-th largest logit among the 8 x tiles candidates of lm_topc_epilogue, the exactness check, (top-p,) the draw.
// ONE WAVE per row (four rows per workgroup, lane l holds the records of tiles l, l + 64, ...: NT of them), so that the eight to twelve histogram passes
// need no workgroup barrier — a wave's LDS operations execute in order; each pass is a few predicated LDS atomics and one 64-lane suffix scan.  (The
// 256-threads-per-row form of the same passes took 20 us per launch against 5 us for the plain sampler's merge kernel.)  Same keys, same radix selects,
// same 32.32 fixed-point mass histograms, same Philox word per column and same arg-max tie rule as topk_sample_kernel on the materialised row: whenever
// the check passes the kept set is the row's true top-k (ties kept) and the sampled token is identical.  A row that fails the check is appended to the
// list in `fb` ({count, -, ..., block flags [kFbMaxBlocks], rows [M]}) and left to the materialised path.
template <int NT, bool NUC>
__global__ __launch_bounds__(256) void topc_reduce_sample_kernel(const uint32_t *__restrict__ cand, int M, int tiles_n, int vocab, int top_k, float top_p,
                                                                 const uint8_t *__restrict__ active, int32_t *__restrict__ token,
                                                                 float *__restrict__ logprob, SampleParams sp, int pad_token, int32_t *__restrict__ fb) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;
    if (active && !active[m]) {
        if (lane == 0) { token[m] = pad_token; if (logprob) logprob[m] = 0.f; }
        return;
    }
    __shared__ uint32_t hist_s[4][256];
    __shared__ unsigned long long mhist_s[4][256];
    __shared__ uint32_t ck_s[4][kTopcCompactCap], cn_s[4][kTopcCompactCap];
    uint32_t *hist = hist_s[wave], *ck = ck_s[wave], *cn = cn_s[wave];
    unsigned long long *mhist = mhist_s[wave];
#define LMRL_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define LMRL_HAND_BACK() do { if (lane == 0) { const int idx_ = atomicAdd(&fb[0], 1); fb[kFbHeader + kFbMaxBlocks + idx_] = m; fb[kFbHeader + m / kLmBM] = 1; } return; } while (0)
    uint32_t key[NT][kTopC], colw[NT][2];
#pragma unroll
    for (int q = 0; q < NT; q++) {
        const int t = lane + 64 * q;
        if (t < tiles_n) {
            const uint4 *src = reinterpret_cast<const uint4 *>(cand + ((size_t)m * tiles_n + t) * kCandWords);  // 64-byte records {key, column} x 8
            const uint4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
            key[q][0] = r0.x; key[q][1] = r0.z; key[q][2] = r1.x; key[q][3] = r1.z; key[q][4] = r2.x; key[q][5] = r2.z; key[q][6] = r3.x; key[q][7] = r3.z;
            colw[q][0] = (r0.y & 255u) | ((r0.w & 255u) << 8) | ((r1.y & 255u) << 16) | ((r1.w & 255u) << 24);
            colw[q][1] = (r2.y & 255u) | ((r2.w & 255u) << 8) | ((r3.y & 255u) << 16) | ((r3.w & 255u) << 24);
        } else {
#pragma unroll
            for (int e = 0; e < kTopC; e++) key[q][e] = 0u;
            colw[q][0] = colw[q][1] = 0u;
        }
    }
    // every candidate this lane holds: k_ its key (0 = empty slot), n its column
#define LMRL_TC_FOREACH(BODY)                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < NT; q_++) {                                                        \
        _Pragma("unroll") for (int e_ = 0; e_ < kTopC; e_++) {                                                 \
            const uint32_t k_ = key[q_][e_];                                                                   \
            if (k_ != 0u) { const int n = (lane + 64 * q_) * kLmBN + (int)((colw[q_][e_ >> 2] >> (8 * (e_ & 3))) & 255u); (void)n; BODY } \
        }                                                                                                      \
    }
    // per tile: its largest key, and — what a FULL record hides — its smallest one: everything the tile did not report lies at or below that
    uint32_t tmaxk[NT], hidden_max = 0, tkey = 0;
#pragma unroll
    for (int q = 0; q < NT; q++) {
        uint32_t mn = key[q][0], mx = key[q][0];
#pragma unroll
        for (int e = 1; e < kTopC; e++) { mn = key[q][e] < mn ? key[q][e] : mn; mx = key[q][e] > mx ? key[q][e] : mx; }
        hidden_max = mn > hidden_max ? mn : hidden_max;        // (a record with an empty slot hides nothing: mn = 0)
        tmaxk[q] = mx;
        tkey = mx > tkey ? mx : tkey;
    }
    uint32_t floor_key = 0;
    {
        // top-k threshold: radix select over the candidates, pre-filtered by a lower bound of the k-th largest candidate: the k-th largest of k or more
        // DISTINCT candidates — the per-lane maxima (k <= 64 lanes), else the per-tile maxima (k <= kTopCMaxK; fewer tiles than k: the bound degrades to 0).
        // Top-p alone: the nucleus of a peaked row is a few dozen tokens — the candidates at or above the 128th largest tile maximum (a few hundred)
        // are searched; a nucleus that needs more is handed back like one that reaches a hidden bound
        const int k_pre = NUC ? 128 : top_k;
        uint32_t fp = 0, frem = (uint32_t)k_pre;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
#pragma unroll
            for (int i = 0; i < 4; i++) hist[4 * lane + i] = 0;
            LMRL_WAVE_SYNC();
            if (k_pre <= 64) {
                if (tkey != 0u && (pass == 0 || (tkey >> (shift + 8)) == (fp >> (shift + 8)))) atomicAdd(&hist[(tkey >> shift) & 255u], 1u);
            } else {
#pragma unroll
                for (int q = 0; q < NT; q++) {
                    const uint32_t t_ = tmaxk[q];
                    if (t_ != 0u && (pass == 0 || (t_ >> (shift + 8)) == (fp >> (shift + 8)))) atomicAdd(&hist[(t_ >> shift) & 255u], 1u);
                }
            }
            LMRL_WAVE_SYNC();
            uint32_t b, above;
            wave_scan_from_top<uint32_t>(hist, frem, 0u, lane, b, above);
            fp |= b << shift; frem -= above;
            LMRL_WAVE_SYNC();
        }
        floor_key = fp;
    }
    if (NUC) {
        // top-p alone: only candidates ABOVE every tile's hidden bound can belong to an exactly known nucleus
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t oh = (uint32_t)__shfl_xor((int)hidden_max, o); hidden_max = oh > hidden_max ? oh : hidden_max; }
        floor_key = floor_key > hidden_max + 1u ? floor_key : hidden_max + 1u;
    }
    // the candidates at or above the floor (a few multiples of k out of 8 x tiles), compacted into LDS: every later pass walks ceil(count / 64) of them per
    // lane instead of 8 x NT predicated slots (most of which hold a candidate in SOME lane, so none could be skipped)
    uint32_t mine_n = 0;
    LMRL_TC_FOREACH({ mine_n += k_ >= floor_key ? 1u : 0u; })
    uint32_t incl = mine_n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    {
        uint32_t at = incl - mine_n;
        if (total <= (uint32_t)kTopcCompactCap)
            LMRL_TC_FOREACH({ if (k_ >= floor_key) { ck[at] = k_; cn[at] = (uint32_t)n; at++; } })
    }
#undef LMRL_TC_FOREACH
    LMRL_WAVE_SYNC();
    uint32_t thr_key = floor_key;
    if (!NUC) {
        uint32_t prefix = 0, remaining = (uint32_t)top_k;
        if (total <= (uint32_t)kTopcCompactCap) {
            for (int pass = 0; pass < 4; pass++) {
                const int shift = 24 - 8 * pass;
#pragma unroll
                for (int i = 0; i < 4; i++) hist[4 * lane + i] = 0;
                LMRL_WAVE_SYNC();
                for (uint32_t i = lane; i < total; i += 64) {
                    const uint32_t k_ = ck[i];
                    if (pass == 0 || (k_ >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(k_ >> shift) & 255u], 1u);
                }
                LMRL_WAVE_SYNC();
                uint32_t b, above;
                wave_scan_from_top<uint32_t>(hist, remaining, 0u, lane, b, above);
                prefix |= b << shift; remaining -= above;
                LMRL_WAVE_SYNC();
            }
        }
        thr_key = prefix;                                // key of the k-th largest candidate
        // exactness: a full record whose smallest key reaches the threshold may hide a logit at or above it (or: too many candidates to compact —
        // degenerate rows, e.g. all logits equal)
        if (__ballot(hidden_max >= thr_key && hidden_max != 0u) != 0ull || total > (uint32_t)kTopcCompactCap) LMRL_HAND_BACK();
    } else if (total > (uint32_t)kTopcCompactCap || total == 0u) LMRL_HAND_BACK();
    if (NUC || (top_p > 0.f && top_p < 1.f)) {
        float rmax = -INFINITY;
        unsigned long long tgt = 0, sabove = 0;
        if (NUC) {
            // the row maximum is a candidate of its tile; the total: every tile's mass, rescaled from the tile's maximum to the row's (tile_mass_rescaled —
            // the same integers the materialised kernels form from the row itself)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t ot = (uint32_t)__shfl_xor((int)tkey, o); tkey = ot > tkey ? ot : tkey; }
            rmax = f32_from_order_key(tkey);
            unsigned long long tot = 0ull;
#pragma unroll
            for (int q = 0; q < NT; q++) {
                const int t = lane + 64 * q;
                if (t < tiles_n && tmaxk[q] != 0u) {
                    const float tm = f32_from_order_key(tmaxk[q]);
                    if (tm != -INFINITY) tot += tile_mass_rescaled(sp.tile_mass[(size_t)m * tiles_n + t], tm, rmax, sp.inv_temperature);
                }
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
            tgt = (unsigned long long)((double)top_p * (double)tot);
            if (tgt == 0) tgt = 1;
        } else {
            for (uint32_t i = lane; i < total; i += 64) { const uint32_t k_ = ck[i]; if (k_ >= thr_key) rmax = fmaxf(rmax, f32_from_order_key(k_)); }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o));
        }
        uint32_t pprefix = 0;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
#pragma unroll
            for (int i = 0; i < 4; i++) mhist[4 * lane + i] = 0ull;
            LMRL_WAVE_SYNC();
            for (uint32_t i = lane; i < total; i += 64) {
                const uint32_t k_ = ck[i];
                if (k_ >= thr_key && (pass == 0 || (k_ >> (shift + 8)) == (pprefix >> (shift + 8))))
                    atomicAdd(&mhist[(k_ >> shift) & 255u], mass_fixed(f32_from_order_key(k_), rmax, sp.inv_temperature));
            }
            LMRL_WAVE_SYNC();
            if (pass == 0 && !NUC) {
                unsigned long long tot = mhist[4 * lane] + mhist[4 * lane + 1] + mhist[4 * lane + 2] + mhist[4 * lane + 3];
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
                tgt = (unsigned long long)((double)top_p * (double)tot);
                if (tgt == 0) tgt = 1;
            }
            uint32_t b;
            unsigned long long ab;
            wave_scan_from_top<unsigned long long>(mhist, tgt, pass == 0 ? 0ull : sabove, lane, b, ab);
            sabove = ab; pprefix |= b << shift;
            LMRL_WAVE_SYNC();
        }
        // top-p alone: the crossing key is the row's true one iff it lies above every hidden bound (then every key at or above it is a candidate and
        // the masses summed down to it are complete); a nucleus that reaches into what the tiles did not report goes back to the materialised path
        if (NUC && pprefix < floor_key) LMRL_HAND_BACK();
        if (pprefix > thr_key) thr_key = pprefix;
    }
#undef LMRL_WAVE_SYNC
#undef LMRL_HAND_BACK
    const uint32_t epoch = sp.epoch ? *sp.epoch : 0u;
    const bool jax = sp.rng == LMRL_RNG_JAX;
    float pmax = -INFINITY, psum = 0.f, best = -INFINITY, best_z = 0.f;
    int best_col = 0x7fffffff;
    for (uint32_t i = lane; i < total; i += 64) {
        const uint32_t k_ = ck[i];
        if (k_ >= thr_key) {
            const int n = (int)cn[i];
            const float raw = f32_from_order_key(k_);
            float vv, sc;
            if (jax) {          // jax.random.categorical on logits / T: one key for the whole [M, vocab] noise array (word index = row * vocab + column)
                vv = raw / sp.temperature;
                sc = vv + gumbel_jax(sp.seed_hi, sp.seed_lo, (uint32_t)m * (uint32_t)vocab + (uint32_t)n, sp.jax_n);
            } else {
                uint32_t rnd[4];
                philox4x32_10((uint32_t)m, (uint32_t)(n >> 2), sp.step, epoch, sp.seed_lo, sp.seed_hi, rnd);
                vv = raw * sp.inv_temperature;
                const uint32_t rw = (n & 3) == 0 ? rnd[0] : ((n & 3) == 1 ? rnd[1] : ((n & 3) == 2 ? rnd[2] : rnd[3]));
                sc = vv + gumbel_from_bits(rw);
            }
            if (sc > best || (sc == best && n < best_col)) { best = sc; best_col = n; best_z = vv; }
            const float nm = fmaxf(pmax, vv);
            psum = psum * ((pmax == -INFINITY) ? 0.f : __expf(pmax - nm)) + __expf(vv - nm);
            pmax = nm;
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float om = __shfl_xor(pmax, o), os = __shfl_xor(psum, o), ob = __shfl_xor(best, o), oz = __shfl_xor(best_z, o);
        const int oc = __shfl_xor(best_col, o);
        const float nm = fmaxf(pmax, om);
        const float e1 = (pmax == -INFINITY) ? 0.f : __expf(pmax - nm), e2 = (om == -INFINITY) ? 0.f : __expf(om - nm);
        psum = psum * e1 + os * e2; pmax = nm;
        if (ob > best || (ob == best && oc < best_col)) { best = ob; best_col = oc; best_z = oz; }
    }
    if (lane == 0) {
        token[m] = best_col;
        if (logprob) logprob[m] = best_z - (pmax + __logf(psum));
    }
}

int g_sampler_variant = 0;      // tools / tests only: 1 = materialised logits + the round-2 strided-row warper kernel, 2 = materialised logits + the register-row kernel (no fused top-k)

// top-k / top-p sampling from materialised logits: the register-row kernel where the row fits (vocab <= 16 float4 x 1024 threads, 16-byte aligned rows)
static void launch_topk_sample(const float *logits_d, int ld, int m, int vocab, int top_k, float top_p, const uint8_t *active_d, int32_t *token_d,
                               float *logprob_d, const SampleParams &sp, int pad_token, hipStream_t s, const int32_t *row_list = nullptr,
                               const int32_t *row_count = nullptr) {
    const int grid = row_list ? (m < 32 ? m : 32) : m;
    const bool reg_ok = g_sampler_variant != 1 && ld % 4 == 0 && (uintptr_t)logits_d % 16 == 0 && vocab <= 16 * 4096 && vocab > 4096;
    if (reg_ok) {
        const int nv = (ld / 4 + 1023) / 1024;      // float4 chunks per thread
#define LMRL_TK_LAUNCH(NV_) hipLaunchKernelGGL(topk_sample_reg_kernel<NV_>, dim3(grid), dim3(1024), 0, s, logits_d, ld, vocab, top_k, top_p, active_d, token_d, logprob_d, sp, pad_token, row_list, row_count)
        if (nv <= 4) LMRL_TK_LAUNCH(4);
        else if (nv <= 8) LMRL_TK_LAUNCH(8);
        else if (nv <= 13) LMRL_TK_LAUNCH(13);
        else if (nv <= 16) LMRL_TK_LAUNCH(16);
        else hipLaunchKernelGGL(topk_sample_kernel, dim3(grid), dim3(256), 0, s, logits_d, ld, vocab, top_k, top_p, active_d, token_d, logprob_d, sp, pad_token, row_list, row_count);
#undef LMRL_TK_LAUNCH
        return;
    }
    hipLaunchKernelGGL(topk_sample_kernel, dim3(grid), dim3(256), 0, s, logits_d, ld, vocab, top_k, top_p, active_d, token_d, logprob_d, sp, pad_token, row_list, row_count);
}

// ---- generic generation bookkeeping (any tokenizer / env): append the sampled token of every live sequence, stop a sequence
// at eos or at `cap` tokens, and emit the next decode step's inputs — so a whole `generate` runs without a host sync.
__global__ void gen_accept_kernel(const int32_t *__restrict__ sampled, uint8_t *__restrict__ active, int32_t *__restrict__ out_tokens,
                                  int32_t *__restrict__ out_len, int32_t *__restrict__ next_tok, int32_t *__restrict__ next_cnt, int eos,
                                  int cap, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int cnt = 0;
    if (active[e]) {
        const int tok = sampled[e];
        const int l = out_len[e];
        out_tokens[(size_t)e * cap + l] = tok;
        out_len[e] = l + 1;
        if ((eos >= 0 && tok == eos) || l + 1 >= cap) active[e] = 0;
        else { cnt = 1; next_tok[e] = tok; }
    }
    next_cnt[e] = cnt;
}

// synthetic-workload hook on MATERIALISED logits (the fused path adds it in its epilogue): logits[row][steer_tok[row]] += strength
__global__ void steer_add_kernel(float *__restrict__ logits, int ld, const int32_t *__restrict__ steer_tok, float strength, int vocab, int m) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int st = steer_tok[r];
    if (st >= 0 && st < vocab) logits[(size_t)r * ld + st] += strength;
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {
void lmrl_sampler_set_variant(int v) { g_sampler_variant = v; }

int lmrl_gen_accept(const int32_t *sampled_d, uint8_t *active_d, int32_t *out_tokens_d, int32_t *out_len_d, int32_t *next_tok_d,
                    int32_t *next_cnt_d, int eos_token, int cap, int n, void *stream) {
    LMRL_REQUIRE(sampled_d && active_d && out_tokens_d && out_len_d && next_tok_d && next_cnt_d && cap > 0 && n > 0, "lmrl_gen_accept: bad argument");
    hipLaunchKernelGGL(gen_accept_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), sampled_d, active_d, out_tokens_d, out_len_d,
                       next_tok_d, next_cnt_d, eos_token, cap, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

// per-(row, tile) partials of the plain sampler (6 floats) or candidate records of the fused top-k path (kCandWords = 16 words: 8 x {order key, tile-local
// column}), then that path's flagged-row list: a header of kFbHeader words ([0] = rows handed back), kFbMaxBlocks block flags, then the row list
size_t lmrl_sample_fb_offset(int m, int vocab_padded) {      // byte offset of the flagged-row header inside the workspace
    return (size_t)m * (size_t)(vocab_padded / kLmBN) * kCandWords * sizeof(uint32_t);
}
static size_t sample_tile_mass_offset(int m, int vocab_padded) {      // after the flagged-row list, 8-byte aligned: [m][tiles] tile masses of the fused top-p path
    const size_t o = lmrl_sample_fb_offset(m, vocab_padded) + (size_t)(kFbHeader + kFbMaxBlocks + m) * sizeof(int32_t);
    return (o + 7) & ~(size_t)7;
}
size_t lmrl_sample_ws_bytes(int m, int vocab_padded) {
    static_assert(kCandWords >= kPartialFloats, "the candidate records are the larger form");
    return sample_tile_mass_offset(m, vocab_padded) + (size_t)m * (size_t)(vocab_padded / kLmBN) * sizeof(unsigned long long);
}

int lmrl_lm_head_sample(const void *hidden_d, const void *wte_d, const void *q_hidden1_d, const void *q_w1_d,
                        const float *q_b1_d, const void *q_hidden2_d, const void *q_w2_d, const float *q_b2_d, int m,
                        int d_model, int vocab, int vocab_padded, const lmrl_sample_params *p, const int32_t *steer_tok_d,
                        const uint8_t *active_d, int32_t *token_d, float *logprob_d, float *logits_out_d, void *ws_d,
                        void *stream) {
    LMRL_REQUIRE(hidden_d && wte_d && p && token_d && ws_d && m > 0, "lmrl_lm_head_sample: null pointer");
    LMRL_REQUIRE(vocab_padded % 128 == 0 && d_model % 64 == 0 && vocab <= vocab_padded, "lmrl_lm_head_sample: bad shape");
    SampleParams sp;
    sp.greedy = (p->temperature <= 0.f) ? 1 : 0;
    sp.inv_temperature = sp.greedy ? 1.f : 1.f / p->temperature;
    sp.seed_lo = (uint32_t)p->seed; sp.seed_hi = (uint32_t)(p->seed >> 32); sp.step = p->step; sp.epoch = p->epoch_d;
    sp.steer_strength = p->steer_strength; sp.beta = p->beta; sp.vocab = vocab;
    sp.temperature = sp.greedy ? 1.f : p->temperature; sp.rng = p->rng; sp.tile_mass = nullptr;
    LMRL_REQUIRE(p->rng == LMRL_RNG_PHILOX || p->rng == LMRL_RNG_JAX, "lmrl_lm_head_sample: unknown rng mode");
    LMRL_REQUIRE(p->rng != LMRL_RNG_JAX || (double)m * vocab < 4294967296.0, "lmrl_lm_head_sample: LMRL_RNG_JAX needs rows * vocab < 2^32 (uint32 iota)");
    sp.jax_n = (uint32_t)m * (uint32_t)vocab;
    hipStream_t s = as_stream(stream);
    const XcdMap xm = make_xcd_map((m + kLmBM - 1) / kLmBM, vocab_padded / kLmBN, 2.0 * m * d_model, 2.0 * (double)vocab_padded * d_model);
    const int tiles = xcd_grid(xm);
    const int nops = (q_hidden1_d && q_w1_d) ? ((q_hidden2_d && q_w2_d) ? 3 : 2) : 1;
    const size_t shmem = (size_t)(nops > 1 ? kLmQStages : kLmStages) * (kLmBM + kLmBN) * 128;
    if (nops > 1) {      // 128 KB of dynamic LDS: a per-device opt-in of every Q-head instantiation, set on the first such launch there (before any graph capture)
        static std::atomic<unsigned long long> q_attr_set{0ull};
        int dev_id = 0;
        LMRL_CHECK_HIP(hipGetDevice(&dev_id));
        const unsigned long long dev_bit = 1ull << (dev_id & 63);
        if (dev_id >= 64 || !(q_attr_set.load(std::memory_order_acquire) & dev_bit)) {
#define LMRL_Q_OPTIN(...) LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lm_head_sample_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
            LMRL_Q_OPTIN(2, true, false); LMRL_Q_OPTIN(2, false, false); LMRL_Q_OPTIN(2, true, true); LMRL_Q_OPTIN(2, false, false, 1); LMRL_Q_OPTIN(2, false, false, 2); LMRL_Q_OPTIN(2, false, false, 0, true);
            LMRL_Q_OPTIN(3, true, false); LMRL_Q_OPTIN(3, false, false); LMRL_Q_OPTIN(3, true, true); LMRL_Q_OPTIN(3, false, false, 1); LMRL_Q_OPTIN(3, false, false, 2); LMRL_Q_OPTIN(3, false, false, 0, true);
#undef LMRL_Q_OPTIN
            q_attr_set.fetch_or(dev_bit, std::memory_order_release);
        }
    }
    float *partials = (float *)ws_d;
    const int tiles_n = vocab_padded / kLmBN, tiles_m = (m + kLmBM - 1) / kLmBM;
    const bool nucleus = p->top_p > 0.f && p->top_p < 1.f && !sp.greedy;
    const bool topk_on = p->top_k > 0 && p->top_k < vocab;
    const bool nuc_only = nucleus && !topk_on;
    bool topc = false;
    int32_t *fb = nullptr;
    const uint16_t *A0 = (const uint16_t *)hidden_d, *W0 = (const uint16_t *)wte_d;
    const uint16_t *A1 = (const uint16_t *)q_hidden1_d, *W1 = (const uint16_t *)q_w1_d;
    const uint16_t *A2 = (const uint16_t *)q_hidden2_d, *W2 = (const uint16_t *)q_w2_d;
    {
    ProfScope ps(PROF_LM_HEAD_SAMPLE, s, 2.0 * (double)m * (double)vocab_padded * (double)d_model * nops);
#define LMRL_LM_LAUNCH(NOPS_, LP_, JAX_)                                                                                                    \
    hipLaunchKernelGGL((lm_head_sample_kernel<NOPS_, LP_, JAX_>), dim3(tiles), dim3(kLmWM * kLmWN * 64), shmem, s, A0, W0, A1, W1, q_b1_d, A2, W2, \
                       q_b2_d, steer_tok_d, partials, logits_out_d, m, vocab_padded, d_model, vocab_padded, sp, xm, (const int32_t *)nullptr, (int32_t *)nullptr)
    const bool lp = logprob_d != nullptr;      // the log-sum-exp (one exp per logit) is computed only when the log-prob is wanted
    // fused warpers (both random streams; the policy-only head and — round 6 — the ILQL value policy's pi_beta + beta min(q1, q2), generation.py:97-119):
    // candidate epilogue + reduce, no logits in HBM, for 0 < top_k <= 256 (with or without top-p) and for top-p ALONE (per-tile probability masses beside
    // the candidates: right for the peaked distributions of a trained policy — a nucleus that reaches past a tile's 8 candidates goes back);
    // `logits_out_d` stays the scratch of the rows the exactness checks hand back to the materialised path.  LMRL_SAMPLE_WANT_LOGITS in `flags` (the
    // caller reads logits_out_d afterwards) or g_sampler_variant 1 / 2 (tools, tests): always materialise.
    topc = !sp.greedy && ((topk_on && p->top_k <= kTopCMaxK && (p->top_k <= 64 || p->top_k <= tiles_n)) || nuc_only) && tiles_n <= 512 &&      // (k > 64: pre-filter by tile maxima, needs k tiles)
           tiles_m <= kFbMaxBlocks && logits_out_d && g_sampler_variant == 0 && !(p->flags & LMRL_SAMPLE_WANT_LOGITS);
    fb = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(ws_d) + lmrl_sample_fb_offset(m, vocab_padded));
    if (topc && nuc_only) sp.tile_mass = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ws_d) + sample_tile_mass_offset(m, vocab_padded));
    // policy-only sampling on the Philox / greedy path (and the candidate form, which draws no noise): the persistent kernel (ring running ahead across
    // tiles); needs K / 64 even and enough tiles to give every one of the 512 resident workgroups at least two.  g_gemm_variant 301 (tools) forces the
    // one-tile-per-workgroup kernel for the A/B.
    const bool persist = nops == 1 && (topc || !(sp.rng == LMRL_RNG_JAX && !sp.greedy)) && (d_model / 64) % 2 == 0 && d_model >= 128 && tiles >= 1024 &&
                         g_gemm_variant != 301;
    if (persist) {
        const int grid = 512;                                          // 2 workgroups per CU x 256 CUs (a multiple of 8: XCD affinity preserved)
        const size_t shp_plain = shmem + (size_t)kLmWM * (kLmBM / kLmWM) * (kLmWN - 1) * 5 * sizeof(float);
        const size_t shp = topc ? shmem + (size_t)kTopcLdsBytes : shp_plain, shp_max = shmem + (size_t)kTopcLdsBytes;
        static_assert((size_t)kTopcLdsBytes >= (size_t)kLmWM * (kLmBM / kLmWM) * (kLmWN - 1) * 5 * sizeof(float), "the opt-in below is sized by the candidate form");
        // the dynamic-LDS opt-in is a PER-DEVICE function attribute: one bit per device ordinal, set on the first launch there (the eager warm-up
        // episode, i.e. before any graph capture on that device); atomic: ranks / lanes may arrive from several host threads
        static std::atomic<unsigned long long> attr_set{0ull};
        int dev_id = 0;
        LMRL_CHECK_HIP(hipGetDevice(&dev_id));
        const unsigned long long dev_bit = 1ull << (dev_id & 63);
        if (dev_id >= 64 || !(attr_set.load(std::memory_order_acquire) & dev_bit)) {
            LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lm_head_sample_persist_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp_max));
            LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lm_head_sample_persist_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp_max));
            LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lm_head_sample_persist_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp_max));
            LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lm_head_sample_persist_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp_max));
            attr_set.fetch_or(dev_bit, std::memory_order_release);
        }
        const int rounds = tiles / grid, n_tail = tiles - rounds * grid;          // ids: xcd_grid(xm) (a multiple of 8), a few of them surplus
        if (topc && nuc_only) hipLaunchKernelGGL((lm_head_sample_persist_kernel<false, 2>), dim3(grid + n_tail), dim3(kLmWM * kLmWN * 64), shp, s, A0, W0, steer_tok_d, partials,
                                                 (float *)nullptr, m, vocab_padded, d_model, vocab_padded, sp, xm, grid, rounds, fb);
        else if (topc) hipLaunchKernelGGL((lm_head_sample_persist_kernel<false, 1>), dim3(grid + n_tail), dim3(kLmWM * kLmWN * 64), shp, s, A0, W0, steer_tok_d, partials,
                                          (float *)nullptr, m, vocab_padded, d_model, vocab_padded, sp, xm, grid, rounds, fb);
        else if (lp) hipLaunchKernelGGL((lm_head_sample_persist_kernel<true, 0>), dim3(grid + n_tail), dim3(kLmWM * kLmWN * 64), shp, s, A0, W0, steer_tok_d, partials,
                                        logits_out_d, m, vocab_padded, d_model, vocab_padded, sp, xm, grid, rounds, (int32_t *)nullptr);
        else hipLaunchKernelGGL((lm_head_sample_persist_kernel<false, 0>), dim3(grid + n_tail), dim3(kLmWM * kLmWN * 64), shp, s, A0, W0, steer_tok_d, partials,
                                logits_out_d, m, vocab_padded, d_model, vocab_padded, sp, xm, grid, rounds, (int32_t *)nullptr);
    }
    else if (topc) {
#define LMRL_TOPC_LAUNCH(NOPS_, T_) hipLaunchKernelGGL((lm_head_sample_kernel<NOPS_, false, false, T_>), dim3(tiles), dim3(kLmWM * kLmWN * 64), shmem, s, A0, W0, A1, W1,  \
                                                   q_b1_d, A2, W2, q_b2_d, steer_tok_d, partials, (float *)nullptr, m, vocab_padded, d_model, vocab_padded, sp, xm,      \
                                                   (const int32_t *)nullptr, fb)
        if (nuc_only) { if (nops == 1) LMRL_TOPC_LAUNCH(1, 2); else if (nops == 2) LMRL_TOPC_LAUNCH(2, 2); else LMRL_TOPC_LAUNCH(3, 2); }
        else { if (nops == 1) LMRL_TOPC_LAUNCH(1, 1); else if (nops == 2) LMRL_TOPC_LAUNCH(2, 1); else LMRL_TOPC_LAUNCH(3, 1); }
#undef LMRL_TOPC_LAUNCH
    }
    else if (sp.rng == LMRL_RNG_JAX && !sp.greedy) {       // parity mode: always with the log-sum-exp variant (one instantiation per operand count)
        if (nops == 1) LMRL_LM_LAUNCH(1, true, true); else if (nops == 2) LMRL_LM_LAUNCH(2, true, true); else LMRL_LM_LAUNCH(3, true, true);
    }
    else if (nops == 1) { if (lp) LMRL_LM_LAUNCH(1, true, false); else LMRL_LM_LAUNCH(1, false, false); }
    else if (nops == 2) { if (lp) LMRL_LM_LAUNCH(2, true, false); else LMRL_LM_LAUNCH(2, false, false); }
    else { if (lp) LMRL_LM_LAUNCH(3, true, false); else LMRL_LM_LAUNCH(3, false, false); }
#undef LMRL_LM_LAUNCH
    }
    LMRL_CHECK_LAUNCH();
    if (topc) {
#define LMRL_TCR_LAUNCH(NT_, NUC_) hipLaunchKernelGGL((topc_reduce_sample_kernel<NT_, NUC_>), dim3((m + 3) / 4), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(ws_d), m, \
                                                     tiles_n, vocab, p->top_k, p->top_p, active_d, token_d, logprob_d, sp, p->pad_token, fb)
        if (nuc_only) { if (tiles_n <= 64) LMRL_TCR_LAUNCH(1, true); else if (tiles_n <= 128) LMRL_TCR_LAUNCH(2, true); else if (tiles_n <= 256) LMRL_TCR_LAUNCH(4, true); else LMRL_TCR_LAUNCH(8, true); }
        else { if (tiles_n <= 64) LMRL_TCR_LAUNCH(1, false); else if (tiles_n <= 128) LMRL_TCR_LAUNCH(2, false); else if (tiles_n <= 256) LMRL_TCR_LAUNCH(4, false); else LMRL_TCR_LAUNCH(8, false); }
#undef LMRL_TCR_LAUNCH
        LMRL_CHECK_LAUNCH();
        // the rows the check flagged (a tile holding eight or more of the row's top k: ~1e-8 of rows for spread-out logits): their 128-row blocks' logits
        // from the SAME tile kernel (bit-identical logits, steer included; greedy flag: no noise is drawn), then the materialised selection on those rows
        SampleParams sg = sp;
        sg.greedy = 1;
#define LMRL_FLAGGED_LAUNCH(NOPS_) hipLaunchKernelGGL((lm_head_sample_kernel<NOPS_, false, false, 0, true>), dim3(tiles_n), dim3(kLmWM * kLmWN * 64), shmem, s, A0, W0, A1, \
                                                      W1, q_b1_d, A2, W2, q_b2_d, steer_tok_d, (float *)nullptr, logits_out_d, m, vocab_padded, d_model, vocab_padded, sg, xm,   \
                                                      (const int32_t *)(fb + kFbHeader), (int32_t *)nullptr)
        if (nops == 1) LMRL_FLAGGED_LAUNCH(1); else if (nops == 2) LMRL_FLAGGED_LAUNCH(2); else LMRL_FLAGGED_LAUNCH(3);
#undef LMRL_FLAGGED_LAUNCH
        LMRL_CHECK_LAUNCH();
        launch_topk_sample(logits_out_d, vocab_padded, m, vocab, p->top_k, p->top_p, active_d, token_d, logprob_d, sp, p->pad_token, s, fb + kFbHeader + kFbMaxBlocks, fb);
    }
    else if ((p->top_k > 0 && p->top_k < vocab) || nucleus) {
        LMRL_REQUIRE(logits_out_d, "lmrl_lm_head_sample: top_k / top_p sampling needs logits_out_d (materialised logits)");
        launch_topk_sample(logits_out_d, vocab_padded, m, vocab, p->top_k, p->top_p, active_d, token_d, logprob_d, sp, p->pad_token, s);
    } else {
        hipLaunchKernelGGL(sample_reduce_kernel, dim3(ceil_div(m, 4)), dim3(256), 0, s, partials, active_d, token_d, logprob_d, m,
                           vocab_padded / kLmBN, p->pad_token);
    }
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_sample_logits_steer(float *logits_d, int ld, int m, int vocab, const lmrl_sample_params *p, const int32_t *steer_tok_d,
                             const uint8_t *active_d, int32_t *token_d, float *logprob_d, void *stream) {
    LMRL_REQUIRE(logits_d && p && m > 0, "lmrl_sample_logits_steer: bad argument");
    if (steer_tok_d && p->steer_strength != 0.f) {
        hipLaunchKernelGGL(steer_add_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, as_stream(stream), logits_d, ld, steer_tok_d, p->steer_strength, vocab, m);
        LMRL_CHECK_LAUNCH();
    }
    return lmrl_sample_logits(logits_d, ld, m, vocab, p, active_d, token_d, logprob_d, stream);
}

// host faces of the LMRL_RNG_JAX stream (CPU-tier known-answer tests; no GPU needed)
void lmrl_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) { threefry2x32_20(key[0], key[1], ctr[0], ctr[1], out[0], out[1]); }
int lmrl_jax_random_bits_host(const uint32_t key[2], uint32_t n, uint32_t i0, uint32_t count, uint32_t *out) {
    if (!key || !out || (uint64_t)i0 + count > n) return LMRL_ERR_ARG;
    for (uint32_t j = 0; j < count; j++) out[j] = jax_random_word(key[0], key[1], i0 + j, n);
    return LMRL_OK;
}

int lmrl_sample_logits(const float *logits_d, int ld, int m, int vocab, const lmrl_sample_params *p, const uint8_t *active_d,
                       int32_t *token_d, float *logprob_d, void *stream) {
    LMRL_REQUIRE(logits_d && p && token_d && m > 0 && vocab > 0 && ld >= vocab, "lmrl_sample_logits: bad argument");
    SampleParams sp;
    sp.greedy = (p->temperature <= 0.f) ? 1 : 0;
    sp.inv_temperature = sp.greedy ? 1.f : 1.f / p->temperature;
    sp.seed_lo = (uint32_t)p->seed; sp.seed_hi = (uint32_t)(p->seed >> 32); sp.step = p->step; sp.epoch = p->epoch_d;
    sp.steer_strength = 0.f; sp.beta = 0.f; sp.vocab = vocab; sp.tile_mass = nullptr;
    sp.temperature = sp.greedy ? 1.f : p->temperature; sp.rng = p->rng;
    LMRL_REQUIRE(p->rng == LMRL_RNG_PHILOX || p->rng == LMRL_RNG_JAX, "lmrl_sample_logits: unknown rng mode");
    LMRL_REQUIRE(p->rng != LMRL_RNG_JAX || (double)m * vocab < 4294967296.0, "lmrl_sample_logits: LMRL_RNG_JAX needs rows * vocab < 2^32");
    sp.jax_n = (uint32_t)m * (uint32_t)vocab;
    launch_topk_sample(logits_d, ld, m, vocab, p->top_k, p->top_p, active_d, token_d, logprob_d, sp, p->pad_token, as_stream(stream));
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
