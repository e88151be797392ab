// rl_reduce.hip — per-token RL reductions as wavefront-shuffle kernels (gfx950).
//
//   lmrl_gae     get_advantages_and_returns + get_action_state_next_state_idxs + scatter
//                (LLM_RL/algorithms/ppo/base_interface.py:230-243, 253-293, 586-606, 635-645)
//   lmrl_rtg     get_rtg over action tokens + MCData scatter (mc_returns/data.py:10-14, 49-74)
//   lmrl_whiten  whiten over all action tokens of all chains (ppo/base_interface.py:245-251, 609-615)
//
// Mapping: one 64-lane wave per chain.  The chain's action tokens are compacted with
// __ballot/popcount ranks into LDS; the reverse linear recurrence  A_t = d_t + c*A_{t+1}
// is evaluated 64 elements at a time with a 6-step shuffle scan (weights c, c^2, c^4 ...),
// chunks chained through a carry; results are scattered back to token positions.
// All of it is HBM-bound: 12 B read + 8 B written per token.
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

typedef float f32x2_t __attribute__((ext_vector_type(2)));      // (clang ext vectors: what __builtin_nontemporal_load / _store accept)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// the values rows have L + 1 floats (the bootstrap slot): with an even L every other row starts 4 bytes off an 8-byte boundary.  gfx950 global loads
// need dword alignment only (unaligned-access mode of the HSA ABI), so the value PAIR of a lane is still ONE 8-byte request
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// ---- round 5: the whole chain in registers.  Lane l of the wave owns token slots t = l + 64 j (j < K): every row read / written by a wave
// instruction is 64 consecutive elements (256 B of floats), ALL loads of a chain are issued before anything depends on them (one memory latency
// per chain instead of the round-1 kernel's load -> ballot -> LDS -> dependent gather chain, which left the launch latency-bound at 0.26-0.44 of
// the HBM peak), and no LDS is used.  The action tokens are not compacted: a non-action slot is the IDENTITY element of both scans —
//   next-state value   nv_t = V at the first action slot after t (bootstrap slot values[len] if none)   : reverse "first flagged" scan
//   advantage          A_t  = b_t + a_t A_{t+1},  (a, b) = (c, delta_t) on action slots, (1, 0) elsewhere : reverse scan of affine maps
// so x * 1 and x + 0 are exact and the result is the compacted recurrence of get_advantages_and_returns (ppo/base_interface.py:253-293) in a
// different (tree) association order.  64-slot chunks are chained from the tail through two carries (A and nv at the chunk's first slot).
struct Aff { float a, b; };

__device__ __forceinline__ Aff wave_rev_affine_scan2(Aff x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float a2 = __shfl_down(x.a, d), b2 = __shfl_down(x.b, d);
        if (lane + d < 64) { x.b = x.b + x.a * b2; x.a = x.a * a2; }        // f_t o (f_{t+1} o ...): b first (it needs the old a)
    }
    return x;
}

// reverse inclusive "first flagged value" scan: (f, v) at lane l = the flagged value at the lowest lane >= l that has one
__device__ __forceinline__ void wave_rev_first_scan(bool &f, float &v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int f2 = __shfl_down((int)f, d);
        const float v2 = __shfl_down(v, d);
        if (lane + d < 64 && !f) { f = f2 != 0; v = v2; }
    }
}

template <bool GAE, int K>          // K = ceil(L / 64) chunks held in registers
__global__ __launch_bounds__(256) void chain_scan_reg_kernel(const float *__restrict__ values, const float *__restrict__ rewards,
                                                              const uint8_t *__restrict__ sta, const int32_t *__restrict__ lens,
                                                              float *__restrict__ out0, float *__restrict__ out1, int B, int L, float gamma, float lam) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int len = lens ? min(max(lens[b], 0), L) : L;
    const uint8_t *srow = sta + (size_t)b * L;
    const float *rrow = rewards + (size_t)b * L;
    const float *vrow = GAE ? values + (size_t)b * (L + 1) : nullptr;
    float v[K], r[K];
    bool s[K];
#pragma unroll
    for (int j = 0; j < K; j++) {            // every load of the chain, back to back
        const int t = lane + 64 * j;
        const bool in = t < len;
        s[j] = in && srow[t] != 0;
        r[j] = in ? rrow[t] : 0.f;
        v[j] = (GAE && in) ? vrow[t] : 0.f;
    }
    const float boot = GAE ? vrow[len] : 0.f;             // values[len]: the bootstrap slot (next-state value of the last action, :230-243)
    const float c = GAE ? gamma * lam : gamma;
    float carry_a = 0.f, carry_nv = boot;
#pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        float d = r[j];
        if (GAE) {
            bool f = s[j];
            float fv = v[j];
            wave_rev_first_scan(f, fv, lane);
            const float incl = f ? fv : carry_nv;                        // first action value at or after this slot
            const float up = __shfl_down(incl, 1);
            const float nv = lane < 63 ? up : carry_nv;                  // ... strictly after it
            carry_nv = __shfl(incl, 0);
            d = r[j] + gamma * nv - v[j];                                // delta, :288
        }
        Aff x{s[j] ? c : 1.f, s[j] ? d : 0.f};
        x = wave_rev_affine_scan2(x, lane);
        const float a = x.b + x.a * carry_a;
        carry_a = __shfl(a, 0);
        const int t = lane + 64 * j;
        if (t < L) {
            out0[(size_t)b * L + t] = s[j] ? a : 0.f;                    // scattered to token positions, zeros elsewhere (:635-645)
            if (GAE) out1[(size_t)b * L + t] = s[j] ? a + v[j] : 0.f;    // returns = advantages + values, :291
        }
    }
}

// ---- chains of up to 128 slots (the rollout's: L = 96): FOUR chains per wave, one per 16-lane DPP row.  Lane g of a row owns slots t = g + 16 j
// (a row's load / store covers 16 consecutive elements = one 64 B segment; a wave instruction four of them).  The scans run over the 16 lanes of a
// row with 4 DPP steps (row_shl:1/2/4/8 — plain VALU, no ds_bpermute: the 64-lane kernel above spends ~50 LDS-crossbar shuffles per chain, this one
// none) and the two carries between 16-slot chunks travel from a row's lane 0 to its lane 15 by one row_mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_keep(float old, float x) {      // lanes whose DPP source lies outside the row keep `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xF, 0xF, false));
}

template <bool GAE, int K>          // K = ceil(L / 16) chunks held in registers
__global__ __launch_bounds__(256) void chain_scan_row_kernel(const float *__restrict__ values, const float *__restrict__ rewards,
                                                              const uint8_t *__restrict__ sta, const int32_t *__restrict__ lens,
                                                              float *__restrict__ out0, float *__restrict__ out1, int B, int L, float gamma, float lam) {
    const int g = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = b < B;                                  // (no early exit: DPP rows of a wave stay in lock step)
    const int bb = live ? b : B - 1;
    const int len = live ? (lens ? min(max(lens[bb], 0), L) : L) : 0;
    const uint8_t *srow = sta + (size_t)bb * L;
    const float *rrow = rewards + (size_t)bb * L;
    const float *vrow = GAE ? values + (size_t)bb * (L + 1) : nullptr;
    float v[K], r[K];
    bool s[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const int t = g + 16 * j;
        const bool in = t < len;
        s[j] = in && srow[t] != 0;
        r[j] = in ? rrow[t] : 0.f;
        v[j] = (GAE && in) ? vrow[t] : 0.f;
    }
    const float boot = GAE ? vrow[len] : 0.f;
    const float c = GAE ? gamma * lam : gamma;
    float carry_a = 0.f, carry_nv = boot;                     // meaningful in lane 15 of the row (everywhere for the first chunk processed)
#pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        float d = r[j];
        if (GAE) {
            // first action value at or after each slot, the chunk's successor (carry) standing behind lane 15
            bool f = s[j] || g == 15;
            float fv = s[j] ? v[j] : carry_nv;
#define LMRL_FIRST_STEP(N)                                                         \
            {                                                                      \
                const float f2 = dpp_keep<0x100 + N>(0.f, f ? 1.f : 0.f);          \
                const float v2 = dpp_keep<0x100 + N>(0.f, fv);                     \
                if (!f && f2 != 0.f) { f = true; fv = v2; }                        \
            }
            LMRL_FIRST_STEP(1) LMRL_FIRST_STEP(2) LMRL_FIRST_STEP(4) LMRL_FIRST_STEP(8)
#undef LMRL_FIRST_STEP
            const float nv = dpp_keep<0x101>(carry_nv, fv);                        // strictly after this slot (lane 15: the carry)
            carry_nv = dpp_keep<0x140>(fv, fv);                                    // row_mirror: lane 15 <- lane 0's inclusive result
            d = r[j] + gamma * nv - v[j];                                          // delta, :288
        }
        float xa = s[j] ? c : 1.f, xb = s[j] ? d : 0.f;
        if (g == 15) xb = xb + xa * carry_a;                                       // chain the later chunk in
#define LMRL_AFF_STEP(N)                                                           \
        {                                                                          \
            const float a2 = dpp_keep<0x100 + N>(1.f, xa), b2 = dpp_keep<0x100 + N>(0.f, xb); \
            xb = xb + xa * b2;                                                     \
            xa = xa * a2;                                                          \
        }
        LMRL_AFF_STEP(1) LMRL_AFF_STEP(2) LMRL_AFF_STEP(4) LMRL_AFF_STEP(8)
#undef LMRL_AFF_STEP
        carry_a = dpp_keep<0x140>(xb, xb);                                         // lane 15 <- A at the chunk's first slot
        const int t = g + 16 * j;
        if (live && t < L) {
            out0[(size_t)b * L + t] = s[j] ? xb : 0.f;
            if (GAE) out1[(size_t)b * L + t] = s[j] ? xb + v[j] : 0.f;
        }
    }
}

// The same with TWO consecutive slots per lane (t = 32 j + 2 g, + 1): half as many cross-lane scans per chain (the DPP steps are most of the
// kernel's VALU work), 8-byte accesses on rewards / flags / outputs (a row instruction = one full 128 B line).  Needs an even L.
// MOM (GAE only): the launch also leaves per-workgroup partial moments (sum, sum of squares, count in fp64) of the advantages on action slots in
// partials[3 * blockIdx.x ..] — what `whiten` over all action tokens needs (ppo/base_interface.py:245-251, 609-615) — so that no separate pass
// re-reads the advantages: lmrl_gae_moments + lmrl_whiten_apply_partials are two launches where lmrl_gae + lmrl_whiten_moments (two) +
// lmrl_whiten_apply were four.  Fixed association order (lane partials, wave tree, the four waves in order): bit-reproducible.
template <bool GAE, int K, bool MOM = false, bool V4 = false>          // K = ceil(L / 32) chunks held in registers; V4: the round-5 value loads (A/B, variant 5)
__global__ __launch_bounds__(256) void chain_scan_row2_kernel(const float *__restrict__ values, const float *__restrict__ rewards,
                                                               const uint8_t *__restrict__ sta, const int32_t *__restrict__ lens,
                                                               float *__restrict__ out0, float *__restrict__ out1, int B, int L, float gamma, float lam,
                                                               double *__restrict__ partials = nullptr) {
    const int g = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = b < B;
    const int bb = live ? b : B - 1;
    const int len = live ? (lens ? min(max(lens[bb], 0), L) : L) : 0;
    const uint8_t *srow = sta + (size_t)bb * L;
    const float *rrow = rewards + (size_t)bb * L;
    const float *vrow = GAE ? values + (size_t)bb * (L + 1) : nullptr;
    float v0[K], v1[K], r0[K], r1[K];
    bool s0[K], s1[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const int t = 32 * j + 2 * g;
        // (t even, L even: t < L implies t + 1 < L — the pair is inside the row; slots at or past `len` are masked after the load)
        f32x2_t rr = {0.f, 0.f};
        unsigned short ss = 0;
        if (t < len) {        // read once, never again: streaming (nt) accesses keep the lines out of the way of the other chains' in L2
            rr = __builtin_nontemporal_load(reinterpret_cast<const f32x2_t *>(rrow + t));
            ss = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(srow + t));
        }
        s0[j] = t < len && (ss & 0xffu) != 0;
        s1[j] = t + 1 < len && (ss >> 8) != 0;
        r0[j] = rr.x; r1[j] = t + 1 < len ? rr.y : 0.f;
        f32x2_t vv = {0.f, 0.f};
        if (V4) {                 // two 4-byte default-policy loads (round 5)
            vv.x = (GAE && t < len) ? vrow[t] : 0.f;
            vv.y = (GAE && t + 1 < len) ? vrow[t + 1] : 0.f;
        } else if (GAE && t < len) vv = __builtin_nontemporal_load(reinterpret_cast<const f32x2_a4 *>(vrow + t));      // (t + 1 <= len: the bootstrap slot at worst)
        v0[j] = vv.x;
        v1[j] = t + 1 < len ? vv.y : 0.f;
    }
    const float boot = GAE ? vrow[len] : 0.f;
    const float c = GAE ? gamma * lam : gamma;
    float carry_a = 0.f, carry_nv = boot;
    double ms = 0.0, mss = 0.0, mc = 0.0;
#pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        float d0 = r0[j], d1 = r1[j];
        if (GAE) {
            bool f = s0[j] || s1[j] || g == 15;
            float fv = s0[j] ? v0[j] : (s1[j] ? v1[j] : carry_nv);
#define LMRL_FIRST_STEP(N)                                                         \
            {                                                                      \
                const float f2 = dpp_keep<0x100 + N>(0.f, f ? 1.f : 0.f);          \
                const float w2 = dpp_keep<0x100 + N>(0.f, fv);                     \
                if (!f && f2 != 0.f) { f = true; fv = w2; }                        \
            }
            LMRL_FIRST_STEP(1) LMRL_FIRST_STEP(2) LMRL_FIRST_STEP(4) LMRL_FIRST_STEP(8)
#undef LMRL_FIRST_STEP
            const float after = dpp_keep<0x101>(carry_nv, fv);                     // first action value after this lane's pair
            carry_nv = dpp_keep<0x140>(fv, fv);
            d1 = r1[j] + gamma * after - v1[j];
            d0 = r0[j] + gamma * (s1[j] ? v1[j] : after) - v0[j];
        }
        const float a0 = s0[j] ? c : 1.f, b0 = s0[j] ? d0 : 0.f, a1 = s1[j] ? c : 1.f, b1 = s1[j] ? d1 : 0.f;
        float xa = a0 * a1, xb = b0 + a0 * b1;                                     // f_t0 o f_t1
        if (g == 15) xb = xb + xa * carry_a;
#define LMRL_AFF_STEP(N)                                                           \
        {                                                                          \
            const float a2 = dpp_keep<0x100 + N>(1.f, xa), b2 = dpp_keep<0x100 + N>(0.f, xb); \
            xb = xb + xa * b2;                                                     \
            xa = xa * a2;                                                          \
        }
        LMRL_AFF_STEP(1) LMRL_AFF_STEP(2) LMRL_AFF_STEP(4) LMRL_AFF_STEP(8)
#undef LMRL_AFF_STEP
        const float next = dpp_keep<0x101>(carry_a, xb);                           // A at the slot after this lane's pair (lane 15: the carry)
        carry_a = dpp_keep<0x140>(xb, xb);
        const float A1 = b1 + a1 * next;
        const int t = 32 * j + 2 * g;
        if (live && t < L) {
            const f32x2_t o0 = {s0[j] ? xb : 0.f, s1[j] ? A1 : 0.f}, o1 = {s0[j] ? xb + v0[j] : 0.f, s1[j] ? A1 + v1[j] : 0.f};
            __builtin_nontemporal_store(o0, reinterpret_cast<f32x2_t *>(out0 + (size_t)b * L + t));
            if (GAE) __builtin_nontemporal_store(o1, reinterpret_cast<f32x2_t *>(out1 + (size_t)b * L + t));
            if (MOM) {
                if (s0[j]) { const double a = (double)xb; ms += a; mss += a * a; mc += 1.0; }
                if (s1[j]) { const double a = (double)A1; ms += a; mss += a * a; mc += 1.0; }
            }
        }
    }
    if (MOM) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { ms += __shfl_down(ms, d); mss += __shfl_down(mss, d); mc += __shfl_down(mc, d); }
        __shared__ double red[3][4];
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][wave] = ms; red[1][wave] = mss; red[2][wave] = mc; }
        __syncthreads();
        if (threadIdx.x == 0) {
            partials[3 * blockIdx.x + 0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            partials[3 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
            partials[3 * blockIdx.x + 2] = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
        }
    }
}

// Reverse inclusive scan of x_i + c*x_{i+1} + c^2*x_{i+2} ... over the 64 lanes of a wave.
__device__ __forceinline__ float wave_rev_affine_scan(float x, float c, int lane) {
    float cd = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float t = __shfl_down(x, d);
        if (lane + d < 64) x = fmaf(cd, t, x);
        cd *= cd;
    }
    return x;
}

// (round 1; chains longer than 512 slots) LDS per wave: pos[L] (int), acc[L] (float), aux[L] (float)
template <bool GAE>
__global__ __launch_bounds__(256) void chain_scan_kernel(const float *__restrict__ values,   // [B][L+1] (GAE only)
                                                          const float *__restrict__ rewards,  // [B][L]
                                                          const uint8_t *__restrict__ sta,    // [B][L]
                                                          const int32_t *__restrict__ lens,   // [B] or null
                                                          float *__restrict__ out0,           // adv / rtg [B][L]
                                                          float *__restrict__ out1,           // ret [B][L] (GAE only)
                                                          int B, int L, float gamma, float lam) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    int *pos = reinterpret_cast<int *>(smem) + (size_t)wave * 3 * L;
    float *acc = reinterpret_cast<float *>(pos + L);
    float *aux = acc + L;
    const int len = lens ? min(lens[b], L) : L;
    const uint8_t *srow = sta + (size_t)b * L;
    const float *rrow = rewards + (size_t)b * L;
    const float *vrow = GAE ? values + (size_t)b * (L + 1) : nullptr;

    // 1. compaction: pos[k] = position of the k-th action token  (np.where(should_take_action))
    int n = 0;
    for (int base = 0; base < len; base += 64) {
        const int t = base + lane;
        const bool f = t < len && srow[t] != 0;
        const unsigned long long bal = __ballot(f);
        if (f) pos[n + __popcll(bal & ((1ull << lane) - 1ull))] = t;
        n += __popcll(bal);
    }
    // pos[] is written and read by the same wave only; a wave-level LDS fence is enough.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // 2. reverse scan over compacted tokens, 64 at a time starting from the tail
    const float c = GAE ? gamma * lam : gamma;
    float carry = 0.f;
    for (int hi = n; hi > 0; hi -= 64) {
        const int lo = hi - 64;           // chunk covers compacted indices [lo, hi), lane i <-> lo + i
        const int k = lo + lane;
        float d = 0.f, vs = 0.f;
        if (k >= 0) {
            const int p = pos[k];
            if (GAE) {
                // state value at the action position, next-state value at the NEXT action position or the
                // bootstrap slot values[len]  (get_action_state_next_state_idxs, :230-243)
                const int pn = (k + 1 < n) ? pos[k + 1] : len;
                vs = vrow[p];
                d = rrow[p] + gamma * vrow[pn] - vs;   // delta, :288
            } else {
                d = rrow[p];
            }
        }
        if (lane == 63) d = fmaf(c, carry, d);          // chain the later chunk in
        const float a = wave_rev_affine_scan(d, c, lane);
        if (k >= 0) {
            acc[k] = a;
            if (GAE) aux[k] = a + vs;                    // returns = advantages + values, :291
        }
        // carry = A at compacted index lo (lane 0 if lo >= 0)
        carry = __shfl(a, lo >= 0 ? 0 : -lo);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // 3. scatter back to token positions, zeros elsewhere (:635-645 / mc_returns/data.py:67-69)
    int seen = 0;
    for (int base = 0; base < L; base += 64) {
        const int t = base + lane;
        const bool f = t < len && srow[t] != 0;
        const unsigned long long bal = __ballot(f);
        const int k = seen + __popcll(bal & ((1ull << lane) - 1ull));
        if (t < L) {
            out0[(size_t)b * L + t] = f ? acc[k] : 0.f;
            if (GAE) out1[(size_t)b * L + t] = f ? aux[k] : 0.f;
        }
        seen += __popcll(bal);
    }
}

// ---- whitening -------------------------------------------------------------------------------
// VEC: 16 bytes of x and 4 mask bytes per lane and iteration (n a multiple of 4, 16-byte aligned x, 4-byte aligned mask); the sums are fp64 (the
// reference's jnp.mean / jnp.var are fp32 pairwise sums: any order within a few ulp is as good, fp64 partials make the result order-independent
// to fp32 accuracy and — with the fixed-order finish — bit-reproducible)
template <bool VEC>
__global__ __launch_bounds__(256) void whiten_moments_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                              double *partials, size_t n, unsigned *ticket, double *__restrict__ moments_out) {
    double s = 0.0, ss = 0.0, cnt = 0.0;
    auto take = [&](f32x4_t v, uint32_t m) {
        const float e[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((m >> (8 * k)) & 0xffu) { const double d = (double)e[k]; s += d; ss += d * d; cnt += 1.0; }
    };
    if (VEC) {
        const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent 16-byte loads in flight per lane
            f32x4_t v[4]; uint32_t m[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(x) + i + u * stride);
                m[u] = mask ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(mask) + i + u * stride) : 0x01010101u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) take(v[u], m[u]);
        }
        for (; i < n4; i += stride)
            take(reinterpret_cast<const f32x4_t *>(x)[i], mask ? reinterpret_cast<const uint32_t *>(mask)[i] : 0x01010101u);
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            if (!mask || mask[i]) {
                const double v = (double)x[i];
                s += v; ss += v * v; cnt += 1.0;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_down(s, d); ss += __shfl_down(ss, d); cnt += __shfl_down(cnt, d);
    }
    __shared__ double red[3][4];
    __shared__ bool is_last;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; red[2][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {   // per-workgroup partial; the LAST workgroup to arrive adds all of them in workgroup order (deterministic, no fp64 atomics)
        double a = 0, b2 = 0, c2 = 0;
        for (int w = 0; w < 4; w++) { a += red[0][w]; b2 += red[1][w]; c2 += red[2][w]; }
        partials[3 * blockIdx.x + 0] = a;
        partials[3 * blockIdx.x + 1] = b2;
        partials[3 * blockIdx.x + 2] = c2;
        is_last = false;
        if (ticket) {
            __threadfence();
            is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
        }
    }
    __syncthreads();
    if (is_last && wave == 0) {
        __threadfence();
        double ts = 0.0, tss = 0.0, tc = 0.0;
        for (int b = lane; b < (int)gridDim.x; b += 64) {      // lane l takes partials l, l + 64, ... then a fixed-order tree: order independent of arrival
            ts += __builtin_nontemporal_load(partials + 3 * b); tss += __builtin_nontemporal_load(partials + 3 * b + 1);
            tc += __builtin_nontemporal_load(partials + 3 * b + 2);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { ts += __shfl_down(ts, d); tss += __shfl_down(tss, d); tc += __shfl_down(tc, d); }
        if (lane == 0) { moments_out[0] = ts; moments_out[1] = tss; moments_out[2] = tc; *ticket = 0u; }
    }
}

// (the one-launch form — last workgroup to arrive adds the partials, `ticket` below — is kept for A/B only: the agent-scope fence every workgroup needs
// before its ticket writes back its XCD's L2, 36 us instead of 8 + 3 at 65 536 chains; a kernel boundary orders the partials for free)
__global__ __launch_bounds__(256) void whiten_finish_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ moments) {
    double s = 0.0, ss = 0.0, cnt = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { s += partials[3 * b]; ss += partials[3 * b + 1]; cnt += partials[3 * b + 2]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_down(s, d); ss += __shfl_down(ss, d); cnt += __shfl_down(cnt, d);
    }
    __shared__ double red[3][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; red[2][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {          // fixed order: thread-strided partial sums, wave trees, then the four waves in order
        moments[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        moments[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        moments[2] = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void whiten_apply_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                            const double *__restrict__ moments, float *__restrict__ y,
                                                            size_t n, int shift_mean) {
    const double cnt = moments[2];
    const double mean = cnt > 0 ? moments[0] / cnt : 0.0;
    double var = cnt > 0 ? moments[1] / cnt - mean * mean : 0.0;   // population variance (jnp.var)
    if (var < 0) var = 0;
    const double inv = 1.0 / sqrt(var + 1e-8);
    auto w1 = [&](float v) {
        double w = ((double)v - mean) * inv;
        if (!shift_mean) w += mean;
        return (float)w;
    };
    if (VEC) {
        const size_t n4 = n / 4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 v = reinterpret_cast<const float4 *>(x)[i];
            const uint32_t m = mask ? reinterpret_cast<const uint32_t *>(mask)[i] : 0x01010101u;
            if (m & 0x000000ffu) v.x = w1(v.x);
            if (m & 0x0000ff00u) v.y = w1(v.y);
            if (m & 0x00ff0000u) v.z = w1(v.z);
            if (m & 0xff000000u) v.w = w1(v.w);
            reinterpret_cast<float4 *>(y)[i] = v;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = (!mask || mask[i]) ? w1(v) : v;
    }
}

// whiten_apply with the moments still as per-workgroup partials (lmrl_gae_moments): every workgroup adds the (few hundred) partials itself, in the
// order whiten_finish_kernel uses — the same three doubles in every workgroup, and the same as the two-launch form's — then applies
template <bool VEC>
__global__ __launch_bounds__(256) void whiten_apply_partials_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                                     const double *__restrict__ partials, int nblocks, float *__restrict__ y,
                                                                     size_t n, int shift_mean) {
    double s = 0.0, ss = 0.0, cn = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { s += partials[3 * b]; ss += partials[3 * b + 1]; cn += partials[3 * b + 2]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_down(s, d); ss += __shfl_down(ss, d); cn += __shfl_down(cn, d); }
    __shared__ double red[3][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; red[2][wave] = cn; }
    __syncthreads();
    const double cnt = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
    const double sum = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], sq = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    const double mean = cnt > 0 ? sum / cnt : 0.0;
    double var = cnt > 0 ? sq / cnt - mean * mean : 0.0;
    if (var < 0) var = 0;
    const double inv = 1.0 / sqrt(var + 1e-8);
    auto w1 = [&](float v) {
        double w = ((double)v - mean) * inv;
        if (!shift_mean) w += mean;
        return (float)w;
    };
    if (VEC) {
        const size_t n4 = n / 4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 v = reinterpret_cast<const float4 *>(x)[i];
            const uint32_t m = mask ? reinterpret_cast<const uint32_t *>(mask)[i] : 0x01010101u;
            if (m & 0x000000ffu) v.x = w1(v.x);
            if (m & 0x0000ff00u) v.y = w1(v.y);
            if (m & 0x00ff0000u) v.z = w1(v.z);
            if (m & 0xff000000u) v.w = w1(v.w);
            reinterpret_cast<float4 *>(y)[i] = v;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = (!mask || mask[i]) ? w1(v) : v;
    }
}

int g_rl_reduce_variant = 0;      // tools / tests only: 1 = the round-1 LDS-compaction kernel for every length, 2 = the 64-lane register kernel also for L <= 128, 3 = one slot per lane in the DPP-row kernel, 5 = GAE at 64 < L <= 96 with the round-5 4-byte value loads (A/B of the unaligned 8-byte pairs)

template <bool GAE>
static int scan_launch_t(const float *values, const float *rewards, const uint8_t *sta, const int32_t *lens, float *o0, float *o1, int b, int l, float gamma,
                         float lam, hipStream_t s) {
    const dim3 grid(ceil_div(b, 4)), block(256);
    const int k = (l + 63) / 64;
    if ((g_rl_reduce_variant == 0 || g_rl_reduce_variant == 3 || g_rl_reduce_variant == 5 || g_rl_reduce_variant == 6) && l <= 128) {
        const dim3 grid16(ceil_div(b, 16));
        // the value pair of a lane: ONE unaligned 8-byte streaming load where the launch is latency-bound (rollout-sized batches: 3.76 vs 4.01 us at 4096
        // chains), two 4-byte default-policy loads where it is bandwidth-bound (19.05 vs 20.56 us at 65 536 chains: the rows of `values` have L + 1
        // floats, a pair straddles a 64-byte sector every 8th lane) — A/B on one box, profiles/r06_rl_reduce_vload_ab.txt; variant 5 / 6 force either form
        const bool v4 = g_rl_reduce_variant == 5 || (g_rl_reduce_variant == 0 && b > 16384);
        if (v4 && g_rl_reduce_variant != 6 && GAE && l % 2 == 0 && l > 64 && l <= 96 && (uintptr_t)rewards % 8 == 0 && (uintptr_t)o0 % 8 == 0 && (uintptr_t)o1 % 8 == 0 && (uintptr_t)sta % 2 == 0) {
            hipLaunchKernelGGL((chain_scan_row2_kernel<GAE, 3, false, true>), grid16, block, 0, s, values, rewards, sta, lens, o0, o1, b, l, gamma, lam, (double *)nullptr);
            LMRL_CHECK_LAUNCH();
            return LMRL_OK;
        }
        if ((g_rl_reduce_variant == 0 || g_rl_reduce_variant == 5 || g_rl_reduce_variant == 6) && l % 2 == 0 && (uintptr_t)rewards % 8 == 0 && (uintptr_t)o0 % 8 == 0 && (!GAE || (uintptr_t)o1 % 8 == 0) && (uintptr_t)sta % 2 == 0) {
#define LMRL_SCAN_ROW2(K_) hipLaunchKernelGGL((chain_scan_row2_kernel<GAE, K_>), grid16, block, 0, s, values, rewards, sta, lens, o0, o1, b, l, gamma, lam)
            switch ((l + 31) / 32) {
                case 1: LMRL_SCAN_ROW2(1); break;
                case 2: LMRL_SCAN_ROW2(2); break;
                case 3: LMRL_SCAN_ROW2(3); break;
                default: LMRL_SCAN_ROW2(4); break;
            }
#undef LMRL_SCAN_ROW2
            LMRL_CHECK_LAUNCH();
            return LMRL_OK;
        }
#define LMRL_SCAN_ROW(K_) hipLaunchKernelGGL((chain_scan_row_kernel<GAE, K_>), grid16, block, 0, s, values, rewards, sta, lens, o0, o1, b, l, gamma, lam)
        switch ((l + 15) / 16) {
            case 1: LMRL_SCAN_ROW(1); break;
            case 2: LMRL_SCAN_ROW(2); break;
            case 3: LMRL_SCAN_ROW(3); break;
            case 4: LMRL_SCAN_ROW(4); break;
            case 5: LMRL_SCAN_ROW(5); break;
            case 6: LMRL_SCAN_ROW(6); break;
            case 7: LMRL_SCAN_ROW(7); break;
            default: LMRL_SCAN_ROW(8); break;
        }
#undef LMRL_SCAN_ROW
        LMRL_CHECK_LAUNCH();
        return LMRL_OK;
    }
    if (g_rl_reduce_variant != 1 && k <= 8) {
#define LMRL_SCAN_REG(K_) hipLaunchKernelGGL((chain_scan_reg_kernel<GAE, K_>), grid, block, 0, s, values, rewards, sta, lens, o0, o1, b, l, gamma, lam)
        switch (k) {
            case 1: LMRL_SCAN_REG(1); break;
            case 2: LMRL_SCAN_REG(2); break;
            case 3: LMRL_SCAN_REG(3); break;
            case 4: LMRL_SCAN_REG(4); break;
            case 5: case 6: LMRL_SCAN_REG(6); break;
            default: LMRL_SCAN_REG(8); break;
        }
#undef LMRL_SCAN_REG
        LMRL_CHECK_LAUNCH();
        return LMRL_OK;
    }
    const size_t shmem = (size_t)4 * 3 * l * sizeof(float);
    if (shmem > 160 * 1024) {
        set_error("chain scan: L=%d needs %zu B of LDS per workgroup (> 160 KiB)", l, shmem);
        return LMRL_ERR_ARG;
    }
    if (shmem > 64 * 1024)
        LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_scan_kernel<GAE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(chain_scan_kernel<GAE>, grid, block, shmem, s, values, rewards, sta, lens, o0, o1, b, l, gamma, lam);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

static int scan_launch(bool gae, const float *values, const float *rewards, const uint8_t *sta, const int32_t *lens,
                       float *o0, float *o1, int b, int l, float gamma, float lam, void *stream) {
    return gae ? scan_launch_t<true>(values, rewards, sta, lens, o0, o1, b, l, gamma, lam, as_stream(stream))
               : scan_launch_t<false>(values, rewards, sta, lens, o0, o1, b, l, gamma, lam, as_stream(stream));
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

void lmrl_rl_reduce_set_variant(int v) { g_rl_reduce_variant = v; }

int lmrl_gae(const float *values_d, const float *rewards_d, const uint8_t *sta_d, const int32_t *len_d, float *adv_d,
             float *ret_d, int b, int l, float gamma, float lam, void *stream) {
    LMRL_REQUIRE(values_d && rewards_d && sta_d && adv_d && ret_d && b >= 0 && l > 0, "lmrl_gae: bad argument");
    if (b == 0) return LMRL_OK;
    return scan_launch(true, values_d, rewards_d, sta_d, len_d, adv_d, ret_d, b, l, gamma, lam, stream);
}

int lmrl_gae_moments_partials(int b, int l) {      // doubles triples lmrl_gae_moments writes (0: shape not covered by the fused kernel)
    return (l <= 128 && l % 2 == 0 && b > 0) ? ceil_div(b, 16) : 0;
}

int lmrl_gae_moments(const float *values_d, const float *rewards_d, const uint8_t *sta_d, const int32_t *len_d, float *adv_d, float *ret_d, int b, int l,
                     float gamma, float lam, double *partials_d, void *stream) {
    LMRL_REQUIRE(values_d && rewards_d && sta_d && adv_d && ret_d && partials_d && b > 0 && l > 0, "lmrl_gae_moments: bad argument");
    LMRL_REQUIRE(lmrl_gae_moments_partials(b, l) > 0 && (uintptr_t)rewards_d % 8 == 0 && (uintptr_t)adv_d % 8 == 0 && (uintptr_t)ret_d % 8 == 0 &&
                     (uintptr_t)sta_d % 2 == 0,
                 "lmrl_gae_moments: chains of an even number of <= 128 slots, 8-byte aligned rows (else lmrl_gae + lmrl_whiten_moments)");
    const dim3 grid16(ceil_div(b, 16)), block(256);
    hipStream_t s = as_stream(stream);
#define LMRL_SCAN_ROW2M(K_) hipLaunchKernelGGL((chain_scan_row2_kernel<true, K_, true>), grid16, block, 0, s, values_d, rewards_d, sta_d, len_d, adv_d, ret_d, b, l, gamma, lam, partials_d)
    if (b > 16384 && (l + 31) / 32 == 3)      // bandwidth-bound launches: the 4-byte value loads (see scan_launch_t)
        hipLaunchKernelGGL((chain_scan_row2_kernel<true, 3, true, true>), grid16, block, 0, s, values_d, rewards_d, sta_d, len_d, adv_d, ret_d, b, l, gamma, lam, partials_d);
    else switch ((l + 31) / 32) {
        case 1: LMRL_SCAN_ROW2M(1); break;
        case 2: LMRL_SCAN_ROW2M(2); break;
        case 3: LMRL_SCAN_ROW2M(3); break;
        default: LMRL_SCAN_ROW2M(4); break;
    }
#undef LMRL_SCAN_ROW2M
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_whiten_finish(const double *partials_d, int n_partials, double *moments_d, void *stream) {
    LMRL_REQUIRE(partials_d && moments_d && n_partials > 0, "lmrl_whiten_finish: bad argument");
    hipLaunchKernelGGL(whiten_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), partials_d, n_partials, moments_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_whiten_apply_partials(const float *x_d, const uint8_t *mask_d, const double *partials_d, int n_partials, float *y_d, size_t n, int shift_mean,
                               void *stream) {
    LMRL_REQUIRE(x_d && partials_d && y_d && n_partials > 0, "lmrl_whiten_apply_partials: bad argument");
    if (n == 0) return LMRL_OK;
    const bool vec = n % 4 == 0 && (uintptr_t)x_d % 16 == 0 && (uintptr_t)y_d % 16 == 0 && (uintptr_t)mask_d % 4 == 0;
    int grid = ceil_div((long)n, vec ? 256 * 4 : 256);
    if (grid > 1024) grid = 1024;      // every workgroup re-adds the partials: keep that redundant read small next to the sweep
    if (vec) hipLaunchKernelGGL(whiten_apply_partials_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, partials_d, n_partials, y_d, n, shift_mean);
    else hipLaunchKernelGGL(whiten_apply_partials_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, partials_d, n_partials, y_d, n, shift_mean);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_rtg(const float *rewards_d, const uint8_t *sta_d, const int32_t *len_d, float *rtg_d, int b, int l, float gamma,
             void *stream) {
    LMRL_REQUIRE(rewards_d && sta_d && rtg_d && b >= 0 && l > 0, "lmrl_rtg: bad argument");
    if (b == 0) return LMRL_OK;
    return scan_launch(false, nullptr, rewards_d, sta_d, len_d, rtg_d, nullptr, b, l, gamma, 0.f, stream);
}

int lmrl_whiten_moments(const float *x_d, const uint8_t *mask_d, double *moments_d, size_t n, void *stream) {
    LMRL_REQUIRE(x_d && moments_d, "lmrl_whiten_moments: null pointer");
    if (n == 0) {
        LMRL_CHECK_HIP(hipMemsetAsync(moments_d, 0, 3 * sizeof(double), as_stream(stream)));
        return LMRL_OK;
    }
    // per-workgroup partials in a process-wide scratch buffer (calls on different streams must not overlap), then a
    // fixed-order final sum: bit-reproducible, no fp64 atomics
    constexpr int kMaxBlocks = 1024;
    static double *partials = nullptr;
    static unsigned *ticket = nullptr;
    if (!partials) {
        LMRL_CHECK_HIP(hipMalloc(&partials, (size_t)kMaxBlocks * 3 * sizeof(double) + 64));
        ticket = reinterpret_cast<unsigned *>(partials + (size_t)kMaxBlocks * 3);
        LMRL_CHECK_HIP(hipMemset(ticket, 0, 64));
    }
    const bool vec = g_rl_reduce_variant != 1 && n % 4 == 0 && (uintptr_t)x_d % 16 == 0 && (uintptr_t)mask_d % 4 == 0;
    int grid = ceil_div((long)n, 256 * (vec ? 16 : 8));
    if (grid > kMaxBlocks) grid = kMaxBlocks;
    if (grid < 1) grid = 1;
    unsigned *tk = g_rl_reduce_variant == 4 ? ticket : nullptr;          // 4 (tools): the one-launch form with the ticket
    if (vec) hipLaunchKernelGGL(whiten_moments_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, partials, n, tk, moments_d);
    else hipLaunchKernelGGL(whiten_moments_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, partials, n, tk, moments_d);
    if (!tk) hipLaunchKernelGGL(whiten_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), partials, grid, moments_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_whiten_apply(const float *x_d, const uint8_t *mask_d, const double *moments_d, float *y_d, size_t n,
                      int shift_mean, void *stream) {
    LMRL_REQUIRE(x_d && moments_d && y_d, "lmrl_whiten_apply: null pointer");
    if (n == 0) return LMRL_OK;
    const bool vec = g_rl_reduce_variant != 1 && n % 4 == 0 && (uintptr_t)x_d % 16 == 0 && (uintptr_t)y_d % 16 == 0 && (uintptr_t)mask_d % 4 == 0;
    int grid = ceil_div((long)n, vec ? 256 * 4 : 256);
    if (grid > 4096) grid = 4096;
    if (vec) hipLaunchKernelGGL(whiten_apply_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, moments_d, y_d, n, shift_mean);
    else hipLaunchKernelGGL(whiten_apply_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, moments_d, y_d, n, shift_mean);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
