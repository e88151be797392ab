// losses.hip — PPO / ILQL / MC loss forward + backward over [B, T-1] token grids (gfx950).
//
//   lmrl_ppo_loss    ppo_loss_fn                         LLM_RL/algorithms/ppo/base_interface.py:72-142
//   lmrl_ilql_loss   ilql_loss (+ get_query_indicators)  LLM_RL/algorithms/ilql/base_interface.py:22-119
//   lmrl_mc_loss     mc_loss                             LLM_RL/algorithms/mc_returns/base_interface.py:19-60
//
// The reference selects action tokens with O(N^2) one-hot "query indicator" matrices (N = B(T-1) = 16 352 -> 1.07 GB
// each); here the k-th action of a row is paired with the next action of the same row by a wave-level ballot
// compaction (the same pairing, see tests/test_oracle_rl.py::test_ilql_loss_selection_equals_per_row_next_action).
// Every kernel emits (a) per-row/-block fp64 partial sums of every logged quantity, reduced on the host in a fixed
// order (deterministic), and (b) the gradient of the scalar loss w.r.t. its inputs.  HBM-bound: ~40 B per token.
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

struct Stat5 { double sum_w, sum_b, sumsq_b, mn, mx; };   // sum(x*maskf), and over mask!=0: sum, sumsq, min, max

__device__ __forceinline__ double wsum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ double wmin_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ double wmax_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    return x;
}

// n = sum(should_take_action * attention_mask)
__global__ __launch_bounds__(256) void mask_sum_kernel(const uint8_t *__restrict__ sta, const float *__restrict__ attn, size_t n, double *out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        s += (sta ? (sta[i] ? 1.0 : 0.0) : 1.0) * (attn ? (double)attn[i] : 1.0);
    s = wsum_d(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// ------------------------------------------------------------------------------------------ PPO
constexpr int kPpoStats = 24;
// partial layout (per block): 0 n | 1 vf_sum 2 vf_clip | 3 kl_sum | 4 pg_sum 5 pg_clip | 6 values_err | 7 ratio_sum | 8 count_b
// 9.. values{sum_w,sum_b,sumsq_b,min,max} | 14.. old_values{...} | 19.. returns{...}
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float *__restrict__ attn, const float *__restrict__ logprobs,
                                                       const float *__restrict__ values, const uint8_t *__restrict__ sta,
                                                       const float *__restrict__ old_logprobs, const float *__restrict__ old_values,
                                                       const float *__restrict__ old_adv, const float *__restrict__ old_ret, size_t n_el,
                                                       float clip_v, float clip, float vcoef, const double *__restrict__ n_ptr,
                                                       double *__restrict__ partials, float *__restrict__ d_logprobs,
                                                       float *__restrict__ d_values) {
    double acc[kPpoStats];
#pragma unroll
    for (int k = 0; k < kPpoStats; k++) acc[k] = 0.0;
    acc[12] = acc[17] = acc[22] = INFINITY; acc[13] = acc[18] = acc[23] = -INFINITY;
    const double n = *n_ptr;
    const float inv_n = (float)(1.0 / n);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (size_t)gridDim.x * blockDim.x) {
        const float mask = (sta[i] ? 1.f : 0.f) * attn[i];
        const bool mb = mask != 0.f;
        const float v = values[i], ov = old_values[i], R = old_ret[i], A = old_adv[i];
        const float lo = ov - clip_v, hi = ov + clip_v;
        const float vc = fminf(fmaxf(v, lo), hi);
        const float vf1 = (v - R) * (v - R), vf2 = (vc - R) * (vc - R);
        const float lr = (logprobs[i] - old_logprobs[i]) * mask;
        const float ratio = expf(lr);
        const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        const float pg1 = -A * ratio, pg2 = -A * rc;
        acc[0] += mask;
        acc[1] += (double)(fmaxf(vf1, vf2) * mask); acc[2] += (vf2 > vf1 ? 1.0 : 0.0) * mask;
        acc[3] += (double)((ratio - 1.f) - lr);
        acc[4] += (double)(fmaxf(pg1, pg2) * mask); acc[5] += (pg2 > pg1 ? 1.0 : 0.0) * mask;
        const float ve = (v - R) * mask;
        acc[6] += (double)(ve * ve); acc[7] += (double)(ratio * mask);
        if (mb) acc[8] += 1.0;
        const float xs[3] = {v, ov, R};
#pragma unroll
        for (int t = 0; t < 3; t++) {
            acc[9 + 5 * t] += (double)(xs[t] * mask);
            if (mb) {
                acc[10 + 5 * t] += xs[t]; acc[11 + 5 * t] += (double)xs[t] * xs[t];
                acc[12 + 5 * t] = fmin(acc[12 + 5 * t], (double)xs[t]); acc[13 + 5 * t] = fmax(acc[13 + 5 * t], (double)xs[t]);
            }
        }
        // ---- gradients (jnp.maximum / jnp.clip split ties evenly; inside the clip range both branches agree)
        const float dpg1 = -A * ratio * mask;                                   // d pg1 / d logprob
        const float dpg2 = (ratio >= 1.f - clip && ratio <= 1.f + clip) ? dpg1 : 0.f;
        const float gpg = pg1 > pg2 ? dpg1 : (pg1 < pg2 ? dpg2 : 0.5f * (dpg1 + dpg2));
        d_logprobs[i] = gpg * mask * inv_n;
        const float dvf1 = 2.f * (v - R);
        const float dvf2 = (v >= lo && v <= hi) ? 2.f * (vc - R) : 0.f;
        const float gvf = vf1 > vf2 ? dvf1 : (vf1 < vf2 ? dvf2 : 0.5f * (dvf1 + dvf2));
        d_values[i] = vcoef * 0.5f * gvf * mask * inv_n;
    }
    __shared__ double red[4][kPpoStats];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kPpoStats; k++) {
        const bool is_min = (k == 12 || k == 17 || k == 22), is_max = (k == 13 || k == 18 || k == 23);
        const double r = is_min ? wmin_d(acc[k]) : (is_max ? wmax_d(acc[k]) : wsum_d(acc[k]));
        if (lane == 0) red[wave][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < kPpoStats) {
        const int k = threadIdx.x;
        const bool is_min = (k == 12 || k == 17 || k == 22), is_max = (k == 13 || k == 18 || k == 23);
        double r = red[0][k];
        for (int w = 1; w < 4; w++) r = is_min ? fmin(r, red[w][k]) : (is_max ? fmax(r, red[w][k]) : r + red[w][k]);
        partials[(size_t)blockIdx.x * kPpoStats + k] = r;
    }
}

// ------------------------------------------------------------------------------------------ ILQL
constexpr int kIlqlStats = 48;
// per-row partials: 0 q1_loss 1 q2_loss 2 v_loss 3 cql1 4 cql2 | 5 n_sa (count of actions) 6 n_ns |
// 7.. q1{sum,sumsq,min,max} 11.. q2 15.. v 19.. target_q 23.. target_q1 27.. target_q2 31.. vns | 35.. rewards{sum_w,sum_b,sumsq_b,min,max} 40 count_mask_b
__global__ __launch_bounds__(256) void ilql_loss_kernel(const float *__restrict__ q1, const float *__restrict__ q2, const float *__restrict__ v,
                                                        const float *__restrict__ v_final, const float *__restrict__ tq1,
                                                        const float *__restrict__ tq2, const float *__restrict__ ce1,
                                                        const float *__restrict__ ce2, const float *__restrict__ attn,
                                                        const uint8_t *__restrict__ sta, const float *__restrict__ rewards, int B,
                                                        int T1, float gamma, float tau, float cql_w, const double *__restrict__ n_ptr,
                                                        double *__restrict__ partials, float *__restrict__ dq1, float *__restrict__ dq2,
                                                        float *__restrict__ dv, float *__restrict__ coef_ce) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    int *pos = reinterpret_cast<int *>(smem) + (size_t)wave * T1;
    const size_t off = (size_t)b * T1;
    const float inv_n = (float)(1.0 / *n_ptr);
    // compaction of the row's action positions
    int na = 0;
    for (int base = 0; base < T1; base += 64) {
        const int t = base + lane;
        const bool f = t < T1 && sta[off + t] != 0;
        const unsigned long long bal = __ballot(f);
        if (f) pos[na + __popcll(bal & ((1ull << lane) - 1ull))] = t;
        na += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double acc[kIlqlStats];
#pragma unroll
    for (int k = 0; k < kIlqlStats; k++) acc[k] = 0.0;
#pragma unroll
    for (int t = 0; t < 7; t++) { acc[9 + 4 * t] = INFINITY; acc[10 + 4 * t] = -INFINITY; }
    acc[38] = INFINITY; acc[39] = -INFINITY;
    // zero-fill / CE coefficients / rewards stats over every position of the row
    for (int t = lane; t < T1; t += 64) {
        const float mask = (sta[off + t] ? 1.f : 0.f) * attn[off + t];
        dq1[off + t] = 0.f; dq2[off + t] = 0.f; dv[off + t] = 0.f;
        coef_ce[off + t] = cql_w * mask * inv_n;
        acc[3] += (double)(mask * ce1[off + t]); acc[4] += (double)(mask * ce2[off + t]);
        const float r = rewards[off + t];
        acc[35] += (double)(r * mask);
        if (mask != 0.f) { acc[36] += r; acc[37] += (double)r * r; acc[38] = fmin(acc[38], (double)r); acc[39] = fmax(acc[39], (double)r); acc[40] += 1.0; }
    }
    // per action k: state = pos[k], next state = pos[k+1] or the v_final slot
    for (int k = lane; k < na; k += 64) {
        const int p = pos[k];
        const bool last = k + 1 >= na;
        const float vns = last ? v_final[b] : v[off + pos[k + 1]];
        const float q1s = q1[off + p], q2s = q2[off + p], vs = v[off + p], t1 = tq1[off + p], t2 = tq2[off + p], rs = rewards[off + p];
        const float target = rs + gamma * vns;
        const float tq = fminf(t1, t2);
        const float w = tq >= vs ? tau : 1.f - tau;
        acc[0] += 0.5 * (double)(q1s - target) * (q1s - target);
        acc[1] += 0.5 * (double)(q2s - target) * (q2s - target);
        acc[2] += 0.5 * (double)(vs - tq) * (vs - tq) * w;
        acc[5] += 1.0; acc[6] += 1.0;
        const float xs[7] = {q1s, q2s, vs, tq, t1, t2, vns};
#pragma unroll
        for (int t = 0; t < 7; t++) {
            acc[7 + 4 * t] += xs[t]; acc[8 + 4 * t] += (double)xs[t] * xs[t];
            acc[9 + 4 * t] = fmin(acc[9 + 4 * t], (double)xs[t]); acc[10 + 4 * t] = fmax(acc[10 + 4 * t], (double)xs[t]);
        }
        dq1[off + p] = (q1s - target) * inv_n;
        dq2[off + p] = (q2s - target) * inv_n;
        dv[off + p] = (vs - tq) * w * inv_n;
    }
#pragma unroll
    for (int k = 0; k < kIlqlStats; k++) {
        const int kk = k - 7;
        const bool in_t = k >= 7 && k < 35;
        const bool is_min = (in_t && (kk & 3) == 2) || k == 38, is_max = (in_t && (kk & 3) == 3) || k == 39;
        const double r = is_min ? wmin_d(acc[k]) : (is_max ? wmax_d(acc[k]) : wsum_d(acc[k]));
        if (lane == 0) partials[(size_t)b * kIlqlStats + k] = r;
    }
}

// ------------------------------------------------------------------------------------------ MC returns
constexpr int kMcStats = 16;   // 0 q_loss 1 cql 2 n_a | 3.. q{sum,sumsq,min,max} 7.. returns{...}
__global__ __launch_bounds__(256) void mc_loss_kernel(const float *__restrict__ q, const float *__restrict__ ce, const float *__restrict__ attn,
                                                      const uint8_t *__restrict__ sta, const float *__restrict__ returns, size_t n_el,
                                                      float cql_w, const double *__restrict__ n_ptr, double *__restrict__ partials,
                                                      float *__restrict__ dq, float *__restrict__ coef_ce) {
    double acc[kMcStats];
#pragma unroll
    for (int k = 0; k < kMcStats; k++) acc[k] = 0.0;
    acc[5] = acc[9] = INFINITY; acc[6] = acc[10] = -INFINITY;
    const float inv_n = (float)(1.0 / *n_ptr);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (size_t)gridDim.x * blockDim.x) {
        const bool a = sta[i] != 0;
        const float mask = (a ? 1.f : 0.f) * attn[i];
        acc[1] += (double)(mask * ce[i]);
        coef_ce[i] = cql_w * mask * inv_n;
        float g = 0.f;
        if (a) {
            const float qs = q[i], rs = returns[i];
            acc[0] += 0.5 * (double)(qs - rs) * (qs - rs); acc[2] += 1.0;
            acc[3] += qs; acc[4] += (double)qs * qs; acc[5] = fmin(acc[5], (double)qs); acc[6] = fmax(acc[6], (double)qs);
            acc[7] += rs; acc[8] += (double)rs * rs; acc[9] = fmin(acc[9], (double)rs); acc[10] = fmax(acc[10], (double)rs);
            g = (qs - rs) * inv_n;
        }
        dq[i] = g;
    }
    __shared__ double red[4][kMcStats];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kMcStats; k++) {
        const bool is_min = (k == 5 || k == 9), is_max = (k == 6 || k == 10);
        const double r = is_min ? wmin_d(acc[k]) : (is_max ? wmax_d(acc[k]) : wsum_d(acc[k]));
        if (lane == 0) red[wave][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < kMcStats) {
        const int k = threadIdx.x;
        const bool is_min = (k == 5 || k == 9), is_max = (k == 6 || k == 10);
        double r = red[0][k];
        for (int w = 1; w < 4; w++) r = is_min ? fmin(r, red[w][k]) : (is_max ? fmax(r, red[w][k]) : r + red[w][k]);
        partials[(size_t)blockIdx.x * kMcStats + k] = r;
    }
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

int lmrl_mask_sum(const uint8_t *sta_d, const float *attn_d, size_t n, double *out_d, void *stream) {
    LMRL_REQUIRE(out_d && (sta_d || attn_d), "lmrl_mask_sum: null pointer");
    LMRL_CHECK_HIP(hipMemsetAsync(out_d, 0, sizeof(double), as_stream(stream)));
    int grid = ceil_div((long)n, 256); if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(mask_sum_kernel, dim3(grid), dim3(256), 0, as_stream(stream), sta_d, attn_d, n, out_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ppo_loss_blocks(size_t n) { long g = (long)((n + 255) / 256); return (int)(g > 256 ? 256 : (g < 1 ? 1 : g)); }
int lmrl_ppo_loss_nstats(void) { return kPpoStats; }
int lmrl_ppo_loss(const float *attn_d, const float *logprobs_d, const float *values_d, const uint8_t *sta_d, const float *old_logprobs_d,
                  const float *old_values_d, const float *old_adv_d, const float *old_ret_d, size_t n, float cliprange_value,
                  float cliprange, float value_loss_coef, const double *n_mask_d, double *partials_d, float *d_logprobs_d,
                  float *d_values_d, void *stream) {
    LMRL_REQUIRE(attn_d && logprobs_d && values_d && sta_d && old_logprobs_d && old_values_d && old_adv_d && old_ret_d && n_mask_d &&
                 partials_d && d_logprobs_d && d_values_d, "lmrl_ppo_loss: null pointer");
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(lmrl_ppo_loss_blocks(n)), dim3(256), 0, as_stream(stream), attn_d, logprobs_d, values_d, sta_d,
                       old_logprobs_d, old_values_d, old_adv_d, old_ret_d, n, cliprange_value, cliprange, value_loss_coef, n_mask_d,
                       partials_d, d_logprobs_d, d_values_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ilql_loss_nstats(void) { return kIlqlStats; }
int lmrl_ilql_loss(const float *q1_d, const float *q2_d, const float *v_d, const float *v_final_d, const float *tq1_d, const float *tq2_d,
                   const float *ce1_d, const float *ce2_d, const float *attn_d, const uint8_t *sta_d, const float *rewards_d, int b, int t1,
                   float gamma, float tau, float cql_weight, const double *n_mask_d, double *partials_d, float *dq1_d, float *dq2_d,
                   float *dv_d, float *coef_ce_d, void *stream) {
    LMRL_REQUIRE(q1_d && q2_d && v_d && v_final_d && tq1_d && tq2_d && ce1_d && ce2_d && attn_d && sta_d && rewards_d && n_mask_d &&
                 partials_d && dq1_d && dq2_d && dv_d && coef_ce_d && b > 0 && t1 > 0, "lmrl_ilql_loss: bad argument");
    const size_t shmem = (size_t)4 * t1 * sizeof(int);
    LMRL_REQUIRE(shmem <= 64 * 1024, "lmrl_ilql_loss: T-1 too large for the LDS compaction buffer");
    hipLaunchKernelGGL(ilql_loss_kernel, dim3(ceil_div(b, 4)), dim3(256), shmem, as_stream(stream), q1_d, q2_d, v_d, v_final_d, tq1_d, tq2_d,
                       ce1_d, ce2_d, attn_d, sta_d, rewards_d, b, t1, gamma, tau, cql_weight, n_mask_d, partials_d, dq1_d, dq2_d, dv_d,
                       coef_ce_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_mc_loss_blocks(size_t n) { return lmrl_ppo_loss_blocks(n); }
int lmrl_mc_loss_nstats(void) { return kMcStats; }
int lmrl_mc_loss(const float *q_d, const float *ce_d, const float *attn_d, const uint8_t *sta_d, const float *returns_d, size_t n,
                 float cql_weight, const double *n_mask_d, double *partials_d, float *dq_d, float *coef_ce_d, void *stream) {
    LMRL_REQUIRE(q_d && ce_d && attn_d && sta_d && returns_d && n_mask_d && partials_d && dq_d && coef_ce_d, "lmrl_mc_loss: null pointer");
    hipLaunchKernelGGL(mc_loss_kernel, dim3(lmrl_mc_loss_blocks(n)), dim3(256), 0, as_stream(stream), q_d, ce_d, attn_d, sta_d, returns_d, n,
                       cql_weight, n_mask_d, partials_d, dq_d, coef_ce_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
