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

// LDS per wave: pos[L] (int), acc[L] (float), aux[L] (float)
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
__global__ __launch_bounds__(256) void whiten_moments_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                              double *moments, size_t n) {
    double s = 0.0, ss = 0.0, cnt = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (!mask || mask[i]) {
            const double v = (double)x[i];
            s += v; ss += v * v; cnt += 1.0;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_down(s, d); ss += __shfl_down(ss, d); cnt += __shfl_down(cnt, d);
    }
    __shared__ double red[3][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; red[2][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {   // per-workgroup partial; whiten_finish_kernel adds them in workgroup order (deterministic)
        double a = 0, b2 = 0, c2 = 0;
        for (int w = 0; w < 4; w++) { a += red[0][w]; b2 += red[1][w]; c2 += red[2][w]; }
        moments[3 * blockIdx.x + 0] = a;
        moments[3 * blockIdx.x + 1] = b2;
        moments[3 * blockIdx.x + 2] = c2;
    }
}

__global__ __launch_bounds__(64) void whiten_finish_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ moments) {
    double s = 0.0, ss = 0.0, cnt = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) { s += partials[3 * b]; ss += partials[3 * b + 1]; cnt += partials[3 * b + 2]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_down(s, d); ss += __shfl_down(ss, d); cnt += __shfl_down(cnt, d);
    }
    if (threadIdx.x == 0) { moments[0] = s; moments[1] = ss; moments[2] = cnt; }
}

__global__ __launch_bounds__(256) void whiten_apply_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                            const double *__restrict__ moments, float *__restrict__ y,
                                                            size_t n, int shift_mean) {
    const double cnt = moments[2];
    const double mean = cnt > 0 ? moments[0] / cnt : 0.0;
    double var = cnt > 0 ? moments[1] / cnt - mean * mean : 0.0;   // population variance (jnp.var)
    if (var < 0) var = 0;
    const double inv = 1.0 / sqrt(var + 1e-8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (!mask || mask[i]) {
            double w = ((double)v - mean) * inv;
            if (!shift_mean) w += mean;
            y[i] = (float)w;
        } else {
            y[i] = v;
        }
    }
}

static int scan_launch(bool gae, const float *values, const float *rewards, const uint8_t *sta, const int32_t *lens,
                       float *o0, float *o1, int b, int l, float gamma, float lam, void *stream) {
    const size_t shmem = (size_t)4 * 3 * l * sizeof(float);
    if (shmem > 160 * 1024) {
        set_error("chain scan: L=%d needs %zu B of LDS per workgroup (> 160 KiB)", l, shmem);
        return LMRL_ERR_ARG;
    }
    if (gae) {
        if (shmem > 64 * 1024)
            LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_scan_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(chain_scan_kernel<true>, dim3(ceil_div(b, 4)), dim3(256), shmem, as_stream(stream), values,
                           rewards, sta, lens, o0, o1, b, l, gamma, lam);
    } else {
        if (shmem > 64 * 1024)
            LMRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&chain_scan_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(chain_scan_kernel<false>, dim3(ceil_div(b, 4)), dim3(256), shmem, as_stream(stream), values,
                           rewards, sta, lens, o0, o1, b, l, gamma, lam);
    }
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

int lmrl_gae(const float *values_d, const float *rewards_d, const uint8_t *sta_d, const int32_t *len_d, float *adv_d,
             float *ret_d, int b, int l, float gamma, float lam, void *stream) {
    LMRL_REQUIRE(values_d && rewards_d && sta_d && adv_d && ret_d && b >= 0 && l > 0, "lmrl_gae: bad argument");
    if (b == 0) return LMRL_OK;
    return scan_launch(true, values_d, rewards_d, sta_d, len_d, adv_d, ret_d, b, l, gamma, lam, stream);
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
    if (!partials) LMRL_CHECK_HIP(hipMalloc(&partials, (size_t)kMaxBlocks * 3 * sizeof(double)));
    int grid = ceil_div((long)n, 256 * 8);
    if (grid > kMaxBlocks) grid = kMaxBlocks;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(whiten_moments_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, partials, n);
    hipLaunchKernelGGL(whiten_finish_kernel, dim3(1), dim3(64), 0, as_stream(stream), partials, grid, moments_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_whiten_apply(const float *x_d, const uint8_t *mask_d, const double *moments_d, float *y_d, size_t n,
                      int shift_mean, void *stream) {
    LMRL_REQUIRE(x_d && moments_d && y_d, "lmrl_whiten_apply: null pointer");
    if (n == 0) return LMRL_OK;
    int grid = ceil_div((long)n, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(whiten_apply_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x_d, mask_d, moments_d, y_d, n,
                       shift_mean);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
