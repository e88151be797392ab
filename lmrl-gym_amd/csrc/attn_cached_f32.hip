// attn_cached_f32.hip — fp32 attention against a persistent fp32 K/V cache: the attention of the fp32 ROLLOUT mode (GPT2EngineF32).
//
// The reference's default rollout arithmetic is float32 (llm_rl_scripts/wordle/bc/eval_bc_gpt2.py:34,69: bf16_activations=False ->
// jnp.float32 params and activations; HF-Flax GPT-2 attention with `init_cache`).  The bf16 engine (csrc/gpt2.hip) is the throughput mode;
// this kernel serves the parity mode in which EVERY sampled token is compared with the float64 oracle, so it is written for exactness
// and clarity first: fp32 operands, fp32 accumulation, one wave per (env, head, new token).
//
// Layout: per layer, kcache / vcache fp32 [B][tmax][H * 64] (token-major, as the bf16 cache); qkv fp32 [B * C][3 * H * 64] holds the C new
// tokens' rows of env b at rows b * C .. b * C + cnt[b] - 1 (slots beyond cnt[b] are padding and are skipped).
// Query j of env b sees the cached positions [0, len[b]) and the new tokens [0, j] of this chunk (causal); its own K / V row is appended
// to the cache at position len[b] + j.  Keys of the chunk are read from `qkv` (not from the cache), so no wave waits for another.
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

constexpr int kDh = 64;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void attn_cached_f32_kernel(const float *__restrict__ qkv, float *__restrict__ kcache, float *__restrict__ vcache,
                                                              const int32_t *__restrict__ len, const int32_t *__restrict__ cnt,
                                                              float *__restrict__ out, int B, int C, int H, int tmax) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // wave id = ((b * C) + j) * H + h
    if (w >= (long)B * C * H) return;
    const int h = (int)(w % H);
    const int j = (int)((w / H) % C);
    const int b = (int)(w / ((long)H * C));
    if (j >= cnt[b]) return;                                           // padding slot (wave-uniform)
    const int d = H * kDh, L = len[b];
    const float *row = qkv + ((size_t)b * C + j) * 3 * d;
    const float *q = row + h * kDh;
    // append this token's K / V rows (lane = dim: one coalesced 256-byte store each)
    const size_t crow = ((size_t)b * tmax + (L + j)) * d + h * kDh;
    kcache[crow + lane] = row[d + h * kDh + lane];
    vcache[crow + lane] = row[2 * d + h * kDh + lane];
    // ---- scores: lane p handles positions p, p + 64, ... of the n = L + j + 1 visible keys
    const int n = L + j + 1;
    float qreg[kDh];
#pragma unroll
    for (int i = 0; i < kDh; i += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(q + i);
        qreg[i] = t.x; qreg[i + 1] = t.y; qreg[i + 2] = t.z; qreg[i + 3] = t.w;
    }
    constexpr int kMaxIter = 16;                                        // up to 1024 visible positions
    float sc[kMaxIter];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < kMaxIter; it++) {
        const int p = it * 64 + lane;
        float s = -INFINITY;
        if (it * 64 < n && p < n) {
            const float *kr = p < L ? kcache + ((size_t)b * tmax + p) * d + h * kDh : qkv + ((size_t)b * C + (p - L)) * 3 * d + d + h * kDh;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < kDh; i += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(kr + i);
                acc = fmaf(qreg[i], t.x, acc); acc = fmaf(qreg[i + 1], t.y, acc); acc = fmaf(qreg[i + 2], t.z, acc); acc = fmaf(qreg[i + 3], t.w, acc);
            }
            s = acc * 0.125f;                                           // 1 / sqrt(64)
        }
        sc[it] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxIter; it++) {
        const float e = (sc[it] == -INFINITY) ? 0.f : expf(sc[it] - mx);
        sc[it] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    // ---- output: lane = dim; probabilities broadcast position by position
    float o = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxIter; it++) {
        if (it * 64 >= n) break;                                        // wave-uniform
        const int lim = min(64, n - it * 64);
        for (int pl = 0; pl < lim; pl++) {
            const float pr = __shfl(sc[it], pl);
            const int p = it * 64 + pl;
            const float *vr = p < L ? vcache + ((size_t)b * tmax + p) * d + h * kDh : qkv + ((size_t)b * C + (p - L)) * 3 * d + 2 * d + h * kDh;
            o = fmaf(pr, vr[lane], o);
        }
    }
    out[((size_t)b * C + j) * d + h * kDh + lane] = o * inv;
}

// last valid row of every env's chunk -> dst[b] (envs with cnt == 0 keep their previous row)
__global__ void gather_last_f32_kernel(const float *__restrict__ x, const int32_t *__restrict__ cnt, float *__restrict__ dst, int B, int C, int d) {
    const int b = blockIdx.x;
    const int c = cnt[b];
    if (c <= 0) return;
    const float *src = x + ((size_t)b * C + (c - 1)) * d;
    for (int i = threadIdx.x; i < d; i += blockDim.x) dst[(size_t)b * d + i] = src[i];
}

// pos[b * C + j] = len[b] + j (clamped to n_pos - 1 on padding slots), ids of padding slots -> 0; then len[b] += cnt[b] is done by
// advance_len_kernel AFTER the layers ran
__global__ void chunk_positions_kernel(const int32_t *__restrict__ len, const int32_t *__restrict__ cnt, int32_t *__restrict__ ids,
                                       int32_t *__restrict__ pos, int B, int C, int n_pos) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B * C) return;
    const int b = r / C, j = r % C;
    const bool live = j < cnt[b];
    int p = len[b] + j;
    pos[r] = (live && p < n_pos) ? p : 0;
    if (!live) ids[r] = 0;
}
__global__ void advance_len_kernel(int32_t *__restrict__ len, const int32_t *__restrict__ cnt, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) len[b] += cnt[b];
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

int lmrl_attn_cached_f32(const float *qkv_d, float *kcache_d, float *vcache_d, const int32_t *len_d, const int32_t *cnt_d, float *out_d, int b, int c,
                         int n_head, int tmax, void *stream) {
    LMRL_REQUIRE(qkv_d && kcache_d && vcache_d && len_d && cnt_d && out_d && b > 0 && c > 0 && n_head > 0 && tmax > 0 && tmax <= 1024,
                 "lmrl_attn_cached_f32: bad argument (tmax <= 1024)");
    const long waves = (long)b * c * n_head;
    hipLaunchKernelGGL(attn_cached_f32_kernel, dim3(ceil_div(waves, 4)), dim3(256), 0, as_stream(stream), qkv_d, kcache_d, vcache_d, len_d, cnt_d, out_d,
                       b, c, n_head, tmax);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_chunk_begin_f32(const int32_t *len_d, const int32_t *cnt_d, int32_t *ids_d, int32_t *pos_d, int b, int c, int n_pos, void *stream) {
    LMRL_REQUIRE(len_d && cnt_d && ids_d && pos_d && b > 0 && c > 0, "lmrl_chunk_begin_f32: bad argument");
    hipLaunchKernelGGL(chunk_positions_kernel, dim3(ceil_div((long)b * c, 256)), dim3(256), 0, as_stream(stream), len_d, cnt_d, ids_d, pos_d, b, c, n_pos);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_chunk_end_f32(const float *x_d, const int32_t *cnt_d, float *last_d, int32_t *len_d, int b, int c, int d, void *stream) {
    LMRL_REQUIRE(x_d && cnt_d && last_d && len_d && b > 0 && c > 0 && d > 0, "lmrl_chunk_end_f32: bad argument");
    hipLaunchKernelGGL(gather_last_f32_kernel, dim3(b), dim3(256), 0, as_stream(stream), x_d, cnt_d, last_d, b, c, d);
    hipLaunchKernelGGL(advance_len_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, as_stream(stream), len_d, cnt_d, b);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

}  // extern "C"
