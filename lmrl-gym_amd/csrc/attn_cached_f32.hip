// attn_cached_f32.hip — fp32 attention against a persistent fp32 K/V cache: the attention of the fp32 ROLLOUT mode (GPT2EngineF32).
//
// The reference's default rollout arithmetic is float32 (llm_rl_scripts/wordle/bc/eval_bc_gpt2.py:34,69: bf16_activations=False ->
// jnp.float32 params and activations; HF-Flax GPT-2 attention with `init_cache`).  The bf16 engine (csrc/gpt2.hip) is the throughput mode;
// this kernel serves the parity mode in which EVERY sampled token is compared with the float64 oracle: fp32 operands, fp32 accumulation,
// one wave per (env, head, new token).
//
// Layout: per layer, kcache / vcache fp32 [B][tmax][H * 64] (token-major, as the bf16 cache); qkv fp32 [B * C][3 * H * 64] holds the C new
// tokens' rows of env b at rows b * C .. b * C + cnt[b] - 1 (slots beyond cnt[b] are padding and are skipped).
// Query j of env b sees the cached positions [0, len[b]) and the new tokens [0, j] of this chunk (causal); its own K / V row is appended
// to the cache at position len[b] + j.  Keys of the chunk are read from `qkv` (not from the cache), so no wave waits for another.
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

constexpr int kDh = 64;

// One wave per (env, head, new token).  A wave instruction covers 4 key positions x 256 contiguous bytes: lane l = (g = l >> 4: position slot,
// c = l & 15: float4 column of the 64-float row), so K / V rows are read as coalesced 256-byte segments.  Scores: 4 fmaf per lane, summed over the
// 16 lanes of a slot (xor 1, 2, 4, 8); online softmax per slot (running max / sum / rescaled output, as the bf16 decode kernel), the 4 slots
// merged once at the end (xor 16, 32).  fp32 throughout.
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
    const int g = lane >> 4, c = lane & 15;
    const float *row = qkv + ((size_t)b * C + j) * 3 * d;
    const float4 q4 = *reinterpret_cast<const float4 *>(row + h * kDh + c * 4);
    // append this token's K / V rows (lane = dim: one coalesced 256-byte store each)
    const size_t crow = ((size_t)b * tmax + (L + j)) * d + h * kDh;
    kcache[crow + lane] = row[d + h * kDh + lane];
    vcache[crow + lane] = row[2 * d + h * kDh + lane];
    const int n = L + j + 1;                                           // visible keys: cached [0, L) + this chunk's tokens [0, j]
    float m = -INFINITY, l = 0.f;
    float4 o = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < n; p0 += 4) {                                // wave-uniform trip count
        const int p = p0 + g;
        const bool ok = p < n;
        const int pc = ok ? p : n - 1;
        const float *kr = pc < L ? kcache + ((size_t)b * tmax + pc) * d + h * kDh : qkv + ((size_t)b * C + (pc - L)) * 3 * d + d + h * kDh;
        const float *vr = pc < L ? vcache + ((size_t)b * tmax + pc) * d + h * kDh : qkv + ((size_t)b * C + (pc - L)) * 3 * d + 2 * d + h * kDh;
        const float4 k4 = *reinterpret_cast<const float4 *>(kr + c * 4);
        const float4 v4 = *reinterpret_cast<const float4 *>(vr + c * 4);
        float s = fmaf(q4.x, k4.x, fmaf(q4.y, k4.y, fmaf(q4.z, k4.z, q4.w * k4.w)));
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        s *= 0.125f;                                                    // 1 / sqrt(64)
        const float m_new = ok ? fmaxf(m, s) : m;
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - m_new);
        const float pr = ok ? expf(s - m_new) : 0.f;
        l = l * alpha + pr;
        o.x = fmaf(pr, v4.x, o.x * alpha); o.y = fmaf(pr, v4.y, o.y * alpha); o.z = fmaf(pr, v4.z, o.z * alpha); o.w = fmaf(pr, v4.w, o.w * alpha);
        m = m_new;
    }
    // merge the 4 position slots
    float mm = fmaxf(m, __shfl_xor(m, 16));
    mm = fmaxf(mm, __shfl_xor(mm, 32));
    const float wgt = (m == -INFINITY) ? 0.f : expf(m - mm);
    float lt = l * wgt;
    lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
    float4 r = {o.x * wgt, o.y * wgt, o.z * wgt, o.w * wgt};
    r.x += __shfl_xor(r.x, 16); r.x += __shfl_xor(r.x, 32);
    r.y += __shfl_xor(r.y, 16); r.y += __shfl_xor(r.y, 32);
    r.z += __shfl_xor(r.z, 16); r.z += __shfl_xor(r.z, 32);
    r.w += __shfl_xor(r.w, 16); r.w += __shfl_xor(r.w, 32);
    if (g == 0) {
        const float inv = 1.f / lt;
        *reinterpret_cast<float4 *>(out + ((size_t)b * C + j) * d + h * kDh + c * 4) = float4{r.x * inv, r.y * inv, r.z * inv, r.w * inv};
    }
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
    LMRL_REQUIRE(qkv_d && kcache_d && vcache_d && len_d && cnt_d && out_d && b > 0 && c > 0 && n_head > 0 && tmax > 0,
                 "lmrl_attn_cached_f32: bad argument");
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
