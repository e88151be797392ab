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
#include "gemm_bf16.h"      // pack_bf16x2 (v_cvt_pk_bf16_f32)

namespace lmrl {

constexpr int kDh = 64;

// bf16x3 mode: the attention output also as the three-term split operand of the projection GEMM — row r of `split` = [hi | lo | hi], each d wide
// (lmrl_split3_bf16's layout and rounding: hi = bf16(x) RNE, lo = bf16(x - hi)) — written by the lanes that hold the output, so no split pass
// re-reads the fp32 tensor.
__device__ __forceinline__ void store_split3(uint16_t *__restrict__ split_row, int d, int col, float4 v) {
    const uint32_t h0 = pack_bf16x2(v.x, v.y), h1 = pack_bf16x2(v.z, v.w);
    const uint32_t l0 = pack_bf16x2(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
    const uint32_t l1 = pack_bf16x2(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
    *reinterpret_cast<uint2 *>(split_row + col) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(split_row + d + col) = make_uint2(l0, l1);
    *reinterpret_cast<uint2 *>(split_row + 2 * d + col) = make_uint2(h0, h1);
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true)); }

// One wave per (env, head, new token).  A wave instruction covers 4 key positions x 256 contiguous bytes: lane l = (g = l >> 4: position slot,
// c = l & 15: float4 column of the 64-float row), so K / V rows are read as coalesced 256-byte segments.  Scores: 4 fmaf per lane, summed over the
// 16 lanes of a slot (xor 1, 2, 4, 8); online softmax per slot (running max / sum / rescaled output, as the bf16 decode kernel), the 4 slots
// merged once at the end (xor 16, 32).  fp32 throughout.
__global__ __launch_bounds__(256) void attn_cached_f32_kernel(const float *__restrict__ qkv, float *__restrict__ kcache, float *__restrict__ vcache,
                                                              const int32_t *__restrict__ len, const int32_t *__restrict__ cnt,
                                                              float *__restrict__ out, int B, int C, int H, int tmax, uint16_t *__restrict__ split) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // wave id = ((b * C) + j) * H + h
    if (w >= (long)B * C * H) return;
    const int h = (int)(w % H);
    const int j = (int)((w / H) % C);
    const int b = (int)(w / ((long)H * C));
    const int d = H * kDh;
    const int g = lane >> 4, c = lane & 15;
    if (j >= cnt[b]) {                                                 // padding slot (wave-uniform): `out` untouched; the split operand of a padding
        if (split && g == 0)                                           // row is zeroed (the buffer is shared by operands of different row pitches)
            store_split3(split + ((size_t)b * C + j) * 3 * d, d, h * kDh + c * 4, float4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const int L = len[b];
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
        const float4 res = float4{r.x * inv, r.y * inv, r.z * inv, r.w * inv};
        *reinterpret_cast<float4 *>(out + ((size_t)b * C + j) * d + h * kDh + c * 4) = res;
        if (split) store_split3(split + ((size_t)b * C + j) * 3 * d, d, h * kDh + c * 4, res);
    }
}

// Chunk forwards (C = 8 / 16 new tokens per env), one wave per (env, head) for ALL of the chunk's queries: every K / V row is loaded once and
// scored against the C queries held in registers (the one-wave-per-query kernel above re-reads the whole context per query: 208 us per launch
// at B = 1024, C = 8, vs ~1/6 of that here).  Same lane layout (g = position slot, c = float4 column), per-query online softmax per slot, slots
// merged at the end; keys of the chunk itself come from `qkv`, causal (query j sees positions <= L + j); new K / V rows appended first.
template <int C>
__global__ __launch_bounds__(256) void attn_chunk_f32_kernel(const float *__restrict__ qkv, float *__restrict__ kcache, float *__restrict__ vcache,
                                                             const int32_t *__restrict__ len, const int32_t *__restrict__ cnt,
                                                             float *__restrict__ out, int B, int H, int tmax, uint16_t *__restrict__ split) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);                 // wave id = b * H + h
    if (w >= B * H) return;
    const int b = w / H, h = w - b * H;
    const int g = lane >> 4, c = lane & 15;
    const int d = H * kDh;
    const int nq = min(cnt[b], C);                                     // wave-uniform
    const float *rows = qkv + (size_t)b * C * 3 * d + h * kDh + c * 4;
    if (split && g == 0)                                               // padding slots: zero split rows (shared buffer, see the kernel above)
        for (int j = max(nq, 0); j < C; j++) store_split3(split + ((size_t)b * C + j) * 3 * d, d, h * kDh + c * 4, float4{0.f, 0.f, 0.f, 0.f});
    if (nq <= 0) return;
    const int L = len[b];
    float4 q[C];
#pragma unroll
    for (int j = 0; j < C; j++) {
        q[j] = *reinterpret_cast<const float4 *>(rows + (size_t)(j < nq ? j : 0) * 3 * d);
        q[j].x *= 0.125f; q[j].y *= 0.125f; q[j].z *= 0.125f; q[j].w *= 0.125f;
    }
    float *kc = kcache + (size_t)b * tmax * d + h * kDh + c * 4;
    float *vc = vcache + (size_t)b * tmax * d + h * kDh + c * 4;
    if (g == 0)                                                        // append the chunk's K / V rows (read back below only from `qkv`)
        for (int j = 0; j < nq; j++)
            if (L + j < tmax) {
                *reinterpret_cast<float4 *>(kc + (size_t)(L + j) * d) = *reinterpret_cast<const float4 *>(rows + (size_t)j * 3 * d + d);
                *reinterpret_cast<float4 *>(vc + (size_t)(L + j) * d) = *reinterpret_cast<const float4 *>(rows + (size_t)j * 3 * d + 2 * d);
            }
    float m[C], l[C];
    float4 o[C];
#pragma unroll
    for (int j = 0; j < C; j++) { m[j] = -1e30f; l[j] = 0.f; o[j] = float4{0.f, 0.f, 0.f, 0.f}; }
    const int n = L + nq;                                              // visible positions of the LAST query
    constexpr int U = 4;                                               // 16 positions requested per batch
    for (int t0 = 0; t0 < n; t0 += 4 * U) {
        float4 kr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (t0 + u * 4 < n) {
                const int t = min(t0 + u * 4 + g, n - 1);
                const float *kp = t < L ? kc + (size_t)t * d : rows + (size_t)(t - L) * 3 * d + d;
                const float *vp = t < L ? vc + (size_t)t * d : rows + (size_t)(t - L) * 3 * d + 2 * d;
                kr[u] = *reinterpret_cast<const float4 *>(kp);
                vr[u] = *reinterpret_cast<const float4 *>(vp);
            }
        // per query: the batch's U scores first, then ONE rescale of the running state (U + 1 exps per U positions instead of 2 U)
#pragma unroll
        for (int j = 0; j < C; j++) {
            if (j < nq) {                                              // wave-uniform
                float sc[U];
                float mb = m[j];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    sc[u] = -1e30f;
                    if (t0 + u * 4 < n) {
                        const int t = t0 + u * 4 + g;
                        float s = fmaf(q[j].x, kr[u].x, fmaf(q[j].y, kr[u].y, fmaf(q[j].z, kr[u].z, q[j].w * kr[u].w)));
                        s += dpp_mov<0xB1>(s); s += dpp_mov<0x4E>(s); s += dpp_mov<0x141>(s); s += dpp_mov<0x140>(s);
                        if (t <= L + j && t < n) { sc[u] = s; mb = fmaxf(mb, s); }          // causal
                    }
                }
                const float alpha = expf(m[j] - mb);
                l[j] *= alpha; o[j].x *= alpha; o[j].y *= alpha; o[j].z *= alpha; o[j].w *= alpha;
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (t0 + u * 4 < n) {
                        const float pr = sc[u] > -1e29f ? expf(sc[u] - mb) : 0.f;
                        l[j] += pr;
                        o[j].x = fmaf(pr, vr[u].x, o[j].x); o[j].y = fmaf(pr, vr[u].y, o[j].y);
                        o[j].z = fmaf(pr, vr[u].z, o[j].z); o[j].w = fmaf(pr, vr[u].w, o[j].w);
                    }
                m[j] = mb;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < C; j++) {
        if (j >= nq) continue;                                         // wave-uniform
        float mm = fmaxf(m[j], __shfl_xor(m[j], 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        const float wgt = expf(m[j] - mm);
        float lt = l[j] * wgt;
        lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
        float4 r = {o[j].x * wgt, o[j].y * wgt, o[j].z * wgt, o[j].w * wgt};
        r.x += __shfl_xor(r.x, 16); r.x += __shfl_xor(r.x, 32);
        r.y += __shfl_xor(r.y, 16); r.y += __shfl_xor(r.y, 32);
        r.z += __shfl_xor(r.z, 16); r.z += __shfl_xor(r.z, 32);
        r.w += __shfl_xor(r.w, 16); r.w += __shfl_xor(r.w, 32);
        if (g == 0) {
            const float inv = 1.f / lt;
            const float4 res = float4{r.x * inv, r.y * inv, r.z * inv, r.w * inv};
            *reinterpret_cast<float4 *>(out + ((size_t)b * C + j) * d + h * kDh + c * 4) = res;
            if (split) store_split3(split + ((size_t)b * C + j) * 3 * d, d, h * kDh + c * 4, res);
        }
    }
}

// Single-token decode (C = 1), one wave per (env, head), the fp32 twin of gpt2.hip's attention_decode_kernel: the scalar length load, then a
// BATCH of vector loads covering up to 4 U cached positions (a wave-uniform number of 4-position blocks: lane (g = lane >> 4, c = lane & 15)
// holds the float4 column c of position t0 + 4 u + g, a wave instruction covers 4 rows x 256 contiguous bytes) before any arithmetic — the
// per-position loop of the chunk kernel above pays one dependent memory latency per 4 positions.  Scores: 4 fmaf + a 16-lane DPP sum; softmax
// per position slot g with ONE rescale per batch (U + 1 exps per U positions instead of 2 U), slots merged once at the end.  The new token's own
// K / V row comes from `qkv`, is attended last by slot 0 and appended to the cache behind the load stream.  fp32 throughout, accurate expf.
template <int U>
__global__ __launch_bounds__(256) void attn_decode_f32_kernel(const float *__restrict__ qkv, float *__restrict__ kcache, float *__restrict__ vcache,
                                                              const int32_t *__restrict__ len, const int32_t *__restrict__ cnt,
                                                              float *__restrict__ out, int B, int H, int tmax, uint16_t *__restrict__ split) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);                 // wave id = b * H + h
    if (w >= B * H) return;
    const int b = w / H, h = w - b * H;
    const int g = lane >> 4, c = lane & 15;
    const int d = H * kDh;
    const float *row = qkv + (size_t)b * 3 * d + h * kDh + c * 4;
    float4 q4 = *reinterpret_cast<const float4 *>(row);
    const float4 knew = *reinterpret_cast<const float4 *>(row + d), vnew = *reinterpret_cast<const float4 *>(row + 2 * d);
    if (cnt[b] <= 0) {                                                 // wave-uniform: finished env (`out` untouched, its split row zeroed)
        if (split && g == 0) store_split3(split + (size_t)b * 3 * d, d, h * kDh + c * 4, float4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const int L = len[b];
    q4.x *= 0.125f; q4.y *= 0.125f; q4.z *= 0.125f; q4.w *= 0.125f;    // 1 / sqrt(64): a power of two, exact
    const float *kc = kcache + (size_t)b * tmax * d + h * kDh + c * 4;
    const float *vc = vcache + (size_t)b * tmax * d + h * kDh + c * 4;
    float m = -1e30f, l = 0.f;
    float4 o = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < L; t0 += 4 * U) {
        float4 kr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (t0 + u * 4 < L) {                                      // wave-uniform: block u holds cached positions
                const int t = t0 + u * 4 + g;
                const size_t ro = (size_t)(t < L ? t : L - 1) * d;
                // (streaming loads: every cached row is read by ONE wave per decode step — as in the bf16 engine's decode attention, csrc/gpt2.hip)
                typedef float f32x4_nt __attribute__((ext_vector_type(4)));
                kr[u] = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(kc + ro)));
                vr[u] = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(vc + ro)));
            }
        float sc[U];
        float mb = m;
#pragma unroll
        for (int u = 0; u < U; u++) {
            sc[u] = -1e30f;
            if (t0 + u * 4 < L) {
                float s = fmaf(q4.x, kr[u].x, fmaf(q4.y, kr[u].y, fmaf(q4.z, kr[u].z, q4.w * kr[u].w)));
                s += dpp_mov<0xB1>(s);       // quad_perm [1,0,3,2]
                s += dpp_mov<0x4E>(s);       // quad_perm [2,3,0,1]
                s += dpp_mov<0x141>(s);      // row_half_mirror: the other quad of the 8-lane half
                s += dpp_mov<0x140>(s);      // row_mirror: the other half of the 16-lane row
                sc[u] = (t0 + u * 4 + g < L) ? s : -1e30f;
                mb = fmaxf(mb, sc[u]);
            }
        }
        const float alpha = expf(m - mb);
        l *= alpha; o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (t0 + u * 4 < L) {
                const float pr = (t0 + u * 4 + g < L) ? expf(sc[u] - mb) : 0.f;
                l += pr;
                o.x = fmaf(pr, vr[u].x, o.x); o.y = fmaf(pr, vr[u].y, o.y); o.z = fmaf(pr, vr[u].z, o.z); o.w = fmaf(pr, vr[u].w, o.w);
            }
        m = mb;
    }
    {   // position L (this step's own token): slot 0 attends it last
        float s = fmaf(q4.x, knew.x, fmaf(q4.y, knew.y, fmaf(q4.z, knew.z, q4.w * knew.w)));
        s += dpp_mov<0xB1>(s); s += dpp_mov<0x4E>(s); s += dpp_mov<0x141>(s); s += dpp_mov<0x140>(s);
        const bool ok = g == 0;
        const float mb = ok ? fmaxf(m, s) : m;
        const float alpha = expf(m - mb), pr = ok ? expf(s - mb) : 0.f;
        l = l * alpha + pr;
        o.x = fmaf(pr, vnew.x, o.x * alpha); o.y = fmaf(pr, vnew.y, o.y * alpha); o.z = fmaf(pr, vnew.z, o.z * alpha); o.w = fmaf(pr, vnew.w, o.w * alpha);
        m = mb;
    }
    // merge the 4 position slots
    float mm = fmaxf(m, __shfl_xor(m, 16));
    mm = fmaxf(mm, __shfl_xor(mm, 32));
    const float wgt = expf(m - mm);
    float lt = l * wgt;
    lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
    float4 r = {o.x * wgt, o.y * wgt, o.z * wgt, o.w * wgt};
    r.x += __shfl_xor(r.x, 16); r.x += __shfl_xor(r.x, 32);
    r.y += __shfl_xor(r.y, 16); r.y += __shfl_xor(r.y, 32);
    r.z += __shfl_xor(r.z, 16); r.z += __shfl_xor(r.z, 32);
    r.w += __shfl_xor(r.w, 16); r.w += __shfl_xor(r.w, 32);
    if (g == 0) {
        const float inv = 1.f / lt;
        const float4 res = float4{r.x * inv, r.y * inv, r.z * inv, r.w * inv};
        *reinterpret_cast<float4 *>(out + (size_t)b * d + h * kDh + c * 4) = res;
        if (split) store_split3(split + (size_t)b * 3 * d, d, h * kDh + c * 4, res);
        if (L < tmax) {                                                // append the new token's K / V row, behind the wave's load stream
            *reinterpret_cast<float4 *>(const_cast<float *>(kc) + (size_t)L * d) = knew;
            *reinterpret_cast<float4 *>(const_cast<float *>(vc) + (size_t)L * d) = vnew;
        }
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

int lmrl_attn_cached_f32_split3(const float *qkv_d, float *kcache_d, float *vcache_d, const int32_t *len_d, const int32_t *cnt_d, float *out_d,
                                void *split3_out_d, int b, int c, int n_head, int tmax, void *stream) {
    LMRL_REQUIRE(qkv_d && kcache_d && vcache_d && len_d && cnt_d && out_d && b > 0 && c > 0 && n_head > 0 && tmax > 0,
                 "lmrl_attn_cached_f32: bad argument");
    const long waves = (long)b * c * n_head;
    if (c == 1)      // single-token decode: the batched-load kernel (32 cached positions requested per wave before any arithmetic)
        hipLaunchKernelGGL(attn_decode_f32_kernel<8>, dim3(ceil_div(waves, 4)), dim3(256), 0, as_stream(stream), qkv_d, kcache_d, vcache_d, len_d, cnt_d,
                           out_d, b, n_head, tmax, (uint16_t *)split3_out_d);
    else if (c == 8)      // the per-turn chunk: one wave per (env, head) scores every K / V row against all 8 queries
        hipLaunchKernelGGL(attn_chunk_f32_kernel<8>, dim3(ceil_div((long)b * n_head, 4)), dim3(256), 0, as_stream(stream), qkv_d, kcache_d, vcache_d, len_d,
                           cnt_d, out_d, b, n_head, tmax, (uint16_t *)split3_out_d);
    else if (c == 16)
        hipLaunchKernelGGL(attn_chunk_f32_kernel<16>, dim3(ceil_div((long)b * n_head, 4)), dim3(256), 0, as_stream(stream), qkv_d, kcache_d, vcache_d, len_d,
                           cnt_d, out_d, b, n_head, tmax, (uint16_t *)split3_out_d);
    else
        hipLaunchKernelGGL(attn_cached_f32_kernel, dim3(ceil_div(waves, 4)), dim3(256), 0, as_stream(stream), qkv_d, kcache_d, vcache_d, len_d, cnt_d,
                           out_d, b, c, n_head, tmax, (uint16_t *)split3_out_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_attn_cached_f32(const float *qkv_d, float *kcache_d, float *vcache_d, const int32_t *len_d, const int32_t *cnt_d, float *out_d, int b, int c,
                         int n_head, int tmax, void *stream) {
    return lmrl_attn_cached_f32_split3(qkv_d, kcache_d, vcache_d, len_d, cnt_d, out_d, nullptr, b, c, n_head, tmax, stream);
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
