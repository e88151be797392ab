// train_ops.hip — fp32 building blocks of the PPO / ILQL / BC train step on gfx950 (everything except the GEMMs, which
// are sgemm_f32.hip): embeddings, LayerNorm, gelu_new, causal softmax, log-softmax/cross-entropy over the vocabulary,
// column reductions for bias/LN gradients, AdamW, Polyak averaging.  All are HBM-bound streaming kernels: one wave per
// row (16-byte accesses, shuffle reductions) or grid-stride elementwise.
//
// They restate, op for op, what jax/flax/optax execute for the reference's `_step` functions
// (LLM_RL/algorithms/ppo/gpt2/interface.py:72-211, LLM_RL/algorithms/ilql/gpt2/interface.py:88-367): HF-Flax GPT-2
// block (pre-LN, gelu_new, causal attention in fp32), optax.softmax_cross_entropy_with_integer_labels,
// optax.adamw(b1, b2, eps, weight_decay) and optax.incremental_update.
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

typedef __attribute__((ext_vector_type(4))) float v4f;

// fp32 -> bf16, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {f, 0.f};
    return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2)) & 0xffffu);
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}

// ------------------------------------------------------------------------------------------ embeddings
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float *__restrict__ wte, const float *__restrict__ wpe,
                                                        const int32_t *__restrict__ ids, const int32_t *__restrict__ pos,
                                                        float *__restrict__ x, int R, int d, int vocab) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    // ids outside [0, vocab) — the pad id of a tokenizer whose `<|pad|>` is the first id AFTER the model's vocabulary (train_ppo_gpt2.py:124-126; the
    // reference resizes the embedding and masks those logits to -inf, ppo/gpt2/interface.py:330) — embed as a zero row and receive no gradient
    const int id = ids[r];
    const bool in = vocab <= 0 || (id >= 0 && id < vocab);
    const float *te = wte + (size_t)(in ? id : 0) * d, *pe = wpe + (size_t)pos[r] * d;
    for (int c = lane * 4; c < d; c += 256) {
        const v4f e = in ? *reinterpret_cast<const v4f *>(te + c) : v4f{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<v4f *>(x + (size_t)r * d + c) = e + *reinterpret_cast<const v4f *>(pe + c);
    }
}
// Row compaction for the vocabulary-wide heads: the RL losses read the Q / policy logits only on rows whose mask is set (should_take_action x
// attention mask), so the heads run on the gathered rows and their input gradient is scattered back.  idx holds DISTINCT rows: no atomics.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, float *__restrict__ dst,
                                                          int n, int d) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float *p = src + (size_t)idx[i] * d;
    if ((d & 3) == 0) for (int c = lane * 4; c < d; c += 256) *reinterpret_cast<v4f *>(dst + (size_t)i * d + c) = *reinterpret_cast<const v4f *>(p + c);
    else for (int c = lane; c < d; c += 64) dst[(size_t)i * d + c] = p[c];
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, float *dst, int n, int d,
                                                           int accumulate) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float *q = dst + (size_t)idx[i] * d;
    const float *p = src + (size_t)i * d;
    if ((d & 3) == 0) {
        for (int c = lane * 4; c < d; c += 256) {
            v4f v = *reinterpret_cast<const v4f *>(p + c);
            if (accumulate) v += *reinterpret_cast<const v4f *>(q + c);
            *reinterpret_cast<v4f *>(q + c) = v;
        }
    } else {
        for (int c = lane; c < d; c += 64) q[c] = accumulate ? q[c] + p[c] : p[c];
    }
}
// Embedding gradients WITHOUT atomics (bit-reproducible): one wave per row r of one table (blockIdx.y: 0 token ids -> dwte, 1 positions -> dwpe).
// The wave owns index ids[r] iff r is its FIRST occurrence (a ballot scan over the earlier rows; the id array is 64 KB and lives in L2); the owner
// then adds the rows that carry the same index in increasing row order and is the only writer of that table row.
// `live` (optional, uint8 [R]): rows whose flag is 0 take no part — the padded positions of a right-padded batch (attention_mask == 0).  Their dx
// is exactly zero (masked as keys, never read by a loss term), but they all carry the SAME token id (pad) and position: without the flag one
// wave adds ~B (T - len) zero rows serially (30 k rows at B = 32, T = 1024 with 70-token episodes: 80 ms of a 124 ms PPO step).
// A row with flag 0 can still carry a gradient when the NEXT position of its sequence is attended: the PPO / BC losses read row t's logits wherever
// attention_mask[t + 1] is set (ppo/base_interface.py:208-214), which with left padding or a mask with holes includes rows whose own flag is 0.
// `t_row` > 0 (the sequence length T of a [B, T] batch) makes such rows live too: live(r) = flag[r] | (flag[r + 1] within the same sequence) — for
// a right-padded batch exactly the flagged rows.  Token ids outside [0, vocab) (embed_fwd_kernel) own no table row.
__device__ __forceinline__ bool embed_row_live(const uint8_t *__restrict__ live, int j, int t_row) {
    if (!live || live[j]) return true;
    return t_row > 0 && (j + 1) % t_row != 0 && live[j + 1] != 0;
}
template <int NPL>   // floats per lane: d <= 64 * NPL
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float *__restrict__ dx, const int32_t *__restrict__ ids,
                                                        const int32_t *__restrict__ pos, const uint8_t *__restrict__ live, float *__restrict__ dwte,
                                                        float *__restrict__ dwpe, int R, int d, int vocab, int t_row) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    if (!embed_row_live(live, r, t_row)) return;
    const int32_t *idx = blockIdx.y ? pos : ids;
    float *table = blockIdx.y ? dwpe : dwte;
    const int id = idx[r];
    if (!blockIdx.y && vocab > 0 && (id < 0 || id >= vocab)) return;                      // zero embedding row: no parameter behind it
    for (int base = 0; base < r; base += 64) {
        const int j = base + lane;
        if (__ballot(j < r && idx[j] == id && embed_row_live(live, j, t_row))) return;      // an earlier row owns this index (wave-uniform exit)
    }
    float acc[NPL];
#pragma unroll
    for (int k = 0; k < NPL; k++) acc[k] = lane + 64 * k < d ? dx[(size_t)r * d + lane + 64 * k] : 0.f;
    for (int base = r + 1; base < R; base += 64) {
        const int j = base + lane;
        unsigned long long m = __ballot(j < R && idx[j] == id && embed_row_live(live, j, t_row));
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float *row = dx + (size_t)(base + b) * d;
#pragma unroll
            for (int k = 0; k < NPL; k++) if (lane + 64 * k < d) acc[k] += row[lane + 64 * k];
        }
    }
#pragma unroll
    for (int k = 0; k < NPL; k++) if (lane + 64 * k < d) table[(size_t)id * d + lane + 64 * k] += acc[k];
}

// ------------------------------------------------------------------------------------------ LayerNorm (fp32, d <= 4096)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                     const float *__restrict__ b, float *__restrict__ y, float *__restrict__ mean,
                                                     float *__restrict__ rstd, int R, int d, float eps, uint16_t *__restrict__ yb, long ldb) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float *xr = x + (size_t)r * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c];
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = xr[c] - mu; q += t * t; }
    const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
    for (int c = lane; c < d; c += 64) {
        const float v = (xr[c] - mu) * rs * g[c] + b[c];
        if (y) y[(size_t)r * d + c] = v;
        if (yb) yb[(size_t)r * ldb + c] = f32_to_bf16_rne(v);      // the bf16 operand of the consuming GEMM (bf16-matmul train mode)
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}
// The same with the row held in registers (d = 256 NV: 768, 1024, 1280, ...): one 16-byte load per lane and 256 columns, the row read ONCE, and
// optionally the block's residual add in front — x[r] += resid[r] (in place: x is the projection's output, afterwards the residual stream the
// backward's LayerNorm reads) — so that no stand-alone `x + proj(...)` pass over [B*T][d] floats runs between a projection and the LayerNorm behind it.
// SPLIT3: yb is the "bf16 x 3" operand [R][3 d] = [hi | lo | hi] of the LayerNorm output (hi = bf16(v), lo = bf16(v - hi): lmrl_split3_bf16's row
// format, GPT2EngineF32's bf16x3 matmul mode) instead of its plain bf16 copy.
__device__ __forceinline__ void split3_pack4(float a, float b, float c, float d, uint2 &hi, uint2 &lo) {
    const uint16_t h0 = f32_to_bf16_rne(a), h1 = f32_to_bf16_rne(b), h2 = f32_to_bf16_rne(c), h3 = f32_to_bf16_rne(d);
    hi.x = (uint32_t)h0 | ((uint32_t)h1 << 16); hi.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
    const float l0 = a - __uint_as_float((uint32_t)h0 << 16), l1 = b - __uint_as_float((uint32_t)h1 << 16);
    const float l2 = c - __uint_as_float((uint32_t)h2 << 16), l3 = d - __uint_as_float((uint32_t)h3 << 16);
    lo.x = (uint32_t)f32_to_bf16_rne(l0) | ((uint32_t)f32_to_bf16_rne(l1) << 16); lo.y = (uint32_t)f32_to_bf16_rne(l2) | ((uint32_t)f32_to_bf16_rne(l3) << 16);
}
template <int NV, bool SPLIT3 = false>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(float *x, const float *__restrict__ resid, const float *__restrict__ g,
                                                         const float *__restrict__ b, float *__restrict__ y, float *__restrict__ mean,
                                                         float *__restrict__ rstd, int R, float eps, uint16_t *__restrict__ yb, long ldb) {
    constexpr int d = NV * 256;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    float *xr = x + (size_t)r * d + lane * 4;
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = *reinterpret_cast<const float4 *>(xr + k * 256);
    if (resid) {
        const float *rr = resid + (size_t)r * d + lane * 4;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const float4 o = *reinterpret_cast<const float4 *>(rr + k * 256);
            v[k].x += o.x; v[k].y += o.y; v[k].z += o.z; v[k].w += o.w;
            *reinterpret_cast<float4 *>(xr + k * 256) = v[k];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const float a0 = v[k].x - mu, a1 = v[k].y - mu, a2 = v[k].z - mu, a3 = v[k].w - mu;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int c = k * 256 + lane * 4;
        const float4 gg = *reinterpret_cast<const float4 *>(g + c), bb = *reinterpret_cast<const float4 *>(b + c);
        float4 o;
        o.x = (v[k].x - mu) * rs * gg.x + bb.x; o.y = (v[k].y - mu) * rs * gg.y + bb.y;
        o.z = (v[k].z - mu) * rs * gg.z + bb.z; o.w = (v[k].w - mu) * rs * gg.w + bb.w;
        if (y) *reinterpret_cast<float4 *>(y + (size_t)r * d + c) = o;
        if (yb) {
            if constexpr (SPLIT3) {
                uint2 hi, lo;
                split3_pack4(o.x, o.y, o.z, o.w, hi, lo);
                uint16_t *q = yb + (size_t)r * ldb + c;
                *reinterpret_cast<uint2 *>(q) = hi; *reinterpret_cast<uint2 *>(q + d) = lo; *reinterpret_cast<uint2 *>(q + 2 * d) = hi;
            } else {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16_rne(o.x) | ((uint32_t)f32_to_bf16_rne(o.y) << 16);
                pk.y = (uint32_t)f32_to_bf16_rne(o.z) | ((uint32_t)f32_to_bf16_rne(o.w) << 16);
                *reinterpret_cast<uint2 *>(yb + (size_t)r * ldb + c) = pk;
            }
        }
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}
// dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)),  dyg = dy * g ; also emits xhat*dy for the gamma gradient
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                     const float *__restrict__ g, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, float *__restrict__ dx,
                                                     float *__restrict__ dy_xhat, int R, int d, int accumulate) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float mu = mean[r], rs = rstd[r];
    const float *xr = x + (size_t)r * d, *dyr = dy + (size_t)r * d;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float xh = (xr[c] - mu) * rs, dg = dyr[c] * g[c];
        s1 += dg; s2 += dg * xh;
    }
    s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
    for (int c = lane; c < d; c += 64) {
        const float xh = (xr[c] - mu) * rs, dg = dyr[c] * g[c];
        const float v = rs * (dg - s1 - xh * s2);
        float *p = dx + (size_t)r * d + c;
        *p = accumulate ? *p + v : v;
        if (dy_xhat) dy_xhat[(size_t)r * d + c] = dyr[c] * xh;
    }
}

// The same with the gamma / beta gradients reduced in the same pass: a workgroup owns a slab of rows, every lane keeps the running column
// sums of dy*xhat and dy for its NC columns in registers across the slab's rows, the four waves are merged through LDS in a fixed order and
// the slab's partial row goes to partial[slab][2][d]; ln_bwd_reduce_kernel then sums the slabs (fixed order: deterministic).  Replaces
// ln_bwd_kernel + a [R][d] dy*xhat scratch matrix + two column-sum passes over R x d (3 launches, 4 extra sweeps of the activations).
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ g,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd, float *dx,
                                                           float *__restrict__ partial, int R, int rows_per_wg, int accumulate,
                                                           uint16_t *__restrict__ dxb, long ldb) {
    constexpr int d = NC * 64;
    __shared__ float sm[4][2][d];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float gv[NC], ag[NC], ab[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) { gv[k] = g[lane + 64 * k]; ag[k] = 0.f; ab[k] = 0.f; }
    const int r_end = min(R, (int)(blockIdx.x + 1) * rows_per_wg);
    for (int r = blockIdx.x * rows_per_wg + wave; r < r_end; r += 4) {
        const float mu = mean[r], rs = rstd[r];
        const float *xr = x + (size_t)r * d, *dyr = dy + (size_t)r * d;
        float xh[NC], dv[NC], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NC; k++) { xh[k] = xr[lane + 64 * k]; dv[k] = dyr[lane + 64 * k]; }
#pragma unroll
        for (int k = 0; k < NC; k++) {
            xh[k] = (xh[k] - mu) * rs;
            const float dg = dv[k] * gv[k];
            s1 += dg; s2 += dg * xh[k];
        }
        s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
        float *dxr = dx + (size_t)r * d;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            float v = rs * (dv[k] * gv[k] - s1 - xh[k] * s2);
            if (accumulate) v += dxr[lane + 64 * k];
            dxr[lane + 64 * k] = v;
            if (dxb) dxb[(size_t)r * ldb + lane + 64 * k] = f32_to_bf16_rne(v);    // the bf16 dy operand of the next linear backward (bf16-matmul mode)
            ag[k] += dv[k] * xh[k];
            ab[k] += dv[k];
        }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) { sm[wave][0][lane + 64 * k] = ag[k]; sm[wave][1][lane + 64 * k] = ab[k]; }
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * 2 * d;
    for (int i = threadIdx.x; i < 2 * d; i += 256) {
        const int which = i >= d, c = i - which * d;
        out[i] = (sm[0][which][c] + sm[1][which][c]) + (sm[2][which][c] + sm[3][which][c]);
    }
}
// the same for d = 256 NV with 16-byte accesses: a lane owns columns 4 lane + 256 k + {0..3} (one float4 load of x / dy / dx and one 8-byte
// bf16 store per 256 columns instead of four 4-byte / 2-byte ones: 70 -> ~45 us on [16384][768], an HBM-bound sweep of 225 MB).  Every
// column's running sums see the same rows in the same order as in the strided kernel: dgamma / dbeta are bit-identical to it.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_fused_vec_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ g,
                                                               const float *__restrict__ mean, const float *__restrict__ rstd, float *dx,
                                                               float *__restrict__ partial, int R, int rows_per_wg, int accumulate,
                                                               uint16_t *__restrict__ dxb, long ldb) {
    constexpr int d = NV * 256;
    __shared__ float sm[4][2][d];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 gv[NV], ag[NV], ab[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        gv[k] = *reinterpret_cast<const float4 *>(g + k * 256 + lane * 4);
        ag[k] = make_float4(0.f, 0.f, 0.f, 0.f); ab[k] = ag[k];
    }
    const int r_end = min(R, (int)(blockIdx.x + 1) * rows_per_wg);
    for (int r = blockIdx.x * rows_per_wg + wave; r < r_end; r += 4) {
        const float mu = mean[r], rs = rstd[r];
        const float *xr = x + (size_t)r * d + lane * 4, *dyr = dy + (size_t)r * d + lane * 4;
        float *dxr = dx + (size_t)r * d + lane * 4;
        float4 xh[NV], dv[NV], acc[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) {
            xh[k] = *reinterpret_cast<const float4 *>(xr + k * 256);
            dv[k] = *reinterpret_cast<const float4 *>(dyr + k * 256);
            if (accumulate) acc[k] = *reinterpret_cast<const float4 *>(dxr + k * 256);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            xh[k].x = (xh[k].x - mu) * rs; xh[k].y = (xh[k].y - mu) * rs; xh[k].z = (xh[k].z - mu) * rs; xh[k].w = (xh[k].w - mu) * rs;
            const float g0 = dv[k].x * gv[k].x, g1 = dv[k].y * gv[k].y, g2 = dv[k].z * gv[k].z, g3 = dv[k].w * gv[k].w;
            s1 += (g0 + g1) + (g2 + g3);
            s2 += (g0 * xh[k].x + g1 * xh[k].y) + (g2 * xh[k].z + g3 * xh[k].w);
        }
        s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            float4 v;
            v.x = rs * (dv[k].x * gv[k].x - s1 - xh[k].x * s2); v.y = rs * (dv[k].y * gv[k].y - s1 - xh[k].y * s2);
            v.z = rs * (dv[k].z * gv[k].z - s1 - xh[k].z * s2); v.w = rs * (dv[k].w * gv[k].w - s1 - xh[k].w * s2);
            if (accumulate) { v.x += acc[k].x; v.y += acc[k].y; v.z += acc[k].z; v.w += acc[k].w; }
            *reinterpret_cast<float4 *>(dxr + k * 256) = v;
            if (dxb) {
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16_rne(v.x) | ((uint32_t)f32_to_bf16_rne(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16_rne(v.z) | ((uint32_t)f32_to_bf16_rne(v.w) << 16);
                *reinterpret_cast<uint2 *>(dxb + (size_t)r * ldb + k * 256 + lane * 4) = pk;
            }
            ag[k].x += dv[k].x * xh[k].x; ag[k].y += dv[k].y * xh[k].y; ag[k].z += dv[k].z * xh[k].z; ag[k].w += dv[k].w * xh[k].w;
            ab[k].x += dv[k].x; ab[k].y += dv[k].y; ab[k].z += dv[k].z; ab[k].w += dv[k].w;
        }
    }
#pragma unroll
    for (int k = 0; k < NV; k++) {
        *reinterpret_cast<float4 *>(&sm[wave][0][k * 256 + lane * 4]) = ag[k];
        *reinterpret_cast<float4 *>(&sm[wave][1][k * 256 + lane * 4]) = ab[k];
    }
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * 2 * d;
    for (int i = threadIdx.x; i < 2 * d; i += 256) {
        const int which = i >= d, c = i - which * d;
        out[i] = (sm[0][which][c] + sm[1][which][c]) + (sm[2][which][c] + sm[3][which][c]);
    }
}
// dgamma[c] (+)= sum_slab partial[slab][0][c], dbeta likewise: 16 columns per workgroup, 16 slab phases per column merged through LDS in a
// fixed order (2d / 16 workgroups: enough of them to pull the 2 x nslab x d partial matrix at HBM rate)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float *__restrict__ partial, int nslab, int d, float *dgamma, float *dbeta,
                                                            int accumulate) {
    __shared__ float sm[16][17];
    const int cl = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;                      // over [0, 2d)
    float s = 0.f;
    if (col < 2 * d)
        for (int k = ph; k < nslab; k += 16) s += partial[(size_t)k * 2 * d + col];
    sm[ph][cl] = s;
    __syncthreads();
    if (ph == 0 && col < 2 * d) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) v += sm[k][cl];
        float *o = col < d ? dgamma + col : dbeta + (col - d);
        *o = accumulate ? *o + v : v;
    }
}

// ------------------------------------------------------------------------------------------ column sums (bias / LN grads)
// out[c] (+)= sum_r x[r][c] ; deterministic two-stage: stage 1 = SPLIT row slabs -> partial[SPLIT][C], stage 2 sums them.
// wrow (optional): per-row weights, out[c] = sum_r wrow[r * ldw] x[r][c] — the weight gradient x^T dy of a Dense layer with ONE output (the V /
// value heads): a matrix-vector product, which the 64 x 64-tile sgemm ran as 12 tiles over K = B*T (820 us per ILQL step)
__global__ __launch_bounds__(256) void colsum_stage1(const float *__restrict__ x, float *__restrict__ partial, int R, int C, int ld,
                                                     int rows_per_slab, const float *__restrict__ wrow, int ldw) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(R, r0 + rows_per_slab);
    // 8 independent loads in flight per thread (a serial "s += x[r]" loop waits a full memory latency per row); fixed association order
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = x[(size_t)(r + u) * ld + c];
        if (wrow) {
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = fmaf(v[u], wrow[(size_t)(r + u) * ldw], a[u]);
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] += v[u];
        }
    }
    for (; r < r1; r++) a[0] += x[(size_t)r * ld + c] * (wrow ? wrow[(size_t)r * ldw] : 1.f);
    partial[(size_t)blockIdx.y * C + c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
__global__ __launch_bounds__(256) void colsum_stage2(const float *__restrict__ partial, float *__restrict__ out, int C, int nslab,
                                                     int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < nslab; k++) s += partial[(size_t)k * C + c];
    out[c] = accumulate ? out[c] + s : s;
}

// ------------------------------------------------------------------------------------------ elementwise
__device__ __forceinline__ float gelu_new_exact(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.f + tanhf(u));
}
// (the elementwise kernels below may be called in place — out == x or out == y: no __restrict__ on the pairs that may alias)
__global__ void gelu_fwd_kernel(const float *x, float *y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = gelu_new_exact(x[i]);
}
// gelu_new(x) written as the bf16 x 3 operand [rows][3 cols] (see SPLIT3 above): gelu + lmrl_split3_bf16 in one pass.  cols a multiple of 4.
__global__ __launch_bounds__(256) void gelu_split3_kernel(const float *__restrict__ x, int rows, int cols, uint16_t *__restrict__ dst) {
    const long n4 = (long)rows * (cols / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int r = (int)(i / (cols / 4)), c = (int)(i - (long)r * (cols / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)r * cols + c);
        uint2 hi, lo;
        split3_pack4(gelu_new_exact(v.x), gelu_new_exact(v.y), gelu_new_exact(v.z), gelu_new_exact(v.w), hi, lo);
        uint16_t *q = dst + (size_t)r * 3 * cols + c;
        *reinterpret_cast<uint2 *>(q) = hi; *reinterpret_cast<uint2 *>(q + cols) = lo; *reinterpret_cast<uint2 *>(q + 2 * cols) = hi;
    }
}
// the same on a [rows][cols] matrix, writing the bf16 copy (row pitch ldb) the consuming GEMM reads and, optionally, the fp32 result
// (bf16-matmul train mode keeps only the bf16 copy: it is both the c_proj operand and, transposed, the operand of its dW product).
// One workgroup per row, 8 columns per lane and iteration.
__global__ __launch_bounds__(256) void gelu_fwd_staged_kernel(const float *__restrict__ x, float *y, uint16_t *__restrict__ yb, long ldb, int rows,
                                                              int cols) {
    const int r = blockIdx.x;
    const float *xr = x + (size_t)r * cols;
    const bool vec = (cols & 3) == 0;
    for (int c0 = threadIdx.x * 8; c0 < cols; c0 += 256 * 8) {
        float a[8], o[8];
        const bool full = vec && c0 + 8 <= cols;
        if (full) {
            const float4 a0 = *reinterpret_cast<const float4 *>(xr + c0), a1 = *reinterpret_cast<const float4 *>(xr + c0 + 4);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = c0 + k < cols ? xr[c0 + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) o[k] = gelu_new_exact(a[k]);
        if (full && (ldb & 7) == 0) {
            uint4 pk;
            pk.x = (uint32_t)f32_to_bf16_rne(o[0]) | ((uint32_t)f32_to_bf16_rne(o[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16_rne(o[2]) | ((uint32_t)f32_to_bf16_rne(o[3]) << 16);
            pk.z = (uint32_t)f32_to_bf16_rne(o[4]) | ((uint32_t)f32_to_bf16_rne(o[5]) << 16);
            pk.w = (uint32_t)f32_to_bf16_rne(o[6]) | ((uint32_t)f32_to_bf16_rne(o[7]) << 16);
            *reinterpret_cast<uint4 *>(yb + (size_t)r * ldb + c0) = pk;
            if (y) {
                *reinterpret_cast<float4 *>(y + (size_t)r * cols + c0) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4 *>(y + (size_t)r * cols + c0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (c0 + k < cols) {
                    yb[(size_t)r * ldb + c0 + k] = f32_to_bf16_rne(o[k]);
                    if (y) y[(size_t)r * cols + c0 + k] = o[k];
                }
        }
    }
}
__global__ void gelu_bwd_kernel(const float *dy, const float *__restrict__ x, float *dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
        const float th = tanhf(u);
        const float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * v * v);
        dx[i] = dy[i] * (0.5f * (1.f + th) + 0.5f * v * (1.f - th * th) * du);
    }
}
// the same written as the bf16 dy operand [rows_dst][ldb] of the next linear backward (bf16-matmul mode: nothing else reads dx); one
// workgroup per destination row, 8 columns per lane and iteration, padding zero-filled
__global__ __launch_bounds__(256) void gelu_bwd_bf16_kernel(const float *__restrict__ dy, const float *__restrict__ x, int rows, int cols,
                                                            uint16_t *__restrict__ dst, long ldb) {
    const int r = blockIdx.x;
    uint16_t *q = dst + (size_t)r * ldb;
    const float *dyr = dy + (size_t)r * cols, *xr = x + (size_t)r * cols;
    const bool vec = r < rows && (cols & 3) == 0;
    for (int c0 = threadIdx.x * 8; c0 < ldb; c0 += 256 * 8) {
        float a[8], b[8], o[8];
        if (vec && c0 + 8 <= cols) {
            const float4 a0 = *reinterpret_cast<const float4 *>(dyr + c0), a1 = *reinterpret_cast<const float4 *>(dyr + c0 + 4);
            const float4 b0 = *reinterpret_cast<const float4 *>(xr + c0), b1 = *reinterpret_cast<const float4 *>(xr + c0 + 4);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool ok = r < rows && c0 + k < cols;
                a[k] = ok ? dyr[c0 + k] : 0.f; b[k] = ok ? xr[c0 + k] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float v = b[k];
            const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
            const float th = tanhf(u);
            const float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * v * v);
            o[k] = a[k] * (0.5f * (1.f + th) + 0.5f * v * (1.f - th * th) * du);
        }
        uint4 pk;
        pk.x = (uint32_t)f32_to_bf16_rne(o[0]) | ((uint32_t)f32_to_bf16_rne(o[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16_rne(o[2]) | ((uint32_t)f32_to_bf16_rne(o[3]) << 16);
        pk.z = (uint32_t)f32_to_bf16_rne(o[4]) | ((uint32_t)f32_to_bf16_rne(o[5]) << 16);
        pk.w = (uint32_t)f32_to_bf16_rne(o[6]) | ((uint32_t)f32_to_bf16_rne(o[7]) << 16);
        *reinterpret_cast<uint4 *>(q + c0) = pk;
    }
}
__global__ void relu_fwd_kernel(const float *x, float *y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float *dy, const float *__restrict__ x, float *dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dx[i] = x[i] > 0.f ? dy[i] : 0.f;
}
// out = a*x + b*y   (residual adds, grad accumulation, Polyak: optax.incremental_update(new, old, s) = s*new + (1-s)*old)
__global__ void axpby_kernel(float a, const float *x, float b, const float *y, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
// optax.adamw: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; mhat = m/(1-b1^t) ; vhat = v/(1-b2^t)
//              p -= lr * (mhat / (sqrt(vhat) + eps) + wd * p)
__global__ void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, size_t n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float upd = (mi / bc1) / (sqrtf(vi / bc2) + eps) + wd * p[i];
        p[i] = p[i] - lr * upd;
    }
}

// The same update over a whole parameter ARENA in one launch: the weight-decay coefficient is a per-tensor property (biases and LayerNorm
// parameters are masked out, train_ilql_gpt2.py:155-186), so the arena comes with a table of segment ends and coefficients.  One workgroup
// per 4096-element chunk; its first lane's segment is found by bisection, lanes then walk forward (a chunk rarely crosses a boundary).
// 16-byte accesses: a lane owns 4 consecutive elements per step (7 x 16 B of traffic instead of 28 scalar accesses — the scalar form ran at ~2 TB/s,
// 4 x 0.68 ms per ILQL step); a group that straddles a segment boundary or the end takes the scalar path.  Optional Polyak target in the same
// sweep (optax.incremental_update(new_params, target, alpha) right after apply_gradients, ilql/gpt2/interface.py:327-347): tgt = (1 - alpha) tgt +
// alpha p_new while p_new is still in registers — no second pass that re-reads the parameters.
__device__ __forceinline__ float adamw_one(float &p, float g, float &m, float &v, float wd, float lr, float b1, float b2, float eps, float bc1, float bc2) {
    const float mi = b1 * m + (1.f - b1) * g;
    const float vi = b2 * v + (1.f - b2) * g * g;
    m = mi; v = vi;
    const float upd = (mi / bc1) / (sqrtf(vi / bc2) + eps) + wd * p;
    p = p - lr * upd;
    return p;
}
__global__ __launch_bounds__(256) void adamw_segments_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                             float *__restrict__ v, long n, const long *__restrict__ seg_end,
                                                             const float *__restrict__ seg_wd, int nseg, float lr, float b1, float b2, float eps,
                                                             float bc1, float bc2, float *__restrict__ tgt, float alpha, float one_m_alpha) {
    const long base = (long)blockIdx.x * 4096;
    int lo = 0, hi = nseg - 1;                       // first segment with seg_end > base
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > base) hi = mid; else lo = mid + 1; }
    int s = lo;
#pragma unroll 2
    for (int k = 0; k < 4; k++) {
        const long i = base + (long)(k * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        while (s < nseg - 1 && i >= seg_end[s]) s++;
        if (i + 3 < n && i + 3 < seg_end[s]) {
            const float wd = seg_wd[s];
            float4 pi = *reinterpret_cast<const float4 *>(p + i), mi = *reinterpret_cast<const float4 *>(m + i), vi = *reinterpret_cast<const float4 *>(v + i);
            const float4 gi = *reinterpret_cast<const float4 *>(g + i);
            adamw_one(pi.x, gi.x, mi.x, vi.x, wd, lr, b1, b2, eps, bc1, bc2);
            adamw_one(pi.y, gi.y, mi.y, vi.y, wd, lr, b1, b2, eps, bc1, bc2);
            adamw_one(pi.z, gi.z, mi.z, vi.z, wd, lr, b1, b2, eps, bc1, bc2);
            adamw_one(pi.w, gi.w, mi.w, vi.w, wd, lr, b1, b2, eps, bc1, bc2);
            *reinterpret_cast<float4 *>(p + i) = pi; *reinterpret_cast<float4 *>(m + i) = mi; *reinterpret_cast<float4 *>(v + i) = vi;
            if (tgt) {
                float4 ti = *reinterpret_cast<const float4 *>(tgt + i);
                ti.x = alpha * pi.x + one_m_alpha * ti.x; ti.y = alpha * pi.y + one_m_alpha * ti.y;
                ti.z = alpha * pi.z + one_m_alpha * ti.z; ti.w = alpha * pi.w + one_m_alpha * ti.w;
                *reinterpret_cast<float4 *>(tgt + i) = ti;
            }
        } else {
            int ss = s;
            for (long j = i; j < i + 4 && j < n; j++) {
                while (ss < nseg - 1 && j >= seg_end[ss]) ss++;
                float pj = p[j], mj = m[j], vj = v[j];
                adamw_one(pj, g[j], mj, vj, seg_wd[ss], lr, b1, b2, eps, bc1, bc2);
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (tgt) tgt[j] = alpha * pj + one_m_alpha * tgt[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ causal softmax (attention)
// S: [nb][T][T] scores (already scaled). P[r][c] = softmax over c <= r with key_mask[b][c] != 0 ; 0 elsewhere.  In place ok.
__global__ __launch_bounds__(256) void softmax_causal_fwd_kernel(const float *S, const uint8_t *__restrict__ key_mask, float *P, int T, int heads,
                                                                 long rows_total) {   // S == P allowed: the reductions finish before the write loop, and each lane reads element c before it writes it
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows_total) return;
    const int r = (int)(row % T);
    const long bh = row / T;
    const int b = (int)(bh / heads);
    const float *s = S + row * T;
    float *p = P + row * T;
    const uint8_t *km = key_mask ? key_mask + (size_t)b * T : nullptr;
    float mx = -INFINITY;
    for (int c = lane; c <= r; c += 64) if (!km || km[c]) mx = fmaxf(mx, s[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c <= r; c += 64) if (!km || km[c]) sum += __expf(s[c] - mx);
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    for (int c = lane; c < T; c += 64) {
        const bool ok = c <= r && (!km || km[c]);
        p[c] = ok ? __expf(s[c] - mx) * inv : 0.f;
    }
}
// dS = P * (dP - sum_c dP*P), written over dP
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float *__restrict__ P, float *dP, int T, long rows_total) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows_total) return;
    const float *p = P + row * T;
    float *dp = dP + row * T;
    float s = 0.f;
    for (int c = lane; c < T; c += 64) s += dp[c] * p[c];
    s = wave_sum(s);
    for (int c = lane; c < T; c += 64) dp[c] = p[c] * (dp[c] - s);
}

// ------------------------------------------------------------------------------------------ log-softmax / CE over the vocabulary
// One workgroup per row: lse = logsumexp(logits[r][:V]); target logit; logprob = target - lse.
__global__ __launch_bounds__(256) void lse_gather_kernel(const float *__restrict__ logits, int ld, int V, const int32_t *__restrict__ targets,
                                                         float *__restrict__ logprob, float *__restrict__ lse, float *__restrict__ target_logit) {
    // ONE pass over the row (200 KB at V = 50 257: a second pass would come from HBM again): per-thread running (max, sum of exp) over float4
    // groups — one rescale per group — merged across the workgroup at the end.
    const int r = blockIdx.x, tid = threadIdx.x;
    const float *row = logits + (size_t)r * ld;
    float mx = -INFINITY, s = 0.f;
    const int head = (int)((16 - (reinterpret_cast<uintptr_t>(row) & 15)) & 15) >> 2;        // scalars before the first 16-byte boundary
    const int nvec = (V - (head < V ? head : V)) >> 2;
    auto add = [&](float x) {
        if (x > mx) { s = s * __expf(mx - x) + 1.f; mx = x; }
        else s += __expf(x - mx);
    };
    for (int c = tid; c < head && c < V; c += 256) add(row[c]);
    const float4 *rv = reinterpret_cast<const float4 *>(row + head);
    for (int q = tid; q < nvec; q += 256) {
        const float4 x = rv[q];
        const float gm = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
        if (gm > mx) { s *= __expf(mx - gm); mx = gm; }
        s += (__expf(x.x - mx) + __expf(x.y - mx)) + (__expf(x.z - mx) + __expf(x.w - mx));
    }
    for (int c = head + 4 * nvec + tid; c < V; c += 256) add(row[c]);
    // merge (max, sum) pairs: wave shuffle, then the four waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o), os = __shfl_xor(s, o);
        const float nm = fmaxf(mx, om);
        s = (mx == -INFINITY ? 0.f : s * __expf(mx - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
        mx = nm;
    }
    __shared__ float rm[4], rs[4];
    if ((tid & 63) == 0) { rm[tid >> 6] = mx; rs[tid >> 6] = s; }
    __syncthreads();
    if (tid == 0) {
        const float m4 = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
        float tot = 0.f;
        for (int w = 0; w < 4; w++) tot += rm[w] == -INFINITY ? 0.f : rs[w] * __expf(rm[w] - m4);
        const float l = m4 + logf(tot);
        int t = targets[r];
        t = t < 0 ? 0 : (t >= V ? V - 1 : t);
        const float tl = row[t];
        if (lse) lse[r] = l;
        if (target_logit) target_logit[r] = tl;
        if (logprob) logprob[r] = tl - l;
    }
}
// dlogits[r][c] = coef_ce[r] * (softmax[r][c] - [c == t]) + coef_gather[r] * [c == t]     (written over the logits)
// coef_ce = d loss / d CE_r ; coef_gather = d loss / d logits[r][t] from a direct gather (Q(s,a)).
__global__ __launch_bounds__(256) void ce_bwd_kernel(float *__restrict__ logits, int ld, int V, const float *__restrict__ lse,
                                                     const int32_t *__restrict__ targets, const float *__restrict__ coef_ce,
                                                     const float *__restrict__ coef_gather) {
    const int r = blockIdx.x;
    float *row = logits + (size_t)r * ld;
    const float l = lse[r], cc = coef_ce ? coef_ce[r] : 0.f, cg = coef_gather ? coef_gather[r] : 0.f;
    int t = targets[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    for (int c = threadIdx.x; c < ld; c += 256) {
        float v = 0.f;
        if (c < V) {
            v = cc == 0.f ? 0.f : cc * expf(row[c] - l);
            if (c == t) v += cg - cc;
        }
        row[c] = v;
    }
}

static int ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g == 0 ? 1 : g));
}

}  // namespace lmrl

using namespace lmrl;
#define ST as_stream(stream)

extern "C" {

int lmrl_embed_fwd(const float *wte_d, const float *wpe_d, const int32_t *ids_d, const int32_t *pos_d, float *x_d, int rows, int d, int vocab,
                   void *stream) {
    LMRL_REQUIRE(wte_d && wpe_d && ids_d && pos_d && x_d && rows > 0 && d % 4 == 0, "lmrl_embed_fwd: bad argument");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, wte_d, wpe_d, ids_d, pos_d, x_d, rows, d, vocab);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_embed_bwd(const float *dx_d, const int32_t *ids_d, const int32_t *pos_d, const uint8_t *live_d, float *dwte_d, float *dwpe_d, int rows, int d,
                   int vocab, int t_row, void *stream) {
    LMRL_REQUIRE(dx_d && ids_d && pos_d && dwte_d && dwpe_d && rows > 0, "lmrl_embed_bwd: bad argument");
    LMRL_REQUIRE(d <= 64 * 32, "lmrl_embed_bwd: d_model up to 2048");
    const dim3 grid(ceil_div(rows, 4), 2);
    if (d <= 64 * 12) hipLaunchKernelGGL(embed_bwd_kernel<12>, grid, dim3(256), 0, ST, dx_d, ids_d, pos_d, live_d, dwte_d, dwpe_d, rows, d, vocab, t_row);
    else if (d <= 64 * 20) hipLaunchKernelGGL(embed_bwd_kernel<20>, grid, dim3(256), 0, ST, dx_d, ids_d, pos_d, live_d, dwte_d, dwpe_d, rows, d, vocab, t_row);
    else hipLaunchKernelGGL(embed_bwd_kernel<32>, grid, dim3(256), 0, ST, dx_d, ids_d, pos_d, live_d, dwte_d, dwpe_d, rows, d, vocab, t_row);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_gather_rows_f32(const float *src_d, const int32_t *idx_d, float *dst_d, int n, int d, void *stream) {
    LMRL_REQUIRE(src_d && idx_d && dst_d && n > 0 && d > 0, "lmrl_gather_rows_f32: bad argument");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, ST, src_d, idx_d, dst_d, n, d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_scatter_rows_f32(const float *src_d, const int32_t *idx_d, float *dst_d, int n, int d, int accumulate, void *stream) {
    LMRL_REQUIRE(src_d && idx_d && dst_d && n > 0 && d > 0, "lmrl_scatter_rows_f32: bad argument");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, ST, src_d, idx_d, dst_d, n, d, accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_layernorm_fwd(const float *x_d, const float *g_d, const float *b_d, float *y_d, float *mean_d, float *rstd_d, int rows, int d,
                       float eps, void *stream) {
    LMRL_REQUIRE(x_d && g_d && b_d && y_d && mean_d && rstd_d && rows > 0 && d > 0, "lmrl_layernorm_fwd: bad argument");
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, x_d, g_d, b_d, y_d, mean_d, rstd_d, rows, d, eps, (uint16_t *)nullptr, 0l);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_layernorm_fwd_staged(const float *x_d, const float *g_d, const float *b_d, float *y_d, float *mean_d, float *rstd_d, void *yb_d, long ldb,
                              int rows, int d, float eps, void *stream) {
    LMRL_REQUIRE(x_d && g_d && b_d && mean_d && rstd_d && yb_d && ldb >= d && rows > 0 && d > 0, "lmrl_layernorm_fwd_staged: bad argument");
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, x_d, g_d, b_d, y_d, mean_d, rstd_d, rows, d, eps, (uint16_t *)yb_d, ldb);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
static int g_train_ops_variant = 0;      // tools / tests only: bit 0 = LayerNorm forward on the strided (round-2) kernel
void lmrl_train_ops_set_variant(int v) { g_train_ops_variant = v; }
int lmrl_layernorm_add_fwd(float *x_d, const float *resid_d, const float *g_d, const float *b_d, float *y_d, float *mean_d, float *rstd_d, void *yb_d,
                           long ldb, int rows, int d, float eps, void *stream) {
    LMRL_REQUIRE(x_d && g_d && b_d && mean_d && rstd_d && (y_d || yb_d) && (!yb_d || (ldb >= d && ldb % 4 == 0)) && rows > 0 && d > 0,
                 "lmrl_layernorm_add_fwd: bad argument");
#define LMRL_LNV(NV_)                                                                                                                   \
    hipLaunchKernelGGL(ln_fwd_vec_kernel<NV_>, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, x_d, resid_d, g_d, b_d, y_d, mean_d, rstd_d, rows, eps, \
                       (uint16_t *)yb_d, ldb)
    switch ((d % 256 == 0 && !(g_train_ops_variant & 1)) ? d / 256 : 0) {
        case 1: LMRL_LNV(1); break;
        case 2: LMRL_LNV(2); break;
        case 3: LMRL_LNV(3); break;
        case 4: LMRL_LNV(4); break;
        case 5: LMRL_LNV(5); break;
        default:                     // other widths: the residual add as its own pass, then the strided kernel
            if (resid_d) hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid((size_t)rows * d)), dim3(256), 0, ST, 1.f, x_d, 1.f, resid_d, x_d, (size_t)rows * d);
            hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, x_d, g_d, b_d, y_d, mean_d, rstd_d, rows, d, eps,
                               (uint16_t *)yb_d, ldb);
    }
#undef LMRL_LNV
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_layernorm_add_fwd_split3(float *x_d, const float *resid_d, const float *g_d, const float *b_d, float *mean_d, float *rstd_d, void *split_d,
                                  int rows, int d, float eps, void *stream) {
    LMRL_REQUIRE(x_d && g_d && b_d && mean_d && rstd_d && split_d && rows > 0 && d > 0 && d % 256 == 0 && d <= 1280,
                 "lmrl_layernorm_add_fwd_split3: bad argument (d a multiple of 256 up to 1280)");
#define LMRL_LNS(NV_)                                                                                                                   \
    hipLaunchKernelGGL((ln_fwd_vec_kernel<NV_, true>), dim3(ceil_div(rows, 4)), dim3(256), 0, ST, x_d, resid_d, g_d, b_d, (float *)nullptr, mean_d, \
                       rstd_d, rows, eps, (uint16_t *)split_d, (long)3 * d)
    switch (d / 256) {
        case 1: LMRL_LNS(1); break;
        case 2: LMRL_LNS(2); break;
        case 3: LMRL_LNS(3); break;
        case 4: LMRL_LNS(4); break;
        default: LMRL_LNS(5); break;
    }
#undef LMRL_LNS
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_gelu_split3(const float *x_d, int rows, int cols, void *split_d, void *stream) {
    LMRL_REQUIRE(x_d && split_d && rows > 0 && cols > 0 && cols % 4 == 0, "lmrl_gelu_split3: bad argument");
    hipLaunchKernelGGL(gelu_split3_kernel, dim3(ew_grid((size_t)rows * cols / 4)), dim3(256), 0, ST, x_d, rows, cols, (uint16_t *)split_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_gelu_fwd_staged(const float *x_d, float *y_d, void *yb_d, long ldb, int rows, int cols, void *stream) {
    LMRL_REQUIRE(x_d && yb_d && rows > 0 && cols > 0 && ldb >= cols, "lmrl_gelu_fwd_staged: bad argument");
    hipLaunchKernelGGL(gelu_fwd_staged_kernel, dim3(rows), dim3(256), 0, ST, x_d, y_d, (uint16_t *)yb_d, ldb, rows, cols);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_layernorm_bwd(const float *dy_d, const float *x_d, const float *g_d, const float *mean_d, const float *rstd_d, float *dx_d,
                       float *dy_xhat_d, int rows, int d, int accumulate_dx, void *stream) {
    LMRL_REQUIRE(dy_d && x_d && g_d && mean_d && rstd_d && dx_d && rows > 0, "lmrl_layernorm_bwd: bad argument");
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, dy_d, x_d, g_d, mean_d, rstd_d, dx_d, dy_xhat_d, rows, d,
                       accumulate_dx);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
static int ln_bwd_rows_per_wg(int rows) { return ((std::max(4, ceil_div(rows, 512)) + 3) / 4) * 4; }   // <= 512 slabs, whole 4-row rounds
int lmrl_layernorm_bwd_fused_supported(int d) { return d == 128 || d == 256 || d == 768 || d == 1024 || d == 1280 || d == 1600; }
size_t lmrl_layernorm_bwd_fused_ws_bytes(int rows, int d) {
    return (size_t)ceil_div(rows, ln_bwd_rows_per_wg(rows)) * 2 * d * sizeof(float);
}
int lmrl_layernorm_bwd_fused(const float *dy_d, const float *x_d, const float *g_d, const float *mean_d, const float *rstd_d, float *dx_d,
                             float *dgamma_d, float *dbeta_d, int rows, int d, int accumulate_dx, int accumulate_dg, float *ws_d, void *dxb_d,
                             long ldb, void *stream) {
    LMRL_REQUIRE(dy_d && x_d && g_d && mean_d && rstd_d && dx_d && dgamma_d && dbeta_d && ws_d && rows > 0, "lmrl_layernorm_bwd_fused: bad argument");
    LMRL_REQUIRE(lmrl_layernorm_bwd_fused_supported(d), "lmrl_layernorm_bwd_fused: d_model must be one of 128, 256, 768, 1024, 1280, 1600");
    LMRL_REQUIRE(!dxb_d || ldb >= d, "lmrl_layernorm_bwd_fused: bf16 copy pitch smaller than d_model");
    const int rpw = ln_bwd_rows_per_wg(rows), nslab = ceil_div(rows, rpw);
#define LMRL_LNB(NC_)                                                                                                                        \
    hipLaunchKernelGGL(ln_bwd_fused_kernel<NC_>, dim3(nslab), dim3(256), 0, ST,                                                               \
                       dy_d, x_d, g_d, mean_d, rstd_d, dx_d, ws_d, rows, rpw, accumulate_dx, \
                       (uint16_t *)dxb_d, ldb)
#define LMRL_LNBV(NV_)                                                                                                                       \
    hipLaunchKernelGGL(ln_bwd_fused_vec_kernel<NV_>, dim3(nslab), dim3(256), 0, ST,                                                           \
                       dy_d, x_d, g_d, mean_d, rstd_d, dx_d, ws_d, rows, rpw, accumulate_dx, \
                       (uint16_t *)dxb_d, ldb)
    const bool vec = !(g_train_ops_variant & 2) && (!dxb_d || ldb % 4 == 0);
    switch (d / 64) {
        case 2: LMRL_LNB(2); break;
        case 4: if (vec) LMRL_LNBV(1); else LMRL_LNB(4); break;
        case 12: if (vec) LMRL_LNBV(3); else LMRL_LNB(12); break;
        case 16: if (vec) LMRL_LNBV(4); else LMRL_LNB(16); break;
        case 20: if (vec) LMRL_LNBV(5); else LMRL_LNB(20); break;
        default: LMRL_LNB(25); break;
    }
#undef LMRL_LNB
#undef LMRL_LNBV
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(2 * d, 16)), dim3(256), 0, ST, ws_d, nslab, d, dgamma_d, dbeta_d, accumulate_dg);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
size_t lmrl_colsum_ws_bytes(int cols) { return (size_t)64 * cols * sizeof(float); }
int lmrl_colsum(const float *x_d, int rows, int cols, int ld, float *out_d, int accumulate, float *ws_d, void *stream) {
    LMRL_REQUIRE(x_d && out_d && ws_d && rows > 0 && cols > 0, "lmrl_colsum: bad argument");
    const int nslab = rows < 64 ? rows : 64;
    const int per = (rows + nslab - 1) / nslab;
    const int slabs = (rows + per - 1) / per;
    hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(cols, 256), slabs), dim3(256), 0, ST, x_d, ws_d, rows, cols, ld, per, (const float *)nullptr, 0);
    hipLaunchKernelGGL(colsum_stage2, dim3(ceil_div(cols, 256)), dim3(256), 0, ST, ws_d, out_d, cols, slabs, accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_colsum_weighted(const float *x_d, int rows, int cols, int ld, const float *wrow_d, int ldw, float *out_d, int accumulate, float *ws_d,
                         void *stream) {
    LMRL_REQUIRE(x_d && wrow_d && out_d && ws_d && rows > 0 && cols > 0 && ldw > 0, "lmrl_colsum_weighted: bad argument");
    const int nslab = rows < 64 ? rows : 64;
    const int per = (rows + nslab - 1) / nslab;
    const int slabs = (rows + per - 1) / per;
    hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(cols, 256), slabs), dim3(256), 0, ST, x_d, ws_d, rows, cols, ld, per, wrow_d, ldw);
    hipLaunchKernelGGL(colsum_stage2, dim3(ceil_div(cols, 256)), dim3(256), 0, ST, ws_d, out_d, cols, slabs, accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_gelu_fwd(const float *x_d, float *y_d, size_t n, void *stream) {
    LMRL_REQUIRE(x_d && y_d, "lmrl_gelu_fwd: null pointer");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x_d, y_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_gelu_bwd(const float *dy_d, const float *x_d, float *dx_d, size_t n, void *stream) {
    LMRL_REQUIRE(dy_d && x_d && dx_d, "lmrl_gelu_bwd: null pointer");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, dy_d, x_d, dx_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_gelu_bwd_bf16(const float *dy_d, const float *x_d, int rows, int cols, void *dst_d, long ldb, int rows_dst, void *stream) {
    LMRL_REQUIRE(dy_d && x_d && dst_d && rows > 0 && cols > 0 && ldb >= cols && ldb % 8 == 0 && rows_dst >= rows, "lmrl_gelu_bwd_bf16: bad argument");
    hipLaunchKernelGGL(gelu_bwd_bf16_kernel, dim3(rows_dst), dim3(256), 0, ST, dy_d, x_d, rows, cols, (uint16_t *)dst_d, ldb);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_relu_fwd(const float *x_d, float *y_d, size_t n, void *stream) {
    LMRL_REQUIRE(x_d && y_d, "lmrl_relu_fwd: null pointer");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x_d, y_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_relu_bwd(const float *dy_d, const float *x_d, float *dx_d, size_t n, void *stream) {
    LMRL_REQUIRE(dy_d && x_d && dx_d, "lmrl_relu_bwd: null pointer");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, dy_d, x_d, dx_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_axpby(float a, const float *x_d, float b, const float *y_d, float *out_d, size_t n, void *stream) {
    LMRL_REQUIRE(x_d && out_d, "lmrl_axpby: null pointer");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, a, x_d, b, y_d, out_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_adamw(float *p_d, const float *g_d, float *m_d, float *v_d, size_t n, float lr, float b1, float b2, float eps, float weight_decay,
               int step, void *stream) {
    LMRL_REQUIRE(p_d && g_d && m_d && v_d && step >= 1, "lmrl_adamw: bad argument");
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, p_d, g_d, m_d, v_d, n, lr, b1, b2, eps, weight_decay, bc1, bc2);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_adamw_segments_polyak(float *p_d, const float *g_d, float *m_d, float *v_d, long n, const long *seg_end_d, const float *seg_wd_d, int nseg, float lr,
                               float b1, float b2, float eps, int step, float *target_d, float alpha, float one_minus_alpha, void *stream) {
    LMRL_REQUIRE(p_d && g_d && m_d && v_d && seg_end_d && seg_wd_d && nseg > 0 && n > 0 && step >= 1, "lmrl_adamw_segments: bad argument");
    LMRL_REQUIRE(((reinterpret_cast<uintptr_t>(p_d) | reinterpret_cast<uintptr_t>(g_d) | reinterpret_cast<uintptr_t>(m_d) | reinterpret_cast<uintptr_t>(v_d) |
                   reinterpret_cast<uintptr_t>(target_d)) & 15) == 0, "lmrl_adamw_segments: arenas must be 16-byte aligned");
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    hipLaunchKernelGGL(adamw_segments_kernel, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, ST, p_d, g_d, m_d, v_d, n, seg_end_d, seg_wd_d, nseg, lr, b1,
                       b2, eps, bc1, bc2, target_d, alpha, one_minus_alpha);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_adamw_segments(float *p_d, const float *g_d, float *m_d, float *v_d, long n, const long *seg_end_d, const float *seg_wd_d, int nseg, float lr,
                        float b1, float b2, float eps, int step, void *stream) {
    return lmrl_adamw_segments_polyak(p_d, g_d, m_d, v_d, n, seg_end_d, seg_wd_d, nseg, lr, b1, b2, eps, step, nullptr, 0.f, 0.f, stream);
}
int lmrl_softmax_causal_fwd(const float *s_d, const uint8_t *key_mask_d, float *p_d, int batch, int heads, int t, void *stream) {
    LMRL_REQUIRE(s_d && p_d && batch > 0 && heads > 0 && t > 0, "lmrl_softmax_causal_fwd: bad argument");
    const long rows = (long)batch * heads * t;
    hipLaunchKernelGGL(softmax_causal_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, s_d, key_mask_d, p_d, t, heads, rows);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_softmax_bwd(const float *p_d, float *dp_d, long rows, int t, void *stream) {
    LMRL_REQUIRE(p_d && dp_d && rows > 0 && t > 0, "lmrl_softmax_bwd: bad argument");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ST, p_d, dp_d, t, rows);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_lse_gather(const float *logits_d, int ld, int vocab, const int32_t *targets_d, float *logprob_d, float *lse_d,
                    float *target_logit_d, int rows, void *stream) {
    LMRL_REQUIRE(logits_d && targets_d && rows > 0 && vocab > 0 && ld >= vocab, "lmrl_lse_gather: bad argument");
    hipLaunchKernelGGL(lse_gather_kernel, dim3(rows), dim3(256), 0, ST, logits_d, ld, vocab, targets_d, logprob_d, lse_d, target_logit_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
int lmrl_ce_bwd(float *logits_d, int ld, int vocab, const float *lse_d, const int32_t *targets_d, const float *coef_ce_d,
                const float *coef_gather_d, int rows, void *stream) {
    LMRL_REQUIRE(logits_d && lse_d && targets_d && rows > 0, "lmrl_ce_bwd: bad argument");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(rows), dim3(256), 0, ST, logits_d, ld, vocab, lse_d, targets_d, coef_ce_d, coef_gather_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
