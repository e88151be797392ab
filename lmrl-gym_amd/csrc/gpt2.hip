// gpt2.hip — GPT-2 forward for lock-step rollouts on gfx950: persistent per-env KV cache,
// chunked prefill / single-token decode through the same code path.
//
// Replaces, for the rollout hot path, the JaxSeq/HF-Flax GPT-2 forward that the reference reaches
// through GPT2PPOPolicy.act -> generate_from_str (LLM_RL/algorithms/ppo/gpt2/interface.py:507-546)
// and GPT2ValuePolicy.act (LLM_RL/algorithms/value_rl_base/gpt2/interface.py:281-320).  The
// reference re-tokenises and re-prefills the whole left-padded history on every turn; here every
// env keeps its K/V in HBM across turns and only the NEW tokens of a turn are forwarded.
//
// A forward call processes B envs x C token slots ("chunk"): env b contributes cnt[b] <= C new tokens
// at positions len[b] .. len[b]+cnt[b]-1 (row r = b*C + j).  C = 1 is decode; C = 8 is the per-turn
// prefill of [action terminator + observation tokens].  Rows j >= cnt[b] are padding (computed, never
// stored to the cache, never attended to).
//
// Data layout in HBM
//   residual stream x   f32  [B*C][d]
//   GEMM activations    bf16 [B*C][d | 3d | d_ff]
//   KV cache            bf16 [layer][2 (K,V)][B][Tmax][H*64]  (token-major: one env's K (or V) for all heads is ONE
//                       contiguous T x 1.5 KiB stream; appends are whole 1.5 KiB rows; a wave reads 8 rows x 128 B per load)
//   weights             bf16 [N][K] ("out x in", K contiguous) so activation and weight fragments are
//                       both K-major for the MFMA operand loads (gemm_bf16.h)
#include "../../include/lmrl_amd.h"
#include "common.h"
#include <hip/hip_ext.h>
#include "gemm_dispatch.h"
#include "skinny_gemm.h"
#ifdef LMRL_TOOLS
#include "ablate_tools.h"
#endif

namespace lmrl {

struct Gpt2Layer {
    const float *ln1_g, *ln1_b;
    const uint16_t *w_qkv; const float *b_qkv;
    const uint16_t *w_proj; const float *b_proj;
    const float *ln2_g, *ln2_b;
    const uint16_t *w_fc; const float *b_fc;
    const uint16_t *w_fc2; const float *b_fc2;
    // LayerNorm folded into the consuming GEMM (owned by the model, built at create):
    //   W' = bf16(W . diag(gamma)), colsum[n] = sum_k W'[n][k], bias' = bias + W beta
    uint16_t *wf_qkv; float *cs_qkv, *bf_qkv;
    uint16_t *wf_fc; float *cs_fc, *bf_fc;
};

struct Gpt2Model {
    lmrl_gpt2_config cfg;
    const uint16_t *wte, *wpe;
    const float *lnf_g, *lnf_b;
    Gpt2Layer *layers;
#ifdef LMRL_TOOLS
    AblateAux ablate_aux;
#endif
};

// ------------------------------------------------------------------------------------------ layernorm
// y[r][:] (bf16) = (x[r][:] - mean) * rsqrt(var + eps) * g + b ; one wave per row; d <= 64*4*MAXV.
// `rows_idx` (optional) gathers source rows (used for the final LN over each env's last token only).
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ gam,
                                                        const float *__restrict__ bet, uint16_t *__restrict__ y,
                                                        const int32_t *__restrict__ rows_idx, int rows, int d, float eps) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    int src = r;
    if (rows_idx) {
        src = rows_idx[r];
        if (src < 0) return;   // env contributed no token
    }
    const float *xr = x + (size_t)src * d;
    f32x4 v[MAXV], g4[MAXV], b4[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {   // issue every load up front: x row, gamma and beta are independent
        const int c = (k * 64 + lane) * 4;
        const bool in = c < d;
        v[k] = in ? *reinterpret_cast<const f32x4 *>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        g4[k] = in ? *reinterpret_cast<const f32x4 *>(gam + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        b4[k] = in ? *reinterpret_cast<const f32x4 *>(bet + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < MAXV; k++) s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float t = v[k][e] - mean; q += t * t; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)d + eps);
    uint16_t *yr = y + (size_t)r * d;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        if (c < d) {
            uint16_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = f32_to_bf16_rn((v[k][e] - mean) * rstd * g4[k][e] + b4[k][e]);
            uint2 pk;
            pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *reinterpret_cast<uint2 *>(yr + c) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm folding
// One wave per output row n:  W'[n][k] = bf16(W[n][k] * gamma[k]) ; colsum[n] = sum_k W'[n][k] ; bias'[n] = bias[n] + sum_k W[n][k] beta[k]
__global__ __launch_bounds__(256) void fold_ln_kernel(const uint16_t *__restrict__ W, const float *__restrict__ gam,
                                                      const float *__restrict__ bet, const float *__restrict__ bias,
                                                      uint16_t *__restrict__ Wf, float *__restrict__ colsum, float *__restrict__ biasf,
                                                      int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float cs = 0.f, bs = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = bf16_to_f32(W[(size_t)n * K + k]);
        const uint16_t wf = f32_to_bf16_rn(w * gam[k]);
        Wf[(size_t)n * K + k] = wf;
        cs += bf16_to_f32(wf);
        bs += w * bet[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cs += __shfl_xor(cs, o); bs += __shfl_xor(bs, o); }
    if (lane == 0) { colsum[n] = cs; biasf[n] = bias[n] + bs; }
}

// embed for the LN-folded path: x = wte[tok] + wpe[pos] (fp32), xb = bf16(x), stats slot 0 = (sum x, sum x^2), other slots 0
template <int MAXV>
__global__ __launch_bounds__(256) void embed_stats_kernel(const uint16_t *__restrict__ wte, const uint16_t *__restrict__ wpe,
                                                          const int32_t *__restrict__ tokens, const int32_t *__restrict__ cnt,
                                                          const int32_t *__restrict__ len, float *__restrict__ x, uint16_t *__restrict__ xb,
                                                          float2 *__restrict__ stats, int nslots, int B, int C, int d, int vocab, int n_pos,
                                                          const int32_t *__restrict__ row_map, const int32_t *__restrict__ m_dev) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= B * C) return;
    int b = r / C, j = r - b * C;
    if (row_map) {                          // ragged prefill: row r of the compacted batch belongs to (env, slot) = row_map[r]
        if (r >= *m_dev) return;
        b = row_map[r] >> 5; j = row_map[r] & 31;
    }
    int tok = tokens[b * C + j];
    int pos = len[b] + j;
    const bool valid = j < cnt[b];
    if (!valid || tok < 0 || tok >= vocab) tok = 0;
    if (!valid || pos >= n_pos) pos = 0;
    const uint16_t *te = wte + (size_t)tok * d, *pe = wpe + (size_t)pos * d;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        if (c < d) {
            const uint2 a = *reinterpret_cast<const uint2 *>(te + c), p = *reinterpret_cast<const uint2 *>(pe + c);
            const f32x4 v = f32x4{bf16_to_f32((uint16_t)(a.x & 0xffff)) + bf16_to_f32((uint16_t)(p.x & 0xffff)),
                                  bf16_to_f32((uint16_t)(a.x >> 16)) + bf16_to_f32((uint16_t)(p.x >> 16)),
                                  bf16_to_f32((uint16_t)(a.y & 0xffff)) + bf16_to_f32((uint16_t)(p.y & 0xffff)),
                                  bf16_to_f32((uint16_t)(a.y >> 16)) + bf16_to_f32((uint16_t)(p.y >> 16))};
            *reinterpret_cast<f32x4 *>(x + (size_t)r * d + c) = v;
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2 *>(xb + (size_t)r * d + c) = o;
            s1 += (v[0] + v[1]) + (v[2] + v[3]);
            s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    for (int sl = lane; sl < nslots; sl += 64) stats[(size_t)r * nslots + sl] = sl == 0 ? make_float2(s1, s2) : make_float2(0.f, 0.f);
}

// ------------------------------------------------------------------------------------------ fused small kernels
// embed + first LayerNorm: x[r] = wte[tok] + wpe[pos] (fp32 residual stream) and h[r] = LN(x[r]) (bf16) in one pass.
template <int MAXV>
__global__ __launch_bounds__(256) void embed_ln_kernel(const uint16_t *__restrict__ wte, const uint16_t *__restrict__ wpe,
                                                       const int32_t *__restrict__ tokens, const int32_t *__restrict__ cnt,
                                                       const int32_t *__restrict__ len, const float *__restrict__ gam,
                                                       const float *__restrict__ bet, float *__restrict__ x, uint16_t *__restrict__ y,
                                                       int B, int C, int d, int vocab, int n_pos, float eps) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= B * C) return;
    const int b = r / C, j = r - b * C;
    int tok = tokens[r];
    int pos = len[b] + j;
    const bool valid = j < cnt[b];
    if (!valid || tok < 0 || tok >= vocab) tok = 0;
    if (!valid || pos >= n_pos) pos = 0;
    const uint16_t *te = wte + (size_t)tok * d, *pe = wpe + (size_t)pos * d;
    f32x4 v[MAXV], g4[MAXV], b4[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f}; g4[k] = v[k]; b4[k] = v[k];
        if (c < d) {
            const uint2 a = *reinterpret_cast<const uint2 *>(te + c), p = *reinterpret_cast<const uint2 *>(pe + c);
            v[k] = f32x4{bf16_to_f32((uint16_t)(a.x & 0xffff)) + bf16_to_f32((uint16_t)(p.x & 0xffff)),
                         bf16_to_f32((uint16_t)(a.x >> 16)) + bf16_to_f32((uint16_t)(p.x >> 16)),
                         bf16_to_f32((uint16_t)(a.y & 0xffff)) + bf16_to_f32((uint16_t)(p.y & 0xffff)),
                         bf16_to_f32((uint16_t)(a.y >> 16)) + bf16_to_f32((uint16_t)(p.y >> 16))};
            g4[k] = *reinterpret_cast<const f32x4 *>(gam + c);
            b4[k] = *reinterpret_cast<const f32x4 *>(bet + c);
            *reinterpret_cast<f32x4 *>(x + (size_t)r * d + c) = v[k];
        }
        s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float t = v[k][e] - mean; q += t * t; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)d + eps);
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        if (c < d) {
            uint16_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = f32_to_bf16_rn((v[k][e] - mean) * rstd * g4[k][e] + b4[k][e]);
            uint2 pk;
            pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *reinterpret_cast<uint2 *>(y + (size_t)r * d + c) = pk;
        }
    }
}

// final LayerNorm of each env's LAST new token + len[b] += cnt[b] (one launch: the gather, the LayerNorm and the cache-length advance)
template <int MAXV>
__global__ __launch_bounds__(256) void final_ln_advance_kernel(const float *__restrict__ x, const float *__restrict__ gam,
                                                               const float *__restrict__ bet, uint16_t *__restrict__ y,
                                                               const int32_t *__restrict__ cnt, int32_t *__restrict__ len, int B, int C,
                                                               int d, float eps, const int32_t *__restrict__ off, int compact) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int n = min(cnt[b], C);
    if (n <= 0) return;
    if (lane == 0) len[b] += n;
    if (!y) return;
    // compact: x holds ONE row per env (the last new token's, gathered before the last layer's projection / MLP), row b
    const float *xr = compact ? x + (size_t)b * d : x + ((size_t)(off ? off[b] : b * C) + n - 1) * d;
    f32x4 v[MAXV], g4[MAXV], b4[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        const bool in = c < d;
        v[k] = in ? *reinterpret_cast<const f32x4 *>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        g4[k] = in ? *reinterpret_cast<const f32x4 *>(gam + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        b4[k] = in ? *reinterpret_cast<const f32x4 *>(bet + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float t = v[k][e] - mean; q += t * t; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)d + eps);
#pragma unroll
    for (int k = 0; k < MAXV; k++) {
        const int c = (k * 64 + lane) * 4;
        if (c < d) {
            uint16_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = f32_to_bf16_rn((v[k][e] - mean) * rstd * g4[k][e] + b4[k][e]);
            uint2 pk;
            pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *reinterpret_cast<uint2 *>(y + (size_t)b * d + c) = pk;
        }
    }
}

// Last layer of a chunk forward: only each env's LAST new token goes on through the output projection and the MLP (its hidden state is all
// the caller reads; the other rows of the chunk only had to leave their K / V rows in the cache, which the attention launch has done).
// One wave per env: att row (bf16) and residual row (fp32) of that token -> row b of the compact buffers; envs without new tokens get zeros.
__global__ __launch_bounds__(256) void gather_last_rows_kernel(const uint16_t *__restrict__ att, const float *__restrict__ x,
                                                               const int32_t *__restrict__ cnt, const int32_t *__restrict__ off, int B, int C, int d,
                                                               uint16_t *__restrict__ att_c, float *__restrict__ x_c, float2 *__restrict__ stats_c,
                                                               int nslots) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    // moment slots of the row: the projection's epilogue fills the real ones, the padding slots (slot pitch 8 * NQ) must read as zero
    for (int k = lane; k < nslots; k += 64) stats_c[(size_t)b * nslots + k] = make_float2(0.f, 0.f);
    const int n = min(cnt[b], C);
    const size_t r = (size_t)(off ? off[b] : b * C) + (n > 0 ? n - 1 : 0);
    for (int c8 = lane * 8; c8 < d; c8 += 512) {
        u32x4 a = u32x4{0u, 0u, 0u, 0u};
        f32x4 x0 = f32x4{0.f, 0.f, 0.f, 0.f}, x1 = x0;
        if (n > 0) {
            a = *reinterpret_cast<const u32x4 *>(att + r * d + c8);
            x0 = *reinterpret_cast<const f32x4 *>(x + r * d + c8);
            x1 = *reinterpret_cast<const f32x4 *>(x + r * d + c8 + 4);
        }
        *reinterpret_cast<u32x4 *>(att_c + (size_t)b * d + c8) = a;
        *reinterpret_cast<f32x4 *>(x_c + (size_t)b * d + c8) = x0;
        *reinterpret_cast<f32x4 *>(x_c + (size_t)b * d + c8 + 4) = x1;
    }
}

// ------------------------------------------------------------------------------------------ attention
// DPP lane move on a float (VALU, no LDS round trip): CTRL as in the ISA (quad_perm 0x00-0xFF, row_half_mirror 0x141 ...)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}

// One wave per (env b, head h).  Keys/values stream from the KV cache (positions < len[b]) and from this
// chunk's own qkv rows (positions >= len[b]); the new K/V rows are appended to the cache on the way.
// Lane (rr = lane>>3, cc = lane&7) holds 8 head-dims (16 B) of row t0+rr: a load instruction covers
// 8 rows x 128 B contiguous.  Scores are reduced over the 8 lanes of a row with 3 xor-shuffles, the
// online-softmax state (m, l) is wave-uniform per query, and the P.V partial sums are reduced over the
// 8 row-groups at the end (xor 8/16/32).
template <int C>
__global__ __launch_bounds__(256) void attention_kernel(const uint16_t *__restrict__ qkv,   // [B*C][3d]
                                                        uint16_t *__restrict__ kcache, uint16_t *__restrict__ vcache,
                                                        const int32_t *__restrict__ cnt, const int32_t *__restrict__ len,
                                                        uint16_t *__restrict__ out,          // [B*C][d]
                                                        int B, int H, int Tmax, int d, const int32_t *__restrict__ off = nullptr) {
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave_id >= B * H) return;
    const int b = wave_id / H, h = wave_id - b * H;
    const int n_new = min(cnt[b], C);
    if (n_new <= 0) return;
    const size_t row0 = off ? (size_t)off[b] : (size_t)b * C;       // first row of this env in the (possibly compacted) batch
    const int L0 = len[b];
    const int T = L0 + n_new;
    const int rr = lane >> 3, cc = lane & 7;
    const size_t ld = (size_t)3 * d;
    uint16_t *kc = kcache + (size_t)b * Tmax * d + (size_t)h * 64;   // token-major cache: row t of head h at kc + t*d
    uint16_t *vc = vcache + (size_t)b * Tmax * d + (size_t)h * 64;
    const uint16_t *qbase = qkv + row0 * ld + (size_t)h * 64;

    // append this chunk's K/V rows to the cache (row j -> position L0 + j)
    for (int j = rr; j < n_new; j += 8) {
        if (L0 + j < Tmax) {
            const uint4 kv = *reinterpret_cast<const uint4 *>(qbase + (size_t)j * ld + d + cc * 8);
            const uint4 vv = *reinterpret_cast<const uint4 *>(qbase + (size_t)j * ld + 2 * d + cc * 8);
            *reinterpret_cast<uint4 *>(kc + (size_t)(L0 + j) * d + cc * 8) = kv;
            *reinterpret_cast<uint4 *>(vc + (size_t)(L0 + j) * d + cc * 8) = vv;
        }
    }

    // this lane's 8-dim slice of every query, pre-scaled by 1/sqrt(64)
    float q[C][8];
#pragma unroll
    for (int j = 0; j < C; j++) {
        uint4 qv = uint4{0, 0, 0, 0};
        if (j < n_new) qv = *reinterpret_cast<const uint4 *>(qbase + (size_t)j * ld + cc * 8);
        const uint32_t w[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            q[j][2 * k] = bf16_to_f32((uint16_t)(w[k] & 0xffff)) * 0.125f;
            q[j][2 * k + 1] = bf16_to_f32((uint16_t)(w[k] >> 16)) * 0.125f;
        }
    }
    // Online softmax state is kept PER ROW-GROUP (the 8 lanes sharing rr): a lane only ever accumulates the keys of its own
    // rows, so its (m, l, o) can be rescaled independently; the 8 groups are merged once at the end with log-sum-exp
    // weights.  Per key block and query this leaves only the 8-lane dot-product reduction, done with DPP moves
    // (quad_perm xor1, xor2, row_half_mirror) instead of ds_bpermute shuffles.
    float m[C], l[C], o[C][8];
#pragma unroll
    for (int j = 0; j < C; j++) {
        m[j] = -1e30f; l[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) o[j][e] = 0.f;
    }

    // U key blocks (8 rows each) per iteration: all 2U loads of a lane are issued before the first use so that a decode
    // wave keeps several 16-byte requests in flight (the kernel is HBM-latency/bandwidth bound, not ALU bound).
    constexpr int U = (C == 1) ? 4 : 1;
    for (int t0 = 0; t0 < T; t0 += 8 * U) {
        uint4 kraw[U], vraw[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * 8 + rr;
            const int tc = t < T ? t : T - 1;
            // position < L0: cache ; otherwise this chunk's own row
            const uint16_t *kp = tc < L0 ? kc + (size_t)tc * d + cc * 8 : qbase + (size_t)(tc - L0) * ld + d + cc * 8;
            const uint16_t *vp = tc < L0 ? vc + (size_t)tc * d + cc * 8 : qbase + (size_t)(tc - L0) * ld + 2 * d + cc * 8;
            kraw[u] = *reinterpret_cast<const uint4 *>(kp);
            vraw[u] = *reinterpret_cast<const uint4 *>(vp);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * 8 + rr;
            const bool in_range = t < T;
            float kf[8], vf[8];
            {
                const uint32_t kw[4] = {kraw[u].x, kraw[u].y, kraw[u].z, kraw[u].w}, vw[4] = {vraw[u].x, vraw[u].y, vraw[u].z, vraw[u].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    kf[2 * k] = bf16_to_f32((uint16_t)(kw[k] & 0xffff)); kf[2 * k + 1] = bf16_to_f32((uint16_t)(kw[k] >> 16));
                    vf[2 * k] = bf16_to_f32((uint16_t)(vw[k] & 0xffff)); vf[2 * k + 1] = bf16_to_f32((uint16_t)(vw[k] >> 16));
                }
            }
#pragma unroll
            for (int j = 0; j < C; j++) {
                if (j >= n_new) continue;   // wave-uniform
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) s = fmaf(q[j][e], kf[e], s);
                s += dpp_f32<0xB1>(s);      // quad_perm [1,0,3,2]  (lane ^ 1)
                s += dpp_f32<0x4E>(s);      // quad_perm [2,3,0,1]  (lane ^ 2)
                s += dpp_f32<0x141>(s);     // row_half_mirror: the other quad of this 8-lane group
                const bool ok = in_range && t <= L0 + j;   // causal: query j sits at position L0 + j
                const float m_new = ok ? fmaxf(m[j], s) : m[j];
                const float alpha = __expf(m[j] - m_new);
                const float p = ok ? __expf(s - m_new) : 0.f;
                l[j] = l[j] * alpha + p;
#pragma unroll
                for (int e = 0; e < 8; e++) o[j][e] = fmaf(p, vf[e], o[j][e] * alpha);
                m[j] = m_new;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < C; j++) {
        if (j >= n_new) continue;
        // merge the 8 row-groups: M = max m_g ; weight w_g = exp(m_g - M)
        float mm = m[j];
        mm = fmaxf(mm, __shfl_xor(mm, 8)); mm = fmaxf(mm, __shfl_xor(mm, 16)); mm = fmaxf(mm, __shfl_xor(mm, 32));
        const float w = __expf(m[j] - mm);
        float lt = l[j] * w;
        lt += __shfl_xor(lt, 8); lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
        const float inv = 1.f / lt;
        float r8[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float a = o[j][e] * w;
            a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            r8[e] = a * inv;
        }
        if (rr == 0) {
            uint4 pk;
            pk.x = pack_bf16x2(r8[0], r8[1]);
            pk.y = pack_bf16x2(r8[2], r8[3]);
            pk.z = pack_bf16x2(r8[4], r8[5]);
            pk.w = pack_bf16x2(r8[6], r8[7]);
            *reinterpret_cast<uint4 *>(out + (row0 + j) * d + (size_t)h * 64 + cc * 8) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------ decode attention, single-sweep form
// Single-token decode (C = 1), one wave per (env, head) as attention_kernel<1>, restructured so that a strand costs TWO dependent memory
// latencies instead of three or four: the scalar length load, then ONE batch of vector loads covering the whole context — ceil(L0 / 8)
// key blocks, a wave-uniform count, so up to 8 U cached positions are requested at once and no row beyond the context is touched
// (speculating past the length was measured: the over-read costs more bandwidth than the saved latency is worth).  The query and the
// new token's own K/V row (taken from the qkv buffer, attended last, appended to the cache) are requested before the length arrives.
// Softmax state per 8-lane row group, groups merged at the end (as above); the QK dot product is v_dot2_f32_bf16 on packed operands.
//
// PFX (indexed prompt prefix, lmrl_gpt2_forward_prefixed): positions [0, pfx.n[b]) of env b are the rows of row pfx.row[b] of ANOTHER
// session's cache (a prompt-prefix cache: one prefill per distinct prompt) and are read from there — nothing is copied per env, and envs
// that share a prompt read the same bytes.  pfx.order (optional) lists the envs grouped by prefix row; workgroup ids are then spread so
// that each XCD walks a CONTIGUOUS stretch of that list: the waves resident on an XCD at any moment share a handful of prompts, whose
// rows stay in that XCD's L2.  Positions >= pfx.n[b] live in the env's own cache at their absolute position, as without a prefix.
struct DecodePrefix {
    const uint16_t *k, *v;       // this layer's K / V block of the prefix session
    const int32_t *row, *n, *order;
    int tmax;
};

// SPLIT (round 6, small batches): 4 waves share ONE (env, head) item — wave w takes the position batches w, w + 4, ... (32 positions each), the four
// (max, sum, o) states meet in LDS and wave 0 writes the row.  At 8 envs x 12 heads the one-wave form is 96 waves walking ~5 dependent load batches
// each (8.3 us per launch for kilobytes, profiles/r06_maze_b8_kernel_stats.csv); split, every batch of an item is in flight at once.
template <int U, bool PFX, bool HALF, int SPLIT = 1>
__device__ __forceinline__ void attention_decode_item(int wave_id, const uint16_t *__restrict__ qkv, uint16_t *__restrict__ kcache, uint16_t *__restrict__ vcache,
                                                      const int32_t *__restrict__ cnt, const int32_t *__restrict__ len, uint16_t *__restrict__ out,
                                                      int B, int H, int Tmax, int d, const int32_t *__restrict__ off, int n_shared, int append,
                                                      const DecodePrefix &pfx) {
    typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    if (wave_id >= B * H) return;
    int b = wave_id / H;
    const int h = wave_id - b * H;
    if (PFX && pfx.order) b = pfx.order[b];
    const int rr = lane >> 3, cc = lane & 7;
    // wave-uniform bases (SGPR): env 0's rows of this head, and this env's element offset from them.  Positions < n_shared hold the same
    // values in every env (a prompt prefix broadcast by lmrl_gpt2_kv_broadcast): they are read from env 0, i.e. from L2, not once per env
    const char *kc = reinterpret_cast<const char *>(kcache + (size_t)h * 64);
    const char *vc = reinterpret_cast<const char *>(vcache + (size_t)h * 64);
    const uint32_t env_row = (uint32_t)b * (uint32_t)Tmax;
    const size_t row0 = off ? (size_t)off[b] : (size_t)b;           // first (only) row of this env in the possibly compacted batch
    const uint16_t *qbase = qkv + row0 * ((size_t)3 * d) + (size_t)h * 64 + cc * 8;
    const uint4 qv = *reinterpret_cast<const uint4 *>(qbase);
    const uint4 knew = *reinterpret_cast<const uint4 *>(qbase + d), vnew = *reinterpret_cast<const uint4 *>(qbase + 2 * d);
    if (cnt[b] <= 0) return;                                         // wave-uniform: finished env (its rows are not in the batch)
    const int L0 = HALF ? (len[b] + 1) / 2 : len[b];
    // PFX: positions < pn come from row prow of the prefix session (wave-uniform scalars; the per-lane choice is two selects)
    uint32_t pn = 0u, prow = 0u;
    const char *pkc = kc, *pvc = vc;
    if (PFX) {
        const int r = pfx.row[b];
        pn = r >= 0 ? (uint32_t)pfx.n[b] : 0u;
        prow = (uint32_t)max(r, 0) * (uint32_t)pfx.tmax;
        pkc = reinterpret_cast<const char *>(pfx.k + (size_t)h * 64);
        pvc = reinterpret_cast<const char *>(pfx.v + (size_t)h * 64);
    }
    uint4 kr[U], vr[U];
    // every cached K / V byte is read by exactly ONE wave per decode step (the few broadcast-header rows aside): streaming (nt) loads — round 5,
    // A/B on one box (tools/_ab, bench.py --steps 10, twice each): 25.2 / 25.7 -> 23.4 / 24.4 us per launch, episode 47.34 / 47.41 -> 46.74 / 46.98 ms.
    // Same bytes, same arithmetic: bit-identical results.  -DLMRL_DEC_NO_NT (LMRL_GPT2_EXTRA of build.py): the default cache policy, for the A/B.
#ifndef LMRL_DEC_NO_NT
    typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
#define LMRL_DEC_LD16(P) __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt *>(P)))
#else
#define LMRL_DEC_LD16(P) (*reinterpret_cast<const uint4 *>(P))
#endif
#define LMRL_DEC_LOAD(TBASE)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < U; u++)                                                             \
        if ((TBASE) + u * 8 < L0) {                                  /* wave-uniform: block u holds cached positions */ \
            const int tt_ = (TBASE) + u * 8 + rr;                                                             \
            const uint32_t tc_ = (uint32_t)(tt_ < L0 ? tt_ : L0 - 1);                                         \
            if (PFX) {                                                                                        \
                const bool in_p_ = tc_ < pn;                                                                  \
                const uint32_t bo_ = (((in_p_ ? prow : env_row) + tc_) * (uint32_t)d + (uint32_t)cc * 8u) * 2u; \
                kr[u] = *reinterpret_cast<const uint4 *>((in_p_ ? pkc : kc) + bo_);                           \
                vr[u] = *reinterpret_cast<const uint4 *>((in_p_ ? pvc : vc) + bo_);                           \
            } else {                                                                                          \
                const uint32_t bo_ = (((tc_ < (uint32_t)n_shared ? 0u : env_row) + tc_) * (uint32_t)d + (uint32_t)cc * 8u) * 2u; \
                kr[u] = LMRL_DEC_LD16(kc + bo_);                                                              \
                vr[u] = LMRL_DEC_LD16(vc + bo_);                                                              \
            }                                                                                                 \
        }
    const int sw = SPLIT > 1 ? (int)(threadIdx.x >> 6) : 0;         // SPLIT: this wave's first position batch (and its stride below)
    const int t_first = sw * 8 * U;
    LMRL_DEC_LOAD(t_first);
    uint32_t qp[4];                                                  // query slice as packed bf16 pairs, pre-scaled by 1/sqrt(64) (exact)
    {
        const uint32_t w[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int k = 0; k < 4; k++) qp[k] = pack_bf16x2(bf16_to_f32((uint16_t)(w[k] & 0xffff)) * 0.125f, bf16_to_f32((uint16_t)(w[k] >> 16)) * 0.125f);
    }
    float m = -1e30f, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = 0.f;
#define LMRL_DEC_ROW(KW, VW, OK)                                                                               \
    do {                                                                                                       \
        const uint32_t kw_[4] = {(KW).x, (KW).y, (KW).z, (KW).w};                                              \
        const uint32_t vw_[4] = {(VW).x, (VW).y, (VW).z, (VW).w};                                              \
        float s_ = 0.f;                                                                                        \
        _Pragma("unroll") for (int k = 0; k < 4; k++)                                                          \
            s_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_v, qp[k]), __builtin_bit_cast(bf16x2_v, kw_[k]), s_, false); \
        s_ += dpp_f32<0xB1>(s_);      /* quad_perm [1,0,3,2]  (lane ^ 1) */                                    \
        s_ += dpp_f32<0x4E>(s_);      /* quad_perm [2,3,0,1]  (lane ^ 2) */                                    \
        s_ += dpp_f32<0x141>(s_);     /* row_half_mirror: the other quad of this 8-lane group */               \
        const float m_new_ = (OK) ? fmaxf(m, s_) : m;                                                          \
        const float alpha_ = __expf(m - m_new_);                                                               \
        const float pr_ = (OK) ? __expf(s_ - m_new_) : 0.f;                                                    \
        l = l * alpha_ + pr_;                                                                                  \
        _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                        \
            o[2 * k] = fmaf(pr_, __uint_as_float(vw_[k] << 16), o[2 * k] * alpha_);                            \
            o[2 * k + 1] = fmaf(pr_, __uint_as_float(vw_[k] & 0xffff0000u), o[2 * k + 1] * alpha_);            \
        }                                                                                                      \
        m = m_new_;                                                                                            \
    } while (0)
    for (int t0 = t_first; t0 < L0; t0 += 8 * U * SPLIT) {
        if (t0 > t_first) LMRL_DEC_LOAD(t0);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (t0 + u * 8 < L0) {                                   // wave-uniform
                const bool ok = t0 + u * 8 + rr < L0;
                LMRL_DEC_ROW(kr[u], vr[u], ok);
            }
    }
    if (sw == 0) {   // position L0 (this step's own token): row group 0 attends it last
        const bool ok = rr == 0;
        LMRL_DEC_ROW(knew, vnew, ok);
    }
#undef LMRL_DEC_LOAD
#undef LMRL_DEC_LD16
#undef LMRL_DEC_ROW
    // merge the 8 row-groups: M = max m_g ; weight w_g = exp(m_g - M)
    float mm = m;
    mm = fmaxf(mm, __shfl_xor(mm, 8)); mm = fmaxf(mm, __shfl_xor(mm, 16)); mm = fmaxf(mm, __shfl_xor(mm, 32));
    const float w = __expf(m - mm);
    float lt = l * w;
    lt += __shfl_xor(lt, 8); lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
    float r8[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float a = o[e] * w;
        a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        r8[e] = a;
    }
    if (SPLIT > 1) {      // the four waves' states of this item: (max, sum) + 64 unnormalised outputs each, merged by wave 0 in wave order
        __shared__ float xs[4][66];
        if (rr == 0) {
#pragma unroll
            for (int e = 0; e < 8; e++) xs[sw][cc * 8 + e] = r8[e];
            if (cc == 0) { xs[sw][64] = mm; xs[sw][65] = lt; }
        }
        __syncthreads();
        if (sw != 0) return;
        float M4 = xs[0][64];
#pragma unroll
        for (int q = 1; q < 4; q++) M4 = fmaxf(M4, xs[q][64]);
        float L4 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) r8[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float sc = __expf(xs[q][64] - M4);
            L4 += xs[q][65] * sc;
#pragma unroll
            for (int e = 0; e < 8; e++) r8[e] += xs[q][cc * 8 + e] * sc;
        }
        lt = L4;
    }
    const float inv = 1.f / lt;
#pragma unroll
    for (int e = 0; e < 8; e++) r8[e] *= inv;
    if (rr == 0) {
        uint4 pk;
        pk.x = pack_bf16x2(r8[0], r8[1]); pk.y = pack_bf16x2(r8[2], r8[3]);
        pk.z = pack_bf16x2(r8[4], r8[5]); pk.w = pack_bf16x2(r8[6], r8[7]);
        *reinterpret_cast<uint4 *>(out + row0 * d + (size_t)h * 64 + cc * 8) = pk;
        if (append && L0 < Tmax) {     // append the new token's K/V row to the cache (unless the qkv GEMM did) — at the very end: the wave's
                                       // stores then follow its load stream instead of sitting inside it
            *reinterpret_cast<uint4 *>(const_cast<char *>(kc) + (((size_t)env_row + L0) * d + cc * 8) * 2) = knew;
            *reinterpret_cast<uint4 *>(const_cast<char *>(vc) + (((size_t)env_row + L0) * d + cc * 8) * 2) = vnew;
        }
    }
}

// NI (round 6, A/B): (env, head) items per wave.  1: one wave per item, B * H waves (1.7 resident rounds of the chip at 1024 envs x 12 heads).  2 / 3: a grid of
// ceil(B * H / NI) waves — 6144 / 4096 at the bench shape, all resident at once — whose wave w sweeps items w, w + W, w + 2 W one after the other
// (LMRL_FWD_ATTN_ITEMS2 / _ITEMS3).  Per-item arithmetic unchanged: bit-identical results.
template <int U, bool PFX, bool HALF = false, int NI = 1>   // HALF: timing-only ablation — only the first half of the cached positions is read (half the cache lines and bytes)
__global__ __launch_bounds__(256) void attention_decode_kernel(const uint16_t *__restrict__ qkv,   // [rows][3d]
                                                               uint16_t *__restrict__ kcache, uint16_t *__restrict__ vcache,
                                                               const int32_t *__restrict__ cnt, const int32_t *__restrict__ len,
                                                               uint16_t *__restrict__ out,          // [rows][d]
                                                               int B, int H, int Tmax, int d, const int32_t *__restrict__ off, int n_shared,
                                                               int append, DecodePrefix pfx) {
    int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (PFX && pfx.order && (gridDim.x & 7) == 0)            // workgroup ids round-robin over the 8 XCDs: XCD x gets ids [x, x + 8, ..]
        wave_id = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4 + (threadIdx.x >> 6);
#pragma unroll 1
    for (int it = 0; it < NI; it++)
        attention_decode_item<U, PFX, HALF>(wave_id + it * (int)gridDim.x * 4, qkv, kcache, vcache, cnt, len, out, B, H, Tmax, d, off, n_shared, append, pfx);
}

template <int U, bool PFX>   // one 4-wave workgroup per (env, head): small batches (LMRL_FWD_SKINNY)
__global__ __launch_bounds__(256) void attention_decode_split_kernel(const uint16_t *__restrict__ qkv, uint16_t *__restrict__ kcache, uint16_t *__restrict__ vcache,
                                                                     const int32_t *__restrict__ cnt, const int32_t *__restrict__ len, uint16_t *__restrict__ out,
                                                                     int B, int H, int Tmax, int d, const int32_t *__restrict__ off, int n_shared, int append,
                                                                     DecodePrefix pfx) {
    attention_decode_item<U, PFX, false, 4>((int)blockIdx.x, qkv, kcache, vcache, cnt, len, out, B, H, Tmax, d, off, n_shared, append, pfx);
}

// ------------------------------------------------------------------------------------------ chunk attention on MFMA
// C = 8 chunk rows per env: the VALU kernel above spends ~40 instructions per (query, 8 keys); here one wave still owns an
// (env, head) but does S^T = K.Q^T and O^T = V^T.P^T with v_mfma_f32_16x16x32_bf16 over 32-key blocks:
//   step A  A-operand = K rows straight from HBM (16 B/lane, row = key, k = head dims), B-operand = the 8 queries padded to 16
//           columns.  MFMA row i of sub-block sb is mapped to key t0 + (i/4)*8 + sb*4 + (i%4) so that after both sub-blocks
//           lane (query j = lane&15, group g = lane>>4) holds the scores of the 8 CONTIGUOUS keys t0 + 8g .. t0 + 8g + 7 —
//   step B  online softmax per query; the max is shared by the 4 lane groups of a query (xor 16/32) because the O^T
//           accumulators of a query live in all 4 groups;
//   step C  — exactly the k-slot order the P^T B-operand needs, so P never moves between lanes; the V^T A-operand (8 keys of
//           one head-dim per lane) is gathered from a per-wave LDS copy of the 32 x 64 V block (row stride 136 B).
// Rows t >= len[b] come from this chunk's own qkv rows, as in the VALU kernel; new K/V rows are appended to the cache.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

template <int C>   // chunk rows per env: 8 (the per-turn Wordle chunk; half of the 16 MFMA query columns) or 16 (all of them)
__global__ __launch_bounds__(256) void attention_chunk_mfma_kernel(const uint16_t *__restrict__ qkv, uint16_t *__restrict__ kcache,
                                                                    uint16_t *__restrict__ vcache, const int32_t *__restrict__ cnt,
                                                                    const int32_t *__restrict__ len, uint16_t *__restrict__ out, int B,
                                                                    int H, int Tmax, int d, const int32_t *__restrict__ off) {
    static_assert(C == 8 || C == 16, "the query block is one 16-column MFMA tile");
    // per-wave [32 keys][64 dims] V block, 128-byte rows; 16-byte chunk c of row r at c ^ chunk_swz(r): the 32 lanes of a ds_read_b64_tr_b16
    // half-wave (rows 8 g + (j >> 2), g in {0, 1} or {2, 3}) then cover all 64 banks once (flash_attn_train.hip: flash_swz)
    constexpr int VROW = 128;
    __shared__ __attribute__((aligned(16))) unsigned char vlds_all[4][32 * VROW];
    auto chunk_swz = [](int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave_id = blockIdx.x * 4 + wave;
    if (wave_id >= B * H) return;
    const int b = wave_id / H, h = wave_id - b * H;
    const int n_new = min(cnt[b], C);
    if (n_new <= 0) return;
    const int L0 = len[b];
    const int T = L0 + n_new;
    const size_t ld = (size_t)3 * d;
    uint16_t *kc = kcache + (size_t)b * Tmax * d + (size_t)h * 64;   // token-major cache: row t of head h at kc + t*d
    uint16_t *vc = vcache + (size_t)b * Tmax * d + (size_t)h * 64;
    const size_t row0 = off ? (size_t)off[b] : (size_t)b * C;      // first row of this env in the (possibly compacted) chunk batch
    const uint16_t *qbase = qkv + row0 * ld + (size_t)h * 64;
    unsigned char *vlds = vlds_all[wave];
    const int j = lane & 15, g = lane >> 4;

    {   // append this chunk's K/V rows to the cache
        const int cc = lane & 7;
#pragma unroll
        for (int rr = lane >> 3; rr < C; rr += 8) {
            if (rr < n_new && L0 + rr < Tmax) {
                *reinterpret_cast<uint4 *>(kc + (size_t)(L0 + rr) * d + cc * 8) = *reinterpret_cast<const uint4 *>(qbase + (size_t)rr * ld + d + cc * 8);
                *reinterpret_cast<uint4 *>(vc + (size_t)(L0 + rr) * d + cc * 8) = *reinterpret_cast<const uint4 *>(qbase + (size_t)rr * ld + 2 * d + cc * 8);
            }
        }
    }
    // B operand of step A: query j (zero beyond n_new), dims kk*32 + g*8 ..
    bf16x8_t qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        u32x4 raw = u32x4{0u, 0u, 0u, 0u};
        if (j < n_new) raw = *reinterpret_cast<const u32x4 *>(qbase + (size_t)j * ld + kk * 32 + g * 8);
        qf[kk] = __builtin_bit_cast(bf16x8_t, raw);
    }
    f32x4 oacc[4];
#pragma unroll
    for (int f = 0; f < 4; f++) oacc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;
    const int qpos = L0 + j;   // position of this lane's query

    // Two register sets, ping-pong: block n + 1's K fragments and V rows are requested from HBM before block n is computed (a wave walks its
    // 2 - 4 blocks strictly in sequence; without the prefetch every block pays the full load latency).
    struct Blk { bf16x8_t kf[2][2]; };
    auto load_blk = [&](int t0, Blk &x) {
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
            int t = t0 + (j >> 2) * 8 + sb * 4 + (j & 3);
            t = t < T ? t : T - 1;
            const uint16_t *kp = t < L0 ? kc + (size_t)t * d : qbase + (size_t)(t - L0) * ld + d;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) x.kf[sb][kk] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4 *>(kp + kk * 32 + g * 8));
        }
    };
    auto proc_blk = [&](int t0, const Blk &x) {
        // ---- this block's V rows [32 keys][64 dims]: requested now, written to LDS after the softmax (their latency sits under steps A and B)
        u32x4 vreg[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int id = lane + 64 * i, kl = id >> 3, c = id & 7;
            int t = t0 + kl; t = t < T ? t : T - 1;
            const uint16_t *vp = t < L0 ? vc + (size_t)t * d + c * 8 : qbase + (size_t)(t - L0) * ld + 2 * d + c * 8;
            vreg[i] = *reinterpret_cast<const u32x4 *>(vp);
        }
        // ---- step A: scores of keys t0 + 8g + (sb*4 + r) for query j
        f32x4 sacc[2];
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
            sacc[sb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; kk++) sacc[sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.kf[sb][kk], qf[kk], sacc[sb], 0, 0, 0);
        }
        // ---- step B: online softmax for query j over its 8 keys, max shared across the 4 lane groups
        float sv[8];
        float bm = -1e30f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int t = t0 + g * 8 + e;
            const bool ok = t < T && t <= qpos && j < n_new;
            sv[e] = ok ? sacc[e >> 2][e & 3] * 0.125f : -1e30f;
            bm = fmaxf(bm, sv[e]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16));
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m, bm);
        const float alpha = __expf(m - m_new);
        float psum = 0.f;
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float p0 = sv[e] > -1e29f ? __expf(sv[e] - m_new) : 0.f;
            const float p1 = sv[e + 1] > -1e29f ? __expf(sv[e + 1] - m_new) : 0.f;
            const uint16_t h0 = f32_to_bf16_rn(p0), h1 = f32_to_bf16_rn(p1);
            psum += bf16_to_f32(h0) + bf16_to_f32(h1);   // sum what the PV product actually uses
            pk[e >> 1] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        }
        l = l * alpha + psum;
        m = m_new;
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, u32x4{pk[0], pk[1], pk[2], pk[3]});
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int id = lane + 64 * i, kl = id >> 3, c = id & 7;
            *reinterpret_cast<u32x4 *>(vlds + kl * VROW + ((c ^ chunk_swz(kl)) << 4)) = vreg[i];
        }
        // The LDS tile was written as 16-byte vectors and is read back transposed by OTHER lanes of this wave: pin the order (LDS itself is
        // in-order per wave; no s_barrier needed because the tile is private to the wave).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- step C: O^T[dim][query] = alpha * O^T + V^T . P^T ; V^T fragment read transposed from LDS
#pragma unroll
        for (int f = 0; f < 4; f++) {
            // lane (j, g) <- keys 8 g .. 8 g + 7 of dim 16 f + j: two transposed LDS reads (a 16-lane group addresses the sixteen 8-byte pieces of
            // 4 rows x 16 dims: lane -> row j >> 2, piece j & 3; the instruction hands lane j column j) instead of sixteen 2-byte gathers
            const int vr = 8 * g + (j >> 2), vp8 = j & 3;
            const unsigned char *va = vlds + vr * VROW + (((f * 2 + (vp8 >> 1)) ^ chunk_swz(vr)) << 4) + (vp8 & 1) * 8;
            typedef __bf16 bf16x4_v __attribute__((ext_vector_type(4)));
            const bf16x4_v vlo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_v *)(va));
            const bf16x4_v vhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_v *)(va + 4 * VROW));
            const bf16x8_t vf = __builtin_shufflevector(vlo, vhi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int r = 0; r < 4; r++) oacc[f][r] *= alpha;
            oacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[f], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all V^T reads done before the next block overwrites the tile
    };
    {
        Blk ba, bb;
        load_blk(0, ba);
        for (int t0 = 0; t0 < T; t0 += 64) {
            const bool second = t0 + 32 < T;
            if (second) load_blk(t0 + 32, bb);
            proc_blk(t0, ba);
            if (t0 + 64 < T) load_blk(t0 + 64, ba);
            if (second) proc_blk(t0 + 32, bb);
        }
    }
    // ---- finish: l over the 4 lane groups, write query j's 16 dims per lane (4 per fragment)
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (j < n_new) {
        const float inv = 1.f / l;
#pragma unroll
        for (int f = 0; f < 4; f++) {
            uint2 o;
            o.x = pack_bf16x2(oacc[f][0] * inv, oacc[f][1] * inv);
            o.y = pack_bf16x2(oacc[f][2] * inv, oacc[f][3] * inv);
            *reinterpret_cast<uint2 *>(out + (row0 + j) * d + (size_t)h * 64 + f * 16 + g * 4) = o;
        }
    }
}

// Ragged prefill: a chunk forward only has cnt[b] <= C valid rows per env (7 of 8 in a Wordle turn, fewer for short
// observations, 0 for finished envs).  The rows are compacted env-major so that the GEMMs see M = sum(cnt) rows: off[b] =
// first row of env b, off[B] = M (the GEMMs read it from device memory: tiles beyond it exit), row_map[r] = (env << 5) | slot.
__global__ __launch_bounds__(1024) void ragged_scan_kernel(const int32_t *__restrict__ cnt, int B, int C, int32_t *__restrict__ off,
                                                           int32_t *__restrict__ row_map) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024, lo = t * per, hi = min(lo + per, B);
    int sum = 0;
    for (int b = lo; b < hi; b++) sum += max(0, min(cnt[b], C));
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // inclusive Hillis-Steele scan of the 1024 segment sums
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;                       // exclusive prefix of this thread's segment
    for (int b = lo; b < hi; b++) {
        const int n = max(0, min(cnt[b], C));
        off[b] = run;
        for (int j = 0; j < n; j++) row_map[run + j] = (b << 5) | j;
        run += n;
    }
    if (t == 1023) off[B] = part[1023];
}

// Shared-prefix broadcast: every env of a batch starts from the same prompt prefix (the Wordle header), so its K/V rows are
// computed ONCE (a 1-env forward on a small session) and copied into all B caches; len and the last hidden state follow.
// One workgroup row per (layer, K|V, env): n_pos rows of d bf16, 16 B per lane.
__global__ __launch_bounds__(256) void kv_broadcast_kernel(const uint16_t *__restrict__ src, int src_tmax, uint16_t *__restrict__ dst,
                                                           int dst_tmax, int B, int n_pos, int d, const uint16_t *__restrict__ src_hidden,
                                                           uint16_t *__restrict__ dst_hidden, int32_t *__restrict__ dst_len) {
    const int b = blockIdx.x, lk = blockIdx.y;        // lk = layer * 2 + (0 K | 1 V)
    const uint16_t *sp = src + (size_t)lk * src_tmax * d;                       // source session has a single env
    uint16_t *dp = dst + ((size_t)lk * B + b) * dst_tmax * d;
    const int chunks = n_pos * d / 8;
    for (int i = threadIdx.x; i < chunks; i += blockDim.x)
        reinterpret_cast<u32x4 *>(dp)[i] = reinterpret_cast<const u32x4 *>(sp)[i];
    if (lk == 0) {
        if (dst_hidden) for (int i = threadIdx.x; i < d / 8; i += blockDim.x)
            reinterpret_cast<u32x4 *>(dst_hidden + (size_t)b * d)[i] = reinterpret_cast<const u32x4 *>(src_hidden)[i];
        if (threadIdx.x == 0) dst_len[b] = n_pos;
    }
}

// Prompt-prefix cache: env b starts from the prompt held by row idx[b] of another session of the same model (K/V rows, cache length and
// last hidden state) — the indexed form of kv_broadcast_kernel.  A text env whose observations form a finite set (Maze: one text per
// (goal, cell)) prefills every distinct prompt ONCE per set of weights; a turn then starts with this copy instead of a prefill.
// idx[b] < 0: the env gets an empty cache (finished episode).  One workgroup per (env, layer x K|V), 16 B per lane.
__global__ __launch_bounds__(256) void kv_gather_kernel(const uint16_t *__restrict__ src, int src_b, int src_tmax, const int32_t *__restrict__ src_len,
                                                        const int32_t *__restrict__ idx, uint16_t *__restrict__ dst, int dst_tmax, int B, int d,
                                                        const uint16_t *__restrict__ src_hidden, uint16_t *__restrict__ dst_hidden,
                                                        int32_t *__restrict__ dst_len) {
    const int b = blockIdx.x, lk = blockIdx.y;
    const int r = idx[b];
    const int n_pos = (r >= 0 && r < src_b) ? min(min(src_len[r], dst_tmax), src_tmax) : 0;
    if (n_pos > 0) {
        const uint16_t *sp = src + ((size_t)lk * src_b + r) * src_tmax * d;
        uint16_t *dp = dst + ((size_t)lk * B + b) * dst_tmax * d;
        const int chunks = n_pos * d / 8;
        for (int i = threadIdx.x; i < chunks; i += blockDim.x)
            reinterpret_cast<u32x4 *>(dp)[i] = reinterpret_cast<const u32x4 *>(sp)[i];
    }
    if (lk == 0) {
        if (dst_hidden && n_pos > 0) for (int i = threadIdx.x; i < d / 8; i += blockDim.x)
            reinterpret_cast<u32x4 *>(dst_hidden + (size_t)b * d)[i] = reinterpret_cast<const u32x4 *>(src_hidden + (size_t)r * d)[i];
        if (threadIdx.x == 0) dst_len[b] = n_pos;
    }
}

// Indexed prompt prefix (lmrl_gpt2_kv_attach): nothing is copied but the last hidden state — env b's cache length becomes the prompt length
// of prefix row idx[b], and the decode attention reads positions below it from that row (attention_decode_kernel<.., true>).
// Blocks [0, B): per-env bookkeeping; block B: the launch order — envs grouped by prefix row (counting sort over the rows; the order inside a
// group does not matter: it only decides which waves run side by side).  idx < 0 / out of range: empty cache, sorted last.
constexpr int kAttachMaxRows = 12288;        // prefix rows the LDS histogram holds (48 KB); more rows: identity order
__global__ __launch_bounds__(1024) void kv_attach_kernel(int src_b, const int32_t *__restrict__ src_len, const int32_t *__restrict__ idx, int B, int d,
                                                         int dst_tmax, const uint16_t *__restrict__ src_hidden, uint16_t *__restrict__ dst_hidden,
                                                         int32_t *__restrict__ dst_len, int32_t *__restrict__ pfx_n, int32_t *__restrict__ order) {
    const int b = blockIdx.x;
    if (b < B) {
        const int r = idx[b];
        const int n = (r >= 0 && r < src_b) ? min(src_len[r], dst_tmax) : 0;
        if (dst_hidden && n > 0) for (int i = threadIdx.x; i < d / 8; i += blockDim.x)
            reinterpret_cast<u32x4 *>(dst_hidden + (size_t)b * d)[i] = reinterpret_cast<const u32x4 *>(src_hidden + (size_t)r * d)[i];
        if (threadIdx.x == 0) { dst_len[b] = n; pfx_n[b] = n; }
        return;
    }
    if (!order) return;
    if (src_b > kAttachMaxRows) {
        for (int i = threadIdx.x; i < B; i += blockDim.x) order[i] = i;
        return;
    }
    __shared__ int32_t hist[kAttachMaxRows + 1];
    __shared__ int32_t part[1024];
    const int nb = src_b + 1;                                // bucket src_b: envs without a prefix
    for (int i = threadIdx.x; i < nb; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const int r = idx[i];
        atomicAdd(&hist[(r >= 0 && r < src_b) ? r : src_b], 1);
    }
    __syncthreads();
    // exclusive scan of the histogram: each thread owns a contiguous stretch of buckets
    const int per = (nb + blockDim.x - 1) / blockDim.x, lo = min((int)threadIdx.x * per, nb), hi = min(lo + per, nb);
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < (int)blockDim.x; i++) { const int v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { const int v = hist[i]; hist[i] = run; run += v; }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const int r = idx[i];
        order[atomicAdd(&hist[(r >= 0 && r < src_b) ? r : src_b], 1)] = i;
    }
}

// profiling only (one launch per forward): algorithmic HBM bytes of the attention launches of this forward =
// per (env, head, layer): K and V rows of every attended position (2 x 128 B) + the chunk's q rows and output rows.
__global__ void attn_bytes_kernel(const int32_t *cnt, const int32_t *len, int B, int C, int heads_x_layers,
                                  unsigned long long *counter, int n_shared) {
    unsigned long long s = 0;
    bool first = true;      // a shared prefix is read from ONE env's rows: its bytes count once per launch, not once per env
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int n = min(cnt[b], C);
        if (n > 0) {
            const int shared = (n_shared > 0 && !(first && threadIdx.x == 0)) ? min(n_shared, len[b]) : 0;
            s += (unsigned long long)(len[b] - shared + n) * 256ull + (unsigned long long)n * 256ull;
            first = false;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(counter, (red[0] + red[1] + red[2] + red[3]) * (unsigned long long)heads_x_layers);
}

// c[m][n] (=|+=) sum_z ws[z][m][n] for n < n_store, z in order (deterministic); ws rows have pitch ldw
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int S, long zstride, int ldw, float *c, int ldc, int M,
                                                            int n_store, int accumulate, const float *__restrict__ bias = nullptr) {
    const int n4 = (n_store + 3) / 4;
    const long total = (long)M * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
        const float *p = ws + (long)m * ldw + n;
        f32x4 a = *reinterpret_cast<const f32x4 *>(p);
        for (int z = 1; z < S; z++) a += *reinterpret_cast<const f32x4 *>(p + z * zstride);
        if (bias) a += *reinterpret_cast<const f32x4 *>(bias + n);          // (bias rows are padded to the GEMM's N, a multiple of 128)
        float *o = c + (long)m * ldc + n;
        if (n + 4 <= n_store && (ldc & 3) == 0) {
            if (accumulate) a += *reinterpret_cast<const f32x4 *>(o);
            *reinterpret_cast<f32x4 *>(o) = a;
        } else {
            for (int k = 0; k < 4 && n + k < n_store; k++) o[k] = accumulate ? o[k] + a[k] : a[k];
        }
    }
}

// Split-K plan for C[m][n] = A[m][k] . W[n][k]^T with few output tiles and a long K: S copies of the tile grid; 0 = not worth splitting.
// kind 0 (the dW products, m, n <= a few thousand, k = B*T): 128 x 128 tiles, ~2 workgroups per CU busy.
// kind 1 (the vocabulary-wide heads' dX, m = rows, n = 768 .. 1536, k = V): 256 x 192 tiles — 128 .. 256 of them — times S = 2 .. so that every
//         CU holds one (unsplit: 96 tiles of 256 x 256 on 256 CUs, 848 us; profiles/r03_train_gemm_tiles.txt).
static int splitk_plan(int m, int n, int k, int *kchunk, int *kind) {
    *kind = 0;
    if (k < 4096 || k % 64 != 0) return 0;
    const int ksteps = k / 64;
    int S = 0;
    const long tiles = (long)((m + 127) / 128) * (n / 128);
    if (n % 128 == 0 && tiles <= 200) {
        S = (int)(512 / tiles);
    } else if (m >= 2048 && n % 192 == 0 && k >= 8192) {
        const long t192 = (long)((m + 255) / 256) * (n / 192);
        if (t192 > 128) return 0;
        *kind = 1;
        S = (int)(256 / t192);
    } else {
        return 0;
    }
    S = std::min(S, ksteps / 16);                 // >= 16 K-steps per copy: the ring prologue / epilogue stay a small part of a copy's life
    if (S < 2) return 0;
    const int per = (ksteps + S - 1) / S;
    *kchunk = per * 64;
    return (ksteps + per - 1) / per;
}

int g_gemm_variant = 0;   // tools/ only (tools/bench_gemm.py tile-configuration sweeps): never touched by the product path
constexpr int kRaggedAutoMinSlots = 2048;   // default policy: forwards with >= this many slots run on the compacted (sum of cnt) rows

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Gpt2Ws {
    float *x; uint16_t *h, *qkv, *att, *ff; int32_t *rows_idx; float2 *stats; int32_t *off, *row_map;
    static int nslots(const lmrl_gpt2_config &c) { const int nq = ln_fusion_nq(c.d_model); return nq ? 8 * nq : 8; }   // padded slot pitch
    static size_t bytes(const lmrl_gpt2_config &c, size_t M, size_t B) {
        return align256(M * c.d_model * 4) + align256(M * c.d_model * 2) + align256(M * 3 * c.d_model * 2) +
               align256(M * c.d_model * 2) + align256(M * c.d_ff * 2) + align256(B * 4) + align256(M * nslots(c) * 8) + align256((B + 1) * 4) + align256(M * 4);
    }
    void carve(void *ws, const lmrl_gpt2_config &c, size_t M, size_t B) {   // same order as bytes()
        char *p = (char *)ws;
        x = (float *)p; p += align256(M * c.d_model * 4);
        h = (uint16_t *)p; p += align256(M * c.d_model * 2);
        qkv = (uint16_t *)p; p += align256(M * 3 * c.d_model * 2);
        att = (uint16_t *)p; p += align256(M * c.d_model * 2);
        ff = (uint16_t *)p; p += align256(M * c.d_ff * 2);
        rows_idx = (int32_t *)p; p += align256(B * 4);
        stats = (float2 *)p; p += align256(M * nslots(c) * 8);
        off = (int32_t *)p; p += align256((B + 1) * 4);
        row_map = (int32_t *)p;
    }
};

}  // namespace lmrl

using namespace lmrl;

struct lmrl_gpt2 : public Gpt2Model {};

extern "C" {

lmrl_gpt2 *lmrl_gpt2_create(const lmrl_gpt2_config *cfg, const void *wte, const void *wpe, const float *lnf_g,
                            const float *lnf_b, const void *const *layer_ptrs) {
    if (!cfg || !wte || !wpe || !lnf_g || !lnf_b || !layer_ptrs) { set_error("lmrl_gpt2_create: null pointer"); return nullptr; }
    if (cfg->d_model != cfg->n_head * 64) { set_error("lmrl_gpt2_create: head dim must be 64"); return nullptr; }
    if (cfg->d_model % 128 || cfg->d_ff % 128 || cfg->vocab_padded % 128 || cfg->d_model > 64 * 4 * 8) {
        set_error("lmrl_gpt2_create: d_model, d_ff, vocab_padded must be multiples of 128 (d_model <= 2048)");
        return nullptr;
    }
    lmrl_gpt2 *m = new lmrl_gpt2();
    m->cfg = *cfg;
    m->wte = (const uint16_t *)wte; m->wpe = (const uint16_t *)wpe; m->lnf_g = lnf_g; m->lnf_b = lnf_b;
    m->layers = new Gpt2Layer[cfg->n_layer];
    for (int l = 0; l < cfg->n_layer; l++) {
        const void *const *p = layer_ptrs + (size_t)l * 12;
        for (int k = 0; k < 12; k++)
            if (!p[k]) { set_error("lmrl_gpt2_create: layer %d pointer %d is null", l, k); delete[] m->layers; delete m; return nullptr; }
        Gpt2Layer &L = m->layers[l];
        L.ln1_g = (const float *)p[0]; L.ln1_b = (const float *)p[1];
        L.w_qkv = (const uint16_t *)p[2]; L.b_qkv = (const float *)p[3];
        L.w_proj = (const uint16_t *)p[4]; L.b_proj = (const float *)p[5];
        L.ln2_g = (const float *)p[6]; L.ln2_b = (const float *)p[7];
        L.w_fc = (const uint16_t *)p[8]; L.b_fc = (const float *)p[9];
        L.w_fc2 = (const uint16_t *)p[10]; L.b_fc2 = (const float *)p[11];
        L.wf_qkv = L.wf_fc = nullptr; L.cs_qkv = L.bf_qkv = L.cs_fc = L.bf_fc = nullptr;
    }
    // LayerNorm folding (ln_1 -> c_attn, ln_2 -> c_fc): model-owned copies, built once
    const int d = cfg->d_model, dff = cfg->d_ff;
    bool ok = hipDeviceSynchronize() == hipSuccess;
    for (int l = 0; l < cfg->n_layer && ok; l++) {
        Gpt2Layer &L = m->layers[l];
        ok = ok && hipMalloc(&L.wf_qkv, (size_t)3 * d * d * 2) == hipSuccess && hipMalloc(&L.cs_qkv, (size_t)3 * d * 4) == hipSuccess &&
             hipMalloc(&L.bf_qkv, (size_t)3 * d * 4) == hipSuccess && hipMalloc(&L.wf_fc, (size_t)dff * d * 2) == hipSuccess &&
             hipMalloc(&L.cs_fc, (size_t)dff * 4) == hipSuccess && hipMalloc(&L.bf_fc, (size_t)dff * 4) == hipSuccess;
        if (!ok) break;
        hipLaunchKernelGGL(fold_ln_kernel, dim3(ceil_div(3 * d, 4)), dim3(256), 0, 0, L.w_qkv, L.ln1_g, L.ln1_b, L.b_qkv, L.wf_qkv, L.cs_qkv,
                           L.bf_qkv, 3 * d, d);
        hipLaunchKernelGGL(fold_ln_kernel, dim3(ceil_div(dff, 4)), dim3(256), 0, 0, L.w_fc, L.ln2_g, L.ln2_b, L.b_fc, L.wf_fc, L.cs_fc,
                           L.bf_fc, dff, d);
    }
    ok = ok && hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        set_error("lmrl_gpt2_create: device allocation / LayerNorm folding failed (is a GPU visible?)");
        lmrl_gpt2_destroy(m);
        return nullptr;
    }
    return m;
}

int lmrl_gpt2_refresh(lmrl_gpt2 *m, void *stream) {
    LMRL_REQUIRE(m && m->layers, "lmrl_gpt2_refresh: bad argument");
    const int d = m->cfg.d_model, dff = m->cfg.d_ff;
    for (int l = 0; l < m->cfg.n_layer; l++) {
        Gpt2Layer &L = m->layers[l];
        hipLaunchKernelGGL(fold_ln_kernel, dim3(ceil_div(3 * d, 4)), dim3(256), 0, as_stream(stream), L.w_qkv, L.ln1_g, L.ln1_b, L.b_qkv, L.wf_qkv, L.cs_qkv,
                           L.bf_qkv, 3 * d, d);
        hipLaunchKernelGGL(fold_ln_kernel, dim3(ceil_div(dff, 4)), dim3(256), 0, as_stream(stream), L.w_fc, L.ln2_g, L.ln2_b, L.b_fc, L.wf_fc, L.cs_fc,
                           L.bf_fc, dff, d);
    }
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

void lmrl_gpt2_destroy(lmrl_gpt2 *m) {
    if (!m) return;
    for (int l = 0; m->layers && l < m->cfg.n_layer; l++) {
        Gpt2Layer &L = m->layers[l];
        (void)hipFree(L.wf_qkv); (void)hipFree(L.cs_qkv); (void)hipFree(L.bf_qkv);
        (void)hipFree(L.wf_fc); (void)hipFree(L.cs_fc); (void)hipFree(L.bf_fc);
    }
#ifdef LMRL_TOOLS
    if (m->ablate_aux.stream) { (void)hipStreamDestroy(m->ablate_aux.stream); (void)hipEventDestroy(m->ablate_aux.fork); (void)hipEventDestroy(m->ablate_aux.join); }
#endif
    delete[] m->layers;
    delete m;
}

size_t lmrl_gpt2_kv_bytes(const lmrl_gpt2 *m, int b, int tmax) {
    return (size_t)m->cfg.n_layer * 2 * b * m->cfg.n_head * tmax * 64 * sizeof(uint16_t);
}

size_t lmrl_gpt2_ws_bytes(const lmrl_gpt2 *m, int b, int c) { return Gpt2Ws::bytes(m->cfg, (size_t)b * c, b); }

static int gpt2_forward_impl(lmrl_gpt2 *m, void *kv_d, int tmax, void *ws_d, const int32_t *tokens_d, const int32_t *cnt_d,
                             int32_t *len_d, int b, int c, void *last_hidden_d, void *all_hidden_d, const lmrl_kv_prefix *pfx, unsigned flags,
                             void *stream) {
    LMRL_REQUIRE(m && kv_d && ws_d && tokens_d && cnt_d && len_d && b > 0, "lmrl_gpt2_forward: bad argument");
    LMRL_REQUIRE(c == 1 || c == 8 || c == 16, "lmrl_gpt2_forward: chunk width must be 1, 8 or 16");
    const lmrl_gpt2_config &cf = m->cfg;
    hipStream_t s = as_stream(stream);
    const int M = b * c, d = cf.d_model;
    Gpt2Ws w; w.carve(ws_d, cf, M, b);
    const size_t kv_layer = (size_t)b * cf.n_head * tmax * 64;

    // positions [0, n_shared) of every env's cache equal env 0's (lmrl_gpt2_kv_broadcast): the decode attention reads them from env 0
    const int n_shared = (int)((flags >> 8) & 0xffu);
    LMRL_REQUIRE(n_shared <= tmax, "lmrl_gpt2_forward: shared prefix longer than the cache");
    if (pfx) {
        LMRL_REQUIRE(c == 1 && !(flags & LMRL_FWD_ATTN_VALU), "lmrl_gpt2_forward_prefixed: single-token decode on the default attention kernel only");
        LMRL_REQUIRE(pfx->kv_d && pfx->row_d && pfx->n_d && pfx->n_rows > 0 && pfx->tmax > 0 && n_shared == 0, "lmrl_gpt2_forward_prefixed: bad prefix");
        LMRL_REQUIRE((size_t)pfx->n_rows * pfx->tmax * d * 2 < ((size_t)1 << 32) && (size_t)b * tmax * d * 2 < ((size_t)1 << 32),
                     "lmrl_gpt2_forward_prefixed: a layer's K block must stay below 4 GiB (32-bit row offsets)");
    }
    const size_t pfx_layer = pfx ? (size_t)pfx->n_rows * cf.n_head * pfx->tmax * 64 : 0;
    if (unsigned long long *ctr = prof_byte_counter(c == 1 ? PROF_ATTN_DECODE : PROF_ATTN_CHUNK))
        hipLaunchKernelGGL(attn_bytes_kernel, dim3(1), dim3(256), 0, s, cnt_d, len_d, b, c, cf.n_head * cf.n_layer, ctr, c == 1 ? n_shared : 0);
    LMRL_REQUIRE(!((flags & LMRL_FWD_RAGGED_ALWAYS) && (flags & LMRL_FWD_RAGGED_NEVER)), "lmrl_gpt2_forward: contradictory ragged flags");
    const bool fused = !(flags & LMRL_FWD_LN_STANDALONE) && g_gemm_variant != 1 && ln_fusion_nq(d) != 0;
    // Bits >= 16 of `flags` are reserved.  Only the -DLMRL_TOOLS build (`python lmrl-gym_amd/build.py --tools` -> liblmrl_amd_tools.so, loaded by
    // tools/bench_ablate_decode.py alone) gives them a meaning: timing-only launch ablations of the single-token decode layers whose RESULTS ARE
    // GARBAGE (csrc/ablate_tools.h).  The product library refuses them.
#ifdef LMRL_TOOLS
    const unsigned ablate = (c == 1) ? ((flags >> LMRL_FWD_ABLATE_SHIFT) & 0x1fffu) : 0u;
#define LMRL_ABL(bit) ((ablate & (bit)) != 0u)
    AblateAux &aux = m->ablate_aux;     // per model (= per device), created on first use, destroyed with the model
    if ((ablate & (LMRL_ABLATE_PROJ_CONCURRENT | LMRL_ABLATE_FC2_SPLITK2 | LMRL_ABLATE_FC2_HALFK | LMRL_ABLATE_PROJ_AUX_SERIAL)) && !aux.stream) {
        LMRL_CHECK_HIP(hipStreamCreateWithFlags(&aux.stream, hipStreamNonBlocking));
        LMRL_CHECK_HIP(hipEventCreateWithFlags(&aux.fork, hipEventDisableTiming));
        LMRL_CHECK_HIP(hipEventCreateWithFlags(&aux.join, hipEventDisableTiming));
    }
    hipStream_t aux_stream = aux.stream; hipEvent_t aux_fork = aux.fork, aux_join = aux.join;
#else
    LMRL_REQUIRE(((flags & ~LMRL_FWD_SKINNY) >> 16) == 0u, "lmrl_gpt2_forward: flag bits 16-29 are reserved (launch ablations exist only in the LMRL_TOOLS build)");
#define LMRL_ABL(bit) false
    constexpr hipStream_t aux_stream = nullptr; constexpr hipEvent_t aux_fork = nullptr, aux_join = nullptr;   // dead branches below still parse
    (void)aux_stream; (void)aux_fork; (void)aux_join;
#endif
    const int nsl = Gpt2Ws::nslots(cf);
    // ragged batches (LN-folded path, unless the caller wants every row's hidden state back): by default only the forwards of
    // large batches qualify (b*c >= 2048); LMRL_FWD_RAGGED_ALWAYS compacts every forward (generation loops whose rows finish at
    // different steps), LMRL_FWD_RAGGED_NEVER none.  A per-call property: nothing here is process state.
    const bool ragged = fused && !all_hidden_d && !(flags & LMRL_FWD_RAGGED_NEVER) &&
                        ((flags & LMRL_FWD_RAGGED_ALWAYS) || M >= kRaggedAutoMinSlots);
    const int32_t *off = ragged ? w.off : nullptr, *row_map = ragged ? w.row_map : nullptr, *m_dev = ragged ? w.off + b : nullptr;
    if (ragged) hipLaunchKernelGGL(ragged_scan_kernel, dim3(1), dim3(1024), 0, s, cnt_d, b, c, w.off, w.row_map);
    auto ln = [&](const float *g, const float *be, uint16_t *y, const int32_t *idx, int rows) {
        if (d <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w.x, g, be, y, idx, rows, d, cf.ln_eps);
        else hipLaunchKernelGGL(layernorm_kernel<8>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w.x, g, be, y, idx, rows, d, cf.ln_eps);
    };
    if (fused) {
        // LN-folded path: w.h holds the bf16 copy of the residual stream, w.stats its per-row (sum, sum^2) slots
        if (d <= 1024) hipLaunchKernelGGL(embed_stats_kernel<4>, dim3(ceil_div(M, 4)), dim3(256), 0, s, m->wte, m->wpe, tokens_d, cnt_d, len_d,
                                          w.x, w.h, w.stats, nsl, b, c, d, cf.vocab, cf.n_pos, row_map, m_dev);
        else hipLaunchKernelGGL(embed_stats_kernel<8>, dim3(ceil_div(M, 4)), dim3(256), 0, s, m->wte, m->wpe, tokens_d, cnt_d, len_d,
                                w.x, w.h, w.stats, nsl, b, c, d, cf.vocab, cf.n_pos, row_map, m_dev);
    } else {
        // embeddings + LN1 of layer 0 in one launch
        if (d <= 1024) hipLaunchKernelGGL(embed_ln_kernel<4>, dim3(ceil_div(M, 4)), dim3(256), 0, s, m->wte, m->wpe, tokens_d, cnt_d, len_d,
                                          m->layers[0].ln1_g, m->layers[0].ln1_b, w.x, w.h, b, c, d, cf.vocab, cf.n_pos, cf.ln_eps);
        else hipLaunchKernelGGL(embed_ln_kernel<8>, dim3(ceil_div(M, 4)), dim3(256), 0, s, m->wte, m->wpe, tokens_d, cnt_d, len_d,
                                m->layers[0].ln1_g, m->layers[0].ln1_b, w.x, w.h, b, c, d, cf.vocab, cf.n_pos, cf.ln_eps);
    }
    LMRL_CHECK_LAUNCH();
    // LMRL_FWD_KV_FROM_GEMM (single-token decode, LN-folded path): the qkv GEMM's epilogue appends the new K / V rows and the attention kernel
    // only reads.  Measured: attention 25.3 -> 23.8 us (0.53 of the HBM roofline) but the qkv GEMM 12.7 -> 15.3 us (its tile scatters 16-byte
    // stores over 128 envs' cache pages), a net loss of 1.2 us per layer — so it is off by default and kept as a per-call variant.
    const bool kv_from_gemm = fused && c == 1 && (flags & LMRL_FWD_KV_FROM_GEMM) && !(flags & LMRL_FWD_ATTN_VALU);
    // LMRL_FWD_SKINNY (single-token decode of <= 16 sequences, LN-folded path): the four Dense products of a layer on skinny_gemm.h (one MFMA row block,
    // K split over the 16 waves of a workgroup, every operand load issued up front) instead of the 64 x 64-tile latency chains
    // (the skinny residual producers write no LayerNorm slots, the skinny consumers read the fp32 rows instead: all of a layer's products switch together;
    //  GPT-2-small / medium / large all qualify: K = d_model / d_ff of all three; other widths keep the whole session on the tile kernels)
    bool skinny = fused && c == 1 && (flags & LMRL_FWD_SKINNY) && M <= 16 && !kv_from_gemm;
    if (skinny) {
        GemmArgs t1{}; t1.M = M; t1.N = 3 * d; t1.K = d; t1.lda = d;
        GemmArgs t2 = t1; t2.N = d; t2.K = cf.d_ff; t2.lda = cf.d_ff;
        GemmArgs t3 = t1; t3.N = cf.d_ff;
        skinny = skinny_ok(t1) && skinny_ok(t2) && skinny_ok(t3);
    }
    const int append_in_attn = kv_from_gemm ? 0 : 1;
    // chunk forwards that return at most the last token's hidden state: the last layer's projection + MLP run on ONE row per env (the same
    // per-row arithmetic as on the full chunk: bit-identical results), or not at all when no hidden state is asked for (a prompt's
    // non-final chunks).  The compact operands live in the qkv scratch, which is dead after the attention launch.
    const bool last_only = fused && c > 1 && !all_hidden_d && !(flags & LMRL_FWD_FULL_LAST_LAYER);
    uint16_t *lc_att = nullptr, *lc_h = nullptr, *lc_ff = nullptr; float *lc_x = nullptr; float2 *lc_stats = nullptr;
    if (last_only) {
        char *p = (char *)w.qkv;
        lc_x = (float *)p; p += align256((size_t)b * d * 4);
        lc_att = (uint16_t *)p; p += align256((size_t)b * d * 2);
        lc_h = (uint16_t *)p; p += align256((size_t)b * d * 2);
        lc_ff = (uint16_t *)p; p += align256((size_t)b * cf.d_ff * 2);
        lc_stats = (float2 *)p; p += align256((size_t)b * nsl * 8);
        LMRL_REQUIRE((size_t)(p - (char *)w.qkv) <= (size_t)M * 3 * d * 2, "lmrl_gpt2_forward: qkv scratch too small for the compact last-layer rows");
    }
    for (int l = 0; l < cf.n_layer; l++) {
        const Gpt2Layer &L = m->layers[l];
        uint16_t *kc = (uint16_t *)kv_d + (size_t)(2 * l) * kv_layer, *vc = kc + kv_layer;
        if (fused && LMRL_ABL(LMRL_ABLATE_QKV)) {
        } else if (fused) {
            GemmArgs g{w.h, L.wf_qkv, L.bf_qkv, w.qkv, M, 3 * d, d, d, 3 * d, 3 * d, w.stats, nullptr, L.cs_qkv, nsl, 1.f / (float)d, cf.ln_eps, m_dev};
            if (skinny) {
                g.resid = w.x; g.ldr = d;                  // the consumer derives (mu, rstd) from the fp32 rows
                LMRL_CHECK_HIP(skinny_launch<EPI_BF16_LN>(g, s));
            } else if (kv_from_gemm) {   // decode: the new K / V rows go to the cache from this GEMM's epilogue, the attention kernel only reads
                g.kv_k = kc; g.kv_v = vc; g.kv_len = len_d; g.kv_cnt = cnt_d; g.kv_rowmap = row_map; g.kv_tmax = tmax; g.kv_d = d;
                LMRL_CHECK_HIP(gemm_launch_ln<EPI_BF16_LN_KV>(g, s));
            } else {
                LMRL_CHECK_HIP(gemm_launch_ln<EPI_BF16_LN>(g, s));
            }
        } else {
            if (l > 0) {
                ln(L.ln1_g, L.ln1_b, w.h, nullptr, M);
                LMRL_CHECK_LAUNCH();
            }
            GemmArgs g{w.h, L.w_qkv, L.b_qkv, w.qkv, M, 3 * d, d, d, 3 * d, 3 * d};
            LMRL_CHECK_HIP(gemm_launch<EPI_BF16>(g, s));
        }
        // algorithmic bytes: K+V rows read once (2 * 128 B per cached position per head) + q/k/v/out rows of the chunk
        hipEvent_t ev_a, ev_b;
        // decode: the single-shot kernel; LMRL_FWD_ATTN_VALU keeps the multi-round-trip per-head kernel as the cross-check
        const bool shot = c == 1 && !(flags & LMRL_FWD_ATTN_VALU);
        if (fused && LMRL_ABL(LMRL_ABLATE_PROJ_CONCURRENT)) {     // fork: proj on the aux stream, behind the qkv GEMM only
            LMRL_CHECK_HIP(hipEventRecord(aux_fork, s));
            LMRL_CHECK_HIP(hipStreamWaitEvent(aux_stream, aux_fork, 0));
            GemmArgs gp{w.att, L.w_proj, L.b_proj, w.x, M, d, d, d, d, d, w.stats, w.h, nullptr, nsl, 0.f, 0.f, m_dev};
            LMRL_CHECK_HIP(gemm_launch_ln<EPI_RESID_F32_STATS>(gp, aux_stream));
            LMRL_CHECK_HIP(hipEventRecord(aux_join, aux_stream));
        }
        if (LMRL_ABL(LMRL_ABLATE_ATTN)) {
        } else if (shot) {
            const bool ev = prof_kernel_events(PROF_ATTN_DECODE, -1.0, &ev_a, &ev_b);   // start/stop events attached to the dispatch itself
            DecodePrefix dp{};
            if (pfx) {
                dp.k = (const uint16_t *)pfx->kv_d + (size_t)(2 * l) * pfx_layer; dp.v = dp.k + pfx_layer;
                dp.row = pfx->row_d; dp.n = pfx->n_d; dp.order = pfx->order_d; dp.tmax = pfx->tmax;
            }
#define LMRL_DEC_LAUNCH(U_, PFX_, NI_)                                                                                                               \
            do {                                                                                                                                     \
                const dim3 grid_(ceil_div(ceil_div(b * cf.n_head, NI_), 4));                                                                         \
                if (ev) hipExtLaunchKernelGGL((attention_decode_kernel<U_, PFX_, false, NI_>), grid_, dim3(256), 0, s, ev_a, ev_b, 0,                 \
                                              (const uint16_t *)w.qkv, kc, vc, cnt_d, (const int32_t *)len_d, w.att, b, cf.n_head, tmax, d, off,     \
                                              n_shared, append_in_attn, dp);                                                                         \
                else hipLaunchKernelGGL((attention_decode_kernel<U_, PFX_, false, NI_>), grid_, dim3(256), 0, s, (const uint16_t *)w.qkv,            \
                                        kc, vc, cnt_d, (const int32_t *)len_d, w.att, b, cf.n_head, tmax, d, off, n_shared, append_in_attn, dp);      \
            } while (0)
            // 32 cached positions per batch of loads, 72 VGPRs -> 7 waves per SIMD (measured best of U = 4 / 6 / 8 / 10)
            if (skinny && !ev) {        // a handful of envs: four waves per (env, head), every position batch of an item in flight at once
                if (pfx) hipLaunchKernelGGL((attention_decode_split_kernel<4, true>), dim3(b * cf.n_head), dim3(256), 0, s, (const uint16_t *)w.qkv, kc, vc, cnt_d,
                                            (const int32_t *)len_d, w.att, b, cf.n_head, tmax, d, off, n_shared, append_in_attn, dp);
                else hipLaunchKernelGGL((attention_decode_split_kernel<4, false>), dim3(b * cf.n_head), dim3(256), 0, s, (const uint16_t *)w.qkv, kc, vc, cnt_d,
                                        (const int32_t *)len_d, w.att, b, cf.n_head, tmax, d, off, n_shared, append_in_attn, dp);
            }
            else if (pfx) LMRL_DEC_LAUNCH(4, true, 1);
#ifdef LMRL_TOOLS
            else if (LMRL_ABL(LMRL_ABLATE_ATTN_HALF_BYTES))
                hipLaunchKernelGGL((attention_decode_kernel<4, false, true>), dim3(ceil_div(b * cf.n_head, 4)), dim3(256), 0, s, (const uint16_t *)w.qkv, kc, vc,
                                   cnt_d, (const int32_t *)len_d, w.att, b, cf.n_head, tmax, d, off, n_shared, append_in_attn, dp);
#endif
            else if (flags & LMRL_FWD_ATTN_ITEMS3) LMRL_DEC_LAUNCH(4, false, 3);
            else if (flags & LMRL_FWD_ATTN_ITEMS2) LMRL_DEC_LAUNCH(4, false, 2);
            else LMRL_DEC_LAUNCH(4, false, 1);
#undef LMRL_DEC_LAUNCH
        } else if (c == 1 && prof_kernel_events(PROF_ATTN_DECODE, -1.0, &ev_a, &ev_b)) {
            // the roofline kernel: start/stop events attached to the dispatch itself (kernel begin -> end, as rocprofv3 reports it)
            hipExtLaunchKernelGGL(attention_kernel<1>, dim3(ceil_div(b * cf.n_head, 4)), dim3(256), 0, s, ev_a, ev_b, 0, (const uint16_t *)w.qkv, kc, vc,
                                  cnt_d, (const int32_t *)len_d, w.att, b, cf.n_head, tmax, d, off);
        } else {
        ProfScope ps(c == 1 ? PROF_ATTN_DECODE : PROF_ATTN_CHUNK, s, -1.0);
        if (c == 1) hipLaunchKernelGGL(attention_kernel<1>, dim3(ceil_div(b * cf.n_head, 4)), dim3(256), 0, s, w.qkv, kc, vc, cnt_d, len_d, w.att, b, cf.n_head, tmax, d, off);
        else if ((flags & LMRL_FWD_ATTN_VALU) && c == 8 && !ragged) hipLaunchKernelGGL(attention_kernel<8>, dim3(ceil_div(b * cf.n_head, 4)), dim3(256), 0, s, w.qkv, kc, vc, cnt_d, len_d, w.att, b, cf.n_head, tmax, d);
        else if (c == 16) hipLaunchKernelGGL(attention_chunk_mfma_kernel<16>, dim3(ceil_div(b * cf.n_head, 4)), dim3(256), 0, s, w.qkv, kc, vc, cnt_d, len_d, w.att, b, cf.n_head, tmax, d, off);
        else hipLaunchKernelGGL(attention_chunk_mfma_kernel<8>, dim3(ceil_div(b * cf.n_head, 4)), dim3(256), 0, s, w.qkv, kc, vc, cnt_d, len_d, w.att, b, cf.n_head, tmax, d, off);
        }
        LMRL_CHECK_LAUNCH();
        if (fused && l + 1 == cf.n_layer && last_only) {
            // last layer of a chunk forward: nothing after the attention for rows whose hidden state nobody reads
            if (last_hidden_d) {
                hipLaunchKernelGGL(gather_last_rows_kernel, dim3(ceil_div(b, 4)), dim3(256), 0, s, (const uint16_t *)w.att, (const float *)w.x, cnt_d, off,
                                   b, c, d, lc_att, lc_x, lc_stats, nsl);
                LMRL_CHECK_LAUNCH();
                GemmArgs gp{lc_att, L.w_proj, L.b_proj, lc_x, b, d, d, d, d, d, lc_stats, lc_h, nullptr, nsl, 0.f, 0.f, nullptr};
                LMRL_CHECK_HIP(gemm_launch_ln<EPI_RESID_F32_STATS>(gp, s));
                GemmArgs gf{lc_h, L.wf_fc, L.bf_fc, lc_ff, b, cf.d_ff, d, d, cf.d_ff, cf.d_ff, lc_stats, nullptr, L.cs_fc, nsl, 1.f / (float)d, cf.ln_eps, nullptr};
                LMRL_CHECK_HIP(gemm_launch_ln<EPI_GELU_BF16_LN>(gf, s));
                GemmArgs g2{lc_ff, L.w_fc2, L.b_fc2, lc_x, b, d, cf.d_ff, cf.d_ff, d, d, lc_stats, lc_h, nullptr, nsl, 0.f, 0.f, nullptr};
                LMRL_CHECK_HIP(gemm_launch<EPI_RESID_F32>(g2, s));
            }
        } else if (fused) {
            GemmArgs gp{w.att, L.w_proj, L.b_proj, w.x, M, d, d, d, d, d, w.stats, w.h, nullptr, nsl, 0.f, 0.f, m_dev};
            if (LMRL_ABL(LMRL_ABLATE_PROJ_AUX_SERIAL)) {        // calibration: the SAME dependency chain routed through the aux stream (fork after the attention)
                LMRL_CHECK_HIP(hipEventRecord(aux_fork, s));
                LMRL_CHECK_HIP(hipStreamWaitEvent(aux_stream, aux_fork, 0));
                LMRL_CHECK_HIP(gemm_launch_ln<EPI_RESID_F32_STATS>(gp, aux_stream));
                LMRL_CHECK_HIP(hipEventRecord(aux_join, aux_stream));
                LMRL_CHECK_HIP(hipStreamWaitEvent(s, aux_join, 0));
            }
            else if (LMRL_ABL(LMRL_ABLATE_PROJ_CONCURRENT)) LMRL_CHECK_HIP(hipStreamWaitEvent(s, aux_join, 0));      // join
            else if (skinny) LMRL_CHECK_HIP(skinny_launch<EPI_RESID_F32_STATS>(gp, s));
            else if (!LMRL_ABL(LMRL_ABLATE_PROJ)) LMRL_CHECK_HIP(gemm_launch_ln<EPI_RESID_F32_STATS>(gp, s));
            GemmArgs gf{w.h, L.wf_fc, L.bf_fc, w.ff, M, cf.d_ff, d, d, cf.d_ff, cf.d_ff, w.stats, nullptr, L.cs_fc, nsl, 1.f / (float)d, cf.ln_eps, m_dev};
            if (skinny) { gf.resid = w.x; gf.ldr = d; LMRL_CHECK_HIP(skinny_launch<EPI_GELU_BF16_LN>(gf, s)); }
            else if (!LMRL_ABL(LMRL_ABLATE_FC)) LMRL_CHECK_HIP(gemm_launch_ln<EPI_GELU_BF16_LN>(gf, s));
            GemmArgs g2{w.ff, L.w_fc2, L.b_fc2, w.x, M, d, cf.d_ff, cf.d_ff, d, d, w.stats, w.h, nullptr, nsl, 0.f, 0.f, m_dev};
            if (LMRL_ABL(LMRL_ABLATE_FC2)) {}
            else if (skinny) {
                if (l + 1 < cf.n_layer) LMRL_CHECK_HIP(skinny_launch<EPI_RESID_F32_STATS>(g2, s));
                else LMRL_CHECK_HIP(skinny_launch<EPI_RESID_F32>(g2, s));
            }
#ifdef LMRL_TOOLS
            else if (LMRL_ABL(LMRL_ABLATE_FC2_SEAM3 | LMRL_ABLATE_FC2_SEAM2 | LMRL_ABLATE_FC2_SEAM6)) {
                static float *seam_ws = nullptr;           // timing only
                if (!seam_ws) LMRL_CHECK_HIP(hipMalloc(&seam_ws, (size_t)6 * 2048 * 1280 * sizeof(float)));
                GemmArgs gs = g2; gs.ldw = cf.d_ff;
                int S = 3;
                if (LMRL_ABL(LMRL_ABLATE_FC2_SEAM6)) { S = 6; LMRL_CHECK_HIP((gemm8_launch_splitk<128, 128, 2, 4, 2>(gs, seam_ws, S, cf.d_ff / S, s))); }
                else {
                    S = LMRL_ABL(LMRL_ABLATE_FC2_SEAM2) ? 2 : 3;
                    LMRL_CHECK_HIP((gemm8_launch_splitk<64, 64, 4, 2, 4>(gs, seam_ws, S, cf.d_ff / S, s)));
                }
                const long total = (long)M * (d / 4);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, (const float *)seam_ws, S,
                                   (long)M * d, d, w.x, d, M, d, 1, L.b_fc2);
            }
#endif
            else if (LMRL_ABL(LMRL_ABLATE_FC2_SPLITK2 | LMRL_ABLATE_FC2_HALFK)) {
                // timing only: fc2 over HALF of K — alone (what a K loop of half the length costs), or as two such launches running concurrently on two
                // streams (racy read-modify-write of x: garbage) = a split-K = 2 form without its reduction seam
                GemmArgs ga = g2; ga.K = cf.d_ff / 2;
                GemmArgs gb = ga; gb.A = ga.A + cf.d_ff / 2; gb.W = ga.W + cf.d_ff / 2; gb.ldw = cf.d_ff;
                ga.ldw = cf.d_ff;
                if (LMRL_ABL(LMRL_ABLATE_FC2_SPLITK2)) {
                    LMRL_CHECK_HIP(hipEventRecord(aux_fork, s));
                    LMRL_CHECK_HIP(hipStreamWaitEvent(aux_stream, aux_fork, 0));
                    LMRL_CHECK_HIP(gemm_launch_ln<EPI_RESID_F32_STATS>(gb, aux_stream));
                    LMRL_CHECK_HIP(hipEventRecord(aux_join, aux_stream));
                }
                LMRL_CHECK_HIP(gemm_launch_ln<EPI_RESID_F32_STATS>(ga, s));
                if (LMRL_ABL(LMRL_ABLATE_FC2_SPLITK2)) LMRL_CHECK_HIP(hipStreamWaitEvent(s, aux_join, 0));
            }
            else if (l + 1 < cf.n_layer) LMRL_CHECK_HIP(gemm_launch_ln<EPI_RESID_F32_STATS>(g2, s));
            else LMRL_CHECK_HIP(gemm_launch<EPI_RESID_F32>(g2, s));      // ln_f reads the fp32 stream directly
        } else {
            GemmArgs gp{w.att, L.w_proj, L.b_proj, w.x, M, d, d, d, d, d};
            LMRL_CHECK_HIP(gemm_launch<EPI_RESID_F32>(gp, s));
            ln(L.ln2_g, L.ln2_b, w.h, nullptr, M);
            LMRL_CHECK_LAUNCH();
            GemmArgs gf{w.h, L.w_fc, L.b_fc, w.ff, M, cf.d_ff, d, d, cf.d_ff, cf.d_ff};
            LMRL_CHECK_HIP(gemm_launch<EPI_GELU_BF16>(gf, s));
            GemmArgs g2{w.ff, L.w_fc2, L.b_fc2, w.x, M, d, cf.d_ff, cf.d_ff, d, d};
            LMRL_CHECK_HIP(gemm_launch<EPI_RESID_F32>(g2, s));
        }
    }
    if (all_hidden_d) {   // before len is advanced / independent of it
        ln(m->lnf_g, m->lnf_b, (uint16_t *)all_hidden_d, nullptr, M);
        LMRL_CHECK_LAUNCH();
    }
    // ln_f of each env's last new token (optional) + len[b] += cnt[b]
    const float *xf = last_only ? lc_x : w.x;
    if (d <= 1024) hipLaunchKernelGGL(final_ln_advance_kernel<4>, dim3(ceil_div(b, 4)), dim3(256), 0, s, xf, m->lnf_g, m->lnf_b,
                                      (uint16_t *)last_hidden_d, cnt_d, len_d, b, c, d, cf.ln_eps, off, last_only ? 1 : 0);
    else hipLaunchKernelGGL(final_ln_advance_kernel<8>, dim3(ceil_div(b, 4)), dim3(256), 0, s, xf, m->lnf_g, m->lnf_b,
                            (uint16_t *)last_hidden_d, cnt_d, len_d, b, c, d, cf.ln_eps, off, last_only ? 1 : 0);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_gpt2_forward(lmrl_gpt2 *m, void *kv_d, int tmax, void *ws_d, const int32_t *tokens_d, const int32_t *cnt_d,
                      int32_t *len_d, int b, int c, void *last_hidden_d, void *all_hidden_d, unsigned flags, void *stream) {
    return gpt2_forward_impl(m, kv_d, tmax, ws_d, tokens_d, cnt_d, len_d, b, c, last_hidden_d, all_hidden_d, nullptr, flags, stream);
}

int lmrl_gpt2_forward_prefixed(lmrl_gpt2 *m, void *kv_d, int tmax, void *ws_d, const int32_t *tokens_d, const int32_t *cnt_d,
                               int32_t *len_d, int b, void *last_hidden_d, const lmrl_kv_prefix *pfx, unsigned flags, void *stream) {
    LMRL_REQUIRE(pfx, "lmrl_gpt2_forward_prefixed: no prefix");
    return gpt2_forward_impl(m, kv_d, tmax, ws_d, tokens_d, cnt_d, len_d, b, 1, last_hidden_d, nullptr, pfx, flags, stream);
}

int lmrl_gpt2_kv_attach(const lmrl_gpt2 *m, int src_b, const int32_t *src_len_d, const void *src_hidden_d, const int32_t *idx_d, int b,
                        int dst_tmax, void *dst_hidden_d, int32_t *dst_len_d, int32_t *pfx_n_d, int32_t *order_d, void *stream) {
    LMRL_REQUIRE(m && src_b > 0 && src_len_d && idx_d && b > 0 && dst_tmax > 0 && dst_len_d && pfx_n_d, "lmrl_gpt2_kv_attach: bad argument");
    LMRL_REQUIRE(!dst_hidden_d || src_hidden_d, "lmrl_gpt2_kv_attach: dst_hidden_d needs src_hidden_d");
    hipLaunchKernelGGL(kv_attach_kernel, dim3(b + 1), dim3(1024), 0, as_stream(stream), src_b, src_len_d, idx_d, b, m->cfg.d_model, dst_tmax,
                       (const uint16_t *)src_hidden_d, (uint16_t *)dst_hidden_d, dst_len_d, pfx_n_d, order_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_gpt2_kv_broadcast(const lmrl_gpt2 *m, const void *src_kv_d, int src_tmax, void *dst_kv_d, int dst_tmax, int b, int n_pos,
                           const void *src_hidden_d, void *dst_hidden_d, int32_t *dst_len_d, void *stream) {
    LMRL_REQUIRE(m && src_kv_d && dst_kv_d && dst_len_d && b > 0 && n_pos > 0 && n_pos <= src_tmax && n_pos <= dst_tmax,
                 "lmrl_gpt2_kv_broadcast: bad argument");
    LMRL_REQUIRE(!dst_hidden_d || src_hidden_d, "lmrl_gpt2_kv_broadcast: dst_hidden_d needs src_hidden_d");
    hipLaunchKernelGGL(kv_broadcast_kernel, dim3(b, 2 * m->cfg.n_layer), dim3(256), 0, as_stream(stream), (const uint16_t *)src_kv_d, src_tmax,
                       (uint16_t *)dst_kv_d, dst_tmax, b, n_pos, m->cfg.d_model, (const uint16_t *)src_hidden_d, (uint16_t *)dst_hidden_d, dst_len_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_gpt2_kv_gather(const lmrl_gpt2 *m, const void *src_kv_d, int src_b, int src_tmax, const int32_t *src_len_d, const void *src_hidden_d,
                        const int32_t *idx_d, void *dst_kv_d, int dst_tmax, int b, void *dst_hidden_d, int32_t *dst_len_d, void *stream) {
    LMRL_REQUIRE(m && src_kv_d && src_len_d && idx_d && dst_kv_d && dst_len_d && b > 0 && src_b > 0 && src_tmax > 0 && dst_tmax > 0,
                 "lmrl_gpt2_kv_gather: bad argument");
    LMRL_REQUIRE(!dst_hidden_d || src_hidden_d, "lmrl_gpt2_kv_gather: dst_hidden_d needs src_hidden_d");
    hipLaunchKernelGGL(kv_gather_kernel, dim3(b, 2 * m->cfg.n_layer), dim3(256), 0, as_stream(stream), (const uint16_t *)src_kv_d, src_b, src_tmax,
                       src_len_d, idx_d, (uint16_t *)dst_kv_d, dst_tmax, b, m->cfg.d_model, (const uint16_t *)src_hidden_d, (uint16_t *)dst_hidden_d,
                       dst_len_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

size_t lmrl_gemm_bf16_splitk_ws_bytes(int m, int n, int k) {
    int kchunk = 0, kind = 0;
    const int S = lmrl::splitk_plan(m, n, k, &kchunk, &kind);
    return S ? (size_t)S * m * n * sizeof(float) : 0;
}

int lmrl_gemm_bf16_splitk(const void *a_d, const void *w_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc, int n_store, int accumulate,
                          void *ws_d, void *stream) {
    LMRL_REQUIRE(a_d && w_d && c_d && ws_d && m > 0 && n > 0 && k > 0 && n_store > 0 && n_store <= n, "lmrl_gemm_bf16_splitk: bad argument");
    int kchunk = 0, kind = 0;
    const int S = lmrl::splitk_plan(m, n, k, &kchunk, &kind);
    LMRL_REQUIRE(S >= 2, "lmrl_gemm_bf16_splitk: no split-K plan for this shape (lmrl_gemm_bf16_splitk_ws_bytes returned 0)");
    hipStream_t s = as_stream(stream);
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, nullptr, ws_d, m, n, k, lda, n, n};
    g.ldw = ldw;
    if (kind == 1) LMRL_CHECK_HIP((gemm8_launch_splitk<256, 192, 2, 4, 2>(g, (float *)ws_d, S, kchunk, s)));
    else LMRL_CHECK_HIP((gemm8_launch_splitk<128, 128, 2, 4, 2>(g, (float *)ws_d, S, kchunk, s)));
    const long total = (long)m * ((n_store + 3) / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, (const float *)ws_d, S,
                       (long)m * n, n, (float *)c_d, ldc, m, n_store, accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

// The same split-K product with a bias: c[m][n] = sum_k a[m][k] w[n][k] + bias[n] — the bf16x3 rollout mode's c_proj of the MLP at decode size
// (M = one row per env, N = d_model: 48 tiles of 128 x 128, K' = 3 d_ff = 9216: 144 K-steps on a fifth of the CUs unsplit) and at chunk size (M = 8192:
// 128 tiles of 256 x 192, split in two: one full round of the CUs).  lmrl_gemm_bf16_splitk_ws_bytes says whether a plan exists — the plain product
// (lmrl_gemm_bf16) is the fallback.
int lmrl_gemm_bf16_splitk_bias(const void *a_d, const void *w_d, const float *bias_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc,
                               void *ws_d, void *stream) {
    LMRL_REQUIRE(a_d && w_d && c_d && ws_d && m > 0 && n > 0 && k > 0 && n % 64 == 0 && ldc >= n, "lmrl_gemm_bf16_splitk_bias: bad argument");
    int kchunk = 0, kind = 0;
    const int S = lmrl::splitk_plan(m, n, k, &kchunk, &kind);
    LMRL_REQUIRE(S >= 2, "lmrl_gemm_bf16_splitk_bias: no split-K plan for this shape (lmrl_gemm_bf16_splitk_ws_bytes returned 0)");
    hipStream_t s = as_stream(stream);
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, nullptr, ws_d, m, n, k, lda, n, n};
    g.ldw = ldw;
    if (kind == 1) LMRL_CHECK_HIP((gemm8_launch_splitk<256, 192, 2, 4, 2>(g, (float *)ws_d, S, kchunk, s)));      // chunk-sized M, N = d_model: 256 x 192 tiles
    else LMRL_CHECK_HIP((gemm8_launch_splitk<128, 128, 2, 4, 2>(g, (float *)ws_d, S, kchunk, s)));
    const long total = (long)m * (n / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, (const float *)ws_d, S,
                       (long)m * n, n, (float *)c_d, ldc, m, n, 0, bias_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

// dW = x^T . dy on the operands as their producers staged them: a_d = x [k][lda] (k = B*T rows, m = layer input width), w_d = dy [k][ldw];
// c[m][n] (=|+=) sum_kk a[kk][m] w[kk][n].  Same split-K plan / reduce as lmrl_gemm_bf16_splitk; the kernel gathers its MFMA operands with
// ds_read_b64_tr_b16 (gemm8_bf16.h, KM), so no transposed copy of either operand exists.
int lmrl_gemm_bf16_splitk_kmajor(const void *a_d, const void *w_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc, int n_store, int accumulate,
                                 void *ws_d, void *stream) {
    LMRL_REQUIRE(a_d && w_d && c_d && ws_d && m > 0 && n > 0 && k > 0 && n_store > 0 && n_store <= n && m % 128 == 0 && n % 128 == 0 && k % 64 == 0 &&
                     lda >= m && ldw >= n && lda % 8 == 0 && ldw % 8 == 0, "lmrl_gemm_bf16_splitk_kmajor: bad argument (m, n multiples of 128, k of 64)");
    int kchunk = 0, kind = 0;
    const int S = lmrl::splitk_plan(m, n, k, &kchunk, &kind);
    LMRL_REQUIRE(S >= 2 && kind == 0, "lmrl_gemm_bf16_splitk_kmajor: no 128 x 128 split-K plan for this shape");
    hipStream_t s = as_stream(stream);
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, nullptr, ws_d, m, n, k, lda, n, n};
    g.ldw = ldw;
    LMRL_CHECK_HIP((gemm8_launch_splitk<128, 128, 2, 4, 2, true>(g, (float *)ws_d, S, kchunk, s)));
    const long total = (long)m * ((n_store + 3) / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, (const float *)ws_d, S,
                       (long)m * n, n, (float *)c_d, ldc, m, n_store, accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

void lmrl_gemm_set_variant(int v) { lmrl::g_gemm_variant = v; }

// Plain bf16 GEMM entry (heads, LM-head logits): C = A.W^T + bias with a selectable epilogue.
int lmrl_gemm_bf16_ld(const void *a_d, const void *w_d, const float *bias_d, void *c_d, int m, int n, int k, int lda, int ldw, int ldc, int n_store,
                      int epilogue, void *stream) {
    LMRL_REQUIRE(a_d && w_d && c_d && m > 0 && n > 0 && k > 0, "lmrl_gemm_bf16: bad argument");
    LMRL_REQUIRE(n % 64 == 0 && k % 64 == 0 && lda % 8 == 0 && ldc % 4 == 0 && lda >= k && (ldw == 0 || (ldw >= k && ldw % 8 == 0)),
                 "lmrl_gemm_bf16: n, k must be multiples of 64 and the operand pitches multiples of 8 elements covering k");
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, bias_d, c_d, m, n, k, lda, ldc, n_store > 0 ? n_store : n};
    g.ldw = ldw;
    hipStream_t s = as_stream(stream);
    switch (epilogue) {
        case EPI_BF16: LMRL_CHECK_HIP(gemm_launch<EPI_BF16>(g, s)); break;
        case EPI_GELU_BF16: LMRL_CHECK_HIP(gemm_launch<EPI_GELU_BF16>(g, s)); break;
        case EPI_RESID_F32: LMRL_CHECK_HIP(gemm_launch<EPI_RESID_F32>(g, s)); break;
        case EPI_F32: LMRL_CHECK_HIP(gemm_launch<EPI_F32>(g, s)); break;
        case EPI_RELU_BF16: LMRL_CHECK_HIP(gemm_launch<EPI_RELU_BF16>(g, s)); break;
        case EPI_GELU_SPLIT3:
            LMRL_REQUIRE(ldc >= 3 * n && ldc % 4 == 0 && g.n_store == n, "lmrl_gemm_bf16: EPI_GELU_SPLIT3 writes [m][3 n] bf16 (ldc >= 3 n, n_store = n)");
            LMRL_CHECK_HIP(gemm_launch<EPI_GELU_SPLIT3>(g, s)); break;
        default: LMRL_REQUIRE(false, "lmrl_gemm_bf16: unknown epilogue");
    }
    return LMRL_OK;
}

int lmrl_gemm_bf16_resid(const void *a_d, const void *w_d, const float *bias_d, const float *resid_d, int ldr, float *c_d, int m, int n, int k, int lda,
                         int ldw, int ldc, int n_store, void *stream) {
    LMRL_REQUIRE(a_d && w_d && c_d && resid_d && m > 0 && n > 0 && k > 0, "lmrl_gemm_bf16_resid: bad argument");
    LMRL_REQUIRE(n % 64 == 0 && k % 64 == 0 && lda % 8 == 0 && ldc % 4 == 0 && ldr % 4 == 0 && lda >= k && (ldw == 0 || (ldw >= k && ldw % 8 == 0)),
                 "lmrl_gemm_bf16_resid: n, k must be multiples of 64 and the operand pitches multiples of 8 (4 for fp32) elements covering k");
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, bias_d, c_d, m, n, k, lda, ldc, n_store > 0 ? n_store : n};
    g.ldw = ldw; g.resid = resid_d; g.ldr = ldr;
    LMRL_CHECK_HIP(gemm_launch<EPI_RESID_F32>(g, as_stream(stream)));
    return LMRL_OK;
}

// ---- train step, bf16-matmul mode: products whose epilogue writes the next kernel's bf16 operand (gemm8_bf16.h EPI_F32_GELU_BF16 /
// EPI_BF16_HEADS / EPI_GELU_BWD_BF16)
static bool train_gemm_args_ok(int m, int n, int k, int lda, int ldw) {
    return m > 0 && n > 0 && k > 0 && n % 128 == 0 && k % 64 == 0 && lda % 8 == 0 && lda >= k && (ldw == 0 || (ldw >= k && ldw % 8 == 0));
}

int lmrl_gemm_bf16_gelu_dual(const void *a_d, const void *w_d, const float *bias_d, float *c_d, int ldc, void *act_bf16_d, int ldact, int m, int n, int k,
                             int lda, int ldw, void *stream) {
    LMRL_REQUIRE(a_d && w_d && act_bf16_d && train_gemm_args_ok(m, n, k, lda, ldw) && (!c_d || (ldc % 4 == 0 && ldc >= n)) && ldact % 8 == 0 && ldact >= n,
                 "lmrl_gemm_bf16_gelu_dual: bad argument (n a multiple of 128, k of 64, pitches covering n / k)");       // c_d null: bf16 gelu output only
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, bias_d, c_d, m, n, k, lda, ldc, n};
    g.ldw = ldw; g.xb = (uint16_t *)act_bf16_d; g.ldxb = ldact;
    LMRL_CHECK_HIP(gemm_launch_train<EPI_F32_GELU_BF16>(g, as_stream(stream)));
    return LMRL_OK;
}

// the same with the pre-activation stored ROUNDED TO bf16 ([m][ldpre] elements): the gelu backward is its only reader, and the reference's bf16 mode keeps
// every activation in bf16 — 100 MB instead of 201 MB per block at the ILQL batch, written here and read back by lmrl_gemm_bf16_gelu_bwd_prebf16
int lmrl_gemm_bf16_gelu_dual_prebf16(const void *a_d, const void *w_d, const float *bias_d, void *pre_bf16_d, int ldpre, void *act_bf16_d, int ldact, int m, int n,
                                     int k, int lda, int ldw, void *stream) {
    LMRL_REQUIRE(a_d && w_d && pre_bf16_d && act_bf16_d && train_gemm_args_ok(m, n, k, lda, ldw) && ldpre % 8 == 0 && ldpre >= n && ldact % 8 == 0 && ldact >= n,
                 "lmrl_gemm_bf16_gelu_dual_prebf16: bad argument (n a multiple of 128, k of 64, pitches multiples of 8 covering n / k)");
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, bias_d, pre_bf16_d, m, n, k, lda, ldpre, n};
    g.ldw = ldw; g.xb = (uint16_t *)act_bf16_d; g.ldxb = ldact; g.pre_bf16 = 1;
    LMRL_CHECK_HIP(gemm_launch_train<EPI_F32_GELU_BF16>(g, as_stream(stream)));
    return LMRL_OK;
}

int lmrl_gemm_bf16_gelu_bwd_prebf16(const void *a_d, const void *w_d, const void *pre_bf16_d, int ldpre, void *c_bf16_d, int ldc, int m, int n, int k, int lda,
                                    int ldw, void *stream) {
    LMRL_REQUIRE(a_d && w_d && pre_bf16_d && c_bf16_d && train_gemm_args_ok(m, n, k, lda, ldw) && ldc % 8 == 0 && ldc >= n && ldpre % 4 == 0 && ldpre >= n,
                 "lmrl_gemm_bf16_gelu_bwd_prebf16: bad argument");
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, nullptr, c_bf16_d, m, n, k, lda, ldc, n};
    g.ldw = ldw; g.resid = (const float *)pre_bf16_d; g.ldr = ldpre; g.pre_bf16 = 1;
    LMRL_CHECK_HIP(gemm_launch_train<EPI_GELU_BWD_BF16>(g, as_stream(stream)));
    return LMRL_OK;
}

int lmrl_gemm_bf16_qkv_heads(const void *a_d, const void *w_d, const float *bias_d, void *q_heads_d, long plane_elems, int m, int k, int lda, int ldw,
                             int heads, int t, void *stream) {
    const int n = 3 * heads * 64, tp = (t + 63) / 64 * 64;
    LMRL_REQUIRE(a_d && w_d && q_heads_d && heads > 0 && t > 0 && m % t == 0 && train_gemm_args_ok(m, n, k, lda, ldw) &&
                     plane_elems >= (long)(m / t) * heads * tp * 64,
                 "lmrl_gemm_bf16_qkv_heads: bad argument (3 * heads * 64 a multiple of 128, m = batch * t)");
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, bias_d, q_heads_d, m, n, k, lda, n, n};
    g.ldw = ldw; g.hd_T = t; g.hd_Tp = tp; g.hd_H = heads; g.hd_plane = plane_elems;
    LMRL_CHECK_HIP(gemm_launch_train<EPI_BF16_HEADS>(g, as_stream(stream)));
    return LMRL_OK;
}

// slabs per row of lmrl_gemm_bf16_ce's log-sum-exp partials for this shape (follows the tile the shape policy picks)
static int ce_slots(int m, int n, int k) {
    switch (pick_train_tile(m, n, k)) {
        case TT_256x256: return n / 256 * 4;
        case TT_256x192: return n / 192 * 2;
        default: return n / 128 * 4;
    }
}
int lmrl_gemm_bf16_ce_slots(int m, int n, int k) { return (n > 0 && n % 128 == 0) ? ce_slots(m, n, k) : 0; }

int lmrl_gemm_bf16_ce(const void *a_d, const void *w_d, const float *bias_d, void *logits_bf16_d, int ldc, int m, int n, int k, int lda, int ldw, int n_store,
                      const int32_t *targets_d, float *tgt_logit_d, void *partials_d, void *stream) {
    LMRL_REQUIRE(a_d && w_d && partials_d && train_gemm_args_ok(m, n, k, lda, ldw) && (!logits_bf16_d || (ldc % 8 == 0 && ldc >= n_store)) && n_store > 0 &&
                     n_store <= n && (!targets_d || tgt_logit_d), "lmrl_gemm_bf16_ce: bad argument");
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, bias_d, logits_bf16_d, m, n, k, lda, ldc, n_store};
    g.ldw = ldw; g.stats = (float2 *)partials_d; g.nslots = ce_slots(m, n, k); g.ce_targets = targets_d; g.ce_tgt_logit = tgt_logit_d;
    LMRL_CHECK_HIP(gemm_launch_train<EPI_BF16_CE>(g, as_stream(stream)));
    return LMRL_OK;
}

int lmrl_gemm_bf16_gelu_bwd(const void *a_d, const void *w_d, const float *pre_d, int ldpre, void *c_bf16_d, int ldc, int m, int n, int k, int lda,
                            int ldw, void *stream) {
    LMRL_REQUIRE(a_d && w_d && pre_d && c_bf16_d && train_gemm_args_ok(m, n, k, lda, ldw) && ldc % 8 == 0 && ldc >= n && ldpre % 4 == 0 && ldpre >= n,
                 "lmrl_gemm_bf16_gelu_bwd: bad argument");
    GemmArgs g{(const uint16_t *)a_d, (const uint16_t *)w_d, nullptr, c_bf16_d, m, n, k, lda, ldc, n};
    g.ldw = ldw; g.resid = pre_d; g.ldr = ldpre;
    LMRL_CHECK_HIP(gemm_launch_train<EPI_GELU_BWD_BF16>(g, as_stream(stream)));
    return LMRL_OK;
}

int lmrl_gemm_bf16(const void *a_d, const void *w_d, const float *bias_d, void *c_d, int m, int n, int k, int lda, int ldc, int n_store, int epilogue,
                   void *stream) {
    return lmrl_gemm_bf16_ld(a_d, w_d, bias_d, c_d, m, n, k, lda, 0, ldc, n_store, epilogue, stream);
}
}
