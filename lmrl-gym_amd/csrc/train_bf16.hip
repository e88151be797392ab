// train_bf16.hip — operand staging for the train step's bf16-MFMA matmul mode (the reference's optional `bf16_activations`,
// train_ilql_gpt2.py:193 / train_ppo_gpt2.py: model dtype bf16 with fp32 parameters): every Dense / Conv1D product of the forward AND
// backward pass runs on `v_mfma_f32_16x16x32_bf16` (the rollout engine's GEMM kernels, C = A . W^T with both operands K-major), with
// fp32 accumulation, fp32 outputs, fp32 master weights / gradients / optimizer state.  The three products of a linear layer
//     y  = x . w          A = bf16(x)   [R][k]     W = bf16(w)^T  [n][k]
//     dx = dy . w^T       A = bf16(dy)  [R][n]     W = bf16(w)    [k][n]
//     dw = x^T . dy       A = bf16(x)^T [k][R]     W = bf16(dy)^T [n][R]
// need each operand as a K-major bf16 matrix whose K extent is a multiple of 64: the kernels here cast (round-to-nearest-even, the MFMA
// input rounding of the rollout engine) and, where needed, transpose, zero-filling the padding.  All are HBM streaming kernels.
// Also: the gathered column product used for the ILQL target Q heads (only Q_target(s, a) of the taken token is ever read,
// ilql/base_interface.py: take_along_axis of the target logits), and a transposing accumulate for gradients produced in [n][k] form.
#include "../../include/lmrl_amd.h"
#include "gemm_bf16.h"

namespace lmrl {

// dst[r][c] = bf16(src[r][c]) for r < rows, c < cols; zero elsewhere in [rows_dst][ld_dst].  One workgroup (128 lanes) per destination row,
// 8 columns (32 B in, 16 B out) per lane and iteration: no index arithmetic beyond one multiply per row.
__global__ __launch_bounds__(128) void cast_bf16_kernel(const float *__restrict__ src, long ld_src, int rows, int cols, uint16_t *__restrict__ dst,
                                                        long ld_dst, int rows_dst) {
    const int r = blockIdx.x;
    const float *p = src + (long)r * ld_src;
    uint16_t *q = dst + (long)r * ld_dst;
    const bool vec = r < rows && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    for (int c0 = threadIdx.x * 8; c0 < ld_dst; c0 += 128 * 8) {
        float v[8];
        if (vec && c0 + 8 <= cols) {
            const float4 a = *reinterpret_cast<const float4 *>(p + c0), b = *reinterpret_cast<const float4 *>(p + c0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (r < rows && c0 + k < cols) ? p[c0 + k] : 0.f;
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(q + c0) = o;
    }
}

// Three-term bf16 split of an fp32 operand for "bf16 x 3" products: x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi); the row
// [hi(x) | lo(x) | hi(x)] (3 * cols bf16) against a weight row [hi(w) | hi(w) | lo(w)] makes ONE bf16 GEMM with K' = 3 K accumulate
// hi.hi + lo.hi + hi.lo in fp32 — every term of x.w but lo.lo (2^-16 relative), i.e. ~16 mantissa bits at three times the bf16 MFMA cost instead
// of the sixteen times of the f32-input MFMA.  One workgroup per row, 8 columns per lane and iteration (32 B in, 3 x 16 B out).
__global__ __launch_bounds__(128) void split3_bf16_kernel(const float *__restrict__ src, long ld_src, int rows, int cols, uint16_t *__restrict__ dst,
                                                          long ld_dst) {
    const int r = blockIdx.x;
    const float *p = src + (long)r * ld_src;
    uint16_t *q = dst + (long)r * ld_dst;
    for (int c0 = threadIdx.x * 8; c0 < cols; c0 += 128 * 8) {
        const float4 a = *reinterpret_cast<const float4 *>(p + c0), b = *reinterpret_cast<const float4 *>(p + c0 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float lo[8];
        uint4 hi;
        hi.x = pack_bf16x2(v[0], v[1]); hi.y = pack_bf16x2(v[2], v[3]); hi.z = pack_bf16x2(v[4], v[5]); hi.w = pack_bf16x2(v[6], v[7]);
        const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            lo[2 * k] = v[2 * k] - __uint_as_float(hw[k] << 16);
            lo[2 * k + 1] = v[2 * k + 1] - __uint_as_float(hw[k] & 0xffff0000u);
        }
        uint4 l;
        l.x = pack_bf16x2(lo[0], lo[1]); l.y = pack_bf16x2(lo[2], lo[3]); l.z = pack_bf16x2(lo[4], lo[5]); l.w = pack_bf16x2(lo[6], lo[7]);
        *reinterpret_cast<uint4 *>(q + c0) = hi;
        *reinterpret_cast<uint4 *>(q + cols + c0) = l;
        *reinterpret_cast<uint4 *>(q + 2 * cols + c0) = hi;
    }
}

// All weight matrices of a parameter arena in ONE launch (the per-step bf16 staging of the fp32 masters: ~170 launches of 7 - 9 us for
// GPT-2-small's policy + target otherwise).  Segment i: src + src_off is a [rows][cols] fp32 matrix (dense); workgroups [tile0, tile0 + tiles)
// own its 64 x 64 tiles; the natural copy goes to dst + nat_off ([rows][ld_nat], skipped when nat_off < 0), the transposed copy to
// dst + t_off ([cols][ld_t], skipped when t_off < 0).  Only the source extent is written: the caller zero-fills the destination once.
__global__ __launch_bounds__(256) void cast_bf16_segments_kernel(const float *__restrict__ src0, const lmrl_cast_seg *__restrict__ segs, int nseg,
                                                                 uint16_t *__restrict__ dst0) {
    __shared__ float tile[64][65];
    int lo = 0, hi = nseg - 1;                  // last segment with tile0 <= blockIdx.x (workgroup-uniform)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const lmrl_cast_seg sg = segs[lo];
    const int tl = blockIdx.x - sg.tile0, tcn = (sg.cols + 63) / 64;
    const int r0 = (tl / tcn) * 64, c0 = (tl % tcn) * 64;
    const float *src = src0 + sg.src_off;
    const bool vec = (sg.cols & 3) == 0 && (sg.src_off & 3) == 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idx = threadIdx.x + 256 * k, rl = idx >> 4, c4 = (idx & 15) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        const int r = r0 + rl, c = c0 + c4;
        if (r < sg.rows) {
            if (vec && c + 4 <= sg.cols) x = *reinterpret_cast<const float4 *>(src + (long)r * sg.cols + c);
            else {
                if (c < sg.cols) x.x = src[(long)r * sg.cols + c];
                if (c + 1 < sg.cols) x.y = src[(long)r * sg.cols + c + 1];
                if (c + 2 < sg.cols) x.z = src[(long)r * sg.cols + c + 2];
                if (c + 3 < sg.cols) x.w = src[(long)r * sg.cols + c + 3];
            }
        }
        tile[rl][c4] = x.x; tile[rl][c4 + 1] = x.y; tile[rl][c4 + 2] = x.z; tile[rl][c4 + 3] = x.w;
    }
    __syncthreads();
    const int e8 = (threadIdx.x & 7) * 8, q = threadIdx.x >> 3;
    if (sg.nat_off >= 0) {                       // natural: row r0 + q (+32), 8 columns from c0 + e8 (ld_nat is a multiple of 64 >= cols: in bounds)
        uint16_t *dn = dst0 + sg.nat_off;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int rl = q + 32 * k;
            if (r0 + rl < sg.rows && c0 + e8 < sg.cols) {
                uint4 o;
                o.x = pack_bf16x2(tile[rl][e8 + 0], tile[rl][e8 + 1]); o.y = pack_bf16x2(tile[rl][e8 + 2], tile[rl][e8 + 3]);
                o.z = pack_bf16x2(tile[rl][e8 + 4], tile[rl][e8 + 5]); o.w = pack_bf16x2(tile[rl][e8 + 6], tile[rl][e8 + 7]);
                *reinterpret_cast<uint4 *>(dn + (long)(r0 + rl) * sg.ld_nat + c0 + e8) = o;
            }
        }
    }
    if (sg.t_off >= 0) {                         // transposed: row c0 + q (+32), 8 source rows from r0 + e8
        uint16_t *dt = dst0 + sg.t_off;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int cl = q + 32 * k;
            if (c0 + cl < sg.cols && r0 + e8 < sg.rows) {
                uint4 o;
                o.x = pack_bf16x2(tile[e8 + 0][cl], tile[e8 + 1][cl]); o.y = pack_bf16x2(tile[e8 + 2][cl], tile[e8 + 3][cl]);
                o.z = pack_bf16x2(tile[e8 + 4][cl], tile[e8 + 5][cl]); o.w = pack_bf16x2(tile[e8 + 6][cl], tile[e8 + 7][cl]);
                *reinterpret_cast<uint4 *>(dt + (long)(c0 + cl) * sg.ld_t + r0 + e8) = o;
            }
        }
    }
}

// dst[c][r] = bf16(src[r][c]) for r < rows, c < cols; zero elsewhere in [rows_dst][ld_dst].  64 x 64 tiles through LDS: coalesced 256-B
// reads along c, 128-B writes along r.
__global__ __launch_bounds__(256) void cast_bf16_t_kernel(const float *__restrict__ src, long ld_src, int rows, int cols, uint16_t *__restrict__ dst,
                                                          long ld_dst, int rows_dst, float *__restrict__ colpart) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if ((ld_src & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && c0 + 64 <= cols) {      // workgroup-uniform: 16-byte loads
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int idx = threadIdx.x + 256 * k, rl = idx >> 4, c4 = (idx & 15) * 4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + rl < rows) x = *reinterpret_cast<const float4 *>(src + (long)(r0 + rl) * ld_src + c0 + c4);
            tile[rl][c4] = x.x; tile[rl][c4 + 1] = x.y; tile[rl][c4 + 2] = x.z; tile[rl][c4 + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int r = r0 + ty + k * 4, c = c0 + tx;
            tile[ty + k * 4][tx] = (r < rows && c < cols) ? src[(long)r * ld_src + c] : 0.f;
        }
    }
    __syncthreads();
    if (colpart && threadIdx.x < 64 && r0 < rows) {      // this 64-row block's share of the column sums (the bias gradient of the layer): fixed order
        float sum = 0.f;
        for (int r = 0; r < 64; r++) sum += tile[r][threadIdx.x];
        if (c0 + (int)threadIdx.x < rows_dst) colpart[(long)blockIdx.x * rows_dst + c0 + threadIdx.x] = sum;
    }
    // out row = c0 + cc, 64 r-values = 128 B: 8 lanes x 16 B per row, 32 rows per pass
    const int rr8 = (threadIdx.x & 7) * 8, cc = threadIdx.x >> 3;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int c = c0 + cc + k * 32;
        if (c < rows_dst && r0 + rr8 < ld_dst) {
            uint4 o;
            const int q = cc + k * 32;
            o.x = pack_bf16x2(tile[rr8 + 0][q], tile[rr8 + 1][q]); o.y = pack_bf16x2(tile[rr8 + 2][q], tile[rr8 + 3][q]);
            o.z = pack_bf16x2(tile[rr8 + 4][q], tile[rr8 + 5][q]); o.w = pack_bf16x2(tile[rr8 + 6][q], tile[rr8 + 7][q]);
            *reinterpret_cast<uint4 *>(dst + (long)c * ld_dst + r0 + rr8) = o;
        }
    }
}

// The same from a bf16 source (an operand a producer already staged, e.g. dlogits written by ce_bwd_bf16_kernel): dst[c][r] = src[r][c].
// Half the read traffic of the fp32 form and no second rounding; colpart sums the bf16 values (what the dW product multiplies).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t *__restrict__ src, long ld_src, int rows, int cols,
                                                             uint16_t *__restrict__ dst, long ld_dst, int rows_dst, float *__restrict__ colpart) {
    __shared__ uint16_t tile[64][72];            // row pitch 144 B: 16-byte aligned rows, column reads spread over the banks
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int idx = threadIdx.x + 256 * k, rl = idx >> 3, c8 = (idx & 7) * 8;
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + rl < rows && c0 + c8 < cols) x = *reinterpret_cast<const uint4 *>(src + (long)(r0 + rl) * ld_src + c0 + c8);   // src padding is zero
        *reinterpret_cast<uint4 *>(&tile[rl][c8]) = x;
    }
    __syncthreads();
    if (colpart && threadIdx.x < 64 && r0 < rows) {
        float sum = 0.f;
        for (int r = 0; r < 64; r++) sum += bf16_to_f32(tile[r][threadIdx.x]);
        if (c0 + (int)threadIdx.x < rows_dst) colpart[(long)blockIdx.x * rows_dst + c0 + threadIdx.x] = sum;
    }
    const int rr8 = (threadIdx.x & 7) * 8, cc = threadIdx.x >> 3;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int c = c0 + cc + k * 32;
        if (c < rows_dst && r0 + rr8 < ld_dst) {
            const int q = cc + k * 32;
            uint4 o;
            o.x = (uint32_t)tile[rr8 + 0][q] | ((uint32_t)tile[rr8 + 1][q] << 16); o.y = (uint32_t)tile[rr8 + 2][q] | ((uint32_t)tile[rr8 + 3][q] << 16);
            o.z = (uint32_t)tile[rr8 + 4][q] | ((uint32_t)tile[rr8 + 5][q] << 16); o.w = (uint32_t)tile[rr8 + 6][q] | ((uint32_t)tile[rr8 + 7][q] << 16);
            *reinterpret_cast<uint4 *>(dst + (long)c * ld_dst + r0 + rr8) = o;
        }
    }
}

// dlogits of the CE / gather losses (ce_bwd_kernel's formula) written as the bf16 A operand [rows_dst][ld_dst] of the head's backward
// products — no fp32 dlogits matrix, no cast pass over it.  One workgroup per destination row; padding rows / columns are zero-filled.
__global__ __launch_bounds__(256) void ce_bwd_bf16_kernel(const float *__restrict__ logits, int ld, int V, const float *__restrict__ lse,
                                                          const int32_t *__restrict__ targets, const float *__restrict__ coef_ce,
                                                          const float *__restrict__ coef_gather, int rows, uint16_t *__restrict__ dst, long ld_dst) {
    const int r = blockIdx.x;
    uint16_t *q = dst + (long)r * ld_dst;
    if (r >= rows) {
        for (int c0 = threadIdx.x * 8; c0 < ld_dst; c0 += 256 * 8) *reinterpret_cast<uint4 *>(q + c0) = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const float *row = logits + (size_t)r * ld;
    const float l = lse[r], cc = coef_ce ? coef_ce[r] : 0.f, cg = coef_gather ? coef_gather[r] : 0.f;
    int t = targets[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const bool vec = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
    for (int c0 = threadIdx.x * 8; c0 < ld_dst; c0 += 256 * 8) {
        float x[8], v[8];
        if (vec && c0 + 8 <= V) {
            const float4 a = *reinterpret_cast<const float4 *>(row + c0), b = *reinterpret_cast<const float4 *>(row + c0 + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = c0 + k < V ? row[c0 + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = c0 + k;
            float g = 0.f;
            if (c < V) {
                g = cc == 0.f ? 0.f : cc * expf(x[k] - l);
                if (c == t) g += cg - cc;
            }
            v[k] = g;
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(q + c0) = o;
    }
}

// The same on bf16 logits, IN PLACE: q[r][c] := bf16(dlogit) for r < rows, c < V; 0 in the padding (rows up to rows_dst, columns up to ld) — the
// vocabulary heads' logits were written once, in bf16, by the head GEMM (EPI_BF16_CE, which also produced lse from its fp32 accumulators), and
// become the dy operand of the head's backward products without another [rows][V] buffer.
__global__ __launch_bounds__(256) void ce_bwd_bf16_inplace_kernel(uint16_t *__restrict__ q0, long ld, int V, const float *__restrict__ lse,
                                                                  const int32_t *__restrict__ targets, const float *__restrict__ coef_ce,
                                                                  const float *__restrict__ coef_gather, int rows) {
    const int r = blockIdx.x;
    uint16_t *q = q0 + (long)r * ld;
    if (r >= rows) {
        for (int c0 = threadIdx.x * 8; c0 < ld; c0 += 256 * 8) *reinterpret_cast<uint4 *>(q + c0) = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const float l = lse[r], cc = coef_ce ? coef_ce[r] : 0.f, cg = coef_gather ? coef_gather[r] : 0.f;
    int t = targets[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    for (int c0 = threadIdx.x * 8; c0 < ld; c0 += 256 * 8) {
        const uint4 in = *reinterpret_cast<const uint4 *>(q + c0);
        const uint32_t w[4] = {in.x, in.y, in.z, in.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = c0 + k;
            const float x = __uint_as_float((k & 1) ? (w[k >> 1] & 0xffff0000u) : (w[k >> 1] << 16));
            float g = 0.f;
            if (c < V) {
                g = cc == 0.f ? 0.f : cc * expf(x - l);
                if (c == t) g += cg - cc;
            }
            v[k] = g;
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(q + c0) = o;
    }
}

// lse[r] = log sum_c exp(logit[r][c]) from the head GEMM's per-slab partials (max, sum exp) [rows][nslots] (slabs of padding columns carry
// max = -inf); logprob[r] = tgt_logit[r] - lse[r] when asked.  One wave per row, fixed order.
__global__ __launch_bounds__(256) void lse_from_partials_kernel(const float2 *__restrict__ part, int nslots, int rows, const float *__restrict__ tgt_logit,
                                                                float *__restrict__ lse, float *__restrict__ logprob) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float2 *p = part + (size_t)r * nslots;
    float m = -INFINITY;
    for (int k = lane; k < nslots; k += 64) m = fmaxf(m, p[k].x);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int k = lane; k < nslots; k += 64) {
        const float2 v = p[k];
        if (v.x != -INFINITY) s += v.y * expf(v.x - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float l = m + logf(s);
        lse[r] = l;
        if (logprob) logprob[r] = tgt_logit[r] - l;
    }
}

// colpart[rb][c] = sum of the 64 rows of block rb of a bf16 matrix [rows][ld] (fixed order): the bias gradient of a Dense layer from its staged dy
// when no transposing pass over dy runs any more (lmrl_gemm_bf16_splitk_kmajor).  64 lanes x 8 columns, 4 row phases merged through LDS.
__global__ __launch_bounds__(256) void colpart_bf16_kernel(const uint16_t *__restrict__ src, long ld, int rows, int cols, float *__restrict__ colpart,
                                                           int ldp) {
    __shared__ float sm[4][64][9];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + tx) * 8, r0 = blockIdx.y * 64;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < cols) {
#pragma unroll 4
        for (int k = 0; k < 16; k++) {
            const int r = r0 + ty + 4 * k;
            if (r < rows) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + (long)r * ld + c0);      // padding columns of a staged operand are zero
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) { a[2 * e] += __uint_as_float(w[e] << 16); a[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) sm[ty][tx][e] = a[e];
    __syncthreads();
    if (ty == 0 && c0 < cols) {
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (c0 + e < ldp) colpart[(long)blockIdx.y * ldp + c0 + e] = (sm[0][tx][e] + sm[1][tx][e]) + (sm[2][tx][e] + sm[3][tx][e]);
    }
}

// out[c] (+)= sum over row blocks of colpart[rb][c] (deterministic: fixed partition and order).  One workgroup per 64 columns, 16 lane groups
// each summing every 16th row block, combined through LDS in group order — the one-thread-per-column form walked 256 dependent strided loads
// from 3 .. 12 workgroups (20.8 us per call, 53 calls per ILQL step: profiles/r03_ilql_bf16_step_kernel_stats_before_fusion.csv).
__global__ __launch_bounds__(1024) void colpart_reduce_kernel(const float *__restrict__ colpart, int nrb, int ldp, int cols, float *__restrict__ out,
                                                              int accumulate) {
    __shared__ float part[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int rb = ty;
        for (; rb + 16 < nrb; rb += 32) { s0 += colpart[(long)rb * ldp + c]; s1 += colpart[(long)(rb + 16) * ldp + c]; }
        if (rb < nrb) s0 += colpart[(long)rb * ldp + c];
    }
    part[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) s += part[k][tx];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// dst[j][i] = beta * dst[j][i] + src[i][j]   (src [n][k], dst [k][n]): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_add_kernel(const float *__restrict__ src, long ld_src, float *__restrict__ dst, long ld_dst, int n,
                                                            int k, float beta) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = i0 + ty + q * 8, j = j0 + tx;
        tile[ty + q * 8][tx] = (i < n && j < k) ? src[(long)i * ld_src + j] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int j = j0 + ty + q * 8, i = i0 + tx;
        if (i < n && j < k) {
            float *p = dst + (long)j * ld_dst + i;
            *p = (beta != 0.f ? beta * *p : 0.f) + tile[tx][ty + q * 8];
        }
    }
}

// out[r] = sum_j a[r][j] * w[j][idx[r]] + bias[idx[r]]: one wave per row
__global__ __launch_bounds__(256) void gather_dot_kernel(const float *__restrict__ a, long lda, const float *__restrict__ w, long ldw,
                                                         const float *__restrict__ bias, const int32_t *__restrict__ idx, float *__restrict__ out,
                                                         int rows, int k, int n) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    int c = idx[r];
    c = c < 0 ? 0 : (c >= n ? n - 1 : c);
    float s = 0.f;
    for (int j = lane; j < k; j += 64) s = fmaf(a[(long)r * lda + j], w[(long)j * ldw + c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = s + (bias ? bias[c] : 0.f);
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

int lmrl_cast_bf16(const float *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, int rows_dst, int transpose, void *stream) {
    LMRL_REQUIRE(src_d && dst_d && rows > 0 && cols > 0 && rows_dst > 0 && ld_dst > 0 && ld_dst % 8 == 0, "lmrl_cast_bf16: bad argument");
    hipStream_t s = as_stream(stream);
    if (!transpose) {
        LMRL_REQUIRE(rows_dst >= rows && ld_dst >= cols, "lmrl_cast_bf16: destination smaller than the source");
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(rows_dst), dim3(128), 0, s, src_d, ld_src, rows, cols, (uint16_t *)dst_d, ld_dst, rows_dst);
    } else {
        LMRL_REQUIRE(rows_dst >= cols && ld_dst >= rows, "lmrl_cast_bf16: destination smaller than the transposed source");
        hipLaunchKernelGGL(cast_bf16_t_kernel, dim3((int)((ld_dst + 63) / 64), (rows_dst + 63) / 64), dim3(256), 0, s, src_d, ld_src, rows, cols,
                           (uint16_t *)dst_d, ld_dst, rows_dst, (float *)nullptr);
    }
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_cast_bf16_segments(const float *src_d, const lmrl_cast_seg *segs_d, int nseg, int total_tiles, void *dst_d, void *stream) {
    LMRL_REQUIRE(src_d && segs_d && dst_d && nseg > 0 && total_tiles > 0, "lmrl_cast_bf16_segments: bad argument");
    hipLaunchKernelGGL(cast_bf16_segments_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream), src_d, segs_d, nseg, (uint16_t *)dst_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_split3_bf16(const float *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, void *stream) {
    LMRL_REQUIRE(src_d && dst_d && rows > 0 && cols > 0 && cols % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0 && ld_dst >= 3L * cols &&
                 ((reinterpret_cast<uintptr_t>(src_d) & 15) == 0), "lmrl_split3_bf16: bad argument (cols % 8, 16-byte aligned rows, ld_dst >= 3 cols)");
    hipLaunchKernelGGL(split3_bf16_kernel, dim3(rows), dim3(128), 0, as_stream(stream), src_d, ld_src, rows, cols, (uint16_t *)dst_d, ld_dst);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

size_t lmrl_cast_bf16_t_colsum_ws_bytes(int rows, int rows_dst) { return (size_t)((rows + 63) / 64) * (size_t)rows_dst * sizeof(float); }

int lmrl_cast_bf16_t_colsum(const float *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, int rows_dst, float *colsum_d,
                            int accumulate, float *ws_d, void *stream) {
    LMRL_REQUIRE(src_d && dst_d && colsum_d && ws_d && rows > 0 && cols > 0 && ld_dst % 8 == 0 && rows_dst >= cols && ld_dst >= rows,
                 "lmrl_cast_bf16_t_colsum: bad argument");
    hipStream_t s = as_stream(stream);
    const int nrb = (rows + 63) / 64;       // row blocks that hold source rows (blocks beyond only zero-fill the padding)
    hipLaunchKernelGGL(cast_bf16_t_kernel, dim3((int)((ld_dst + 63) / 64), (rows_dst + 63) / 64), dim3(256), 0, s, src_d, ld_src, rows, cols,
                       (uint16_t *)dst_d, ld_dst, rows_dst, ws_d);
    hipLaunchKernelGGL(colpart_reduce_kernel, dim3((cols + 63) / 64), dim3(1024), 0, s, (const float *)ws_d, nrb, rows_dst, cols, colsum_d, accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_transpose_bf16_colsum(const void *src_d, long ld_src, int rows, int cols, void *dst_d, long ld_dst, int rows_dst, float *colsum_d,
                               int accumulate, float *ws_d, void *stream) {
    LMRL_REQUIRE(src_d && dst_d && rows > 0 && cols > 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && rows_dst >= cols && ld_dst >= rows &&
                 (!colsum_d || ws_d), "lmrl_transpose_bf16_colsum: bad argument");
    hipStream_t s = as_stream(stream);
    const int nrb = (rows + 63) / 64;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((int)((ld_dst + 63) / 64), (rows_dst + 63) / 64), dim3(256), 0, s, (const uint16_t *)src_d, ld_src,
                       rows, cols, (uint16_t *)dst_d, ld_dst, rows_dst, colsum_d ? ws_d : (float *)nullptr);
    if (colsum_d)
        hipLaunchKernelGGL(colpart_reduce_kernel, dim3((cols + 63) / 64), dim3(1024), 0, s, (const float *)ws_d, nrb, rows_dst, cols, colsum_d,
                           accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ce_bwd_bf16_inplace(void *logits_bf16_d, long ld, int vocab, const float *lse_d, const int32_t *targets_d, const float *coef_ce_d,
                             const float *coef_gather_d, int rows, int rows_dst, void *stream) {
    LMRL_REQUIRE(logits_bf16_d && lse_d && targets_d && rows > 0 && vocab > 0 && ld >= vocab && ld % 8 == 0 && rows_dst >= rows,
                 "lmrl_ce_bwd_bf16_inplace: bad argument");
    hipLaunchKernelGGL(ce_bwd_bf16_inplace_kernel, dim3(rows_dst), dim3(256), 0, as_stream(stream), (uint16_t *)logits_bf16_d, ld, vocab, lse_d, targets_d,
                       coef_ce_d, coef_gather_d, rows);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_lse_from_partials(const void *partials_d, int nslots, int rows, const float *tgt_logit_d, float *lse_d, float *logprob_d, void *stream) {
    LMRL_REQUIRE(partials_d && lse_d && nslots > 0 && rows > 0 && (!logprob_d || tgt_logit_d), "lmrl_lse_from_partials: bad argument");
    hipLaunchKernelGGL(lse_from_partials_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), (const float2 *)partials_d, nslots, rows,
                       tgt_logit_d, lse_d, logprob_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_colsum_bf16(const void *src_d, long ld_src, int rows, int cols, float *colsum_d, int accumulate, float *ws_d, void *stream) {
    LMRL_REQUIRE(src_d && colsum_d && ws_d && rows > 0 && cols > 0 && ld_src % 8 == 0 && ld_src >= (cols + 7) / 8 * 8, "lmrl_colsum_bf16: bad argument");
    hipStream_t s = as_stream(stream);
    const int nrb = (rows + 63) / 64;
    hipLaunchKernelGGL(colpart_bf16_kernel, dim3((cols + 511) / 512, nrb), dim3(256), 0, s, (const uint16_t *)src_d, ld_src, rows, cols, ws_d, cols);
    hipLaunchKernelGGL(colpart_reduce_kernel, dim3((cols + 63) / 64), dim3(1024), 0, s, (const float *)ws_d, nrb, cols, cols, colsum_d, accumulate);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ce_bwd_bf16(const float *logits_d, int ld, int vocab, const float *lse_d, const int32_t *targets_d, const float *coef_ce_d,
                     const float *coef_gather_d, int rows, void *dst_d, long ld_dst, int rows_dst, void *stream) {
    LMRL_REQUIRE(logits_d && lse_d && targets_d && dst_d && rows > 0 && vocab > 0 && ld >= vocab && ld_dst >= vocab && ld_dst % 8 == 0 &&
                 rows_dst >= rows, "lmrl_ce_bwd_bf16: bad argument");
    hipLaunchKernelGGL(ce_bwd_bf16_kernel, dim3(rows_dst), dim3(256), 0, as_stream(stream), logits_d, ld, vocab, lse_d, targets_d, coef_ce_d,
                       coef_gather_d, rows, (uint16_t *)dst_d, ld_dst);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_transpose_add_f32(const float *src_d, long ld_src, float *dst_d, long ld_dst, int n, int k, float beta, void *stream) {
    LMRL_REQUIRE(src_d && dst_d && n > 0 && k > 0, "lmrl_transpose_add_f32: bad argument");
    hipLaunchKernelGGL(transpose_add_kernel, dim3((n + 31) / 32, (k + 31) / 32), dim3(256), 0, as_stream(stream), src_d, ld_src, dst_d, ld_dst, n, k,
                       beta);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_gather_dot_f32(const float *a_d, long lda, const float *w_d, long ldw, const float *bias_d, const int32_t *idx_d, float *out_d, int rows,
                        int k, int n, void *stream) {
    LMRL_REQUIRE(a_d && w_d && idx_d && out_d && rows > 0 && k > 0 && n > 0, "lmrl_gather_dot_f32: bad argument");
    hipLaunchKernelGGL(gather_dot_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), a_d, lda, w_d, ldw, bias_d, idx_d, out_d, rows, k, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

}  // extern "C"
