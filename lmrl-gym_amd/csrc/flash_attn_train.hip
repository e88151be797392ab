// flash_attn_train.hip — causal self-attention of the TRAIN step (forward and backward) without the [B*H][T][T] score / probability
// tensors: the HF-Flax GPT-2 attention the reference's `_step` functions differentiate (ppo/gpt2/interface.py:72-211,
// ilql/gpt2/interface.py:88-367) — softmax(Q K^T / sqrt(64) + causal + key-padding mask) V, head dim 64 — as an online-softmax tile
// sweep on the matrix cores.  One source for both arithmetic modes of the train step:
//     E = F32  : exact fp32 operands on v_mfma_f32_16x16x4_f32 (the reference's default arithmetic)
//     E = BF16 : bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation / softmax / outputs (`bf16_activations`)
// A lane's MFMA operand is always "8 consecutive k-elements of row (lane & 15), starting at 8*(lane >> 4)" of a 32-wide k slab: the bf16
// instruction consumes it directly; the fp32 path issues 8 16x16x4 MFMAs whose k index is paired (lane >> 4, s) <-> 8*(lane >> 4) + s on
// both operands (a permutation of the summation index).  Operands are staged once per call by a streaming pre-pass into per-head
// K-major matrices (natural [T][64] and transposed [64][T]), so every tile load is a run of 16-byte row chunks.
//
// Forward, one workgroup per (64 queries, head): 4 waves x 16 queries; per 64-key block S = K.Q^T lands as lane (query = lane & 15) x
// 8 CONSECUTIVE keys (the K rows fed to the two 16-row MFMAs of a 32-key slab are interleaved 4 by 4), i.e. exactly the next MFMA's
// operand layout: P goes from registers straight into O += V^T.P with no LDS round trip.  Saves lse = m + log(l) per query.
// Backward, two kernels (no atomics, deterministic): dQ per query block (recompute S, P; dP = V.dO^T; dS = P (dP - D); dQ = K^T.dS)
// and dK/dV per key block (lane owns a key: S^T, dP^T; dV += dO^T.P^T, dK += Q^T.dS^T), D = rowsum(dO o O) from the pre-pass.
// Masking matches softmax_causal_fwd_kernel (train_ops.hip): keys c <= r with key_mask[b][c] != 0; a row with no valid key gives 0.
#include "../../include/lmrl_amd.h"
#include "gemm_bf16.h"

namespace lmrl {

struct ElemF32 {
    typedef float T;
    static constexpr int SZ = 4, CPR = 16, ROWB = 256, TILE = 64 * 256;
    struct Frag { f32x4 a, b; };
};
struct ElemBF16 {
    typedef uint16_t T;
    static constexpr int SZ = 2, CPR = 8, ROWB = 128, TILE = 64 * 128;
    struct Frag { bf16x8 v; };
};

// 64 x 64 tile in LDS: 16-byte chunks, chunk index XOR-swizzled with the row so that the 16-lane groups of ds_read_b128 hit 16 distinct
// chunk columns (the GEMM kernels' scheme; fp32 rows are 16 chunks wide, so the swizzle uses 4 row bits)
template <class E>
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * E::ROWB + ((chunk ^ (row & (E::CPR - 1))) << 4); }

// the same in two halves — global -> registers now, registers -> LDS later — so that the loads of the NEXT tile are in flight while the
// MFMAs of the current one run (the sweeps are latency-bound otherwise: a 64 x 64 tile is 16 MFMAs per wave between two barriers)
template <class E>
struct TileRegs { u32x4 v[64 * E::CPR / 256]; };
template <class E>
__device__ __forceinline__ void tile_fetch(TileRegs<E> &r, const typename E::T *src, long ld) {
#pragma unroll
    for (int i = 0; i < 64 * E::CPR / 256; i++) {
        const int c = threadIdx.x + 256 * i, row = c / E::CPR, ch = c % E::CPR;
        r.v[i] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(src + (long)row * ld) + ch * 16);
    }
}
template <class E>
__device__ __forceinline__ void tile_commit(char *tile, const TileRegs<E> &r) {
#pragma unroll
    for (int i = 0; i < 64 * E::CPR / 256; i++) {
        const int c = threadIdx.x + 256 * i, row = c / E::CPR, ch = c % E::CPR;
        *reinterpret_cast<u32x4 *>(tile + tile_off<E>(row, ch)) = r.v[i];
    }
}

template <class E>
__device__ __forceinline__ void load_tile(char *tile, const typename E::T *src, long ld) {
#pragma unroll
    for (int c = threadIdx.x; c < 64 * E::CPR; c += 256) {
        const int row = c / E::CPR, ch = c % E::CPR;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(src + (long)row * ld) + ch * 16);
        *reinterpret_cast<u32x4 *>(tile + tile_off<E>(row, ch)) = v;
    }
}

__device__ __forceinline__ ElemBF16::Frag ld_frag_lds(ElemBF16, const char *tile, int row, int e0) {
    ElemBF16::Frag f;
    f.v = *reinterpret_cast<const bf16x8 *>(tile + tile_off<ElemBF16>(row, e0 >> 3));
    return f;
}
__device__ __forceinline__ ElemF32::Frag ld_frag_lds(ElemF32, const char *tile, int row, int e0) {
    ElemF32::Frag f;
    f.a = *reinterpret_cast<const f32x4 *>(tile + tile_off<ElemF32>(row, e0 >> 2));
    f.b = *reinterpret_cast<const f32x4 *>(tile + tile_off<ElemF32>(row, (e0 >> 2) + 1));
    return f;
}
__device__ __forceinline__ ElemBF16::Frag ld_frag_glb(ElemBF16, const uint16_t *p) {
    ElemBF16::Frag f;
    f.v = *reinterpret_cast<const bf16x8 *>(p);
    return f;
}
__device__ __forceinline__ ElemF32::Frag ld_frag_glb(ElemF32, const float *p) {
    ElemF32::Frag f;
    f.a = *reinterpret_cast<const f32x4 *>(p);
    f.b = *reinterpret_cast<const f32x4 *>(p + 4);
    return f;
}
__device__ __forceinline__ ElemBF16::Frag make_frag(ElemBF16, const float (&v)[8]) {
    u32x4 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    ElemBF16::Frag f;
    f.v = __builtin_bit_cast(bf16x8, w);
    return f;
}
__device__ __forceinline__ ElemF32::Frag make_frag(ElemF32, const float (&v)[8]) {
    ElemF32::Frag f;
    f.a = f32x4{v[0], v[1], v[2], v[3]};
    f.b = f32x4{v[4], v[5], v[6], v[7]};
    return f;
}
// acc[i = 4*(lane>>4) + e][j = lane & 15] += sum_k A[i][k] B[j][k] over a 32-wide k slab
__device__ __forceinline__ f32x4 mma(ElemBF16, f32x4 acc, const ElemBF16::Frag &a, const ElemBF16::Frag &b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(ElemF32, f32x4 acc, const ElemF32::Frag &a, const ElemF32::Frag &b) {
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.a[s], b.a[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.b[s], b.b[s], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void st_elem(uint16_t *p, float v) { *p = f32_to_bf16_rn(v); }
__device__ __forceinline__ void st_elem(float *p, float v) { *p = v; }

// key-validity bytes of one 64-key block -> LDS (1 = key index < T and key_mask set).  Branch-free on purpose: the per-element
// `ok ? score : -inf` selects below must stay selects (a divergent short-circuit load around AGPR moves was miscompiled by hipcc 7.2).
__device__ __forceinline__ uint8_t key_valid_fetch(const uint8_t *kmb, int k0, int T) {      // threads 0..63; branch-free as below
    const int kk = k0 + (threadIdx.x & 63);
    const int kc = kk < T ? kk : T - 1;
    const uint8_t mv = kmb ? kmb[kc] : (uint8_t)1;
    return (kk < T && mv != 0) ? 1 : 0;
}
__device__ __forceinline__ void load_key_valid(uint8_t *sM, const uint8_t *kmb, int k0, int T) {
    if (threadIdx.x < 64) {
        const int kk = k0 + threadIdx.x;
        const int kc = kk < T ? kk : T - 1;
        const uint8_t mv = kmb ? kmb[kc] : (uint8_t)1;
        sM[threadIdx.x] = (kk < T && mv != 0) ? 1 : 0;
    }
}

// rows of a 32-row slab fed to the two 16-row MFMAs so that the lane holding output rows 4*lq + e gets slab rows 8*lq + e (first MFMA)
// and 8*lq + 4 + e (second): 8 consecutive rows per lane
__device__ __forceinline__ int slab_row(int lr) { return (lr >> 2) * 8 + (lr & 3); }

// ------------------------------------------------------------------------------------------ operand staging
// src [B*T][ld] fp32, columns col0 + h*64 + c  ->  natural Xn [BH][Tp][64] and transposed XT [BH][64][Tp] (zero rows / columns for t >= T)
// 8 consecutive elements per lane: one 16-byte store (bf16) or two (fp32)
__device__ __forceinline__ void st_vec8(uint16_t *p, const float (&v)[8]) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16_rn(v[0]) | ((uint32_t)f32_to_bf16_rn(v[1]) << 16); o.y = (uint32_t)f32_to_bf16_rn(v[2]) | ((uint32_t)f32_to_bf16_rn(v[3]) << 16);
    o.z = (uint32_t)f32_to_bf16_rn(v[4]) | ((uint32_t)f32_to_bf16_rn(v[5]) << 16); o.w = (uint32_t)f32_to_bf16_rn(v[6]) | ((uint32_t)f32_to_bf16_rn(v[7]) << 16);
    *reinterpret_cast<uint4 *>(p) = o;
}
__device__ __forceinline__ void st_vec8(float *p, const float (&v)[8]) {
    *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4 *>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

template <class E>
__global__ __launch_bounds__(256) void flash_stage_kernel(const float *__restrict__ src, long ld, int col0, float scale, typename E::T *__restrict__ Xn,
                                                          typename E::T *__restrict__ XT, const float *__restrict__ other, float *__restrict__ rowdot,
                                                          int H, int T, int Tp, const uint16_t *__restrict__ other_b = nullptr, long ldob = 0) {
    // other_b (instead of other): the second factor of the row dot products as bf16 [B*T][ldob] — D = rowsum(dO o O) from the bf16 attention output the
    // forward wrote for the output projection (then no fp32 copy of O exists)
    __shared__ float tile[64][65];
    __shared__ float prod[64][65];
    const int t0 = blockIdx.x * 64, bh = blockIdx.y, b = bh / H, h = bh - b * H;
    // in: 16 lanes x 16 B per token row (the 64 floats of this head), 16 rows per pass
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = threadIdx.x + 256 * k, r = i >> 4, c4 = (i & 15) * 4, t = t0 + r;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f}, o = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            const long at = ((long)b * T + t) * ld + col0 + h * 64 + c4;
            v = *reinterpret_cast<const f32x4 *>(src + at) * scale;
            if (other) o = *reinterpret_cast<const f32x4 *>(other + at);
            else if (other_b) {
                const uint2 u = *reinterpret_cast<const uint2 *>(other_b + ((long)b * T + t) * ldob + col0 + h * 64 + c4);
                o = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) tile[r][c4 + e] = v[e];
        if (other || other_b) {
#pragma unroll
            for (int e = 0; e < 4; e++) prod[r][c4 + e] = v[e] * o[e];
        }
    }
    __syncthreads();
    // out: 8 lanes x 8 elements per row, 32 rows per pass — natural rows (token, head dims) and transposed rows (head dim, tokens)
    const int c8 = (threadIdx.x & 7) * 8, rr = threadIdx.x >> 3;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = rr + 32 * k;
        float a[8], bt[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { a[e] = tile[r][c8 + e]; bt[e] = tile[c8 + e][r]; }
        st_vec8(Xn + ((long)bh * Tp + t0 + r) * 64 + c8, a);
        if (XT) st_vec8(XT + ((long)bh * 64 + r) * Tp + t0 + c8, bt);      // null: the round-4 bf16 sweeps read the natural matrix both ways
    }
    if ((other || other_b) && threadIdx.x < 64) {
        float s = 0.f;
        for (int c = 0; c < 64; c++) s += prod[threadIdx.x][c];
        rowdot[(long)bh * Tp + t0 + threadIdx.x] = s;
    }
}

// bf16 only: the q / k / v matrices were written in their natural per-head form [3][BH][Tp][64] by the c_attn GEMM's epilogue
// (gemm8_bf16.h EPI_BF16_HEADS: no fp32 qkv tensor exists); this pass adds the transposed forms [3][BH][64][Tp] and zeroes the rows t >= T of
// both (the GEMM stores token rows only).  One workgroup per (64 tokens, head, matrix); 8 elements per lane both ways.
__global__ __launch_bounds__(256) void flash_transpose_staged_kernel(uint16_t *__restrict__ Xn0, uint16_t *__restrict__ XT0, long plane, int T, int Tp) {
    __shared__ uint32_t tile[64][65];          // one element per dword, odd pitch: both passes are conflict-free (bank = row + column)
    const int t0 = blockIdx.x * 64, bh = blockIdx.y;
    uint16_t *Xn = Xn0 + blockIdx.z * plane + ((long)bh * Tp + t0) * 64;
    uint16_t *XT = XT0 + blockIdx.z * plane + (long)bh * 64 * Tp + t0;
    const int c8 = (threadIdx.x & 7) * 8, rr = threadIdx.x >> 3;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = rr + 32 * k;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (t0 + r < T) v = *reinterpret_cast<const uint4 *>(Xn + (long)r * 64 + c8);
        else *reinterpret_cast<uint4 *>(Xn + (long)r * 64 + c8) = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { tile[r][c8 + 2 * e] = w[e] & 0xffffu; tile[r][c8 + 2 * e + 1] = w[e] >> 16; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = rr + 32 * k;             // head dimension
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; e++) w[e] = tile[c8 + 2 * e][r] | (tile[c8 + 2 * e + 1][r] << 16);
        *reinterpret_cast<uint4 *>(XT + (long)r * Tp + c8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// the round-4 sweeps read only the natural matrices: all that is left of the pass above is zeroing their rows t in [T, Tp) (Tp > T only)
__global__ __launch_bounds__(256) void flash_zero_pad_rows_kernel(uint16_t *__restrict__ Xn0, long plane, int T, int Tp) {
    const int bh = blockIdx.x, n = (Tp - T) * 8;                  // 16-byte pieces of the pad rows of one head
    uint16_t *Xn = Xn0 + blockIdx.y * plane + ((long)bh * Tp + T) * 64;
    for (int i = threadIdx.x; i < n; i += 256) reinterpret_cast<uint4 *>(Xn)[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------------ forward
template <class E, bool PF>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const typename E::T *__restrict__ Qn, const typename E::T *__restrict__ Kn,
                                                        const typename E::T *__restrict__ VT, const uint8_t *__restrict__ km, float *__restrict__ att,
                                                        float *__restrict__ lse, int H, int T, int Tp, int d, uint16_t *__restrict__ att_b, long ldb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sK = smem, *sV = smem + E::TILE;
    uint8_t *sM = reinterpret_cast<uint8_t *>(smem + 2 * E::TILE);
    const int qb = gridDim.x - 1 - blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H;      // longest sweeps first
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
    const int qi = qb * 64 + wave * 16 + lr;
    typename E::Frag qf[2];
#pragma unroll
    for (int s = 0; s < 2; s++) qf[s] = ld_frag_glb(E(), Qn + ((long)bh * Tp + qi) * 64 + s * 32 + lq * 8);
    const uint8_t *kmb = km ? km + (long)b * T : nullptr;
    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ra = slab_row(lr);
    TileRegs<E> rk, rv;
    uint8_t rm = 0;
    if (PF) {
        tile_fetch<E>(rk, Kn + ((long)bh * Tp) * 64, 64);
        tile_fetch<E>(rv, VT + (long)bh * 64 * Tp, Tp);
        rm = key_valid_fetch(kmb, 0, T);
    }
    for (int kb = 0; kb <= qb; kb++) {
        __syncthreads();
        if (PF) {
            tile_commit<E>(sK, rk);
            tile_commit<E>(sV, rv);
            if (threadIdx.x < 64) sM[threadIdx.x] = rm;
        } else {
            load_tile<E>(sK, Kn + ((long)bh * Tp + kb * 64) * 64, 64);
            load_tile<E>(sV, VT + (long)bh * 64 * Tp + kb * 64, Tp);
            load_key_valid(sM, kmb, kb * 64, T);
        }
        __syncthreads();
        if (PF && kb < qb) {        // next key block: requested now, committed to LDS after this block's MFMAs
            tile_fetch<E>(rk, Kn + ((long)bh * Tp + (kb + 1) * 64) * 64, 64);
            tile_fetch<E>(rv, VT + (long)bh * 64 * Tp + (kb + 1) * 64, Tp);
            rm = key_valid_fetch(kmb, (kb + 1) * 64, T);
        }
        // blocks strictly below the diagonal whose 64 keys are all valid need no masking at all (most blocks of a long sweep): the
        // per-element compare / byte extract / select is a third of the VALU work of a block.  Workgroup-uniform (SGPR) condition.
        bool plain = kb < qb;
        {
            const unsigned long long *m8 = reinterpret_cast<const unsigned long long *>(sM);
            unsigned long long all = 0x0101010101010101ull;
#pragma unroll
            for (int i = 0; i < 8; i++) all &= m8[i];
            plain = plain && __builtin_amdgcn_readfirstlane((int)(all == 0x0101010101010101ull)) != 0;
        }
        float s[2][8];
        float mloc = -INFINITY;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa;
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                sa = mma(E(), sa, ld_frag_lds(E(), sK, p * 32 + ra, sl * 32 + lq * 8), qf[sl]);
                sb = mma(E(), sb, ld_frag_lds(E(), sK, p * 32 + ra + 4, sl * 32 + lq * 8), qf[sl]);
            }
            if (plain) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float v = e < 4 ? sa[e] : sb[e - 4];
                    s[p][e] = v;
                    mloc = fmaxf(mloc, v);
                }
            } else {
                const unsigned long long mb = *reinterpret_cast<const unsigned long long *>(sM + p * 32 + lq * 8);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int kk = kb * 64 + p * 32 + lq * 8 + e;
                    const bool ok = (kk <= qi) & (((mb >> (8 * e)) & 0xffull) != 0);
                    const float v = ok ? (e < 4 ? sa[e] : sb[e - 4]) : -INFINITY;
                    s[p][e] = v;
                    mloc = fmaxf(mloc, v);
                }
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m, mloc);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m - m_safe);
        float rs = 0.f;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 8; e++) { s[p][e] = __expf(s[p][e] - m_safe); rs += s[p][e]; }
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] *= alpha;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const typename E::Frag pf = make_frag(E(), s[p]);
#pragma unroll
            for (int db = 0; db < 4; db++) o[db] = mma(E(), o[db], ld_frag_lds(E(), sV, db * 16 + lr, p * 32 + lq * 8), pf);
        }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (qi < T) {
#pragma unroll
        for (int db = 0; db < 4; db++)
            *reinterpret_cast<f32x4 *>(att + ((long)b * T + qi) * d + h * 64 + db * 16 + lq * 4) = o[db] * inv;
        if (att_b) {     // bf16 copy: the operand of the output projection in the bf16-matmul train mode
#pragma unroll
            for (int db = 0; db < 4; db++) {
                const f32x4 v = o[db] * inv;
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(att_b + ((long)b * T + qi) * ldb + h * 64 + db * 16 + lq * 4) = pk;
            }
        }
    }
    if (lq == 0) lse[(long)bh * Tp + qi] = (qi < T && l > 0.f) ? m + __logf(l) : INFINITY;
}

// ------------------------------------------------------------------------------------------ forward, bf16, round 4
// Same arithmetic and operand layouts as flash_fwd_kernel<ElemBF16>, rebuilt around what tools/pmc_flash.sh and the timing-only ablations of
// tools/bench_flash_ablate.py showed (the sweep is bound by VALU issue + exposed latencies, not by MFMA rate; profiles/r04_flash_*):
//   * K / V tiles (natural rows, see below) go L2 -> LDS with global_load_lds into a 2-slot ring (the GEMM kernels' lane-linear image + source-side
//     XOR swizzle): the next block's tiles land while the current block is computed; ONE barrier per key block, no staging registers, no ds_write
//     pass — and with that 109-118 VGPRs: four waves per SIMD (the register-staged kernel holds 124 + the MFMA results in AGPRs)
//   * XCD-aware 1-D grid: all query tiles of one (batch, head) run on ONE XCD, longest sweep first, so its K / V enter that L2 once
//   * key validity: one byte per key in LDS for the whole row (written once per workgroup), a 64-bit ballot per block; blocks below the diagonal
//     with all keys valid take a straight-line path in which all 8 QK^T MFMAs are issued before the first score is read
//   * exp(s - m) as v_exp_f32(fma(s, log2 e, -m log2 e)); the 4-row max / sum reductions with v_permlane16_swap / v_permlane32_swap
//     (VALU) instead of ds_bpermute
//   * one chunk swizzle (flash_swz, below) that is conflict-free for both kinds of fragment read
// Measured (B = 32, H = 12, 10 back to back): T = 512 70.7 -> 46.9 us, T = 1024 174.9 -> 130.9 us (8-wave workgroups; 4 waves: 48.3 / 134.6).  Also measured: two query groups per wave
// (128-query workgroups, every fragment read feeds two MFMAs) 63 / 189 us at 168 VGPRs = three waves per SIMD — occupancy beats reuse here.
// Every tile of the round-4 sweeps is a NATURAL [64 rows][64 head dims] block (128-byte rows) read two ways from the same LDS image:
//   * row fragments ("8 consecutive head dims of row r", the QK^T / dP operand: rows slab_row(lr) + 4 hf + 32 p, chunk 4 sl + lq) by ds_read_b128
//   * column fragments ("8 consecutive ROWS of head dim c", the PV / dQ / dK / dV operand) by two ds_read_b64_tr_b16 — no transposed copy of
//     K, V, Q or dO exists in memory (round 3 staged six extra matrices per layer and DMA'd their tiles as well)
// One chunk swizzle serves both: 16-byte chunk c of row r sits at c ^ flash_swz(r), flash_swz(r) = (bit 1 of r | bit 3 of r << 1) << 1.  The row
// set of one ds_read_b128 lane group ({0-3, 24-27} with chunk c, {8-11, 16-19} with c ^ 1) then covers 16 distinct 16-byte bank groups, and the
// 32 lanes of a ds_read_b64_tr_b16 half-wave (rows 8 lq + (lr >> 2), lq in {0, 1} or {2, 3}: four 32-byte pieces on each row parity) all 64 banks once.
typedef __bf16 flash_bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int flash_swz(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }
template <int NW>
__device__ __forceinline__ void flash_dma_tile(char *dst, const uint16_t *src, int wave, int lane) {       // src: 64 contiguous rows of 128 B
    const int lrow = lane >> 3;
#pragma unroll
    for (int i = 0; i < 8 / NW; i++) {
        const int seg = wave + NW * i, row = seg * 8 + lrow;
        const int src_c = (lane & 7) ^ flash_swz(row);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + row * 64 + src_c * 8),
                                         (__attribute__((address_space(3))) void *)(dst + seg * 1024), 16, 0, 0);
    }
}
__device__ __forceinline__ ElemBF16::Frag flash_k_frag(const char *tile, int row, int e0) {
    ElemBF16::Frag f;
    f.v = *reinterpret_cast<const bf16x8 *>(tile + row * 128 + (((e0 >> 3) ^ flash_swz(row)) << 4));
    return f;
}
// column fragment: lane (lr, lq) <- rows r0 + 8 lq + 0..7 of head dim db * 16 + lr.  A 16-lane group addresses the sixteen 8-byte pieces of
// 4 rows x 16 columns (lane -> row lr >> 2, piece lr & 3) and ds_read_b64_tr_b16 hands lane lr column lr of that block (gemm8_bf16.h g8_tr_frag)
__device__ __forceinline__ ElemBF16::Frag flash_t_frag(const char *tile, int r0, int db, int lr, int lq) {
    const int row = r0 + 8 * lq + (lr >> 2), j = lr & 3;
    const char *p = tile + row * 128 + (((db * 2 + (j >> 1)) ^ flash_swz(row)) << 4) + (j & 1) * 8;      // flash_swz(row + 4) == flash_swz(row) here
    const flash_bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) flash_bf16x4_t *)(p));
    const flash_bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) flash_bf16x4_t *)(p + 4 * 128));
    ElemBF16::Frag f;
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
}
// max / sum over the four 16-lane rows of a wave (the lanes that share lane & 15), all VALU: v_permlane16_swap pairs rows (0,1) and (2,3),
// v_permlane32_swap the two halves — no LDS round trip (ds_bpermute) on the softmax's critical path
__device__ __forceinline__ float quad_row_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
__device__ __forceinline__ float quad_row_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}
// one 64-key block for this wave's 16 queries.  MASKED false: every key of the block is valid and at or below every query of the wave
template <bool MASKED, int ABL>
__device__ __forceinline__ void flash_fwd2_block(const char *sK, const char *sV, unsigned long long vm, int kb, int qi, const ElemBF16::Frag (&qf)[2],
                                                 float &m, float &l, f32x4 (&o)[4], int lr, int lq, int ra) {
    typedef ElemBF16 E;
    constexpr float LOG2E = 1.4426950408889634f;
    f32x4 sc[2][2];                 // [32-key slab][half]: this lane's keys slab * 32 + lq * 8 + half * 4 + e of query lr
#pragma unroll
    for (int sl = 0; sl < 2; sl++)
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                const E::Frag kf = flash_k_frag(sK, p * 32 + ra + 4 * hf, sl * 32 + lq * 8);
                const f32x4 c = sl == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sc[p][hf];
                if (ABL & 1) { sc[p][hf] = c; sc[p][hf][0] += __builtin_bit_cast(f32x4, kf.v)[0] * __builtin_bit_cast(f32x4, qf[sl].v)[0]; }
                else sc[p][hf] = mma(E(), c, kf, qf[sl]);
            }
    if (MASKED) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const unsigned mb = (unsigned)(vm >> (p * 32 + lq * 8)) & 0xffu;          // validity bits of this lane's 8 keys of the slab
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int kk = kb * 64 + p * 32 + lq * 8 + e;
                const bool ok = (kk <= qi) & (((mb >> e) & 1u) != 0);
                sc[p][e >> 2][e & 3] = ok ? sc[p][e >> 2][e & 3] : -INFINITY;
            }
        }
    }
    float mloc = fmaxf(fmaxf(sc[0][0][0], sc[0][0][1]), fmaxf(sc[0][0][2], sc[0][0][3]));
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
            if (p + hf > 0) mloc = fmaxf(mloc, fmaxf(fmaxf(sc[p][hf][0], sc[p][hf][1]), fmaxf(sc[p][hf][2], sc[p][hf][3])));
    mloc = quad_row_max(mloc);
    const float m_new = fmaxf(m, mloc);
    const float m_safe = (MASKED && m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f((m - m_safe) * LOG2E);
    const float mb2 = -m_safe * LOG2E;
    float rs = 0.f;
    E::Frag pf[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        float e8[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float a = __builtin_fmaf(sc[p][e >> 2][e & 3], LOG2E, mb2);
            e8[e] = (ABL & 2) ? a : __builtin_amdgcn_exp2f(a);
            rs += e8[e];
        }
        pf[p] = make_frag(E(), e8);
    }
    l = l * alpha + rs;
    m = m_new;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] *= alpha;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int db = 0; db < 4; db++) {
            const E::Frag vf = flash_t_frag(sV, p * 32, db, lr, lq);
            if (ABL & 4) o[db][0] += __builtin_bit_cast(f32x4, vf.v)[p] * __builtin_bit_cast(f32x4, pf[p].v)[db];
            else o[db] = mma(E(), o[db], vf, pf[p]);
        }
}
// NW waves per workgroup, 16 queries per wave
// NW waves per workgroup, 16 queries per wave; RS ring slots: RS - 1 key blocks in flight ahead of the one being computed (counted vmcnt waits).
// RS = 3 measured no faster than 2 (T = 512: 119.0 vs 117.3 us incl. staging, T = 1024: 293 vs 280): the parked time is waves waiting for each other at
// the barrier, not tiles still in flight
template <int NW, int ABL, int RS = 2>
__global__ __launch_bounds__(64 * NW, 4) void flash_fwd2_bf16_kernel(const uint16_t *__restrict__ Qn, const uint16_t *__restrict__ Kn,
                                                                     const uint16_t *__restrict__ Vn, const uint8_t *__restrict__ km,
                                                                     float *__restrict__ att, float *__restrict__ lse, int BH, int H, int T, int Tp, int d,
                                                                     uint16_t *__restrict__ att_b, long ldb) {
    typedef ElemBF16 E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLOT = 2 * E::TILE, QT = 16 * NW, OPS = 2 * (8 / NW);     // OPS: tile DMAs per wave per key block
    uint8_t *sKey = reinterpret_cast<uint8_t *>(smem + RS * SLOT);             // [Tp] key validity of the whole row of this batch element
    // XCD-aware 1-D grid: workgroup id -> XCD id % 8 (observed placement, as in gemm_bf16.h).  All query tiles of one (batch, head) go to ONE XCD,
    // consecutively and longest sweep first: its K / V (2 x Tp x 128 B) enter that XCD's L2 once and the other tiles' fills hit there,
    // instead of eight L2s each pulling every head's tiles over the fabric.  Heads beyond the last full group of 8 wrap onto the XCDs in order.
    const int nq = (Tp + QT - 1) / QT;
    int bh, qb;
    {
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;          // k-th workgroup of this XCD
        const int full = (BH / 8) * 8;
        if (k < (BH / 8) * nq) { bh = (k / nq) * 8 + xcd; qb = nq - 1 - k % nq; }
        else {                                                         // the BH % 8 remaining heads: their tiles dealt round-robin
            const int r = (k - (BH / 8) * nq) * 8 + xcd;
            bh = full + r / nq; qb = nq - 1 - r % nq;
            if (bh >= BH) return;                                      // workgroup-uniform
        }
    }
    const int b = bh / H, h = bh - b * H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 15, lq = lane >> 4;
    const int nkb = min(((qb + 1) * QT + 63) / 64, Tp / 64);      // key blocks of this query tile
    const uint16_t *Kb = Kn + (long)bh * Tp * 64, *Vb = Vn + (long)bh * Tp * 64;
    const uint8_t *kmb = km ? km + (long)b * T : nullptr;
    for (int i = threadIdx.x; i < Tp; i += 64 * NW) sKey[i] = (i < T && (!kmb || kmb[i] != 0)) ? 1 : 0;      // visible after the first barrier
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < RS - 1; j++)
        if (j < nkb) {
            flash_dma_tile<NW>(smem + j * SLOT, Kb + (long)j * 64 * 64, wave, lane);
            flash_dma_tile<NW>(smem + j * SLOT + E::TILE, Vb + (long)j * 64 * 64, wave, lane);
        }
    const int q0 = qb * QT + wave * 16, qi = q0 + lr;
    E::Frag qf[2];
    {
        const int qc = qi < Tp ? qi : Tp - 1;
#pragma unroll
        for (int s = 0; s < 2; s++) qf[s] = ld_frag_glb(E(), Qn + ((long)bh * Tp + qc) * 64 + s * 32 + lq * 8);
    }
    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ra = slab_row(lr);
    int slot = 0;
    for (int kb = 0; kb < nkb; kb++) {
        const char *sK = smem + slot * SLOT, *sV = sK + E::TILE;
        if (!(ABL & 8) || kb == 0) {        // tools, bit 3: no waits, no barriers after the first block
            // block kb's tiles must have landed; the RS - 2 younger blocks (fewer at the end of the sweep) stay in flight
            const int ahead = nkb - 1 - kb;
            if (RS >= 4 && ahead >= 2) wait_vmcnt<2 * OPS>();
            else if (RS >= 3 && ahead >= 1) wait_vmcnt<OPS>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (first block: the sKey stores)
            __builtin_amdgcn_s_barrier();          // raw barrier (__syncthreads would drain vmcnt): ... for every wave; and every wave is done with the slot of block kb - 1
            asm volatile("" ::: "memory");
        }
        if (kb + RS - 1 < nkb && !(ABL & 16)) {  // tools, bit 4: no tile traffic after the prologue
            char *nx = smem + (slot == 0 ? RS - 1 : slot - 1) * SLOT;          // the slot block kb - 1 just left
            flash_dma_tile<NW>(nx, Kb + (long)(kb + RS - 1) * 64 * 64, wave, lane);
            flash_dma_tile<NW>(nx + E::TILE, Vb + (long)(kb + RS - 1) * 64 * 64, wave, lane);
        }
        slot = slot + 1 == RS ? 0 : slot + 1;
        const unsigned long long vm = __ballot(sKey[kb * 64 + lane] != 0);     // key validity of block kb, one bit per key
        // wave-uniform (SGPR) case split
        if (kb * 64 > q0 + 15) continue;                          // NW = 8: the block lies entirely above this wave's queries
        if (vm == ~0ull && kb * 64 + 63 <= q0) flash_fwd2_block<false, ABL>(sK, sV, vm, kb, qi, qf, m, l, o, lr, lq, ra);
        else flash_fwd2_block<true, ABL>(sK, sV, vm, kb, qi, qf, m, l, o, lr, lq, ra);
    }
    l = quad_row_sum(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (qi < T) {
#pragma unroll
        for (int db = 0; db < 4; db++) {
            const f32x4 v = o[db] * inv;
            if (att) *reinterpret_cast<f32x4 *>(att + ((long)b * T + qi) * d + h * 64 + db * 16 + lq * 4) = v;      // null: a forward nobody differentiates
            if (att_b) {     // bf16 copy: the operand of the output projection in the bf16-matmul train mode
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(att_b + ((long)b * T + qi) * ldb + h * 64 + db * 16 + lq * 4) = pk;
            }
        }
    }
    if (lq == 0 && qi < Tp) lse[(long)bh * Tp + qi] = (qi < T && l > 0.f) ? m + __logf(l) : INFINITY;
}

// ------------------------------------------------------------------------------------------ backward: dQ
template <class E, bool PF>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const typename E::T *__restrict__ Qn, const typename E::T *__restrict__ Kn,
                                                           const typename E::T *__restrict__ Vn, const typename E::T *__restrict__ KT,
                                                           const typename E::T *__restrict__ dOn, const float *__restrict__ Dsum,
                                                           const float *__restrict__ lse, const uint8_t *__restrict__ km, float *__restrict__ dqkv,
                                                           uint16_t *__restrict__ dqb, long ldb,
                                                           int H, int T, int Tp, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sK = smem, *sV = smem + E::TILE, *sKT = smem + 2 * E::TILE;
    uint8_t *sM = reinterpret_cast<uint8_t *>(smem + 3 * E::TILE);
    const int qb = gridDim.x - 1 - blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
    const int qi = qb * 64 + wave * 16 + lr;
    typename E::Frag qf[2], dof[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        qf[s] = ld_frag_glb(E(), Qn + ((long)bh * Tp + qi) * 64 + s * 32 + lq * 8);
        dof[s] = ld_frag_glb(E(), dOn + ((long)bh * Tp + qi) * 64 + s * 32 + lq * 8);
    }
    const float Lq = lse[(long)bh * Tp + qi], Dq = Dsum[(long)bh * Tp + qi];
    const uint8_t *kmb = km ? km + (long)b * T : nullptr;
    f32x4 dq[4];
#pragma unroll
    for (int i = 0; i < 4; i++) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ra = slab_row(lr);
    TileRegs<E> rk, rv, rkt;
    uint8_t rm = 0;
    if (PF) {
        tile_fetch<E>(rk, Kn + ((long)bh * Tp) * 64, 64);
        tile_fetch<E>(rv, Vn + ((long)bh * Tp) * 64, 64);
        tile_fetch<E>(rkt, KT + (long)bh * 64 * Tp, Tp);
        rm = key_valid_fetch(kmb, 0, T);
    }
    for (int kb = 0; kb <= qb; kb++) {
        __syncthreads();
        if (PF) {
            tile_commit<E>(sK, rk);
            tile_commit<E>(sV, rv);
            tile_commit<E>(sKT, rkt);
            if (threadIdx.x < 64) sM[threadIdx.x] = rm;
        } else {
            load_tile<E>(sK, Kn + ((long)bh * Tp + kb * 64) * 64, 64);
            load_tile<E>(sV, Vn + ((long)bh * Tp + kb * 64) * 64, 64);
            load_tile<E>(sKT, KT + (long)bh * 64 * Tp + kb * 64, Tp);
            load_key_valid(sM, kmb, kb * 64, T);
        }
        __syncthreads();
        if (PF && kb < qb) {
            tile_fetch<E>(rk, Kn + ((long)bh * Tp + (kb + 1) * 64) * 64, 64);
            tile_fetch<E>(rv, Vn + ((long)bh * Tp + (kb + 1) * 64) * 64, 64);
            tile_fetch<E>(rkt, KT + (long)bh * 64 * Tp + (kb + 1) * 64, Tp);
            rm = key_valid_fetch(kmb, (kb + 1) * 64, T);
        }
        bool plain = kb < qb && qb * 64 + 63 < T;
        {
            const unsigned long long *m8 = reinterpret_cast<const unsigned long long *>(sM);
            unsigned long long all = 0x0101010101010101ull;
#pragma unroll
            for (int i = 0; i < 8; i++) all &= m8[i];
            plain = plain && __builtin_amdgcn_readfirstlane((int)(all == 0x0101010101010101ull)) != 0;
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa, pa = sa, pb = sa;
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                sa = mma(E(), sa, ld_frag_lds(E(), sK, p * 32 + ra, sl * 32 + lq * 8), qf[sl]);
                sb = mma(E(), sb, ld_frag_lds(E(), sK, p * 32 + ra + 4, sl * 32 + lq * 8), qf[sl]);
                pa = mma(E(), pa, ld_frag_lds(E(), sV, p * 32 + ra, sl * 32 + lq * 8), dof[sl]);
                pb = mma(E(), pb, ld_frag_lds(E(), sV, p * 32 + ra + 4, sl * 32 + lq * 8), dof[sl]);
            }
            float ds[8];
            if (plain) {      // below the diagonal, all 64 keys valid, all 64 queries real: no masking (see flash_fwd_kernel)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float sv = e < 4 ? sa[e] : sb[e - 4], dp = e < 4 ? pa[e] : pb[e - 4];
                    ds[e] = __expf(sv - Lq) * (dp - Dq);
                }
            } else {
                const unsigned long long mb = *reinterpret_cast<const unsigned long long *>(sM + p * 32 + lq * 8);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int kk = kb * 64 + p * 32 + lq * 8 + e;
                    const bool ok = (kk <= qi) & (qi < T) & (((mb >> (8 * e)) & 0xffull) != 0);
                    const float sv = e < 4 ? sa[e] : sb[e - 4], dp = e < 4 ? pa[e] : pb[e - 4];
                    const float pr = ok ? __expf(sv - Lq) : 0.f;
                    ds[e] = pr * (dp - Dq);
                }
            }
            const typename E::Frag dsf = make_frag(E(), ds);
#pragma unroll
            for (int db = 0; db < 4; db++) dq[db] = mma(E(), dq[db], ld_frag_lds(E(), sKT, db * 16 + lr, p * 32 + lq * 8), dsf);
        }
    }
    if (qi < T) {
#pragma unroll
        for (int db = 0; db < 4; db++) {
            const f32x4 v = dq[db] * 0.125f;
            if (dqb)       // bf16-matmul train mode: dqkv is only the dy operand of the c_attn backward products — written as that operand
                *reinterpret_cast<uint2 *>(dqb + ((long)b * T + qi) * ldb + h * 64 + db * 16 + lq * 4) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            else
                *reinterpret_cast<f32x4 *>(dqkv + ((long)b * T + qi) * 3 * d + h * 64 + db * 16 + lq * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
template <class E, bool PF>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const typename E::T *__restrict__ Qn, const typename E::T *__restrict__ Kn,
                                                            const typename E::T *__restrict__ Vn, const typename E::T *__restrict__ QT,
                                                            const typename E::T *__restrict__ dOn, const typename E::T *__restrict__ dOT,
                                                            const float *__restrict__ Dsum, const float *__restrict__ lse,
                                                            const uint8_t *__restrict__ km, float *__restrict__ dqkv, uint16_t *__restrict__ dqb,
                                                            long ldb, int H, int T, int Tp, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sQ = smem, *sdO = smem + E::TILE, *sQT = smem + 2 * E::TILE, *sdOT = smem + 3 * E::TILE;
    float *sL = reinterpret_cast<float *>(smem + 4 * E::TILE), *sD = sL + 64;
    const int kb = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H;                         // key block 0 has the longest sweep: first
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
    const int kj = kb * 64 + wave * 16 + lr;
    typename E::Frag kf[2], vf[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        kf[s] = ld_frag_glb(E(), Kn + ((long)bh * Tp + kj) * 64 + s * 32 + lq * 8);
        vf[s] = ld_frag_glb(E(), Vn + ((long)bh * Tp + kj) * 64 + s * 32 + lq * 8);
    }
    const uint8_t kmv = km ? km[(long)b * T + (kj < T ? kj : T - 1)] : (uint8_t)1;
    const bool key_ok = (kj < T) & (kmv != 0);
    const bool all_keys_ok = __syncthreads_and((int)key_ok) != 0;          // all 64 keys of this workgroup's block are valid
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = dk[i]; }
    const int ra = slab_row(lr);
    const int nqb = Tp / 64;
    TileRegs<E> rq, rdo, rqt, rdot;
    float rl = 0.f, rd = 0.f;
    auto fetch = [&](int qb_) {
        tile_fetch<E>(rq, Qn + ((long)bh * Tp + qb_ * 64) * 64, 64);
        tile_fetch<E>(rdo, dOn + ((long)bh * Tp + qb_ * 64) * 64, 64);
        tile_fetch<E>(rqt, QT + (long)bh * 64 * Tp + qb_ * 64, Tp);
        tile_fetch<E>(rdot, dOT + (long)bh * 64 * Tp + qb_ * 64, Tp);
        rl = lse[(long)bh * Tp + qb_ * 64 + (threadIdx.x & 63)];
        rd = Dsum[(long)bh * Tp + qb_ * 64 + (threadIdx.x & 63)];
    };
    if (PF) fetch(kb);
    for (int qb = kb; qb < nqb; qb++) {
        __syncthreads();
        if (PF) {
            tile_commit<E>(sQ, rq);
            tile_commit<E>(sdO, rdo);
            tile_commit<E>(sQT, rqt);
            tile_commit<E>(sdOT, rdot);
            if (threadIdx.x < 64) { sL[threadIdx.x] = rl; sD[threadIdx.x] = rd; }
        } else {
            load_tile<E>(sQ, Qn + ((long)bh * Tp + qb * 64) * 64, 64);
            load_tile<E>(sdO, dOn + ((long)bh * Tp + qb * 64) * 64, 64);
            load_tile<E>(sQT, QT + (long)bh * 64 * Tp + qb * 64, Tp);
            load_tile<E>(sdOT, dOT + (long)bh * 64 * Tp + qb * 64, Tp);
            if (threadIdx.x < 64) {
                sL[threadIdx.x] = lse[(long)bh * Tp + qb * 64 + threadIdx.x];
                sD[threadIdx.x] = Dsum[(long)bh * Tp + qb * 64 + threadIdx.x];
            }
        }
        __syncthreads();
        if (PF && qb + 1 < nqb) fetch(qb + 1);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa, pa = sa, pb = sa;
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                sa = mma(E(), sa, ld_frag_lds(E(), sQ, p * 32 + ra, sl * 32 + lq * 8), kf[sl]);
                sb = mma(E(), sb, ld_frag_lds(E(), sQ, p * 32 + ra + 4, sl * 32 + lq * 8), kf[sl]);
                pa = mma(E(), pa, ld_frag_lds(E(), sdO, p * 32 + ra, sl * 32 + lq * 8), vf[sl]);
                pb = mma(E(), pb, ld_frag_lds(E(), sdO, p * 32 + ra + 4, sl * 32 + lq * 8), vf[sl]);
            }
            float pr[8], ds[8];
            if (all_keys_ok && qb > kb && qb * 64 + 63 < T) {      // workgroup-uniform: no masking on this block
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int ql = p * 32 + lq * 8 + e;
                    const float sv = e < 4 ? sa[e] : sb[e - 4], dp = e < 4 ? pa[e] : pb[e - 4];
                    pr[e] = __expf(sv - sL[ql]);
                    ds[e] = pr[e] * (dp - sD[ql]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int ql = p * 32 + lq * 8 + e, qq = qb * 64 + ql;
                    const bool ok = key_ok & (qq >= kj) & (qq < T);
                    const float sv = e < 4 ? sa[e] : sb[e - 4], dp = e < 4 ? pa[e] : pb[e - 4];
                    pr[e] = ok ? __expf(sv - sL[ql]) : 0.f;
                    ds[e] = pr[e] * (dp - sD[ql]);
                }
            }
            const typename E::Frag pf = make_frag(E(), pr), dsf = make_frag(E(), ds);
#pragma unroll
            for (int db = 0; db < 4; db++) {
                dv[db] = mma(E(), dv[db], ld_frag_lds(E(), sdOT, db * 16 + lr, p * 32 + lq * 8), pf);
                dk[db] = mma(E(), dk[db], ld_frag_lds(E(), sQT, db * 16 + lr, p * 32 + lq * 8), dsf);
            }
        }
    }
    if (kj < T) {
#pragma unroll
        for (int db = 0; db < 4; db++) {
            if (dqb) {
                uint16_t *row = dqb + ((long)b * T + kj) * ldb + h * 64 + db * 16 + lq * 4;
                *reinterpret_cast<uint2 *>(row + d) = make_uint2(pack_bf16x2(dk[db][0], dk[db][1]), pack_bf16x2(dk[db][2], dk[db][3]));
                *reinterpret_cast<uint2 *>(row + 2 * d) = make_uint2(pack_bf16x2(dv[db][0], dv[db][1]), pack_bf16x2(dv[db][2], dv[db][3]));
            } else {
                float *row = dqkv + ((long)b * T + kj) * 3 * d + h * 64 + db * 16 + lq * 4;
                *reinterpret_cast<f32x4 *>(row + d) = dk[db];
                *reinterpret_cast<f32x4 *>(row + 2 * d) = dv[db];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward, bf16, round 4
// The two backward sweeps rebuilt like flash_fwd2_bf16_kernel: 8 waves x 16 queries (dQ) / 16 keys (dK, dV) per workgroup, global_load_lds
// tile ring with one barrier per block, XCD-local heads, ballot / lse-based masking, exp2 with the log2 e factor folded into an FMA, all
// QK^T and dP MFMAs of a block issued before the first score is read.  Same operands, same summation order per output element as the
// round-3 kernels (the bf16 results differ only through exp2(fma) vs exp(sub): ~1 ulp of P before its bf16 rounding).
__device__ __forceinline__ int flash_xcd_tile(int BH, int nt, int &bh, int &tile) {       // -> 0 if this workgroup has no tile; tile 0 first
    const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
    if (k < (BH / 8) * nt) { bh = (k / nt) * 8 + xcd; tile = k % nt; return 1; }
    const int r = (k - (BH / 8) * nt) * 8 + xcd;
    bh = (BH / 8) * 8 + r / nt; tile = r % nt;
    return bh < BH;
}
inline int flash_xcd_grid(int bh, int nt) { return (bh / 8) * nt * 8 + ((bh % 8) * nt + 7) / 8 * 8; }

template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void flash_bwd2_dq_bf16_kernel(const uint16_t *__restrict__ Qn, const uint16_t *__restrict__ Kn,
                                                                        const uint16_t *__restrict__ Vn,
                                                                        const uint16_t *__restrict__ dOn, const float *__restrict__ Dsum,
                                                                        const float *__restrict__ lse, const uint8_t *__restrict__ km,
                                                                        float *__restrict__ dqkv, uint16_t *__restrict__ dqb, long ldb, int BH, int H,
                                                                        int T, int Tp, int d) {
    typedef ElemBF16 E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLOT = 2 * E::TILE, QT = 16 * NW;
    constexpr float LOG2E = 1.4426950408889634f;
    const int nq = (Tp + QT - 1) / QT;
    int bh, t_;
    if (!flash_xcd_tile(BH, nq, bh, t_)) return;
    const int qb = nq - 1 - t_;                                   // longest sweeps first
    const int b = bh / H, h = bh - b * H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 15, lq = lane >> 4;
    const int nkb = min(((qb + 1) * QT + 63) / 64, Tp / 64);
    const uint16_t *Kb = Kn + (long)bh * Tp * 64, *Vb = Vn + (long)bh * Tp * 64;
    const uint8_t *kmb = km ? km + (long)b * T : nullptr;
    flash_dma_tile<NW>(smem, Kb, wave, lane);
    flash_dma_tile<NW>(smem + E::TILE, Vb, wave, lane);
    uint8_t rm = key_valid_fetch(kmb, 0, T);
    const int q0 = qb * QT + wave * 16, qi = q0 + lr, qc = qi < Tp ? qi : Tp - 1;
    E::Frag qf[2], dof[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        qf[s] = ld_frag_glb(E(), Qn + ((long)bh * Tp + qc) * 64 + s * 32 + lq * 8);
        dof[s] = ld_frag_glb(E(), dOn + ((long)bh * Tp + qc) * 64 + s * 32 + lq * 8);
    }
    // lse = +inf marks a query without a valid key (and the rows t >= T): exp2(s c - inf) = 0, no separate test
    const float nL2 = -(qi < T ? lse[(long)bh * Tp + qc] : INFINITY) * LOG2E, Dq = Dsum[(long)bh * Tp + qc];
    f32x4 dq[4];
#pragma unroll
    for (int i = 0; i < 4; i++) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ra = slab_row(lr);
    for (int kb = 0; kb < nkb; kb++) {
        const char *sK = smem + (kb & 1) * SLOT, *sV = sK + E::TILE;
        const unsigned long long vm = __ballot(rm != 0);
        wait_vmcnt<0>();
        __syncthreads();
        if (kb + 1 < nkb) {
            char *nx = smem + ((kb + 1) & 1) * SLOT;
            flash_dma_tile<NW>(nx, Kb + (long)(kb + 1) * 64 * 64, wave, lane);
            flash_dma_tile<NW>(nx + E::TILE, Vb + (long)(kb + 1) * 64 * 64, wave, lane);
            rm = key_valid_fetch(kmb, (kb + 1) * 64, T);
        }
        if (kb * 64 > q0 + 15) continue;                          // the block lies entirely above this wave's queries (wave-uniform)
        const bool masked = !(vm == ~0ull && kb * 64 + 63 <= q0);
        f32x4 sc[2][2], dp[2][2];
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                    sc[p][hf] = mma(E(), sl == 0 ? z : sc[p][hf], flash_k_frag(sK, p * 32 + ra + 4 * hf, sl * 32 + lq * 8), qf[sl]);
                    dp[p][hf] = mma(E(), sl == 0 ? z : dp[p][hf], flash_k_frag(sV, p * 32 + ra + 4 * hf, sl * 32 + lq * 8), dof[sl]);
                }
        E::Frag dsf[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            float ds[8];
            const unsigned mb = (unsigned)(vm >> (p * 32 + lq * 8)) & 0xffu;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[p][e >> 2][e & 3], LOG2E, nL2));
                if (masked) {                                     // SGPR condition
                    const int kk = kb * 64 + p * 32 + lq * 8 + e;
                    pr = ((kk <= qi) & (((mb >> e) & 1u) != 0)) ? pr : 0.f;
                }
                ds[e] = pr * (dp[p][e >> 2][e & 3] - Dq);
            }
            dsf[p] = make_frag(E(), ds);
        }
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int db = 0; db < 4; db++) dq[db] = mma(E(), dq[db], flash_t_frag(sK, p * 32, db, lr, lq), dsf[p]);
    }
    if (qi < T) {
#pragma unroll
        for (int db = 0; db < 4; db++) {
            const f32x4 v = dq[db] * 0.125f;
            if (dqb)       // bf16-matmul train mode: dqkv is only the dy operand of the c_attn backward products — written as that operand
                *reinterpret_cast<uint2 *>(dqb + ((long)b * T + qi) * ldb + h * 64 + db * 16 + lq * 4) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            else
                *reinterpret_cast<f32x4 *>(dqkv + ((long)b * T + qi) * 3 * d + h * 64 + db * 16 + lq * 4) = v;
        }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void flash_bwd2_dkv_bf16_kernel(const uint16_t *__restrict__ Qn, const uint16_t *__restrict__ Kn,
                                                                         const uint16_t *__restrict__ Vn, const uint16_t *__restrict__ dOn,
                                                                         const float *__restrict__ Dsum, const float *__restrict__ lse,
                                                                         const uint8_t *__restrict__ km, float *__restrict__ dqkv,
                                                                         uint16_t *__restrict__ dqb, long ldb, int BH, int H, int T, int Tp, int d) {
    typedef ElemBF16 E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLOT = 2 * E::TILE + 512, KT_ = 16 * NW;       // per slot: Q and dO tiles + 64 lse + 64 D floats
    constexpr float LOG2E = 1.4426950408889634f;
    const int nkt = (Tp + KT_ - 1) / KT_;
    int bh, kt;
    if (!flash_xcd_tile(BH, nkt, bh, kt)) return;                 // key tile 0 has the longest sweep: first
    const int b = bh / H, h = bh - b * H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lr = lane & 15, lq = lane >> 4;
    const int k0 = kt * KT_ + wave * 16, kj = k0 + lr, kc = kj < Tp ? kj : Tp - 1;
    const uint16_t *Qb = Qn + (long)bh * Tp * 64, *dOb = dOn + (long)bh * Tp * 64;
    const float *Lb = lse + (long)bh * Tp, *Db = Dsum + (long)bh * Tp;
    const int qb0 = kt * KT_ / 64, nqb = Tp / 64;
    auto issue = [&](int qb, char *slot) {
        flash_dma_tile<NW>(slot, Qb + (long)qb * 64 * 64, wave, lane);
        flash_dma_tile<NW>(slot + E::TILE, dOb + (long)qb * 64 * 64, wave, lane);
        if (wave == 0)          // 64 floats = one 4-byte-per-lane DMA
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Lb + qb * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(slot + 2 * E::TILE), 4, 0, 0);
        if (wave == 1)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Db + qb * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(slot + 2 * E::TILE + 256), 4, 0, 0);
    };
    issue(qb0, smem);
    E::Frag kf[2], vf[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        kf[s] = ld_frag_glb(E(), Kn + ((long)bh * Tp + kc) * 64 + s * 32 + lq * 8);
        vf[s] = ld_frag_glb(E(), Vn + ((long)bh * Tp + kc) * 64 + s * 32 + lq * 8);
    }
    const uint8_t kmv = km ? km[(long)b * T + (kj < T ? kj : T - 1)] : (uint8_t)1;
    const bool key_ok = (kj < T) & (kmv != 0);
    const bool keys_ok = __ballot(key_ok) == ~0ull;               // all 16 keys of this wave valid (each key appears in 4 lanes)
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = dk[i]; }
    const int ra = slab_row(lr);
    for (int qb = qb0; qb < nqb; qb++) {
        const int it = qb - qb0;
        const char *sQ = smem + (it & 1) * SLOT, *sdO = sQ + E::TILE;
        const float *sL = reinterpret_cast<const float *>(sQ + 2 * E::TILE), *sD = sL + 64;
        wait_vmcnt<0>();
        __syncthreads();
        if (qb + 1 < nqb) issue(qb + 1, smem + ((it + 1) & 1) * SLOT);
        if (qb * 64 + 63 < k0) continue;                          // every query of the block precedes this wave's keys (wave-uniform)
        // queries t >= T and queries without a valid key carry lse = +inf: P = 0 without a test; what is left to mask is the diagonal and invalid keys
        const bool masked = !(keys_ok && qb * 64 >= k0 + 15);
        f32x4 sc[2][2], dp[2][2];
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                    sc[p][hf] = mma(E(), sl == 0 ? z : sc[p][hf], flash_k_frag(sQ, p * 32 + ra + 4 * hf, sl * 32 + lq * 8), kf[sl]);
                    dp[p][hf] = mma(E(), sl == 0 ? z : dp[p][hf], flash_k_frag(sdO, p * 32 + ra + 4 * hf, sl * 32 + lq * 8), vf[sl]);
                }
        E::Frag pf[2], dsf[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            float pr[8], ds[8];
            const f32x4 l0 = *reinterpret_cast<const f32x4 *>(sL + p * 32 + lq * 8), l1 = *reinterpret_cast<const f32x4 *>(sL + p * 32 + lq * 8 + 4);
            const f32x4 d0 = *reinterpret_cast<const f32x4 *>(sD + p * 32 + lq * 8), d1 = *reinterpret_cast<const f32x4 *>(sD + p * 32 + lq * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float Lq = e < 4 ? l0[e & 3] : l1[e & 3], Dq = e < 4 ? d0[e & 3] : d1[e & 3];
                float x = __builtin_amdgcn_exp2f((sc[p][e >> 2][e & 3] - Lq) * LOG2E);
                if (masked) {                                     // SGPR condition
                    const int qq = qb * 64 + p * 32 + lq * 8 + e;
                    x = (key_ok & (qq >= kj)) ? x : 0.f;
                }
                pr[e] = x;
                ds[e] = x * (dp[p][e >> 2][e & 3] - Dq);
            }
            pf[p] = make_frag(E(), pr);
            dsf[p] = make_frag(E(), ds);
        }
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int db = 0; db < 4; db++) {
                dv[db] = mma(E(), dv[db], flash_t_frag(sdO, p * 32, db, lr, lq), pf[p]);
                dk[db] = mma(E(), dk[db], flash_t_frag(sQ, p * 32, db, lr, lq), dsf[p]);
            }
    }
    if (kj < T) {
#pragma unroll
        for (int db = 0; db < 4; db++) {
            if (dqb) {
                uint16_t *row = dqb + ((long)b * T + kj) * ldb + h * 64 + db * 16 + lq * 4;
                *reinterpret_cast<uint2 *>(row + d) = make_uint2(pack_bf16x2(dk[db][0], dk[db][1]), pack_bf16x2(dk[db][2], dk[db][3]));
                *reinterpret_cast<uint2 *>(row + 2 * d) = make_uint2(pack_bf16x2(dv[db][0], dv[db][1]), pack_bf16x2(dv[db][2], dv[db][3]));
            } else {
                float *row = dqkv + ((long)b * T + kj) * 3 * d + h * 64 + db * 16 + lq * 4;
                *reinterpret_cast<f32x4 *>(row + d) = dk[db];
                *reinterpret_cast<f32x4 *>(row + 2 * d) = dv[db];
            }
        }
    }
}

// which sweeps fetch the next tile through registers under the current tile's MFMAs — A/B per kernel and arithmetic mode on one box
// (tools/bench_flash_train.py under rocprofv3, B = 32, H = 12, T = 512 / 1024; plain -> prefetch): bf16 forward 75 -> 68 / 245 -> 212 us, bf16 dQ 108 -> 94 /
// 341 -> 292 us, fp32 forward 282 -> 255 / 965 -> 926 us, fp32 dQ 412 -> 373 / 1413 -> 1348 us; the dK/dV kernels (four tiles per step: the
// fetch registers cost them a wave of occupancy) are neutral in bf16 (168 -> 153 / 562 -> 582 us) and lose in fp32 (632 -> 658 / 2263 -> 2409 us)
// and keep the plain load.  -DLMRL_FLASH_PF=<mask> to A/B (bit 0 fwd, 1 dq, 2 dkv; bits 3-5 the same for fp32).
#ifndef LMRL_FLASH_PF
#define LMRL_FLASH_PF 0x1b
#endif
template <class E> struct FlashPrefetch;
template <> struct FlashPrefetch<ElemBF16> {
    static constexpr bool fwd = (LMRL_FLASH_PF & 1) != 0, dq = (LMRL_FLASH_PF & 2) != 0, dkv = (LMRL_FLASH_PF & 4) != 0;
};
template <> struct FlashPrefetch<ElemF32> {
    static constexpr bool fwd = (LMRL_FLASH_PF & 8) != 0, dq = (LMRL_FLASH_PF & 16) != 0, dkv = (LMRL_FLASH_PF & 32) != 0;
};

int g_flash_variant = 0;        // tools / tests only (lmrl_flash_set_variant): bit 0 = the round-3 forward sweep for bf16 operands
struct FlashWs {
    char *Qn, *Kn, *Vn, *QT, *KT, *VT, *dOn, *dOT;
    float *D;
    static size_t mat_bytes(int bh, int tp, int esz) { return ((size_t)bh * tp * 64 * esz + 255) & ~(size_t)255; }
    static size_t bytes(int bh, int tp, int esz) { return 8 * mat_bytes(bh, tp, esz) + (size_t)bh * tp * sizeof(float); }
    void carve(void *ws, int bh, int tp, int esz) {
        char *p = static_cast<char *>(ws);
        const size_t m = mat_bytes(bh, tp, esz);
        Qn = p; Kn = p + m; Vn = p + 2 * m; QT = p + 3 * m; KT = p + 4 * m; VT = p + 5 * m; dOn = p + 6 * m; dOT = p + 7 * m;
        D = reinterpret_cast<float *>(p + 8 * m);
    }
};

template <class K>
static hipError_t allow_lds(K kernel, size_t bytes) {
    return bytes > 64 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
                             : hipSuccess;
}

template <class E>
static int flash_stage_qkv(const float *qkv, const FlashWs &w, int batch, int heads, int t, int tp, hipStream_t s) {
    typedef typename E::T T;
    const int d = heads * 64;
    const dim3 grid(tp / 64, batch * heads);
    hipLaunchKernelGGL(flash_stage_kernel<E>, grid, dim3(256), 0, s, qkv, (long)3 * d, 0, 0.125f, (T *)w.Qn, (T *)w.QT, (const float *)nullptr,
                       (float *)nullptr, heads, t, tp);
    hipLaunchKernelGGL(flash_stage_kernel<E>, grid, dim3(256), 0, s, qkv, (long)3 * d, d, 1.f, (T *)w.Kn, (T *)w.KT, (const float *)nullptr,
                       (float *)nullptr, heads, t, tp);
    hipLaunchKernelGGL(flash_stage_kernel<E>, grid, dim3(256), 0, s, qkv, (long)3 * d, 2 * d, 1.f, (T *)w.Vn, (T *)w.VT, (const float *)nullptr,
                       (float *)nullptr, heads, t, tp);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

template <class E>
static int flash_fwd(const float *qkv, const uint8_t *km, float *att, float *lse, void *ws, int batch, int heads, int t, hipStream_t s,
                     void *att_b = nullptr, long ldb = 0) {
    typedef typename E::T T;
    const int tp = (t + 63) / 64 * 64, bh = batch * heads, d = heads * 64;
    FlashWs w; w.carve(ws, bh, tp, E::SZ);
    if (qkv) {             // null: the six q / k / v matrices are already staged in ws (lmrl_gemm_bf16_qkv_heads + lmrl_flash_attn_finish_staging)
        int rc = flash_stage_qkv<E>(qkv, w, batch, heads, t, tp, s);
        if (rc) return rc;
    }
    if constexpr (E::SZ == 2) {
        if (!(g_flash_variant & 1)) {
            const int rs = (g_flash_variant & 16) ? 3 : 2;                                // ring slots; tools: bit 4 = 3 slots (two blocks ahead: no faster)
            size_t lds2 = (size_t)rs * 2 * E::TILE + (size_t)tp;                          // + key validity bytes of one row
            const int nw = (g_flash_variant & 8) ? 4 : 8;                                 // 8-wave workgroups (128 queries); tools: bit 3 = 4 waves (64 queries: 3 % slower)
            const int nq2 = (tp + 16 * nw - 1) / (16 * nw);
            const int fwd2_grid = (bh / 8) * nq2 * 8 + ((bh % 8) * nq2 + 7) / 8 * 8;      // whole rounds of the 8 XCDs
#define LMRL_FWD2(NW_, ABL_)                                                                                                                     \
    hipLaunchKernelGGL((flash_fwd2_bf16_kernel<NW_, ABL_>), dim3(fwd2_grid), dim3(64 * NW_), lds2, s, (const uint16_t *)w.Qn, (const uint16_t *)w.Kn, \
                       (const uint16_t *)w.Vn, km, att, lse, bh, heads, t, tp, d, (uint16_t *)att_b, ldb)
#ifdef LMRL_TOOLS           /* timing-only ablations (tools/bench_flash_ablate.py): results are garbage; bits 16-23: extra (unused) LDS per workgroup, KB */
            lds2 += (size_t)((g_flash_variant >> 16) & 0xff) * 1024;
            LMRL_CHECK_HIP(allow_lds(flash_fwd2_bf16_kernel<8, 0>, lds2));
            switch ((g_flash_variant >> 8) & 0xff) {
                case 1: LMRL_FWD2(8, 1); break;
                case 2: LMRL_FWD2(8, 2); break;
                case 4: LMRL_FWD2(8, 4); break;
                case 7: LMRL_FWD2(8, 7); break;
                case 8: LMRL_FWD2(8, 8); break;
                case 24: LMRL_FWD2(8, 24); break;
                case 31: LMRL_FWD2(8, 31); break;
                default:
                    if (rs == 3) hipLaunchKernelGGL((flash_fwd2_bf16_kernel<8, 0, 3>), dim3(fwd2_grid), dim3(512), lds2, s, (const uint16_t *)w.Qn,
                                                    (const uint16_t *)w.Kn, (const uint16_t *)w.Vn, km, att, lse, bh, heads, t, tp, d, (uint16_t *)att_b, ldb);
                    else if (nw == 8) LMRL_FWD2(8, 0); else LMRL_FWD2(4, 0);
            }
#else
            if (rs == 3) hipLaunchKernelGGL((flash_fwd2_bf16_kernel<8, 0, 3>), dim3(fwd2_grid), dim3(512), lds2, s, (const uint16_t *)w.Qn, (const uint16_t *)w.Kn,
                                            (const uint16_t *)w.Vn, km, att, lse, bh, heads, t, tp, d, (uint16_t *)att_b, ldb);
            else if (nw == 8) LMRL_FWD2(8, 0); else LMRL_FWD2(4, 0);
#endif
#undef LMRL_FWD2
            LMRL_CHECK_LAUNCH();
            return LMRL_OK;
        }
    }
    const size_t lds = 2 * E::TILE + 64;
    constexpr bool PF = FlashPrefetch<E>::fwd;
    LMRL_CHECK_HIP(allow_lds(flash_fwd_kernel<E, PF>, lds));
    hipLaunchKernelGGL((flash_fwd_kernel<E, PF>), dim3(tp / 64, bh), dim3(256), lds, s, (const T *)w.Qn, (const T *)w.Kn, (const T *)w.VT, km, att, lse, heads,
                       t, tp, d, (uint16_t *)att_b, ldb);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

template <class E>
static int flash_bwd(const float *qkv, const uint8_t *km, const float *att, const float *datt, const float *lse, float *dqkv, uint16_t *dqb,
                     long ldb, void *ws, int batch, int heads, int t, hipStream_t s, int qkv_staged, const uint16_t *att_b = nullptr, long ld_attb = 0) {
    typedef typename E::T T;
    const int tp = (t + 63) / 64 * 64, bh = batch * heads, d = heads * 64;
    FlashWs w; w.carve(ws, bh, tp, E::SZ);
    if (!qkv_staged) {       // qkv_staged: ws still holds the six q / k / v matrices the forward of these same qkv staged
        int rc = flash_stage_qkv<E>(qkv, w, batch, heads, t, tp, s);
        if (rc) return rc;
    }
    if (E::SZ == 2 && qkv_staged && (g_flash_variant & 6)) {
        // a round-3 bf16 sweep is selected NOW: it reads Q^T / K^T, which the forward's staging wrote only if a round-3 bit was set THEN
        // (lmrl_flash_attn_finish_staging).  Derive them here from the natural forms, so a variant flipped between a forward and its backward
        // (tools / A-B tests) cannot make the sweeps read uninitialised transposes; the default path (variant 0) never gets here.
        hipLaunchKernelGGL(flash_transpose_staged_kernel, dim3(tp / 64, bh, 3), dim3(256), 0, s, (uint16_t *)w.Qn, (uint16_t *)w.QT,
                           (long)(FlashWs::mat_bytes(bh, tp, 2) / 2), t, tp);
    }
    const bool need_dot = E::SZ != 2 || (g_flash_variant & 6) != 0;         // fp32 sweeps and the round-3 bf16 dQ / dK/dV read dO^T
    hipLaunchKernelGGL(flash_stage_kernel<E>, dim3(tp / 64, bh), dim3(256), 0, s, datt, (long)d, 0, 1.f, (T *)w.dOn, need_dot ? (T *)w.dOT : (T *)nullptr, att, w.D,
                       heads, t, tp, att ? (const uint16_t *)nullptr : att_b, ld_attb);
    LMRL_CHECK_LAUNCH();
    const size_t lds_q = 3 * E::TILE + 64, lds_kv = 4 * E::TILE + 512;
    constexpr bool PFQ = FlashPrefetch<E>::dq, PFKV = FlashPrefetch<E>::dkv;
    constexpr bool BF = E::SZ == 2;
    constexpr int NW = 8;
    const int nt = (tp + 16 * NW - 1) / (16 * NW);
    if (BF && !(g_flash_variant & 2)) {
        const size_t lds2 = 4 * E::TILE;
        LMRL_CHECK_HIP(allow_lds(flash_bwd2_dq_bf16_kernel<NW>, lds2));
        hipLaunchKernelGGL(flash_bwd2_dq_bf16_kernel<NW>, dim3(flash_xcd_grid(bh, nt)), dim3(64 * NW), lds2, s, (const uint16_t *)w.Qn, (const uint16_t *)w.Kn,
                           (const uint16_t *)w.Vn, (const uint16_t *)w.dOn, (const float *)w.D, lse, km, dqkv, dqb, ldb, bh, heads, t, tp, d);
    } else {
        LMRL_CHECK_HIP(allow_lds(flash_bwd_dq_kernel<E, PFQ>, lds_q));
        hipLaunchKernelGGL((flash_bwd_dq_kernel<E, PFQ>), dim3(tp / 64, bh), dim3(256), lds_q, s, (const T *)w.Qn, (const T *)w.Kn, (const T *)w.Vn, (const T *)w.KT,
                           (const T *)w.dOn, (const float *)w.D, lse, km, dqkv, dqb, ldb, heads, t, tp, d);
    }
    if (BF && !(g_flash_variant & 4)) {
        const size_t lds2 = 2 * (2 * E::TILE + 512);
        LMRL_CHECK_HIP(allow_lds(flash_bwd2_dkv_bf16_kernel<NW>, lds2));
        hipLaunchKernelGGL(flash_bwd2_dkv_bf16_kernel<NW>, dim3(flash_xcd_grid(bh, nt)), dim3(64 * NW), lds2, s, (const uint16_t *)w.Qn, (const uint16_t *)w.Kn,
                           (const uint16_t *)w.Vn, (const uint16_t *)w.dOn, (const float *)w.D, lse, km, dqkv, dqb, ldb, bh, heads, t, tp, d);
    } else {
        LMRL_CHECK_HIP(allow_lds(flash_bwd_dkv_kernel<E, PFKV>, lds_kv));
        hipLaunchKernelGGL((flash_bwd_dkv_kernel<E, PFKV>), dim3(tp / 64, bh), dim3(256), lds_kv, s, (const T *)w.Qn, (const T *)w.Kn, (const T *)w.Vn,
                           (const T *)w.QT, (const T *)w.dOn, (const T *)w.dOT, (const float *)w.D, lse, km, dqkv, dqb, ldb, heads, t, tp, d);
    }
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

void lmrl_flash_set_variant(int v) { g_flash_variant = v; }
size_t lmrl_flash_attn_ws_bytes(int batch, int heads, int t, int bf16) {
    return FlashWs::bytes(batch * heads, (t + 63) / 64 * 64, bf16 ? 2 : 4);
}
size_t lmrl_flash_attn_lse_bytes(int batch, int heads, int t) { return (size_t)batch * heads * ((t + 63) / 64 * 64) * sizeof(float); }

int lmrl_flash_attn_stage_ptrs(void *ws_d, int batch, int heads, int t, void **q_heads_out, long *plane_elems_out) {
    LMRL_REQUIRE(ws_d && q_heads_out && plane_elems_out && batch > 0 && heads > 0 && t > 0, "lmrl_flash_attn_stage_ptrs: bad argument");
    const int tp = (t + 63) / 64 * 64;
    FlashWs w; w.carve(ws_d, batch * heads, tp, 2);
    *q_heads_out = w.Qn;
    *plane_elems_out = (long)(FlashWs::mat_bytes(batch * heads, tp, 2) / 2);
    return LMRL_OK;
}

int lmrl_flash_attn_finish_staging(void *ws_d, int batch, int heads, int t, void *stream) {
    LMRL_REQUIRE(ws_d && batch > 0 && heads > 0 && t > 0, "lmrl_flash_attn_finish_staging: bad argument");
    const int tp = (t + 63) / 64 * 64, bh = batch * heads;
    FlashWs w; w.carve(ws_d, bh, tp, 2);
    if (g_flash_variant & 7)        // tools / tests: a round-3 sweep is selected — it reads the transposed forms as well
        hipLaunchKernelGGL(flash_transpose_staged_kernel, dim3(tp / 64, bh, 3), dim3(256), 0, as_stream(stream), (uint16_t *)w.Qn, (uint16_t *)w.QT,
                           (long)(FlashWs::mat_bytes(bh, tp, 2) / 2), t, tp);
    else if (tp > t)
        hipLaunchKernelGGL(flash_zero_pad_rows_kernel, dim3(bh, 3), dim3(256), 0, as_stream(stream), (uint16_t *)w.Qn,
                           (long)(FlashWs::mat_bytes(bh, tp, 2) / 2), t, tp);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_flash_attn_fwd(const float *qkv_d, const uint8_t *key_mask_d, float *att_d, float *lse_d, void *ws_d, int batch, int heads, int t, int bf16,
                        void *stream) {
    LMRL_REQUIRE(qkv_d && att_d && lse_d && ws_d && batch > 0 && heads > 0 && t > 0, "lmrl_flash_attn_fwd: bad argument");
    return bf16 ? flash_fwd<ElemBF16>(qkv_d, key_mask_d, att_d, lse_d, ws_d, batch, heads, t, as_stream(stream))
                : flash_fwd<ElemF32>(qkv_d, key_mask_d, att_d, lse_d, ws_d, batch, heads, t, as_stream(stream));
}

int lmrl_flash_attn_fwd_staged(const float *qkv_d, const uint8_t *key_mask_d, float *att_d, float *lse_d, void *ws_d, void *att_bf16_d, long ldb, int batch,
                               int heads, int t, int bf16, void *stream) {
    LMRL_REQUIRE((qkv_d || bf16) && (att_d || (bf16 && !(g_flash_variant & 1))) && lse_d && ws_d && att_bf16_d && ldb >= heads * 64 && ldb % 4 == 0 && batch > 0 &&
                     heads > 0 && t > 0,
                 "lmrl_flash_attn_fwd_staged: bad argument");       // qkv_d null (bf16): q / k / v already staged in ws_d by the c_attn GEMM; att_d null (bf16): no fp32 output
    return bf16 ? flash_fwd<ElemBF16>(qkv_d, key_mask_d, att_d, lse_d, ws_d, batch, heads, t, as_stream(stream), att_bf16_d, ldb)
                : flash_fwd<ElemF32>(qkv_d, key_mask_d, att_d, lse_d, ws_d, batch, heads, t, as_stream(stream), att_bf16_d, ldb);
}

int lmrl_flash_attn_bwd(const float *qkv_d, const uint8_t *key_mask_d, const float *att_d, const float *datt_d, const float *lse_d, float *dqkv_d,
                        void *ws_d, int batch, int heads, int t, int bf16, int qkv_staged, void *stream) {
    LMRL_REQUIRE(qkv_d && att_d && datt_d && lse_d && dqkv_d && ws_d && batch > 0 && heads > 0 && t > 0, "lmrl_flash_attn_bwd: bad argument");
    return bf16 ? flash_bwd<ElemBF16>(qkv_d, key_mask_d, att_d, datt_d, lse_d, dqkv_d, nullptr, 0, ws_d, batch, heads, t, as_stream(stream), qkv_staged)
                : flash_bwd<ElemF32>(qkv_d, key_mask_d, att_d, datt_d, lse_d, dqkv_d, nullptr, 0, ws_d, batch, heads, t, as_stream(stream), qkv_staged);
}

int lmrl_flash_attn_bwd_staged(const float *qkv_d, const uint8_t *key_mask_d, const float *att_d, const float *datt_d, const float *lse_d,
                               void *dqkv_bf16_d, long ldb, void *ws_d, int batch, int heads, int t, int qkv_staged, void *stream) {
    LMRL_REQUIRE((qkv_d || qkv_staged) && att_d && datt_d && lse_d && dqkv_bf16_d && ws_d && ldb >= 3 * heads * 64 && ldb % 4 == 0 && batch > 0 &&
                     heads > 0 && t > 0, "lmrl_flash_attn_bwd_staged: bad argument");
    return flash_bwd<ElemBF16>(qkv_d, key_mask_d, att_d, datt_d, lse_d, nullptr, (uint16_t *)dqkv_bf16_d, ldb, ws_d, batch, heads, t, as_stream(stream),
                               qkv_staged);
}

// lmrl_flash_attn_bwd_staged with D = rowsum(dO o O) taken from the bf16 attention output the forward wrote for the output projection
// (att_bf16_d [batch * t][ld_att]): the forward then needs no fp32 copy of O at all (lmrl_flash_attn_fwd_staged with att_d = NULL)
int lmrl_flash_attn_bwd_staged_attb(const float *qkv_d, const uint8_t *key_mask_d, const void *att_bf16_d, long ld_att, const float *datt_d, const float *lse_d,
                                    void *dqkv_bf16_d, long ldb, void *ws_d, int batch, int heads, int t, int qkv_staged, void *stream) {
    LMRL_REQUIRE((qkv_d || qkv_staged) && att_bf16_d && ld_att >= heads * 64 && ld_att % 4 == 0 && datt_d && lse_d && dqkv_bf16_d && ws_d && ldb >= 3 * heads * 64 &&
                     ldb % 4 == 0 && batch > 0 && heads > 0 && t > 0, "lmrl_flash_attn_bwd_staged_attb: bad argument");
    return flash_bwd<ElemBF16>(qkv_d, key_mask_d, nullptr, datt_d, lse_d, nullptr, (uint16_t *)dqkv_bf16_d, ldb, ws_d, batch, heads, t, as_stream(stream),
                               qkv_staged, (const uint16_t *)att_bf16_d, ld_att);
}

}  // extern "C"
