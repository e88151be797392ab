// ppo_data.hip — the rollout -> PPO-data step on records that are already in HBM (gfx950).
//
// Reference: PPOInference.get_ppo_data_from_token_trajectory_chain (LLM_RL/algorithms/ppo/base_interface.py:464-669), fed by the task
// scripts' ppo_dataset_loader (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:301-353), then PPOData.block / PPODataset (ppo/data.py:9-114)
// and the masks the train step derives from a batch (base_interface.py:172-228).  There the rollout's TokenTrajectory arrays go through
// Python lists, per-chain numpy loops and a re-tokenisation; here the rollout engines leave `tokens / is_action / reward / n_tok / done`
// records in device memory and everything between them and the PPO batches stays there:
//
//   lmrl_ppo_count   per trajectory: effective length (Truncation.RIGHT at max_length), rows with a next token, action tokens; their
//                    exclusive scans (= where a trajectory's rows sit in the compacted row list / the KL list)
//   lmrl_ppo_block   block_sequences(Padding.RIGHT) + initialize_attn_mask_pos_ids + the row list / next-token targets of the LM head
//   lmrl_ppo_shape   log-ratio, KL terms, KL-shaped rewards, the chain-concatenated value / reward / mask rows of the GAE (with the
//                    bootstrap slot, :554-570) and the token-aligned PPOData rows (ids, should_take_action, old_logprobs, old_values)
//   lmrl_ppo_unroll  advantages / returns from the chain rows back to the trajectories' PPOData rows (unroll_arr, :335-343)
//   lmrl_seq_mask_pos, lmrl_masked_rows, lmrl_gather_rows_bytes   what GPT2PPOTrain.step needs of a batch that never visits the host
//
// (lmrl_gae / lmrl_whiten_* in rl_reduce.hip run between shape and unroll.)  Mapping: one 64-lane wave per trajectory; compaction ranks by
// __ballot / popcount; every array row of a wave is contiguous, so a wave instruction reads / writes one or two full cache lines.
// All of it is HBM-bound and tiny next to the two forward passes it sits between (a few MB per 1024 trajectories).
#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {

__device__ __forceinline__ int lanes_below(unsigned long long bal, int lane) { return __popcll(bal & ((1ull << lane) - 1ull)); }

__device__ __forceinline__ int traj_len(const int32_t *__restrict__ n_tok, int k, int cap, int max_len) {
    int n = n_tok[k];
    n = n < 0 ? 0 : (n > cap ? cap : n);
    return (max_len > 0 && n > max_len) ? max_len : n;            // Truncation.RIGHT at max_length (base_interface.py:500-512)
}

// meta: [0] rows with a next token (sum of max(len - 1, 0)), [1] action tokens, [2] longest effective length, [3] pad ids found below a length,
// [4] action tokens CUT by max_len (is_action[1:][max_len - 1:], the reference's 'trajectory truncation error', base_interface.py:318-327),
// [5] trajectories that continue a chain but start with an action token (the same assert's other arm), [6] the longest chain concatenation
// (max over trajectories of pos + len - 1: what `lc` of lmrl_ppo_shape / lmrl_ppo_unroll must reach), [7] reserved (0)
__global__ __launch_bounds__(256) void ppo_count_kernel(const int32_t *__restrict__ tokens, const uint8_t *__restrict__ is_action,
                                                        const int32_t *__restrict__ n_tok, const int32_t *__restrict__ chain,
                                                        const int32_t *__restrict__ pos, int n, int cap, int max_len, int pad,
                                                        int32_t *__restrict__ cnt_rows, int32_t *__restrict__ cnt_act, int32_t *__restrict__ meta) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wave;
    if (k >= n) return;
    const int ne = traj_len(n_tok, k, cap, max_len);
    const int nraw = traj_len(n_tok, k, cap, 0);
    int acts = 0, pads = 0, cut = 0;
    for (int base = 0; base < nraw; base += 64) {
        const int t = base + lane;
        const bool in = t < ne;
        const bool act = t < nraw && t >= 1 && is_action[(size_t)k * cap + t] != 0;      // should_take_action = is_action[1:]
        const bool a = in && act;
        const bool p = in && tokens[(size_t)k * cap + t] == pad;
        acts += __popcll(__ballot(a));
        pads += __popcll(__ballot(p));
        cut += __popcll(__ballot(act && !in));
    }
    if (lane == 0) {
        cnt_rows[k] = ne > 0 ? ne - 1 : 0;
        cnt_act[k] = acts;
        atomicMax(&meta[2], ne);
        if (pads) atomicAdd(&meta[3], pads);
        if (cut) atomicAdd(&meta[4], cut);
        if (chain && k > 0 && chain[k] == chain[k - 1] && nraw > 0 && is_action[(size_t)k * cap] != 0) atomicAdd(&meta[5], 1);
        atomicMax(&meta[6], (pos ? pos[k] : 0) + (ne > 0 ? ne - 1 : 0));
    }
}

// exclusive scans of a (and b, optional) over n entries by ONE workgroup: oa / ob have n + 1 entries (the last = the total); meta[0] / meta[1] = totals
__global__ __launch_bounds__(1024) void scan2_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int n, int32_t *__restrict__ oa,
                                                     int32_t *__restrict__ ob, int32_t *__restrict__ meta) {
    __shared__ int sa[16], sb[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int carry_a = 0, carry_b = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int va = i < n ? a[i] : 0, vb = (b && i < n) ? b[i] : 0;
        int xa = va, xb = vb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ta = __shfl_up(xa, d), tb = __shfl_up(xb, d);
            if (lane >= d) { xa += ta; xb += tb; }
        }
        if (lane == 63) { sa[wave] = xa; sb[wave] = xb; }
        __syncthreads();
        int wa = 0, wb = 0, ta = 0, tb = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            if (w < wave) { wa += sa[w]; wb += sb[w]; }
            ta += sa[w]; tb += sb[w];
        }
        if (i < n) {
            oa[i] = carry_a + wa + xa - va;
            if (b) ob[i] = carry_b + wb + xb - vb;
        }
        carry_a += ta; carry_b += tb;
        __syncthreads();
    }
    if (tid == 0) {
        oa[n] = carry_a;
        if (b) ob[n] = carry_b;
        if (meta) { meta[0] = carry_a; if (b) meta[1] = carry_b; }
    }
}

__global__ __launch_bounds__(256) void ppo_block_kernel(const int32_t *__restrict__ tokens, const int32_t *__restrict__ n_tok, int n, int cap, int max_len,
                                                        int pad, int tf, const int32_t *__restrict__ off_rows, int32_t *__restrict__ ids,
                                                        uint8_t *__restrict__ am, int32_t *__restrict__ pos, int32_t *__restrict__ rows_idx,
                                                        int32_t *__restrict__ tgt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wave;
    if (k >= n) return;
    const int ne = min(traj_len(n_tok, k, cap, max_len), tf);
    const int roff = off_rows[k];
    const int32_t *trow = tokens + (size_t)k * cap;
    for (int base = 0; base < tf; base += 64) {
        const int t = base + lane;
        if (t >= tf) break;
        const bool in = t < ne;
        const size_t o = (size_t)k * tf + t;
        ids[o] = in ? trow[t] : pad;
        am[o] = in ? 1 : 0;
        pos[o] = in ? t : (ne > 0 ? ne - 1 : 0);           // clip(cumsum(mask) - 1, 0) on a right-padded row
        if (t + 1 < ne) {                                  // row (k, t) predicts token t + 1 (token_logprobs_from_logits, :396-403)
            rows_idx[roff + t] = k * tf + t;
            tgt[roff + t] = trow[t + 1];
        }
    }
}

struct PpoRecords {               // lmrl_ppo_records, device pointers
    const int32_t *tokens; const uint8_t *is_action; const float *reward; const int32_t *n_tok; const uint8_t *done;
    const int32_t *chain, *pos; const uint8_t *last;
    int n, cap, n_chains;
};

__global__ __launch_bounds__(256) void ppo_shape_kernel(PpoRecords r, int max_len, int tf, const int32_t *__restrict__ off_rows,
                                                        const int32_t *__restrict__ off_act, const float *__restrict__ lp,
                                                        const float *__restrict__ init_lp, const float *__restrict__ values, float kl_weight, int lc,
                                                        float *__restrict__ cv, float *__restrict__ cr, uint8_t *__restrict__ cs,
                                                        int32_t *__restrict__ chain_len, float *__restrict__ kls, int pad, int tp,
                                                        int32_t *__restrict__ ds_ids, uint8_t *__restrict__ ds_sta, float *__restrict__ ds_lp,
                                                        float *__restrict__ ds_val) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wave;
    if (k >= r.n) return;
    const int ne = min(traj_len(r.n_tok, k, r.cap, max_len), tf);
    const int L = ne > 0 ? ne - 1 : 0;
    const int c = r.chain ? r.chain[k] : k, p0 = r.pos ? r.pos[k] : 0;
    const bool is_last = r.last ? r.last[k] != 0 : true;
    const int roff = off_rows[k], koff = off_act[k];
    const int32_t *trow = r.tokens + (size_t)k * r.cap;
    const uint8_t *arow = r.is_action + (size_t)k * r.cap;
    const float *wrow = r.reward + (size_t)k * r.cap;
    int seen = 0;
    for (int base = 0; base < tp; base += 64) {
        const int t = base + lane;
        if (t < tp) ds_ids[(size_t)k * tp + t] = t < ne ? trow[t] : pad;
        const bool in = t < L;
        const bool a = in && arow[t + 1] != 0;
        const float l = in ? lp[roff + t] : 0.f, li = in ? init_lp[roff + t] : 0.f;
        const float v = in ? values[(size_t)k * tf + t] : 0.f;
        const float lr = a ? l - li : 0.f;                               // (logprobs - initial_policy_logprobs) * should_take_action, :573-575
        if (t < tp - 1) {
            const size_t o = (size_t)k * (tp - 1) + t;
            ds_sta[o] = a ? 1 : 0;
            ds_lp[o] = l;
            ds_val[o] = v;
        }
        if (in && p0 + t < lc) {
            const float kl_term = kl_weight * lr;
            cv[(size_t)c * (lc + 1) + p0 + t] = v;
            cr[(size_t)c * lc + p0 + t] = wrow[t + 1] - kl_term;          // rewards - kl_weight * log_ratio, :580-584
            cs[(size_t)c * lc + p0 + t] = a ? 1 : 0;
        }
        const unsigned long long bal = __ballot(a);
        if (a) kls[koff + seen + lanes_below(bal, lane)] = expf(lr) - 1.f - lr;   // all_kls over np.argwhere(should_take_action), :576-579
        seen += __popcll(bal);
    }
    if (is_last && lane == 0) {
        // the chain's bootstrap slot: value of the last token of its last trajectory, zeroed when the episode is done (:554-570)
        const float lv = ne > 0 ? values[(size_t)k * tf + ne - 1] : 0.f;
        const int end = min(p0 + L, lc);
        cv[(size_t)c * (lc + 1) + end] = lv * (1.f - (r.done[c] ? 1.f : 0.f));
        chain_len[c] = end;
    }
}

__global__ __launch_bounds__(256) void ppo_unroll_kernel(PpoRecords r, int max_len, int tf, int lc, const float *__restrict__ cadv,
                                                         const float *__restrict__ cret, int tp, float *__restrict__ ds_adv, float *__restrict__ ds_ret) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wave;
    if (k >= r.n) return;
    const int ne = min(traj_len(r.n_tok, k, r.cap, max_len), tf);
    const int L = ne > 0 ? ne - 1 : 0;
    const int c = r.chain ? r.chain[k] : k, p0 = r.pos ? r.pos[k] : 0;
    for (int t = lane; t < tp - 1; t += 64) {
        const bool in = t < L && p0 + t < lc;
        const size_t o = (size_t)k * (tp - 1) + t;
        ds_adv[o] = in ? cadv[(size_t)c * lc + p0 + t] : 0.f;
        ds_ret[o] = in ? cret[(size_t)c * lc + p0 + t] : 0.f;
    }
}

// ---- what a train step needs of a device-resident batch
__global__ __launch_bounds__(256) void seq_mask_pos_kernel(const int32_t *__restrict__ ids, int pad, uint8_t *__restrict__ am, int32_t *__restrict__ pos,
                                                           float *__restrict__ am_next, int b, int t) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= b) return;
    int carry = 0;
    for (int base = 0; base < t; base += 64) {
        const int i = base + lane;
        const bool m = i < t && ids[(size_t)row * t + i] != pad;
        const unsigned long long bal = __ballot(m);
        if (i < t) {
            const int cum = carry + lanes_below(bal, lane) + (m ? 1 : 0);      // inclusive cumsum of the mask
            am[(size_t)row * t + i] = m ? 1 : 0;
            pos[(size_t)row * t + i] = cum > 0 ? cum - 1 : 0;
            if (am_next && i >= 1) am_next[(size_t)row * (t - 1) + i - 1] = m ? 1.f : 0.f;      // attention_mask[:, 1:] as the loss reads it
        }
        carry += __popcll(bal);
    }
}

// right-padded rows of known lengths: am[b][i] = i < len[b], pos = clip(cumsum(am) - 1, 0) — the masks a data build made, carried to the train step
__global__ __launch_bounds__(256) void len_mask_pos_kernel(const int32_t *__restrict__ len, int b, int t, uint8_t *__restrict__ am, int32_t *__restrict__ pos) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= b) return;
    int n = len[row];
    n = n < 0 ? 0 : (n > t ? t : n);
    for (int i = lane; i < t; i += 64) {
        am[(size_t)row * t + i] = i < n ? 1 : 0;
        pos[(size_t)row * t + i] = i < n ? i : (n > 0 ? n - 1 : 0);
    }
}

__global__ void add_i32_kernel(const int32_t *__restrict__ in, int c, int32_t *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + c;
}

// mask[b][t] = sta[b][t] && am[b][t + 1] on the shifted grid (t < T - 1): count per row, then (after the scan) the row list + targets
template <bool WRITE>
__global__ __launch_bounds__(256) void masked_rows_kernel(const uint8_t *__restrict__ sta, const uint8_t *__restrict__ am, const int32_t *__restrict__ ids,
                                                          int b, int t, int32_t *__restrict__ cnt, const int32_t *__restrict__ off,
                                                          int32_t *__restrict__ idx, int32_t *__restrict__ tgt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= b) return;
    int seen = 0;
    const int o0 = WRITE ? off[row] : 0;
    for (int base = 0; base < t - 1; base += 64) {
        const int i = base + lane;
        const bool m = i < t - 1 && sta[(size_t)row * (t - 1) + i] != 0 && (!am || am[(size_t)row * t + i + 1] != 0);
        const unsigned long long bal = __ballot(m);
        if (WRITE && m) {
            const int o = o0 + seen + lanes_below(bal, lane);
            idx[o] = row * t + i;
            if (tgt) tgt[o] = ids[(size_t)row * t + i + 1];
        }
        seen += __popcll(bal);
    }
    if (!WRITE && lane == 0) cnt[row] = seen;
}

__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const uint8_t *__restrict__ src, const int32_t *__restrict__ idx, uint8_t *__restrict__ dst,
                                                                int n, long row_bytes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const uint8_t *s = src + (size_t)idx[i] * row_bytes;
    uint8_t *d = dst + (size_t)i * row_bytes;
    if (row_bytes % 4 == 0 && ((uintptr_t)s % 4 == 0) && ((uintptr_t)d % 4 == 0)) {
        for (long j = lane; j < row_bytes / 4; j += 64) reinterpret_cast<uint32_t *>(d)[j] = reinterpret_cast<const uint32_t *>(s)[j];
    } else {
        for (long j = lane; j < row_bytes; j += 64) d[j] = s[j];
    }
}

// The task scripts' length rule on single-trajectory chains (llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py:323-341): while a trajectory has more than
// three texts and its tokenisation reaches max_len, its last two texts (action, observation) are dropped, their rewards x gamma are folded into the
// previous action and the chain is no longer done; trajectories left with fewer than three texts, or still too long, are skipped.  On a token
// record a "text" is a maximal run of equal is_action flags (header, action, observation, action, ... alternate), an item's reward sits on its
// last token (TokenTrajectory.from_text_trajectory, environment.py:349-380).  One lane per trajectory: a few short backward scans over <= cap
// flags.  The fold runs in double, as the script's Python floats do, and is rounded to float32 once (np.array(reward, dtype=np.float32)).
__global__ __launch_bounds__(256) void ppo_truncate_turns_kernel(const uint8_t *__restrict__ is_action, const int32_t *__restrict__ n_tok,
                                                                 const uint8_t *__restrict__ done, int n, int cap, int max_len, double gamma,
                                                                 float *__restrict__ reward, int32_t *__restrict__ n_tok_out, uint8_t *__restrict__ done_out,
                                                                 uint8_t *__restrict__ keep, int32_t *__restrict__ meta) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint8_t *ia = is_action + (size_t)k * cap;
    float *rw = reward + (size_t)k * cap;
    int nt = n_tok[k];
    nt = nt < 0 ? 0 : (nt > cap ? cap : nt);
    int items = nt > 0 ? 1 : 0;
    for (int t = 1; t < nt; t++) items += (ia[t] != 0) != (ia[t - 1] != 0);
    bool dn = done[k] != 0;
    int target = -1, dropped = 0;      // token whose reward is held in `carry` (double) instead of rw[]
    double carry = 0.0;
    while (items > 3 && nt >= max_len) {
        int s1 = nt - 1;
        while (s1 > 0 && (ia[s1 - 1] != 0) == (ia[nt - 1] != 0)) s1--;          // last text = [s1, nt)
        int s2 = s1 - 1;
        while (s2 > 0 && (ia[s2 - 1] != 0) == (ia[s1 - 1] != 0)) s2--;          // the one before = [s2, s1)
        double fold = 0.0;                                                       // sum(reward[-2:]) in the script's order
        for (int t = s2; t < nt; t++) fold += (t == target) ? carry : (double)rw[t];
        if (target >= s2) target = -1;                                           // the held token leaves with its text
        nt = s2; items -= 2; dn = false; dropped += 2;
        int t1 = nt - 1;
        while (t1 > 0 && (ia[t1 - 1] != 0) == (ia[nt - 1] != 0)) t1--;           // new last text = [t1, nt); new_reward[-2] sits on token t1 - 1
        const int tg = t1 - 1;
        if (tg >= 0) {
            if (target >= 0 && target != tg) rw[target] = (float)carry;
            carry = ((tg == target) ? carry : (double)rw[tg]) + fold * gamma;
            target = tg;
        }
    }
    if (target >= 0) rw[target] = (float)carry;
    const bool kp = items >= 3 && nt < max_len;
    n_tok_out[k] = nt;
    done_out[k] = dn ? 1 : 0;
    keep[k] = kp ? 1 : 0;
    if (dropped) atomicAdd(&meta[0], 1);
    if (!kp) atomicAdd(&meta[1], 1);
}

// idx[j] = the j-th k with flags[k] != 0 (increasing), count[0] = their number; one workgroup (n up to a few 100 k)
__global__ __launch_bounds__(1024) void compact_flags_kernel(const uint8_t *__restrict__ flags, int n, int32_t *__restrict__ idx, int32_t *__restrict__ count) {
    __shared__ int sw[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const bool f = i < n && flags[i] != 0;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) sw[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { if (w < wave) before += sw[w]; total += sw[w]; }
        if (f) idx[carry + before + lanes_below(bal, lane)] = i;
        carry += total;
        __syncthreads();
    }
    if (tid == 0) count[0] = carry;
}

static PpoRecords records_of(const lmrl_ppo_records *r) {
    return PpoRecords{r->tokens, r->is_action, r->reward, r->n_tok, r->done, r->chain, r->pos, r->last, r->n, r->cap, r->n_chains};
}

static bool records_ok(const lmrl_ppo_records *r) {
    return r && r->tokens && r->is_action && r->reward && r->n_tok && r->n > 0 && r->cap > 0 && r->n_chains > 0 && r->n_chains <= r->n;
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

int lmrl_ppo_count(const lmrl_ppo_records *rec, int max_len, int pad, int32_t *cnt_d, int32_t *off_rows_d, int32_t *off_act_d, int32_t *meta_d,
                   void *stream) {
    LMRL_REQUIRE(records_ok(rec) && cnt_d && off_rows_d && off_act_d && meta_d, "lmrl_ppo_count: bad argument");
    hipStream_t s = as_stream(stream);
    LMRL_CHECK_HIP(hipMemsetAsync(meta_d, 0, 8 * sizeof(int32_t), s));
    hipLaunchKernelGGL(ppo_count_kernel, dim3(ceil_div(rec->n, 4)), dim3(256), 0, s, rec->tokens, rec->is_action, rec->n_tok, rec->chain, rec->pos, rec->n,
                       rec->cap, max_len, pad, cnt_d, cnt_d + rec->n, meta_d);
    hipLaunchKernelGGL(scan2_kernel, dim3(1), dim3(1024), 0, s, cnt_d, cnt_d + rec->n, rec->n, off_rows_d, off_act_d, meta_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ppo_block(const lmrl_ppo_records *rec, int max_len, int pad, int tf, const int32_t *off_rows_d, int32_t *ids_d, uint8_t *am_d, int32_t *pos_d,
                   int32_t *rows_idx_d, int32_t *tgt_d, void *stream) {
    LMRL_REQUIRE(records_ok(rec) && tf > 0 && off_rows_d && ids_d && am_d && pos_d && rows_idx_d && tgt_d, "lmrl_ppo_block: bad argument");
    hipLaunchKernelGGL(ppo_block_kernel, dim3(ceil_div(rec->n, 4)), dim3(256), 0, as_stream(stream), rec->tokens, rec->n_tok, rec->n, rec->cap, max_len, pad,
                       tf, off_rows_d, ids_d, am_d, pos_d, rows_idx_d, tgt_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ppo_shape(const lmrl_ppo_records *rec, int max_len, int tf, const int32_t *off_rows_d, const int32_t *off_act_d, const float *logprobs_d,
                   const float *init_logprobs_d, const float *values_d, float kl_weight, int lc, float *chain_values_d, float *chain_rewards_d,
                   uint8_t *chain_sta_d, int32_t *chain_len_d, float *kls_d, int pad, int tp, int32_t *ds_ids_d, uint8_t *ds_sta_d, float *ds_logprobs_d,
                   float *ds_values_d, void *stream) {
    LMRL_REQUIRE(records_ok(rec) && rec->done && tf > 0 && tp >= 2 && lc > 0 && off_rows_d && off_act_d && logprobs_d && init_logprobs_d && values_d &&
                     chain_values_d && chain_rewards_d && chain_sta_d && chain_len_d && kls_d && ds_ids_d && ds_sta_d && ds_logprobs_d && ds_values_d,
                 "lmrl_ppo_shape: bad argument");
    hipStream_t s = as_stream(stream);
    // positions past a chain's length: mask 0 (the whitening's mask), values / rewards 0
    LMRL_CHECK_HIP(hipMemsetAsync(chain_sta_d, 0, (size_t)rec->n_chains * lc, s));
    LMRL_CHECK_HIP(hipMemsetAsync(chain_values_d, 0, (size_t)rec->n_chains * (lc + 1) * sizeof(float), s));
    LMRL_CHECK_HIP(hipMemsetAsync(chain_rewards_d, 0, (size_t)rec->n_chains * lc * sizeof(float), s));
    LMRL_CHECK_HIP(hipMemsetAsync(chain_len_d, 0, (size_t)rec->n_chains * sizeof(int32_t), s));
    hipLaunchKernelGGL(ppo_shape_kernel, dim3(ceil_div(rec->n, 4)), dim3(256), 0, s, records_of(rec), max_len, tf, off_rows_d, off_act_d, logprobs_d,
                       init_logprobs_d, values_d, kl_weight, lc, chain_values_d, chain_rewards_d, chain_sta_d, chain_len_d, kls_d, pad, tp, ds_ids_d, ds_sta_d,
                       ds_logprobs_d, ds_values_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ppo_unroll(const lmrl_ppo_records *rec, int max_len, int tf, int lc, const float *chain_adv_d, const float *chain_ret_d, int tp, float *ds_adv_d,
                    float *ds_ret_d, void *stream) {
    LMRL_REQUIRE(records_ok(rec) && tf > 0 && tp >= 2 && lc > 0 && chain_adv_d && chain_ret_d && ds_adv_d && ds_ret_d, "lmrl_ppo_unroll: bad argument");
    hipLaunchKernelGGL(ppo_unroll_kernel, dim3(ceil_div(rec->n, 4)), dim3(256), 0, as_stream(stream), records_of(rec), max_len, tf, lc, chain_adv_d, chain_ret_d,
                       tp, ds_adv_d, ds_ret_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_ppo_truncate_turns(const uint8_t *is_action_d, const int32_t *n_tok_d, const uint8_t *done_d, int n, int cap, int max_len, double gamma,
                            float *reward_d, int32_t *n_tok_out_d, uint8_t *done_out_d, uint8_t *keep_d, int32_t *meta_d, void *stream) {
    LMRL_REQUIRE(is_action_d && n_tok_d && done_d && reward_d && n_tok_out_d && done_out_d && keep_d && meta_d && n > 0 && cap > 0 && max_len > 0,
                 "lmrl_ppo_truncate_turns: bad argument");
    hipStream_t s = as_stream(stream);
    LMRL_CHECK_HIP(hipMemsetAsync(meta_d, 0, 2 * sizeof(int32_t), s));
    hipLaunchKernelGGL(ppo_truncate_turns_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, is_action_d, n_tok_d, done_d, n, cap, max_len, gamma, reward_d,
                       n_tok_out_d, done_out_d, keep_d, meta_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_compact_flags(const uint8_t *flags_d, int n, int32_t *idx_d, int32_t *count_d, void *stream) {
    LMRL_REQUIRE(flags_d && idx_d && count_d && n > 0, "lmrl_compact_flags: bad argument");
    hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, as_stream(stream), flags_d, n, idx_d, count_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_seq_mask_pos(const int32_t *ids_d, int pad, uint8_t *am_d, int32_t *pos_d, float *am_next_f32_d, int b, int t, void *stream) {
    LMRL_REQUIRE(ids_d && am_d && pos_d && b > 0 && t > 0, "lmrl_seq_mask_pos: bad argument");
    hipLaunchKernelGGL(seq_mask_pos_kernel, dim3(ceil_div(b, 4)), dim3(256), 0, as_stream(stream), ids_d, pad, am_d, pos_d, am_next_f32_d, b, t);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_len_mask_pos(const int32_t *len_d, int b, int t, uint8_t *am_d, int32_t *pos_d, void *stream) {
    LMRL_REQUIRE(len_d && am_d && pos_d && b > 0 && t > 0, "lmrl_len_mask_pos: bad argument");
    hipLaunchKernelGGL(len_mask_pos_kernel, dim3(ceil_div(b, 4)), dim3(256), 0, as_stream(stream), len_d, b, t, am_d, pos_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_add_i32(const int32_t *in_d, int c, int32_t *out_d, int n, void *stream) {
    LMRL_REQUIRE(in_d && out_d && n > 0, "lmrl_add_i32: bad argument");
    hipLaunchKernelGGL(add_i32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), in_d, c, out_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_masked_rows(const uint8_t *sta_d, const uint8_t *am_d, const int32_t *ids_d, int b, int t, int32_t *cnt_d, int32_t *off_d, int32_t *idx_d,
                     int32_t *tgt_d, void *stream) {
    LMRL_REQUIRE(sta_d && b > 0 && t >= 2 && cnt_d && off_d && idx_d && (!tgt_d || ids_d), "lmrl_masked_rows: bad argument");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(masked_rows_kernel<false>, dim3(ceil_div(b, 4)), dim3(256), 0, s, sta_d, am_d, ids_d, b, t, cnt_d, (const int32_t *)nullptr,
                       (int32_t *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(scan2_kernel, dim3(1), dim3(1024), 0, s, cnt_d, (const int32_t *)nullptr, b, off_d, (int32_t *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(masked_rows_kernel<true>, dim3(ceil_div(b, 4)), dim3(256), 0, s, sta_d, am_d, ids_d, b, t, (int32_t *)nullptr, off_d, idx_d, tgt_d);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_exclusive_scan_i32(const int32_t *in_d, int32_t *out_d, int n, void *stream) {
    LMRL_REQUIRE(in_d && out_d && n > 0, "lmrl_exclusive_scan_i32: bad argument");
    hipLaunchKernelGGL(scan2_kernel, dim3(1), dim3(1024), 0, as_stream(stream), in_d, (const int32_t *)nullptr, n, out_d, (int32_t *)nullptr, (int32_t *)nullptr);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_gather_rows_bytes(const void *src_d, const int32_t *idx_d, void *dst_d, int n, long row_bytes, void *stream) {
    LMRL_REQUIRE(src_d && idx_d && dst_d && n >= 0 && row_bytes > 0, "lmrl_gather_rows_bytes: bad argument");
    if (n == 0) return LMRL_OK;
    hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, as_stream(stream), (const uint8_t *)src_d, idx_d, (uint8_t *)dst_d, n,
                       row_bytes);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

}  // extern "C"
