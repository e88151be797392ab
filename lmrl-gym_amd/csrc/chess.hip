// chess.hip — batched chess env stepping (one game per lane) and the host faces of the same rules (chess_rules.h).
//
// Replaces python-chess inside the reference's chess env (llm_rl_scripts/chess/env/env.py:28-185): Board(fen), push_san, san, fen,
// is_checkmate, is_game_over.  The opponent (Stockfish over UCI, env.py:157-170) stays a host process pool — it thinks 100 ms per move — and
// hands its move back in UCI form; both half-steps of `ChessEnv.step` run here for all games at once, including the SAN of the opponent's
// move and the FEN observation.  The game is engine-bound, not kernel-bound: these kernels exist so that the board never leaves the device
// between the policy's generated tokens and the next prompt, and so that 4096 boards step in one launch; they are not tuned further.
#include "../../include/lmrl_amd.h"
#include "common.h"
#include "chess_rules.h"
#include <string.h>

using namespace lmrl_chess;

namespace lmrl {

constexpr int kFen = LMRL_CHESS_FEN_BYTES, kAct = LMRL_CHESS_ACTION_BYTES;

__device__ __host__ inline int cstr_len(const char *s, int cap) {
    int n = 0;
    while (n < cap && s[n]) n++;
    return n;
}

__global__ void chess_reset_kernel(Pos *pos, const char *fens, uint8_t *ok, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    ok[e] = parse_fen(fens + (size_t)e * kFen, pos[e]) ? 1 : 0;
}

__global__ void chess_agent_step_kernel(Pos *pos, const char *actions, const uint8_t *active, float *reward, uint8_t *done, uint8_t *result,
                                        char *fen_out, char *uci_out, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (active && !active[e]) { result[e] = 255; return; }
    const char *a = actions + (size_t)e * kAct;
    float r; int d;
    const int res = agent_half_step(pos[e], a, cstr_len(a, kAct), &r, &d, uci_out ? uci_out + (size_t)e * 8 : nullptr);
    reward[e] = r; done[e] = (uint8_t)d; result[e] = (uint8_t)res;
    fen(pos[e], fen_out + (size_t)e * kFen);
}

__global__ void chess_opponent_step_kernel(Pos *pos, const char *ucis, const uint8_t *active, float *reward, uint8_t *done, uint8_t *ok,
                                           char *san_out, char *fen_out, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (active && !active[e]) { ok[e] = 255; return; }
    const char *u = ucis + (size_t)e * 8;
    float r = 0.f; int d = 0;
    ok[e] = opponent_half_step(pos[e], u, cstr_len(u, 8), san_out + (size_t)e * kAct, &r, &d) ? 1 : 0;
    reward[e] = r; done[e] = (uint8_t)d;
    fen(pos[e], fen_out + (size_t)e * kFen);
}

// board.legal_moves / board.san / board.fen / is_check ... for every game at once: what the env's random opponent, the move-accuracy
// evaluators and the parity tests read.  One game per lane; the move list goes out as fixed-pitch strings.
__global__ void chess_describe_kernel(const Pos *pos, char *uci_out, char *san_out, int32_t *count, uint8_t *status, char *fen_out, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const Pos &p = pos[e];
    Move mv[kMaxMoves];
    const int k = gen_legal(p, mv);
    if (count) count[e] = k;
    for (int i = 0; i < k; i++) {
        if (uci_out) uci(mv[i], uci_out + ((size_t)e * kMaxMoves + i) * 8);
        if (san_out) san(p, mv[i], san_out + ((size_t)e * kMaxMoves + i) * kAct);
    }
    if (status) {
        const bool chk = in_check(p, p.stm);
        status[e] = (uint8_t)((chk ? 1 : 0) | ((chk && !k) ? 2 : 0) | (is_game_over(p) ? 4 : 0) | (is_insufficient_material(p) ? 8 : 0) |
                              ((!chk && !k) ? 16 : 0) | (is_repetition(p, 5) ? 32 : 0) | (p.halfmove >= 150 ? 64 : 0));
    }
    if (fen_out) fen(p, fen_out + (size_t)e * kFen);
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

size_t lmrl_chess_pos_bytes(void) { return sizeof(Pos); }

int lmrl_chess_reset(void *pos_d, const char *fens_d, uint8_t *ok_d, int n, void *stream) {
    LMRL_REQUIRE(pos_d && fens_d && ok_d && n > 0, "lmrl_chess_reset: bad argument");
    hipLaunchKernelGGL(chess_reset_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream), (Pos *)pos_d, fens_d, ok_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_chess_agent_step(void *pos_d, const char *actions_d, const uint8_t *active_d, float *reward_d, uint8_t *done_d, uint8_t *result_d,
                          char *fen_out_d, char *uci_out_d, int n, void *stream) {
    LMRL_REQUIRE(pos_d && actions_d && reward_d && done_d && result_d && fen_out_d && n > 0, "lmrl_chess_agent_step: bad argument");
    hipLaunchKernelGGL(chess_agent_step_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream), (Pos *)pos_d, actions_d, active_d, reward_d, done_d,
                       result_d, fen_out_d, uci_out_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_chess_opponent_step(void *pos_d, const char *uci_d, const uint8_t *active_d, float *reward_d, uint8_t *done_d, uint8_t *ok_d, char *san_out_d,
                             char *fen_out_d, int n, void *stream) {
    LMRL_REQUIRE(pos_d && uci_d && reward_d && done_d && ok_d && san_out_d && fen_out_d && n > 0, "lmrl_chess_opponent_step: bad argument");
    hipLaunchKernelGGL(chess_opponent_step_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream), (Pos *)pos_d, uci_d, active_d, reward_d, done_d,
                       ok_d, san_out_d, fen_out_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_chess_max_moves(void) { return kMaxMoves; }

int lmrl_chess_describe(const void *pos_d, char *uci_out_d, char *san_out_d, int32_t *count_d, uint8_t *status_d, char *fen_out_d, int n, void *stream) {
    LMRL_REQUIRE(pos_d && n > 0 && (uci_out_d || san_out_d || count_d || status_d || fen_out_d), "lmrl_chess_describe: bad argument");
    hipLaunchKernelGGL(chess_describe_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream), (const Pos *)pos_d, uci_out_d, san_out_d, count_d,
                       status_d, fen_out_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

// ---- host faces of the same functions (one position in host memory): CPU-tier tests, oracle comparisons, tools
int lmrl_chess_host_from_fen(const char *fen_str, void *pos) {
    if (!fen_str || !pos) return -1;
    return parse_fen(fen_str, *(Pos *)pos) ? LMRL_OK : LMRL_ERR_ARG;
}
int lmrl_chess_host_fen(const void *pos, char *out) {
    if (!pos || !out) return -1;
    return fen(*(const Pos *)pos, out);
}
int lmrl_chess_host_legal_moves(const void *pos, char *out_uci, char *out_san) {
    if (!pos || !out_uci) return -1;
    Move mv[kMaxMoves];
    const int k = gen_legal(*(const Pos *)pos, mv);
    for (int i = 0; i < k; i++) {
        uci(mv[i], out_uci + (size_t)i * 8);
        if (out_san) san(*(const Pos *)pos, mv[i], out_san + (size_t)i * kAct);
    }
    return k;
}
int lmrl_chess_host_agent_step(void *pos, const char *san_str, float *reward, int *done) {
    if (!pos || !san_str || !reward || !done) return -1;
    return agent_half_step(*(Pos *)pos, san_str, (int)strlen(san_str), reward, done);
}
int lmrl_chess_host_opponent_step(void *pos, const char *uci_str, char *san_out, float *reward, int *done) {
    if (!pos || !uci_str || !reward || !done) return -1;
    return opponent_half_step(*(Pos *)pos, uci_str, (int)strlen(uci_str), san_out, reward, done) ? 1 : 0;
}
int lmrl_chess_host_status(const void *pos) {
    if (!pos) return -1;
    const Pos &p = *(const Pos *)pos;
    Move mv[kMaxMoves];
    const int k = gen_legal(p, mv);
    const bool chk = in_check(p, p.stm);
    return (chk ? 1 : 0) | ((chk && !k) ? 2 : 0) | (is_game_over(p) ? 4 : 0) | (is_insufficient_material(p) ? 8 : 0) | ((!chk && !k) ? 16 : 0) |
           (is_repetition(p, 5) ? 32 : 0) | (p.halfmove >= 150 ? 64 : 0);
}

}  // extern "C"
