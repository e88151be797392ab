// chess_rules.h — the rules of chess as the reference's chess env uses them through python-chess (llm_rl_scripts/chess/env/env.py:28-185):
// Board(fen), push_san / parse_san, san(move), fen(), is_checkmate, is_game_over.  Plain C-style code (no allocation, no library calls) that
// compiles for the host AND as device code: the batched env steps one game per lane (chess.hip), the CPU tests drive the same functions
// through the host entry points.  python-chess is NOT in the image; the move generator is pinned against Stockfish 15.1 built from the
// reference's own sources (oracle/_ref/stockfish: `go perft 1` move lists, `d` positions), the SAN / FEN / termination conventions are restated
// from python-chess 1.x and marked where they cannot be pinned.
//
// Square index: a1 = 0, b1 = 1, ..., h8 = 63.  Pieces: 0 empty, 1..6 = white P N B R Q K, -1..-6 = black.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LMRL_HD __host__ __device__
#else
#define LMRL_HD
#endif

namespace lmrl_chess {

enum { P = 1, N = 2, B = 3, R = 4, Q = 5, K = 6 };
enum { WK = 1, WQ = 2, BK = 4, BQ = 8 };
constexpr int kMaxMoves = 256;       // upper bound on legal moves of a position (known maximum: 218)
constexpr int kHistory = 160;        // reversible plies whose repetition keys are kept (the 75-move rule ends the game at 150)

struct Move {
    int8_t from, to, promo;          // promo: 0 or N..Q (piece type)
    int8_t flag;                     // 0 normal, 1 en passant, 2 castle king side, 3 castle queen side, 4 double pawn push
};

struct Pos {
    int8_t sq[64];
    int8_t stm;                      // 0 white to move, 1 black
    int8_t castle;                   // WK | WQ | BK | BQ as given (cleaned on output, like python-chess)
    int8_t ep;                       // square behind a pawn that just advanced two squares, or -1 (shown in the FEN only if a legal capture exists)
    int8_t pad_;
    int32_t halfmove, fullmove;
    int32_t n_hist;                  // repetition keys of the positions since the last irreversible move (this position excluded)
    uint64_t hist[kHistory];
};

LMRL_HD inline int file_of(int s) { return s & 7; }
LMRL_HD inline int rank_of(int s) { return s >> 3; }
LMRL_HD inline int sgn(int v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }
LMRL_HD inline int absi(int v) { return v < 0 ? -v : v; }

// is `s` attacked by a piece of colour `by` (0 white, 1 black)?
LMRL_HD inline bool attacked(const Pos &p, int s, int by) {
    const int f = file_of(s), r = rank_of(s), me = by ? -1 : 1;
    // pawns: a white pawn on (f +- 1, r - 1) attacks s
    const int pr = r - me;
    if (pr >= 0 && pr < 8) {
        if (f > 0 && p.sq[pr * 8 + f - 1] == me * P) return true;
        if (f < 7 && p.sq[pr * 8 + f + 1] == me * P) return true;
    }
    const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
    for (int i = 0; i < 8; i++) {
        const int ff = f + kn[i][0], rr = r + kn[i][1];
        if (ff >= 0 && ff < 8 && rr >= 0 && rr < 8 && p.sq[rr * 8 + ff] == me * N) return true;
    }
    for (int df = -1; df <= 1; df++)
        for (int dr = -1; dr <= 1; dr++) {
            if (!df && !dr) continue;
            int ff = f + df, rr = r + dr;
            if (ff >= 0 && ff < 8 && rr >= 0 && rr < 8 && p.sq[rr * 8 + ff] == me * K) return true;
            const bool diag = df && dr;
            while (ff >= 0 && ff < 8 && rr >= 0 && rr < 8) {
                const int v = p.sq[rr * 8 + ff];
                if (v) {
                    if (v == me * Q || v == me * (diag ? B : R)) return true;
                    break;
                }
                ff += df; rr += dr;
            }
        }
    return false;
}

LMRL_HD inline int king_sq(const Pos &p, int color) {
    const int k = color ? -K : K;
    for (int s = 0; s < 64; s++) if (p.sq[s] == k) return s;
    return -1;
}
LMRL_HD inline bool in_check(const Pos &p, int color) {
    const int k = king_sq(p, color);
    return k >= 0 && attacked(p, k, color ^ 1);
}

// castling rights that are still meaningful: king and rook on their original squares (python-chess clean_castling_rights)
LMRL_HD inline int clean_castle(const Pos &p) {
    int c = p.castle;
    if (p.sq[4] != K) c &= ~(WK | WQ);
    if (p.sq[7] != R) c &= ~WK;
    if (p.sq[0] != R) c &= ~WQ;
    if (p.sq[60] != -K) c &= ~(BK | BQ);
    if (p.sq[63] != -R) c &= ~BK;
    if (p.sq[56] != -R) c &= ~BQ;
    return c;
}

// board part of a move (no clocks / history): used for legality tests and by make()
LMRL_HD inline void apply(Pos &p, const Move &m) {
    const int me = p.stm ? -1 : 1;
    const int pc = p.sq[m.from];
    p.sq[m.from] = 0;
    if (m.flag == 1) p.sq[m.to - 8 * me] = 0;                        // en passant: the captured pawn stands behind the target
    p.sq[m.to] = m.promo ? (int8_t)(me * m.promo) : (int8_t)pc;
    if (m.flag == 2) { p.sq[m.to + 1] = 0; p.sq[m.to - 1] = (int8_t)(me * R); }    // O-O: rook h -> f
    if (m.flag == 3) { p.sq[m.to - 2] = 0; p.sq[m.to + 1] = (int8_t)(me * R); }    // O-O-O: rook a -> d
}

LMRL_HD inline bool legal_after(const Pos &p, const Move &m) {
    Pos q;
    for (int i = 0; i < 64; i++) q.sq[i] = p.sq[i];
    q.stm = p.stm;
    apply(q, m);
    return !in_check(q, p.stm);
}

// all legal moves of the side to move, in a fixed order (by origin square, then piece-specific target order)
LMRL_HD inline int gen_legal(const Pos &p, Move *out) {
    int n = 0;
    const int me = p.stm ? -1 : 1, us = p.stm;
    const int cr = clean_castle(p);
    auto push = [&](int from, int to, int promo, int flag) {
        Move m;
        m.from = (int8_t)from; m.to = (int8_t)to; m.promo = (int8_t)promo; m.flag = (int8_t)flag;
        if (n < kMaxMoves && legal_after(p, m)) out[n++] = m;
    };
    for (int s = 0; s < 64; s++) {
        const int v = p.sq[s];
        if (!v || sgn(v) != me) continue;
        const int t = absi(v), f = file_of(s), r = rank_of(s);
        if (t == P) {
            const int r1 = r + me;
            if (r1 < 0 || r1 > 7) continue;
            const bool last = (r1 == (us ? 0 : 7));
            const int fw = r1 * 8 + f;
            if (!p.sq[fw]) {
                if (last) { for (int pr = Q; pr >= N; pr--) push(s, fw, pr, 0); }
                else {
                    push(s, fw, 0, 0);
                    if (r == (us ? 6 : 1) && !p.sq[fw + 8 * me]) push(s, fw + 8 * me, 0, 4);
                }
            }
            for (int df = -1; df <= 1; df += 2) {
                const int ff = f + df;
                if (ff < 0 || ff > 7) continue;
                const int to = r1 * 8 + ff;
                if (p.sq[to] && sgn(p.sq[to]) == -me) {
                    if (last) { for (int pr = Q; pr >= N; pr--) push(s, to, pr, 0); }
                    else push(s, to, 0, 0);
                } else if (to == p.ep && !p.sq[to] && p.sq[to - 8 * me] == -me * P) {
                    push(s, to, 0, 1);
                }
            }
        } else if (t == N) {
            const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
            for (int i = 0; i < 8; i++) {
                const int ff = f + kn[i][0], rr = r + kn[i][1];
                if (ff < 0 || ff > 7 || rr < 0 || rr > 7) continue;
                const int to = rr * 8 + ff;
                if (!p.sq[to] || sgn(p.sq[to]) == -me) push(s, to, 0, 0);
            }
        } else if (t == K) {
            for (int df = -1; df <= 1; df++)
                for (int dr = -1; dr <= 1; dr++) {
                    if (!df && !dr) continue;
                    const int ff = f + df, rr = r + dr;
                    if (ff < 0 || ff > 7 || rr < 0 || rr > 7) continue;
                    const int to = rr * 8 + ff;
                    if (!p.sq[to] || sgn(p.sq[to]) == -me) push(s, to, 0, 0);
                }
            const int home = us ? 60 : 4;
            if (s == home && !attacked(p, s, us ^ 1)) {
                if ((cr & (us ? BK : WK)) && !p.sq[home + 1] && !p.sq[home + 2] && !attacked(p, home + 1, us ^ 1)) push(s, home + 2, 0, 2);
                if ((cr & (us ? BQ : WQ)) && !p.sq[home - 1] && !p.sq[home - 2] && !p.sq[home - 3] && !attacked(p, home - 1, us ^ 1))
                    push(s, home - 2, 0, 3);
            }
        } else {
            for (int df = -1; df <= 1; df++)
                for (int dr = -1; dr <= 1; dr++) {
                    if (!df && !dr) continue;
                    const bool diag = df && dr;
                    if (t == B && !diag) continue;
                    if (t == R && diag) continue;
                    int ff = f + df, rr = r + dr;
                    while (ff >= 0 && ff < 8 && rr >= 0 && rr < 8) {
                        const int to = rr * 8 + ff;
                        if (!p.sq[to]) push(s, to, 0, 0);
                        else { if (sgn(p.sq[to]) == -me) push(s, to, 0, 0); break; }
                        ff += df; rr += dr;
                    }
                }
        }
    }
    return n;
}

// does a legal en-passant capture exist? (python-chess has_legal_en_passant: decides whether the FEN / the repetition key show the square)
LMRL_HD inline bool has_legal_ep(const Pos &p) {
    if (p.ep < 0) return false;
    const int me = p.stm ? -1 : 1;
    const int r = rank_of(p.ep) - me, f = file_of(p.ep);
    if (r < 0 || r > 7 || p.sq[p.ep] || p.sq[p.ep - 8 * me] != -me * P) return false;
    for (int df = -1; df <= 1; df += 2) {
        const int ff = f + df;
        if (ff < 0 || ff > 7) continue;
        const int s = r * 8 + ff;
        if (p.sq[s] == me * P) {
            Move m;
            m.from = (int8_t)s; m.to = p.ep; m.promo = 0; m.flag = 1;
            if (legal_after(p, m)) return true;
        }
    }
    return false;
}

// repetition key: piece placement, side to move, cleaned castling rights, legal en-passant square (python-chess _transposition_key)
LMRL_HD inline uint64_t rep_key(const Pos &p) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; h ^= h >> 29; };
    for (int s = 0; s < 64; s++) mix((uint64_t)(uint8_t)(p.sq[s] + 8) + 31ull * (uint64_t)s);
    mix((uint64_t)p.stm + 1000ull);
    mix((uint64_t)clean_castle(p) + 2000ull);
    mix(has_legal_ep(p) ? (uint64_t)p.ep + 3000ull : 2999ull);
    return h;
}

// play a legal move: board, castling rights, en passant, clocks, repetition history (python-chess Board.push)
LMRL_HD inline void make(Pos &p, const Move &m) {
    const int me = p.stm ? -1 : 1;
    const int pc = absi(p.sq[m.from]);
    const bool capture = p.sq[m.to] != 0 || m.flag == 1;
    const bool irreversible = capture || pc == P;
    const uint64_t key = rep_key(p);
    const int cr_before = clean_castle(p);
    apply(p, m);
    if (pc == K) p.castle &= (me > 0) ? ~(WK | WQ) : ~(BK | BQ);
    const int touched[2] = {m.from, m.to};
    for (int i = 0; i < 2; i++) {
        if (touched[i] == 7) p.castle &= ~WK;
        if (touched[i] == 0) p.castle &= ~WQ;
        if (touched[i] == 63) p.castle &= ~BK;
        if (touched[i] == 56) p.castle &= ~BQ;
    }
    p.ep = (m.flag == 4) ? (int8_t)(m.from + 8 * me) : (int8_t)-1;
    p.halfmove = irreversible ? 0 : p.halfmove + 1;
    if (p.stm) p.fullmove++;
    p.stm ^= 1;
    // python-chess walks its move stack back until an irreversible move (capture, pawn move, lost castling right): keep exactly those keys
    if (irreversible || clean_castle(p) != cr_before) p.n_hist = 0;
    else if (p.n_hist < kHistory) p.hist[p.n_hist++] = key;
}

LMRL_HD inline bool is_checkmate(const Pos &p) {
    if (!in_check(p, p.stm)) return false;
    Move mv[kMaxMoves];
    return gen_legal(p, mv) == 0;
}

// python-chess is_insufficient_material(): both sides lack mating material (restated from python-chess 1.x has_insufficient_material; unpinned)
LMRL_HD inline bool insufficient_side(const Pos &p, int color) {
    const int me = color ? -1 : 1;
    int mine = 0, knights = 0, bishops = 0, opp_other = 0, all_pawns = 0, all_knights = 0;
    bool light = false, dark = false;
    for (int s = 0; s < 64; s++) {
        const int v = p.sq[s];
        if (!v) continue;
        const int t = absi(v);
        if (t == P) all_pawns++;
        if (t == N) all_knights++;
        if (t == B) { if ((file_of(s) + rank_of(s)) & 1) light = true; else dark = true; }
        if (sgn(v) == me) {
            mine++;
            if (t == P || t == R || t == Q) return false;
            if (t == N) knights++;
            if (t == B) bishops++;
        } else if (t != K && t != Q) {
            opp_other++;
        }
    }
    if (knights) return mine <= 2 && opp_other == 0;
    if (bishops) return !(light && dark) && !all_pawns && !all_knights;
    return true;
}
LMRL_HD inline bool is_insufficient_material(const Pos &p) { return insufficient_side(p, 0) && insufficient_side(p, 1); }

LMRL_HD inline bool is_repetition(const Pos &p, int count) {
    const uint64_t key = rep_key(p);
    int n = 1;
    for (int i = p.n_hist - 1; i >= 0; i--) if (p.hist[i] == key && ++n >= count) return true;
    return n >= count;
}

// python-chess Board.is_game_over() (claim_draw = False): checkmate, stalemate, insufficient material, 75-move rule, fivefold repetition
LMRL_HD inline bool is_game_over(const Pos &p) {
    Move mv[kMaxMoves];
    if (gen_legal(p, mv) == 0) return true;
    if (is_insufficient_material(p)) return true;
    if (p.halfmove >= 150) return true;
    return is_repetition(p, 5);
}

// ------------------------------------------------------------------------------------------ text: FEN
LMRL_HD inline char piece_char(int v) {
    const char *w = " PNBRQK";
    const char c = w[absi(v)];
    return v < 0 ? (char)(c + 32) : c;
}
LMRL_HD inline int piece_from_char(char c) {
    const char *w = "PNBRQK";
    for (int i = 0; i < 6; i++) {
        if (c == w[i]) return i + 1;
        if (c == w[i] + 32) return -(i + 1);
    }
    return 0;
}
LMRL_HD inline int put_int(char *o, int v) {
    char tmp[12];
    int n = 0;
    if (v <= 0) tmp[n++] = '0';
    while (v > 0) { tmp[n++] = (char)('0' + v % 10); v /= 10; }
    for (int i = 0; i < n; i++) o[i] = tmp[n - 1 - i];
    return n;
}

// python-chess Board.fen(): en-passant square only if a legal en-passant capture exists; cleaned castling rights.  Returns the length.
LMRL_HD inline int fen(const Pos &p, char *o) {
    int n = 0;
    for (int r = 7; r >= 0; r--) {
        int run = 0;
        for (int f = 0; f < 8; f++) {
            const int v = p.sq[r * 8 + f];
            if (!v) { run++; continue; }
            if (run) { o[n++] = (char)('0' + run); run = 0; }
            o[n++] = piece_char(v);
        }
        if (run) o[n++] = (char)('0' + run);
        if (r) o[n++] = '/';
    }
    o[n++] = ' '; o[n++] = p.stm ? 'b' : 'w'; o[n++] = ' ';
    const int c = clean_castle(p);
    if (!c) o[n++] = '-';
    else {
        if (c & WK) o[n++] = 'K';
        if (c & WQ) o[n++] = 'Q';
        if (c & BK) o[n++] = 'k';
        if (c & BQ) o[n++] = 'q';
    }
    o[n++] = ' ';
    if (has_legal_ep(p)) { o[n++] = (char)('a' + file_of(p.ep)); o[n++] = (char)('1' + rank_of(p.ep)); }
    else o[n++] = '-';
    o[n++] = ' ';
    n += put_int(o + n, p.halfmove);
    o[n++] = ' ';
    n += put_int(o + n, p.fullmove);
    o[n] = 0;
    return n;
}

// chess.Board(fen).  Returns false on a malformed string.
LMRL_HD inline bool parse_fen(const char *s, Pos &p) {
    for (int i = 0; i < 64; i++) p.sq[i] = 0;
    p.stm = 0; p.castle = 0; p.ep = -1; p.pad_ = 0; p.halfmove = 0; p.fullmove = 1; p.n_hist = 0;
    int r = 7, f = 0, i = 0;
    for (; s[i] && s[i] != ' '; i++) {
        const char c = s[i];
        if (c == '/') { if (f != 8) return false; r--; f = 0; if (r < 0) return false; }
        else if (c >= '1' && c <= '8') { f += c - '0'; if (f > 8) return false; }
        else {
            const int v = piece_from_char(c);
            if (!v || f > 7) return false;
            p.sq[r * 8 + f++] = (int8_t)v;
        }
    }
    if (r != 0 || f != 8) return false;
    if (!s[i]) return true;
    i++;
    if (s[i] == 'b') p.stm = 1; else if (s[i] != 'w') return false;
    i++;
    if (!s[i]) return true;
    i++;
    for (; s[i] && s[i] != ' '; i++) {
        if (s[i] == 'K') p.castle |= WK; else if (s[i] == 'Q') p.castle |= WQ; else if (s[i] == 'k') p.castle |= BK;
        else if (s[i] == 'q') p.castle |= BQ; else if (s[i] != '-') return false;
    }
    if (!s[i]) return true;
    i++;
    if (s[i] >= 'a' && s[i] <= 'h' && s[i + 1] >= '1' && s[i + 1] <= '8') { p.ep = (int8_t)((s[i + 1] - '1') * 8 + (s[i] - 'a')); i += 2; }
    else if (s[i] == '-') i++;
    else return false;
    if (!s[i]) return true;
    i++;
    int v = 0;
    for (; s[i] >= '0' && s[i] <= '9'; i++) v = v * 10 + (s[i] - '0');
    p.halfmove = v;
    if (!s[i]) return true;
    i++;
    v = 0;
    for (; s[i] >= '0' && s[i] <= '9'; i++) v = v * 10 + (s[i] - '0');
    p.fullmove = v > 0 ? v : 1;
    return true;
}

// ------------------------------------------------------------------------------------------ text: SAN
LMRL_HD inline int uci(const Move &m, char *o) {
    int n = 0;
    o[n++] = (char)('a' + file_of(m.from)); o[n++] = (char)('1' + rank_of(m.from));
    o[n++] = (char)('a' + file_of(m.to)); o[n++] = (char)('1' + rank_of(m.to));
    if (m.promo) o[n++] = " pnbrqk"[m.promo];
    o[n] = 0;
    return n;
}

// python-chess Board.san(move) for a LEGAL move: piece letter, minimal disambiguation among the legal moves, 'x', target, '=Q', '+' / '#'
LMRL_HD inline int san(const Pos &p, const Move &m, char *o) {
    int n = 0;
    const int pc = absi(p.sq[m.from]);
    if (m.flag == 2) { o[n++] = 'O'; o[n++] = '-'; o[n++] = 'O'; }
    else if (m.flag == 3) { o[n++] = 'O'; o[n++] = '-'; o[n++] = 'O'; o[n++] = '-'; o[n++] = 'O'; }
    else {
        const bool capture = p.sq[m.to] != 0 || m.flag == 1;
        if (pc != P) {
            o[n++] = " PNBRQK"[pc];
            Move mv[kMaxMoves];
            const int k = gen_legal(p, mv);
            bool others = false, same_file = false, same_rank = false;
            for (int i = 0; i < k; i++) {
                if (mv[i].to != m.to || mv[i].from == m.from || absi(p.sq[mv[i].from]) != pc) continue;
                others = true;
                if (file_of(mv[i].from) == file_of(m.from)) same_file = true;
                if (rank_of(mv[i].from) == rank_of(m.from)) same_rank = true;
            }
            if (others) {
                if (!same_file) o[n++] = (char)('a' + file_of(m.from));
                else if (!same_rank) o[n++] = (char)('1' + rank_of(m.from));
                else { o[n++] = (char)('a' + file_of(m.from)); o[n++] = (char)('1' + rank_of(m.from)); }
            }
        } else if (capture) {
            o[n++] = (char)('a' + file_of(m.from));
        }
        if (capture) o[n++] = 'x';
        o[n++] = (char)('a' + file_of(m.to)); o[n++] = (char)('1' + rank_of(m.to));
        if (m.promo) { o[n++] = '='; o[n++] = " PNBRQK"[m.promo]; }
    }
    Pos q = p;
    make(q, m);
    if (in_check(q, q.stm)) {
        Move mv[kMaxMoves];
        o[n++] = gen_legal(q, mv) ? '+' : '#';
    }
    o[n] = 0;
    return n;
}

enum SanStatus { SAN_OK = 0, SAN_NULL = 1, SAN_INVALID = 2, SAN_ILLEGAL = 3, SAN_AMBIGUOUS = 4 };

// python-chess Board.parse_san (1.x): castling with O or 0 (optional + / #), the null moves "--" and "Z0", otherwise
// ^([NBKRQ])?([a-h])?([1-8])?[\-x]?([a-h][1-8])(=?[nbrqkNBRQK])?[\+#]?$ matched against the legal moves (piece type, target, origin file /
// rank filters, promotion); more than one match is ambiguous.  `len` characters of `s` are read (no terminator needed).
LMRL_HD inline int parse_san(const Pos &p, const char *s, int len, Move *out) {
    auto eq = [&](const char *t) { int i = 0; for (; t[i]; i++) if (i >= len || s[i] != t[i]) return false; return i == len; };
    Move mv[kMaxMoves];
    if (eq("--") || eq("Z0")) return SAN_NULL;
    int castle = 0;
    {
        const char *ks[6] = {"O-O", "O-O+", "O-O#", "0-0", "0-0+", "0-0#"};
        const char *qs[6] = {"O-O-O", "O-O-O+", "O-O-O#", "0-0-0", "0-0-0+", "0-0-0#"};
        for (int i = 0; i < 6; i++) { if (eq(ks[i])) castle = 2; if (eq(qs[i])) castle = 3; }
    }
    const int k = gen_legal(p, mv);
    if (castle) {
        for (int i = 0; i < k; i++) if (mv[i].flag == castle) { *out = mv[i]; return SAN_OK; }
        return SAN_ILLEGAL;
    }
    int i = 0, piece = P, ff = -1, fr = -1, promo = 0;
    if (i < len && (s[i] == 'N' || s[i] == 'B' || s[i] == 'K' || s[i] == 'R' || s[i] == 'Q')) piece = piece_from_char(s[i++]);
    // strip the optional check suffix, then the optional promotion ("=Q", or "Q" directly behind the target square)
    int end = len;
    if (end > i && (s[end - 1] == '+' || s[end - 1] == '#')) end--;
    if (end - i >= 3) {
        const int pt = absi(piece_from_char(s[end - 1]));
        if (pt >= N) {
            if (s[end - 2] == '=') { promo = pt; end -= 2; }
            else if (s[end - 3] >= 'a' && s[end - 3] <= 'h' && s[end - 2] >= '1' && s[end - 2] <= '8') { promo = pt; end -= 1; }
        }
    }
    if (end - i < 2) return SAN_INVALID;
    const char tf = s[end - 2], tr = s[end - 1];
    if (tf < 'a' || tf > 'h' || tr < '1' || tr > '8') return SAN_INVALID;
    const int to = (tr - '1') * 8 + (tf - 'a');
    int j = i;
    const int mid_end = end - 2;
    if (j < mid_end && s[j] >= 'a' && s[j] <= 'h') ff = s[j++] - 'a';
    if (j < mid_end && s[j] >= '1' && s[j] <= '8') fr = s[j++] - '1';
    if (j < mid_end && (s[j] == '-' || s[j] == 'x')) j++;
    if (j != mid_end) return SAN_INVALID;
    if (promo == K) return SAN_ILLEGAL;                       // "=K" passes the pattern but matches no legal move
    int found = -1;
    for (int q = 0; q < k; q++) {
        const Move &m = mv[q];
        if (m.to != to || absi(p.sq[m.from]) != piece) continue;
        if (ff >= 0 && file_of(m.from) != ff) continue;
        if (fr >= 0 && rank_of(m.from) != fr) continue;
        if (m.flag == 2 || m.flag == 3) { if (piece != K) continue; }
        if (m.promo != promo) continue;                          // a promotion needs its piece; a non-promotion must not name one
        if (piece == P && ff < 0 && file_of(m.from) != file_of(to)) continue;   // "e5" never means a capture: pawn captures name their file
        if (found >= 0) return SAN_AMBIGUOUS;
        found = q;
    }
    if (found < 0) return SAN_ILLEGAL;
    *out = mv[found];
    return SAN_OK;
}

// ------------------------------------------------------------------------------------------ the env step (chess/env/env.py:91-143)
// ChessEnv.step for the agent's half: the action (already stripped of blanks) is a SAN string.  result: 0 illegal / unparsable (reward -1, not done,
// board unchanged), 1 legal and the game goes on (the opponent is to move), 2 legal and the game is over (reward 1 if checkmate else 0),
// 3 null move (reward -1, done).
LMRL_HD inline int agent_half_step(Pos &p, const char *san_str, int len, float *reward, int *done, char *uci_out = nullptr) {
    Move m;
    const int st = parse_san(p, san_str, len, &m);
    if (uci_out) { if (st == SAN_OK) uci(m, uci_out); else uci_out[0] = 0; }
    if (st == SAN_NULL) { *reward = -1.f; *done = 1; return 3; }
    if (st != SAN_OK) { *reward = -1.f; *done = 0; return 0; }
    make(p, m);
    if (is_game_over(p)) { *reward = is_checkmate(p) ? 1.f : 0.f; *done = 1; return 2; }
    *reward = 0.f; *done = 0;
    return 1;
}
// ... and for the opponent's reply (a legal move of the engine, UCI form): reward -1 if the agent is mated, done = is_game_over
LMRL_HD inline bool opponent_half_step(Pos &p, const char *uci_str, int len, char *san_out, float *reward, int *done) {
    Move mv[kMaxMoves];
    const int k = gen_legal(p, mv);
    for (int q = 0; q < k; q++) {
        char u[8];
        const int n = uci(mv[q], u);
        bool same = n == len;
        for (int i = 0; same && i < n; i++) same = u[i] == uci_str[i];
        if (!same) continue;
        if (san_out) san(p, mv[q], san_out);
        make(p, mv[q]);
        *reward = is_checkmate(p) ? -1.f : 0.f;
        *done = is_game_over(p) ? 1 : 0;
        return true;
    }
    return false;
}

}  // namespace lmrl_chess
