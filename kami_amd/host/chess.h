// chess.h — the rules the search needs: position, legal moves, the 73-plane action code, FEN.
//
// Own implementation (bitboards, ray loops, copy-make), NOT the reference's neocortex library; it has
// to agree with it where kami's Env (kami/env.h) exposes it:
//   * the legal-move SET of every position              (env.h:398-423; tests/golden/*.npz)
//   * the action code and its side-to-move point of view (env.h:60-200)
//   * make-move bookkeeping that reaches the network input or the terminal test: half-move clock
//     (reset on pawn moves and captures), castle rights (revoked when e1/a1/h1/e8/a8/h8 are a
//     move's source or destination), en-passant square (set after EVERY double push), and what a
//     repetition compares: placement, side, castle rights, en-passant square
//     (kami/chess/neocortex/position.c:167-318, 1347-1357)
//   * FEN text                                           (position.c:104-163)
// Squares are rank * 8 + file (a1 = 0); colours 0 white / 1 black; piece types P N B R Q K = 0..5,
// the order of kh_board::piece_occ.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

namespace kami {
namespace chess {

enum { PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING };
enum { WHITE, BLACK };
enum { CASTLE_WK = 1, CASTLE_WQ = 2, CASTLE_BK = 4, CASTLE_BQ = 8 };
constexpr int MAX_MOVES = 256;

struct Move {
    uint8_t src, dst;
    uint8_t promo;          // 0 = none, else the piece type promoted to (KNIGHT..QUEEN)
    bool operator==(const Move& o) const { return src == o.src && dst == o.dst && promo == o.promo; }
};

inline uint64_t bit(int sq) { return 1ull << sq; }
inline int lsb(uint64_t b) { return __builtin_ctzll(b); }
inline int popcount(uint64_t b) { return __builtin_popcountll(b); }
constexpr uint64_t FILE_A = 0x0101010101010101ull, FILE_H = FILE_A << 7;
constexpr uint64_t RANK_1 = 0xffull, RANK_8 = RANK_1 << 56;

// direction order shared by the tables: N S E W NE NW SE SW (the action code's order too)
constexpr int DIR_DR[8] = { 1, -1, 0, 0, 1, 1, -1, -1 };
constexpr int DIR_DF[8] = { 0, 0, 1, -1, 1, -1, 1, -1 };

struct Tables {
    uint64_t knight[64], king[64];
    uint64_t ray[8][64];            // squares strictly beyond `sq` in a direction, to the board edge
    Tables()
    {
        for (int s = 0; s < 64; ++s) {
            const int r = s >> 3, f = s & 7;
            knight[s] = king[s] = 0;
            const int kn[8][2] = { { 1, 2 }, { 2, 1 }, { 2, -1 }, { 1, -2 }, { -1, -2 }, { -2, -1 }, { -2, 1 }, { -1, 2 } };
            for (auto& d : kn)
                if (r + d[0] >= 0 && r + d[0] < 8 && f + d[1] >= 0 && f + d[1] < 8) knight[s] |= bit((r + d[0]) * 8 + f + d[1]);
            for (int dr = -1; dr <= 1; ++dr)
                for (int df = -1; df <= 1; ++df)
                    if ((dr || df) && r + dr >= 0 && r + dr < 8 && f + df >= 0 && f + df < 8) king[s] |= bit((r + dr) * 8 + f + df);
            for (int d = 0; d < 8; ++d) {
                ray[d][s] = 0;
                for (int rr = r + DIR_DR[d], ff = f + DIR_DF[d]; rr >= 0 && rr < 8 && ff >= 0 && ff < 8; rr += DIR_DR[d], ff += DIR_DF[d])
                    ray[d][s] |= bit(rr * 8 + ff);
            }
        }
    }
};
inline const Tables& tables() { static const Tables t; return t; }

// squares a slider on `sq` reaches in direction d until (and including) the first blocker: the ray minus
// the blocker's own ray (first blocker = lowest set bit for directions that increase the square index,
// highest for the others)
inline uint64_t ray_attacks(int d, int sq, uint64_t occ)
{
    const Tables& t = tables();
    const uint64_t r = t.ray[d][sq], blockers = r & occ;
    if (!blockers) return r;
    const bool up = d == 0 || d == 2 || d == 4 || d == 5;       // N, E, NE, NW increase the index
    const int b = up ? __builtin_ctzll(blockers) : 63 - __builtin_clzll(blockers);
    return r ^ t.ray[d][b];
}
inline uint64_t rook_attacks(int sq, uint64_t occ)
{
    return ray_attacks(0, sq, occ) | ray_attacks(1, sq, occ) | ray_attacks(2, sq, occ) | ray_attacks(3, sq, occ);
}
inline uint64_t bishop_attacks(int sq, uint64_t occ)
{
    return ray_attacks(4, sq, occ) | ray_attacks(5, sq, occ) | ray_attacks(6, sq, occ) | ray_attacks(7, sq, occ);
}

struct Position {
    uint64_t pc[6];         // by piece type, both colours
    uint64_t col[2];
    uint8_t ctm;            // side to move
    uint8_t castle;         // CASTLE_* bits
    int8_t ep;              // en-passant target square or -1
    int32_t halfmove;       // plies since the last pawn move or capture
    int32_t fullmove;

    uint64_t occ() const { return col[0] | col[1]; }

    static Position start()
    {
        Position p;
        p.pc[PAWN] = 0x00ff00000000ff00ull; p.pc[KNIGHT] = 0x4200000000000042ull; p.pc[BISHOP] = 0x2400000000000024ull;
        p.pc[ROOK] = 0x8100000000000081ull; p.pc[QUEEN] = 0x0800000000000008ull; p.pc[KING] = 0x1000000000000010ull;
        p.col[WHITE] = 0xffffull; p.col[BLACK] = 0xffffull << 48;
        p.ctm = WHITE; p.castle = 15; p.ep = -1; p.halfmove = 0; p.fullmove = 1;
        return p;
    }

    int piece_at(int sq) const
    {
        const uint64_t b = bit(sq);
        for (int t = 0; t < 6; ++t)
            if (pc[t] & b) return t;
        return -1;
    }
    // the same for a square known to be occupied, without branches
    int type_at(int sq) const
    {
        return (int)(((pc[1] >> sq) & 1) + 2 * ((pc[2] >> sq) & 1) + 3 * ((pc[3] >> sq) & 1) + 4 * ((pc[4] >> sq) & 1) + 5 * ((pc[5] >> sq) & 1));
    }

    // is `sq` attacked by a piece of colour `by`?
    bool attacked(int sq, int by) const
    {
        const uint64_t them = col[by], o = occ();
        const uint64_t b = bit(sq);
        // a pawn of colour `by` attacks sq from one rank behind it (seen from `by`)
        const uint64_t from = by == WHITE ? (((b & ~FILE_A) >> 9) | ((b & ~FILE_H) >> 7)) : (((b & ~FILE_A) << 7) | ((b & ~FILE_H) << 9));
        if (from & pc[PAWN] & them) return true;
        if (tables().knight[sq] & pc[KNIGHT] & them) return true;
        if (tables().king[sq] & pc[KING] & them) return true;
        if (rook_attacks(sq, o) & (pc[ROOK] | pc[QUEEN]) & them) return true;
        if (bishop_attacks(sq, o) & (pc[BISHOP] | pc[QUEEN]) & them) return true;
        return false;
    }
    int king_sq(int c) const { return lsb(pc[KING] & col[c]); }
    bool in_check() const { return attacked(king_sq(ctm), !ctm); }

    // what a repetition compares (position.c:302-311): placement, en-passant square, castle rights, side
    uint64_t key() const
    {
        // six independent multiplies (no chain from one word to the next: the search computes a key per
        // move of every walk), folded and finished with one avalanche step
        const uint64_t a = (pc[0] + 0x9e3779b97f4a7c15ull) * 0xff51afd7ed558ccdull, b = (pc[1] + 0xc2b2ae3d27d4eb4full) * 0xc4ceb9fe1a85ec53ull,
                       c = (pc[2] + 0x165667b19e3779f9ull) * 0x9fb21c651e98df25ull, d = (pc[3] + 0x27d4eb2f165667c5ull) * 0xd6e8feb86659fd93ull,
                       e = (pc[4] + 0x85ebca77c2b2ae63ull) * 0xbf58476d1ce4e5b9ull, f = (pc[5] + 0x2545f4914f6cdd1dull) * 0x94d049bb133111ebull,
                       g = (col[0] + 0x632be59bd9b4e019ull) * 0xe7037ed1a0b428dbull,
                       m = ((((uint64_t)(uint8_t)ep << 16) | ((uint64_t)castle << 8) | ctm) + 0x3c79ac492ba7b653ull) * 0x1c69b3f74ac4ae35ull;
        auto rot = [](uint64_t v, int r) { return (v << r) | (v >> (64 - r)); };
        uint64_t h = a ^ rot(b, 9) ^ rot(c, 18) ^ rot(d, 27) ^ rot(e, 36) ^ rot(f, 45) ^ rot(g, 54) ^ rot(m, 31);
        h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
        return h;
    }

    void remove(int sq, int type, int c) { pc[type] &= ~bit(sq); col[c] &= ~bit(sq); }
    void place(int sq, int type, int c) { pc[type] |= bit(sq); col[c] |= bit(sq); }

    // Apply a pseudo-legal move (position.c:167-318 bookkeeping); apply() does not look at the mover's king.
    void apply(Move m)
    {
        const int us = ctm, them = !ctm;
        const int type = type_at(m.src);
        const int victim = (col[them] & bit(m.dst)) ? type_at(m.dst) : -1;
        ++halfmove;
        if (us == BLACK) ++fullmove;
        const int old_ep = ep;
        ep = -1;
        remove(m.src, type, us);
        if (type == PAWN) halfmove = 0;
        if (type == PAWN && m.dst == old_ep) {                       // en-passant capture
            remove((m.src & ~7) | (m.dst & 7), PAWN, them);
            halfmove = 0;
        }
        if (type == KING && std::abs((m.src & 7) - (m.dst & 7)) > 1) {   // castling: move the rook too
            const int rank = m.src & ~7, ks = m.dst > m.src;
            remove(rank | (ks ? 7 : 0), ROOK, us);
            place(rank | (ks ? 5 : 3), ROOK, us);
        }
        if (victim >= 0) { remove(m.dst, victim, them); halfmove = 0; }
        place(m.dst, m.promo ? m.promo : type, us);
        if (type == KING) castle &= us == WHITE ? ~3 : ~12;
        const uint64_t touched = bit(m.src) | bit(m.dst);
        if (touched & (bit(4) | bit(7))) castle &= ~CASTLE_WK;
        if (touched & (bit(4) | bit(0))) castle &= ~CASTLE_WQ;
        if (touched & (bit(60) | bit(63))) castle &= ~CASTLE_BK;
        if (touched & (bit(60) | bit(56))) castle &= ~CASTLE_BQ;
        if (type == PAWN && std::abs((m.dst >> 3) - (m.src >> 3)) > 1) ep = (int8_t)(us == WHITE ? m.dst - 8 : m.dst + 8);
        ctm = (uint8_t)them;
    }
    // ... and make() says whether it left the mover's king attacked (the position is then garbage for the caller)
    bool make(Move m)
    {
        const int us = ctm;
        apply(m);
        return !attacked(king_sq(us), !us);
    }

    // pseudo-legal moves of the side to move
    int pseudo_legal(Move* out) const
    {
        int n = 0;
        const int us = ctm, them = !ctm;
        const uint64_t mine = col[us], theirs = col[them], o = occ(), empty = ~o;
        auto add = [&](int s, int d, int promo = 0) { out[n++] = Move{ (uint8_t)s, (uint8_t)d, (uint8_t)promo }; };
        auto add_pawn = [&](int s, int d) {
            if ((d >> 3) == (us == WHITE ? 7 : 0)) { add(s, d, QUEEN); add(s, d, KNIGHT); add(s, d, ROOK); add(s, d, BISHOP); }
            else add(s, d);
        };
        const int up = us == WHITE ? 8 : -8;
        const uint64_t targets = theirs | (ep >= 0 ? bit(ep) : 0);
        for (uint64_t b = pc[PAWN] & mine; b; b &= b - 1) {
            const int s = lsb(b), f = s & 7, r = s >> 3;
            if (r == (us == WHITE ? 7 : 0)) continue;     // a pawn left on its last rank (see decode_action) never moves
            if (empty & bit(s + up)) {
                add_pawn(s, s + up);
                if (r == (us == WHITE ? 1 : 6) && (empty & bit(s + 2 * up))) add(s, s + 2 * up);
            }
            if (f > 0 && (targets & bit(s + up - 1))) add_pawn(s, s + up - 1);
            if (f < 7 && (targets & bit(s + up + 1))) add_pawn(s, s + up + 1);
        }
        for (uint64_t b = pc[KNIGHT] & mine; b; b &= b - 1) {
            const int s = lsb(b);
            for (uint64_t t = tables().knight[s] & ~mine; t; t &= t - 1) add(s, lsb(t));
        }
        for (uint64_t b = (pc[BISHOP] | pc[QUEEN]) & mine; b; b &= b - 1) {
            const int s = lsb(b);
            for (uint64_t t = bishop_attacks(s, o) & ~mine; t; t &= t - 1) add(s, lsb(t));
        }
        for (uint64_t b = (pc[ROOK] | pc[QUEEN]) & mine; b; b &= b - 1) {
            const int s = lsb(b);
            for (uint64_t t = rook_attacks(s, o) & ~mine; t; t &= t - 1) add(s, lsb(t));
        }
        const int k = king_sq(us);
        for (uint64_t t = tables().king[k] & ~mine; t; t &= t - 1) add(k, lsb(t));
        // castling (position.c:523-557): rights, empty squares between, king's path not attacked
        const int rank = us == WHITE ? 0 : 56;
        if (k == rank + 4) {
            if ((castle & (us == WHITE ? CASTLE_WK : CASTLE_BK)) && !(o & (bit(rank + 5) | bit(rank + 6))) &&
                !attacked(rank + 4, them) && !attacked(rank + 5, them) && !attacked(rank + 6, them))
                add(k, rank + 6);
            if ((castle & (us == WHITE ? CASTLE_WQ : CASTLE_BQ)) && !(o & (bit(rank + 1) | bit(rank + 2) | bit(rank + 3))) &&
                !attacked(rank + 4, them) && !attacked(rank + 3, them) && !attacked(rank + 2, them))
                add(k, rank + 2);
        }
        return n;
    }

    // own pieces that stand alone between the king and an enemy slider aiming at it
    uint64_t pinned() const
    {
        const int us = ctm, them = !ctm, k = king_sq(us);
        const uint64_t o = occ();
        const Tables& t = tables();
        uint64_t pins = 0;
        for (int d = 0; d < 8; ++d) {
            const uint64_t sliders = (d < 4 ? (pc[ROOK] | pc[QUEEN]) : (pc[BISHOP] | pc[QUEEN])) & col[them];
            if (!(t.ray[d][k] & sliders)) continue;
            const uint64_t first = ray_attacks(d, k, o) & o;               // first blocker from the king
            if (!(first & col[us])) continue;
            const uint64_t second = ray_attacks(d, lsb(first), o) & o;     // the piece behind it
            if (second & sliders) pins |= first;
        }
        return pins;
    }

    // legal moves: pseudo-legal ones that do not leave the king attacked.  Out of check, a move of an
    // unpinned piece other than the king (and other than an en-passant capture, which removes a pawn
    // beside the mover) cannot expose the king, so only the rest is played out and tested.
    int legal(Move* out) const
    {
        Move pl[MAX_MOVES];
        const int npl = pseudo_legal(pl);
        const bool check = in_check();
        const uint64_t pins = check ? ~0ull : pinned();
        const int k = king_sq(ctm);
        int n = 0;
        for (int i = 0; i < npl; ++i) {
            const Move m = pl[i];
            const bool risky = check || m.src == k || (pins & bit(m.src)) || (ep >= 0 && m.dst == ep && (pc[PAWN] & bit(m.src)));
            if (!risky) { out[n++] = m; continue; }
            Position q = *this;
            if (q.make(m)) out[n++] = m;
        }
        return n;
    }

    // FEN as ncPositionToFen writes it (position.c:104-163)
    std::string fen() const
    {
        static const char sym[] = "pnbrqk";
        std::string s;
        for (int r = 7; r >= 0; --r) {
            int gap = 0;
            for (int f = 0; f < 8; ++f) {
                const int sq = r * 8 + f, t = piece_at(sq);
                if (t < 0) { ++gap; continue; }
                if (gap) { s += (char)('0' + gap); gap = 0; }
                s += (col[WHITE] & bit(sq)) ? (char)(sym[t] - 32) : sym[t];
            }
            if (gap) s += (char)('0' + gap);
            if (r) s += '/';
        }
        s += ctm == WHITE ? " w " : " b ";
        if (!castle) s += '-';
        else {
            if (castle & CASTLE_WK) s += 'K';
            if (castle & CASTLE_WQ) s += 'Q';
            if (castle & CASTLE_BK) s += 'k';
            if (castle & CASTLE_BQ) s += 'q';
        }
        s += ' ';
        if (ep < 0) s += '-';
        else { s += (char)('a' + (ep & 7)); s += (char)('1' + (ep >> 3)); }
        s += ' ' + std::to_string(halfmove) + ' ' + std::to_string(fullmove);
        return s;
    }

    static bool from_fen(const std::string& fen, Position& p)
    {
        std::memset(&p, 0, sizeof(p));
        p.ep = -1; p.fullmove = 1;
        size_t i = 0;
        int r = 7, f = 0;
        for (; i < fen.size() && fen[i] != ' '; ++i) {
            const char c = fen[i];
            if (c == '/') { --r; f = 0; continue; }
            if (c >= '1' && c <= '8') { f += c - '0'; continue; }
            const char* sym = "pnbrqk";
            const char lc = (char)(c | 32);
            const char* at = std::strchr(sym, lc);
            if (!at || r < 0 || f > 7) return false;
            p.place(r * 8 + f, (int)(at - sym), c == lc ? BLACK : WHITE);
            ++f;
        }
        if (i >= fen.size()) return false;
        // move generation asks for each side's king square: a board without one is refused here
        if (!(p.pc[KING] & p.col[WHITE]) || !(p.pc[KING] & p.col[BLACK])) return false;
        ++i;
        p.ctm = fen[i] == 'b' ? BLACK : WHITE;
        i += 2;
        for (; i < fen.size() && fen[i] != ' '; ++i) {
            if (fen[i] == 'K') p.castle |= CASTLE_WK;
            if (fen[i] == 'Q') p.castle |= CASTLE_WQ;
            if (fen[i] == 'k') p.castle |= CASTLE_BK;
            if (fen[i] == 'q') p.castle |= CASTLE_BQ;
        }
        ++i;
        if (i < fen.size() && fen[i] != '-') {
            // an en-passant square is a file letter and rank 3 or 6; anything else is not a FEN
            if (i + 1 >= fen.size() || fen[i] < 'a' || fen[i] > 'h' || (fen[i + 1] != '3' && fen[i + 1] != '6')) return false;
            p.ep = (int8_t)((fen[i] - 'a') + 8 * (fen[i + 1] - '1'));
            i += 2;
        } else ++i;
        if (i < fen.size()) {
            p.halfmove = std::atoi(fen.c_str() + i);
            const size_t sp = fen.find(' ', i + 1);
            if (sp != std::string::npos) p.fullmove = std::atoi(fen.c_str() + sp + 1);
        }
        // the side that has just moved may not have left its king attacked: otherwise the king could be captured and the
        // next position would have none
        if (p.attacked(p.king_sq(!p.ctm), p.ctm)) return false;
        return true;
    }
};

// ---- the 73-plane action code, side-to-move point of view (env.h:60-200) ----------------------
// action = 73 * src + type, squares flipped (63 - sq) when black is to move.
//   type 0..55: queen-like move, direction N S E W NE NW SE SW (x7) + (distance - 1)
//   type 56..63: knight move, order W-NW, N-NW, E-NE, N-NE, W-SW, S-SW, E-SE, S-SE
//   type 64..72: pawn under-promotion: piece (N, B, R) x 3 + (NW, N, NE)
inline int encode_action(const Position& p, Move m)
{
    int src = m.src, dst = m.dst;
    const int type = p.type_at(m.src);
    if (p.ctm == BLACK) { src = 63 - src; dst = 63 - dst; }
    const int dr = (dst >> 3) - (src >> 3), df = (dst & 7) - (src & 7);
    if (type == PAWN && m.promo && m.promo != QUEEN) {
        const int base = m.promo == KNIGHT ? 1 : (m.promo == BISHOP ? 4 : 7);
        return 73 * src + 64 + df + base;
    }
    if (type == KNIGHT) {
        int ind = 0;
        if (dr < 0) ind += 4;
        if (df > 0) ind += 2;
        ind += std::abs(dr) - 1;
        return 73 * src + 56 + ind;
    }
    const int dist = std::max(std::abs(dr), std::abs(df)) - 1;
    int dir;
    if (df == 0) dir = dr > 0 ? 0 : 1;
    else if (dr == 0) dir = df > 0 ? 2 : 3;
    else if (dr > 0) dir = df > 0 ? 4 : 5;
    else dir = df > 0 ? 6 : 7;
    return 73 * src + 7 * dir + dist;
}

inline Move decode_action(const Position& p, int action)
{
    int src = action / 73, dst;
    const int t = action % 73;
    int promo = 0;
    if (t < 56) {
        static const int step[8] = { 8, -8, 1, -1, 9, 7, -7, -9 };
        dst = src + step[t / 7] * (t % 7 + 1);
    } else if (t < 64) {
        static const int step[8] = { -1 + 7, 8 + 7, 1 + 9, 8 + 9, -1 - 9, -8 - 9, 1 - 7, -8 - 7 };
        dst = src + step[t - 56];
    } else {
        static const int step[3] = { 7, 8, 9 };
        static const int piece[3] = { KNIGHT, BISHOP, ROOK };
        dst = src + step[(t - 64) % 3];
        promo = piece[(t - 64) / 3];
    }
    if (p.ctm == BLACK) { src = 63 - src; dst = 63 - dst; }
    // REFERENCE QUIRK (env.h:178-199): a queen-like code decodes to a move WITHOUT a promotion piece, so
    // a pawn "promoted to a queen" through Env::push stays a pawn on its last rank for the rest of the
    // game (2 881 of the 14 500 plies of tests/golden/games.npz show one).  Reproduced, not fixed.
    return Move{ (uint8_t)src, (uint8_t)dst, (uint8_t)promo };
}

inline std::string uci(Move m)
{
    std::string s;
    s += (char)('a' + (m.src & 7)); s += (char)('1' + (m.src >> 3));
    s += (char)('a' + (m.dst & 7)); s += (char)('1' + (m.dst >> 3));
    if (m.promo) s += "nbrq"[m.promo - KNIGHT];
    return s;
}

inline uint64_t perft(const Position& p, int depth)
{
    Move mv[MAX_MOVES];
    const int n = p.legal(mv);
    if (depth <= 1) return depth == 1 ? (uint64_t)n : 1;
    uint64_t t = 0;
    for (int i = 0; i < n; ++i) {
        Position q = p;
        q.make(mv[i]);
        t += perft(q, depth - 1);
    }
    return t;
}

}  // namespace chess
}  // namespace kami
