// env.h — host-side mirror of kami::Env (kami/env.h:41-485) over the rules in chess.h: same public
// method names and meaning, so the search code reads like the reference's.
//
// Differences, all deliberate:
//   * observe(float*) is replaced by record(kh_board*): the observation leaves the host as the 80-byte
//     compact record and the planes (env.h:202-262) are built on the device (csrc/encode.hip, or inside
//     the forward kernel) — there is no CPU implementation of the encoder in the product.
//   * actions() returns the legal action codes in ASCENDING order; the reference returns them in
//     neocortex's move-ordering heuristic's order (env.h:402-404).  The set is identical
//     (tests/golden/observe_playouts.npz, games.npz); order only matters for exact ties in the search.
//   * pgn() (thc library) and bootstrap_value() (neocortex's static evaluation) are not part of this path.
#pragma once
#include "chess.h"
#include "kami_hip.h"           // include/ of this repository on the include path

#include <string>
#include <vector>

namespace kami {

constexpr int NFEATURES = KH_NFEATURES;     // env.h:19
constexpr int PSIZE = KH_PSIZE;             // env.h:20
constexpr int WIDTH = 8, HEIGHT = 8;
constexpr int OBSIZE = WIDTH * HEIGHT * NFEATURES;

class Env {
    float curturn = 1.0f;
    struct Frame { chess::Position pos; uint64_t key; };
    std::vector<Frame> stack;               // stack.back() is the current position (position.c keeps plies the same way)
    std::vector<int> cur_actions;
    bool actions_utd = false;

public:
    Env() { const chess::Position p = chess::Position::start(); stack.push_back({ p, p.key() }); }

    const chess::Position& position() const { return stack.back().pos; }
    int ply() const { return (int)stack.size() - 1; }                       // env.h:58 (history.size())
    int encode(chess::Move m) const { return chess::encode_action(position(), m); }   // env.h:60-143
    chess::Move decode(int action) const { return chess::decode_action(position(), action); }   // env.h:145-200

    // what Env::observe reads (env.h:202-262), as the engine's compact record
    void record(kh_board* b) const
    {
        const chess::Position& p = position();
        for (int t = 0; t < 6; ++t) b->piece_occ[t] = p.pc[t];
        b->color_occ[0] = p.col[0]; b->color_occ[1] = p.col[1];
        b->ply = ply(); b->halfmove_clock = p.halfmove;
        b->ctm = p.ctm; b->castle_rights = p.castle;
        for (auto& x : b->pad) x = 0;
    }

    void push(int action)                                                    // env.h:264-271
    {
        chess::Position p = position();
        p.apply(decode(action));                                             // (the reference does not test legality here either)
        stack.push_back({ p, p.key() });
        curturn = -curturn;
        actions_utd = false;
    }
    void pop()                                                               // env.h:273-279
    {
        stack.pop_back();
        curturn = -curturn;
        actions_utd = false;
    }
    std::string debug_action(int action) const { return chess::uci(decode(action)); }   // env.h:281-286

    // earlier plies with the current position's key (ncPositionRepCount, position.c:1347-1357)
    int rep_count() const
    {
        // a pawn move or capture cannot be undone, so nothing before the last one can equal the current position
        const size_t n = stack.size(), span = std::min<size_t>(n - 1, (size_t)stack.back().pos.halfmove);
        int c = 0;
        for (size_t i = n - 1 - span; i + 1 < n; ++i) c += stack[i].key == stack.back().key;
        return c;
    }

    bool terminal_str(float* value, std::string& out)                        // env.h:288-384
    {
        using namespace chess;
        const Position& p = position();
        if (p.halfmove >= 50) { *value = 0; out = "Draw by 50-move rule"; return true; }          // (sic: 50 plies)
        if (rep_count() > 3) { *value = 0; out = "Draw by threefold repetition"; return true; }   // (sic: fifth occurrence)
        const uint64_t kings = p.pc[KING], knights = p.pc[KNIGHT], bishops = p.pc[BISHOP], all = p.occ();
        const bool even = popcount(p.col[WHITE]) == popcount(p.col[BLACK]);
        if (kings == all ||
            (all == (kings | bishops) && (popcount(bishops) == 1 || (even && popcount(bishops) == 2))) ||
            (all == (kings | knights) && (popcount(knights) == 1 || (even && popcount(knights) == 2)))) {
            *value = 0; out = "Draw by insufficient material"; return true;
        }
        if (!actions().empty()) return false;
        if (p.in_check()) {
            if (p.ctm == WHITE) { *value = -1.0f; out = "White is checkmated"; }
            else { *value = 1.0f; out = "Black is checkmated"; }
            return true;
        }
        *value = 0.0f;
        out = p.ctm == WHITE ? "White is stalemated" : "Black is stalemated";
        return true;
    }
    bool terminal(float* value) { std::string unused; return terminal_str(value, unused); }   // env.h:386-390

    void prefetch() const { __builtin_prefetch(&stack.back()); __builtin_prefetch(reinterpret_cast<const char*>(&stack.back()) + 64); }

    float turn() const { return curturn; }                                   // env.h:392-395

    std::vector<int>& actions()                                              // env.h:397-423
    {
        if (!actions_utd) {
            chess::Move mv[chess::MAX_MOVES];
            const int n = position().legal(mv);
            // ascending codes by insertion: ~30 codes that arrive in runs (the generator goes piece type by piece
            // type, squares in order), a quarter of std::sort's time here
            int code[chess::MAX_MOVES];
            for (int i = 0; i < n; ++i) {
                const int c = encode(mv[i]);
                int j = i;
                for (; j > 0 && code[j - 1] > c; --j) code[j] = code[j - 1];
                code[j] = c;
            }
            cur_actions.assign(code, code + n);
            // under-promotions and the queen promotion of one pawn move share nothing; but two moves can
            // never share a code either (src and type identify the destination and the piece)
            actions_utd = true;
        }
        return cur_actions;
    }

    std::string print() const { return position().fen(); }                   // env.h:425-430
};

}  // namespace kami
