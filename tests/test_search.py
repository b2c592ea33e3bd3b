"""Host search row: the rules / MCTS mirrors against the reference's own fixtures.

Fixtures (oracle/gen_golden.py, from the unmodified reference through oracle/ref_harness.cpp):
  games.npz            40 random games: played action, FEN and Env::terminal verdict at every ply
  observe_playouts.npz legal action lists (reference order) + planes of 1149 positions
  mcts_ref_*.txt       kami::MCTS (kami/mcts.h) under a deterministic synthetic evaluator, noise off
CPU only: nothing here needs the GPU (the pool test that does lives in test_gpu_parity.py)."""
import os
import re

import numpy as np
import pytest

from kami_amd import search as S

GOLD = os.path.join(os.path.dirname(__file__), "golden")

PERFT = [   # standard known answers (chessprogramming.org perft results)
    ("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", [20, 400, 8902, 197281]),
    ("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", [48, 2039, 97862]),
    ("8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", [14, 191, 2812, 43238]),
    ("r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", [6, 264, 9467]),
    ("rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8", [44, 1486, 62379]),
    ("r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10", [46, 2079, 89890]),
]


@pytest.mark.parametrize("fen,counts", PERFT)
def test_perft_known_answers(fen, counts):
    for d, want in enumerate(counts, 1):
        assert S.perft(fen, d) == want


def test_games_replay_matches_reference_state_and_terminal_verdicts():
    """Replay the reference's 40 random games move by move: FEN (placement, castle rights, en-passant
    square, half-move clock, move number), the terminal verdict and its value agree at all 14 500
    plies — including the reference's quirks: 50-move draw at 50 plies, repetition draw at the fifth
    occurrence, a queen promotion through Env::push leaving a pawn on the last rank."""
    d = np.load(os.path.join(GOLD, "games.npz"))
    env = None
    checked_terminal = 0
    for ply, action, term, value, fen in zip(d["ply"], d["action"], d["terminal"], d["value"], d["fen"]):
        if ply == 0:
            env = S.Env()
        assert env.ply() == ply
        assert env.print() == fen.decode()
        t, v = env.terminal()
        assert t == bool(term)
        if t:
            assert v == value
            checked_terminal += 1
        if action >= 0:
            env.push(int(action))          # raises if my rules call the reference's move illegal
    assert checked_terminal == int(d["terminal"].sum()) > 1000


def test_legal_action_sets_match_reference():
    """Env::actions (env.h:397-423): all 1149 positions of observe_playouts.npz — legal moves generated from
    the FEN, encoded with the side-to-move action code, equal the reference's list as a SET (castling,
    en passant, under-promotions, checks and pins; order differs by design, see kami_amd/host/env.h)."""
    o = np.load(os.path.join(GOLD, "observe_playouts.npz"))
    black = 0
    for fen, n, acts in zip(o["fen"], o["nact"], o["actions"]):
        mine = S.fen_actions(fen.decode())
        assert len(mine) == int(n)
        assert set(mine) == set(int(a) for a in acts[:n]), fen
        assert mine == sorted(mine)
        black += b" b " in fen
    assert black > 400


def test_env_actions_along_games_contain_the_played_move():
    d = np.load(os.path.join(GOLD, "games.npz"))
    env = None
    for ply, action, fen in zip(d["ply"][:3000], d["action"][:3000], d["fen"][:3000]):
        if ply == 0:
            env = S.Env()
        acts = env.actions()
        assert acts == S.fen_actions(fen.decode())          # history-free and in-game generation agree
        if action >= 0:
            assert int(action) in acts
            env.push(int(action))


def parse(text):
    moves = []
    for line in text.strip().splitlines():
        w = line.split()
        if w[0] == "move":
            moves.append({"fen": " ".join(w[3:]), "children": {}})
        elif w[0] == "root":
            moves[-1]["n"] = int(w[2]); moves[-1]["w"] = float(w[4])
        elif w[0] == "child":
            moves[-1]["children"][int(w[1])] = (int(w[3]), float(w[5]), float(w[7]))
        elif w[0] == "pick":
            moves[-1]["pick"] = int(w[1])
        elif w[0] == "terminal":
            moves[-1]["terminal"] = float(w[1])
    return moves


@pytest.mark.parametrize("name,nodes,nmoves", [("mcts_ref_300x12.txt", 300, 12), ("mcts_ref_64x120.txt", 64, 120)])
def test_mcts_matches_reference_search(name, nodes, nmoves):
    """kami::MCTS (kami/mcts.h): same visit counts, accumulated values and priors as the reference's own
    search on every move of the fixture — PUCT rule with its float/double mix, the q() default for
    unvisited children, backprop's 0.5 + v*turn/2, the value sign convention, terminal leaves, tree
    reuse across moves.  The reference's picks are replayed (ties in the visit count resolve by child
    order, and the reference orders children by its move-ordering heuristic)."""
    ref = parse(open(os.path.join(GOLD, name)).read())
    picks = [m["pick"] for m in ref]
    mine = parse(S.mcts_synthetic(nodes, nmoves, 1, picks))
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        assert a["fen"] == b["fen"]
        assert a["n"] == b["n"]
        assert a["w"] == pytest.approx(b["w"], rel=2e-6)
        assert set(a["children"]) == set(b["children"])
        for act, (n, w, p) in b["children"].items():
            an, aw, ap = a["children"][act]
            assert an == n, (a["fen"], act)
            assert aw == pytest.approx(w, rel=2e-6, abs=1e-6)
            assert ap == pytest.approx(p, rel=2e-6)
        # the reference's pick is a most-visited child of my tree too
        assert a["children"][b["pick"]][0] == max(c[0] for c in a["children"].values())
        assert a.get("terminal") == b.get("terminal")


def test_mcts_several_leaves_in_flight_is_consistent():
    """Virtual visits: with 8 leaves of one tree in flight per step the search still spends exactly
    `nodes` visits per move, every child keeps n >= 0 and sum(n_children) + 1 == n_root, priors unchanged."""
    ref = parse(S.mcts_synthetic(200, 6, 1))
    par = parse(S.mcts_synthetic(200, 6, 8, [m["pick"] for m in ref]))
    for a, b in zip(par, ref):
        assert a["fen"] == b["fen"]
        assert a["n"] >= 200 and a["n"] <= 200 + 8
        assert sum(c[0] for c in a["children"].values()) == a["n"] - 1
        for act, (n, w, p) in a["children"].items():
            assert n >= 0 and p == pytest.approx(b["children"][act][2], rel=1e-6)
        # the two schedules agree on where most of the search went
        top_ref = max(b["children"], key=lambda k: b["children"][k][0])
        assert a["children"][top_ref][0] >= 0.5 * b["children"][top_ref][0]


def test_env_record_is_what_observe_reads():
    """Env::record: the compact kh_board of a position reached by playing moves equals the record the
    oracle derives from the reference's FEN + ply (pinned on the reference's planes in
    test_oracle_golden.py), and its planes (oracle encoder) carry the reference's quirks."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import pyoracle as ko
    d = np.load(os.path.join(GOLD, "games.npz"))
    env, recs, fens, plies = None, [], [], []
    for ply, action, fen in zip(d["ply"][:1500], d["action"][:1500], d["fen"][:1500]):
        if ply == 0:
            env = S.Env()
        recs.append(env.record()[0])
        fens.append(fen.decode()); plies.append(int(ply))
        if action >= 0:
            env.push(int(action))
    mine = np.array(recs)
    want = ko.boards_from_fens(fens, plies)
    for f in ("piece_occ", "color_occ", "ply", "halfmove_clock", "ctm", "castle_rights"):
        assert np.array_equal(mine[f], want[f]), f
    assert np.array_equal(ko.observe(mine), ko.observe(want))


@pytest.mark.parametrize("fen", [
    "rnbq1bnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",       # no black king
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQ1BNR w KQkq - 0 1",       # no white king
    "rnbqkbnr/pppp1ppp/8/4Q3/8/8/PPPP1PPP/RNB1KBNR w KQkq - 0 1",     # white to move while black's king is attacked
    "rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f9 0 3",  # en-passant square off the board
    "rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq z6 0 3",
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR",                    # board only
    "", "k", "8/8/8/8/8/8/8/8 w - - 0 1",
])
def test_positions_the_rules_cannot_play_are_refused(fen):
    """Move generation asks for both kings' squares and assumes the side that just moved is not in check (the reference's
    neocortex assumes the same): the FEN reader refuses anything else instead of running on it (found by fuzzing the
    parser under ASan / UBSan: __builtin_ctzll(0))."""
    with pytest.raises(ValueError):
        S.fen_actions(fen)
    with pytest.raises(ValueError):
        S.perft(fen, 1)
