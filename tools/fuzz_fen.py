"""Mutated FENs through the host rules (parser, move generation, perft 2) — run by tools/sanitize.sh against an
ASan / UBSan build of libkamisearch.so:  python tools/fuzz_fen.py <libkamisearch.so> <seed> <iterations>"""
import sys, os, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kami_amd import search as S
S.LIB_PATH = os.path.abspath(sys.argv[1])
rnd = random.Random(int(sys.argv[2]))
alphabet = "rnbqkpRNBQKP12345678/ wb-KQkqabcdefgh0123456789"
bases = ["r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", "rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f6 0 3",
         "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8"]
parsed = refused = 0
for it in range(int(sys.argv[3])):
    f = list(rnd.choice(bases))
    for _ in range(rnd.choice((1, 2, 3, 8))):
        op = rnd.random()
        if op < 0.7: f[rnd.randrange(len(f))] = rnd.choice(alphabet)
        elif op < 0.85 and len(f) > 1: del f[rnd.randrange(len(f))]
        else: f.insert(rnd.randrange(len(f) + 1), rnd.choice(alphabet))
    try:
        S.fen_actions("".join(f)); parsed += 1
        if parsed % 5 == 0: S.perft("".join(f), 2)
    except (ValueError, RuntimeError):
        refused += 1
print("seed", sys.argv[2], "parsed", parsed, "refused", refused)
