#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/kami_ref).

Run in the build container only (needs /root/reference to have built oracle/_ref):
    make -C oracle && python oracle/gen_golden.py
The fixtures hold data only: seeds/weights/inputs we generate and the outputs the reference
computed for them (NN::infer nn.cpp:155-187, NNModule::forward nn.cpp:59-91,
Env::observe env.h:202-262, Env::actions env.h:398-423).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kami_amd import weights as W  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "kami_ref")
OUT = os.path.join(ROOT, "tests", "golden")

# name, F, C, R, B, seed, peaky, input kind
NETS = [
    ("net_f30_c8_r1", 30, 8, 1, 3, 11, 1.0, "uniform"),
    ("net_f30_c16_r2_peaky", 30, 16, 2, 4, 12, 30.0, "uniform"),
    ("net_f30_c64_r2", 30, 64, 2, 2, 13, 1.0, "uniform"),
    ("net_f119_c32_r1_peaky", 119, 32, 1, 2, 14, 30.0, "uniform"),
    ("net_f30_c64_r1_planes", 30, 64, 1, 5, 15, 30.0, "planes"),
    # B > 256 exercises the second row of the Q10 value copy-out (nn.cpp:186)
    ("net_f30_c8_r0_b300", 30, 8, 0, 300, 16, 30.0, "u8"),
]


def run_infer(tmp, blob, F, C, R, gen, x):
    wpath = os.path.join(tmp, "w.bin")
    W.save(wpath, blob, F, C, R, gen)
    x.astype("<f4").tofile(os.path.join(tmp, "x.f32"))
    outs = [os.path.join(tmp, n) for n in ("p.f32", "v.f32", "vf.f32")]
    subprocess.check_call([REF, "infer", wpath, os.path.join(tmp, "x.f32"), str(x.shape[0])] + outs)
    B = x.shape[0]
    p = np.fromfile(outs[0], "<f4").reshape(B, W.PSIZE)
    v = np.fromfile(outs[1], "<f4").reshape(B)
    vf = np.fromfile(outs[2], "<f4").reshape(B, W.VALUE_WIDTH)
    return p, v, vf


def gen_observe(tmp):
    path = os.path.join(tmp, "obs.bin")
    # 6 random games, up to 330 plies each (continues past draw conditions -> ply > 255)
    subprocess.check_call([REF, "observe", "20240607", "6", "330", path])
    rec = np.dtype([("ply", "<i4"), ("nact", "<i4"), ("fen", "S104"),
                    ("actions", "<i4", (128,)), ("obs", "<f4", (1920,))])
    r = np.fromfile(path, rec)
    obs = r["obs"]
    assert np.all(np.isin(obs, [0, 1, 2, 4, 8])), "observe() produced a value outside {0,1,2,4,8}"
    # thin out: keep every position of game 1 plus every 3rd of the rest, plus all with ply > 250
    keep = np.zeros(len(r), bool)
    first_game_end = int(np.argmax(r["ply"][1:] == 0)) + 1
    keep[:first_game_end] = True
    keep[::3] = True
    keep[r["ply"] > 250] = True
    r = r[keep]
    nmax = int(r["nact"].max())
    np.savez_compressed(os.path.join(OUT, "observe_playouts.npz"),
                        ply=r["ply"], fen=r["fen"], nact=r["nact"],
                        actions=r["actions"][:, :nmax].astype(np.int16),
                        obs=r["obs"].astype(np.uint8))
    print("observe fixtures:", len(r), "positions; max ply", r["ply"].max(),
          "black-to-move", int((r["ply"] % 2 == 1).sum()))
    return r


def gen_search(tmp):
    """Fixtures for the host search row: played games with the reference's terminal
    verdict at every ply (Env::terminal, env.h:286-384) and the reference's own MCTS (kami/mcts.h)
    run under a deterministic synthetic evaluator with the root noise switched off."""
    path = os.path.join(tmp, "games.bin")
    # 40 random games, up to 400 plies each, played on through the draw verdicts while a move exists
    subprocess.check_call([REF, "games", "7", "40", "400", path])
    rec = np.dtype([("ply", "<i4"), ("action", "<i4"), ("terminal", "<i4"), ("value", "<f4"), ("fen", "S104")])
    r = np.fromfile(path, rec)
    np.savez_compressed(os.path.join(OUT, "games.npz"), ply=r["ply"].astype(np.int16), action=r["action"].astype(np.int16),
                        terminal=r["terminal"].astype(np.int8), value=r["value"], fen=r["fen"])
    print("games fixture:", len(r), "plies,", int((r["ply"] == 0).sum()), "games,", int(r["terminal"].sum()), "terminal verdicts,",
          "values", sorted(set(r["value"][r["terminal"] == 1].tolist())))
    for nodes, nmoves, name in ((300, 12, "mcts_ref_300x12.txt"), (64, 120, "mcts_ref_64x120.txt")):
        subprocess.check_call([REF, "mcts", str(nodes), str(nmoves), os.path.join(OUT, name)])
        print("mcts fixture:", name, os.path.getsize(os.path.join(OUT, name)), "bytes")


def gen_train(tmp):
    """NN::train (nn.cpp:224-377) on the reference: trained parameters after a few SGD epochs.
    One batch per epoch (n == training_batchsize) on purpose: on the CPU device the reference's batch
    tensors of an epoch all alias ONE stack buffer (`from_blob(next_input).to(kCPU)` does not copy,
    nn.cpp:296-312), so with several batches per epoch every step of the epoch would train on the last
    batch built; on its CUDA path `.to(device)` copies and the batches are distinct.  kh_train follows the
    CUDA path, and single-batch epochs are where the two coincide."""
    cases = (("train_f30_c16_r1", 30, 16, 1, 8, 8, 4, 5, 11), ("train_f30_c8_r2", 30, 8, 2, 6, 6, 5, 20, 12),
             # kami's default width (options.def.yml: filters 64): the shapes the matrix-core training kernels tile for
             ("train_f30_c64_r2", 30, 64, 2, 8, 8, 3, 5, 13))
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--train-case=")]
    for name, F, C, R, n, tbatch, epochs, mlr, seed in cases:
        if only and name not in only:
            continue
        rng = np.random.default_rng(seed)
        blob = W.random_weights(F, C, R, seed=seed, peaky=3.0)
        x = (rng.integers(0, 256, (n, 8, 8, F)).astype(np.float32) / 256.0)
        # visit distributions over ~30 random "legal" actions per sample, targets in {-1, 0, 1}
        idx = np.stack([rng.choice(4672, 32, replace=False) for _ in range(n)]).astype(np.int16)
        val = rng.random((n, 32)).astype(np.float32)
        val /= val.sum(1, keepdims=True)
        obs_p = np.zeros((n, 4672), np.float32)
        for i in range(n):
            obs_p[i, idx[i]] = val[i]
        obs_v = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), n)
        wpath = os.path.join(tmp, "w.bin")
        W.save(wpath, blob, F, C, R, 3)
        paths = [os.path.join(tmp, f) for f in ("x.f32", "p.f32", "v.f32", "out.f32")]
        x.tofile(paths[0]); obs_p.tofile(paths[1]); obs_v.tofile(paths[2])
        subprocess.check_call([REF, "train", wpath, str(n), paths[0], paths[1], paths[2], str(mlr), str(epochs), str(tbatch), paths[3]],
                              stdout=subprocess.DEVNULL)
        trained = np.fromfile(paths[3], "<f4")
        assert trained.size == blob.size and np.isfinite(trained).all()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), features=F, filters=C, residuals=R, blob=blob, trained=trained,
                            x_u8=(x * 256).astype(np.uint8), obs_idx=idx, obs_val=val, obs_v=obs_v,
                            tbatch=tbatch, epochs=epochs, mlr=mlr)
        print("train fixture:", name, "max |delta w|", float(np.abs(trained - blob).max()))


def gen_checkpoint(tmp):
    """A checkpoint exactly as kami leaves it on disk: the blob of the net_f30_c8_r1 fixture pushed through the
    reference's NN::read and written by its own NN::write (nn.cpp:189-222; `kami_ref export`), and the same
    file converted back by the reference (`kami_ref convert`) as the cross-check of the cross-check."""
    name, F, C, R, B, seed, peaky, kind = NETS[0]
    blob = W.random_weights(F, C, R, seed=seed, peaky=peaky)
    wpath = os.path.join(tmp, "w.bin")
    W.save(wpath, blob, F, C, R, 7)
    out = os.path.join(OUT, f"ref_checkpoint_f{F}_c{C}_r{R}.pt")
    subprocess.check_call([REF, "export", wpath, out], stdout=subprocess.DEVNULL)
    back = os.path.join(tmp, "back.bin")
    subprocess.check_call([REF, "convert", out, str(F), str(C), str(R), back], stdout=subprocess.DEVNULL)
    b2, F2, C2, R2, g2 = W.load(back)
    assert (F2, C2, R2, g2) == (F, C, R, 7) and np.array_equal(b2, blob)
    print("checkpoint fixture:", out, os.path.getsize(out), "bytes")


def main():
    os.makedirs(OUT, exist_ok=True)
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/kami_ref missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as tmp:
        if "--search-only" in sys.argv:
            gen_search(tmp)
            return
        if "--train-only" in sys.argv:
            gen_train(tmp)
            return
        if "--checkpoint-only" in sys.argv:
            gen_checkpoint(tmp)
            return
        gen_search(tmp)
        gen_train(tmp)
        gen_checkpoint(tmp)
        recs = gen_observe(tmp)
        planes = recs["obs"].reshape(-1, 8, 8, 30)
        for name, F, C, R, B, seed, peaky, kind in NETS:
            blob = W.random_weights(F, C, R, seed=seed, peaky=peaky)
            rng = np.random.default_rng(seed + 1000)
            if kind == "uniform":
                x = rng.random((B, 8, 8, F), dtype=np.float32)
            elif kind == "u8":  # k/256 grid: exact in fp32, stored as uint8 (x_u8) to keep the file small
                x = (rng.integers(0, 256, (B, 8, 8, F)).astype(np.float32) / 256.0)
            else:  # real encoder planes (values 0/1/2/4/8), as the reference feeds its net
                idx = rng.choice(len(planes), B, replace=False)
                x = planes[idx].astype(np.float32)
            gen = 7
            p, v, vf = run_infer(tmp, blob, F, C, R, gen, x)
            assert np.allclose(p.sum(1), 1.0, atol=1e-4)
            # big batches: keep the policy rows around the 256-row seam only (file size)
            rows = np.arange(B) if B <= 16 else np.array([0, 1, 255, 256, 257, B - 1])
            np.savez_compressed(os.path.join(OUT, name + ".npz"),
                                features=F, filters=C, residuals=R, generation=gen,
                                seed=seed, peaky=peaky, blob=blob,
                                **({"x_u8": (x * 256).astype(np.uint8)} if kind == "u8" else {"x": x}),
                                policy_rows=rows, policy=p[rows], value=v, value_full=vf)
            print(name, "policy range", p.min(), p.max(), "value[0..2]", v[:3])


if __name__ == "__main__":
    main()
