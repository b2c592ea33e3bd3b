#!/bin/bash
# Round 2's tree (git worktree add -f ab_r2 078cf80 && (cd ab_r2 && python -m kami_amd.build)) against this one on ONE box:
# the side benches, alternating.  A regression hunt (round 3 found the fp32 path's this way).
R=$PWD
for t in host_path_bench encode_bench train_bench; do
  for d in ab_r2 . ab_r2 .; do
    echo "=== $t [$d]"
    ( cd $R/$d && GRAFT_REPO_ROOT=$R/$d timeout -k 10 280 python tools/$t.py 2>&1 | tail -14 ) || exit 1
  done
done
