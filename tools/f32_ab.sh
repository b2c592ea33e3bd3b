#!/bin/bash
# Round 1's tree (git worktree of its last commit, built there: ab_r1/) against this tree, the exact-fp32 path of the
# headline step, alternating on ONE box: is 0.51 (round 1) -> 0.47 (rounds 2-3) the kernel or the box?
# Set-up (in the build container):  git worktree add -f ab_r1 b4a4551 && (cd ab_r1 && python -m kami_amd.build)
# (ab_r1/ travels to the GPU box with the snapshot; remove it afterwards: git worktree remove --force ab_r1)
for i in 1 2 3; do
  for d in ab_r1 .; do
    ( cd $d && python bench.py --dtype f32 --steps 300 --warmup 50 --no-cpu-baseline $( [ $d = . ] && echo --no-variants --no-legs ) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d'.ljust(6), 'f32 %.3f M evals/s  %.4f ms  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac']))" ) || exit 1
  done
done
