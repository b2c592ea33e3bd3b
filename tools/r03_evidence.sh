#!/bin/bash
# Round-3 evidence files for profiles/ in one GPU-box run (after tools/build_diag.sh): the split-channel tower, the heads inside
# tower128's launch, the completion word, the queue's settings.  Outputs under gpurun_out/ev/.
O=gpurun_out/ev; rm -rf $O; mkdir -p $O
{
  echo "# tower2s_kernel (20x256 f16, two workgroups per board pair) against tower2b_kernel (KAMI_WIDE_VARIANT=6) and its own timing-only ablation (KAMI_T2S_ABL=1: no global exchange)"
  for v in '6 0' '7 0' '7 1' '7 0' '6 0'; do set -- $v; KAMI_WIDE_VARIANT=$1 KAMI_T2S_ABL=$2 timeout -k 10 100 python tools/t2s_time.py 256 || exit 1; done
  echo "# every batch: the default path (tower2s up to 256) against the per-layer kernels (KAMI_WIDE_VARIANT=1)"
  for b in 1 32 128 254; do for v in 1 0; do if [ $v = 0 ]; then unset KAMI_WIDE_VARIANT; else export KAMI_WIDE_VARIANT=$v; fi; timeout -k 10 100 python tools/t2s_time.py $b || exit 1; done; done; unset KAMI_WIDE_VARIANT
  echo "# outputs against tower2b's summation order (per-layer kernels at batch 255), twenty blocks"
  KAMI_WIDE_VARIANT= timeout -k 10 200 python - <<'PY'
import os, subprocess, sys
os.environ.pop("KAMI_WIDE_VARIANT", None)
PY
  timeout -k 10 200 python tools/split_check.py 20 255 f16 | sed 's/default/variant 0 (default = split)/'
  echo "# in-kernel stamps (diagnostic build)"
  timeout -k 10 100 python tools/t2s_stamps.py
} > $O/tower2s.txt 2>&1
{
  echo "# tower128_kernel<T, HEAD>: 10x128 bf16 batch 1024, heads inside the launch (default) against policy_head4_kernel behind it (KAMI_T128_HEAD=0)"
  for hd in 1 0 1 0; do echo "KAMI_T128_HEAD=$hd"; KAMI_T128_HEAD=$hd timeout -k 10 100 python tools/wide_time.py 128 10 1024 bf16 || exit 1; done
  echo "# phases (diagnostic build)"
  timeout -k 10 200 python tools/t128_stamps.py 128 1024 2>&1 | cut -c1-700
} > $O/t128_head.txt 2>&1
{ echo "# tools/completion_probe.hip: a 1 us kernel; when does the host learn that it has finished?"; timeout -k 10 60 ./tools/completion_probe.bin; } > $O/completion_probe.txt 2>&1
{
  echo "# configs[1] literally (256 games, 800 visits, two leaves per tree) under queue settings: threads pipeline callers(0 = half the workers) inflight target/wait"
  printf '14 1 0 4 512 80\n14 1 0 4 512 80\n14 1 0 4 512 80\n14 1 14 4 512 80\n14 1 14 4 512 80\n14 1 0 2 512 80\n14 3 0 4 512 80\n14 4 0 4 512 80\n15 1 0 4 512 80\n' | bash tools/sp_sweep.sh
} > $O/queue_sweep.txt 2>&1
tail -3 $O/tower2s.txt; tail -2 $O/t128_head.txt | cut -c1-200; tail -2 $O/queue_sweep.txt | cut -c1-160
