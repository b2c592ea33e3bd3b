#!/bin/bash
# sclk / power under sustained load (run ON the GPU box): the headline kernel and the two wide-net towers.  The chip
# holds its clock down under MFMA-dense load (MI355X_MICROARCH.md, DVFS give-back): the denser the loop, the lower.
for cfg in "64 6 512 bf16" "128 10 1024 bf16" "256 20 2048 f16"; do
  python tools/load_loop.py $cfg 9 > gpurun_out/clk_load.txt 2>&1 &
  P=$!
  sleep 3
  for i in 1 2 3 4 5; do
    sleep 1
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/  */ /g' | tr '\n' ' '; echo
  done
  wait $P
  cat gpurun_out/clk_load.txt
done
