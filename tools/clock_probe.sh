#!/bin/bash
# sclk / power under the headline kernel's sustained load (run ON the GPU box): the same kernel measures
# 33.6-36.0 us on different boxes of the pool
python tools/ab_bench.py 0 80 > gpurun_out/clk_ab.txt 2>&1 &
P=$!
for i in $(seq 1 14); do
  sleep 1
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo
done
wait $P
cat gpurun_out/clk_ab.txt
