#!/bin/bash
# one gpurun call: configs[1] literally (256 games, 800 visits, two leaves per tree) under queue settings given as
# "threads pipeline callers inflight target wait" lines on stdin (callers 0: the pool's own value = half its workers)
while read t p c f tg w; do
  [ -z "$t" ] && continue
  echo "== threads $t pipeline $p callers $c inflight $f target $tg/$w"
  SP_CASE=256,$t,2,800,$p,$tg,$w KAMI_CO_CALLERS=$c KAMI_CO_INFLIGHT=$f KAMI_CO_TRACE=1 timeout -k 10 100 python tools/selfplay_bench.py 2>&1 | cut -c1-250 || exit 1
done
