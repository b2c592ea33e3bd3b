#!/bin/bash
# per-kernel times of the exact-fp32 headline step: round 1's tree (ab_r1/) and this one, same box
export TMPDIR=/tmp
R=$PWD
for d in ab_r1 .; do
  cd $R/$d
  rm -rf /tmp/f32prof; 
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/f32prof -- python3 bench.py --dtype f32 --steps 100 --warmup 20 --prewarm 0 --no-cpu-baseline $( [ $d = . ] && echo --no-variants --no-legs --repeats 1 ) > /tmp/f32prof.log 2>&1 || { tail /tmp/f32prof.log; exit 1; }
  echo "== $d"
  python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/f32prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]:
    print('%-90s calls %6s avg %10.1f ns  %5.1f%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
PY
done
