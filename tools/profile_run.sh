#!/bin/bash
# Runs ON THE GPU BOX (one gpurun call): GPU tests, smoke, rocprofv3 kernel trace + PMC passes of the
# default bench, the bench variants and the side benches.  Everything lands under gpurun_out/prof/;
# tools/make_profiles.py (run in the build container) digests it into profiles/rNN_*.
set -o pipefail
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline"          # default K / W / clock pre-warm: the command the driver runs
timeout -k 10 400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
cat $O/bench_default.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $B > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || { tail $O/pmc_fetch.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --no-cpu-baseline > $O/pmc_write.log 2>&1 || { tail $O/pmc_write.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --no-cpu-baseline > $O/pmc_sq.log 2>&1 || { tail $O/pmc_sq.log; exit 1; }
timeout -k 10 120 python bench.py $B --dtype f16 > $O/bench_f16.json 2>/dev/null
timeout -k 10 120 python bench.py $B --features 30 > $O/bench_f30.json 2>/dev/null
timeout -k 10 120 python bench.py $B --batch 2048 > $O/bench_b2048.json 2>/dev/null
timeout -k 10 120 python bench.py $B --dtype f32 > $O/bench_f32.json 2>/dev/null
timeout -k 10 120 python bench.py $B --filters 128 --residuals 10 --batch 1024 > $O/bench_10x128_b1024.json 2>/dev/null
timeout -k 10 120 python bench.py $B --filters 256 --residuals 20 --batch 256 --dtype f16 > $O/bench_20x256_b256_f16.json 2>/dev/null
timeout -k 10 120 python tools/fused_bench.py > $O/fused_bench.txt 2>&1
timeout -k 10 120 python tools/encode_bench.py > $O/encode_bench.txt 2>&1
timeout -k 10 200 python tools/host_path_bench.py > $O/host_path_bench.txt 2>&1
timeout -k 10 200 python tools/selfplay_bench.py > $O/selfplay_bench.txt 2>&1
timeout -k 10 200 python tools/train_bench.py > $O/train_bench.txt 2>&1
timeout -k 10 200 python tools/wide_ab.py > $O/wide_bench.txt 2>&1
# keep only the small csv summaries of the rocprof runs (the merge-back limit is 64 MiB)
for w in "200 20 0" "2000 20 0" "2000 2000 0" "200 20 0.3"; do set -- $w; timeout -k 10 100 python bench.py --no-cpu-baseline --steps $1 --warmup $2 --prewarm $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps', d['steps'], 'warmup', d['warmup'], 'prewarm_s', $3, 'us_per_step', round(d['ms_per_step']*1e3, 2))"; done > $O/clock_ramp.txt
cat $O/clock_ramp.txt
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $O | tail -1
