#!/bin/bash
# Runs ON THE GPU BOX (one gpurun call): rocprofv3 kernel trace + PMC passes of the default bench and of the wide-net
# configurations, the bench itself and the side benches.  Everything lands under gpurun_out/prof/;
# tools/make_profiles.py (run in the build container) digests it into profiles/rNN_*.
# Counters are collected in their own passes (--pmc only, no trace domains), the program directly after `--`.
set -o pipefail
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-variants --no-legs"
SQ="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $B > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --repeats 1 $B > $O/pmc_fetch.log 2>&1 || { tail $O/pmc_fetch.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --repeats 1 $B > $O/pmc_write.log 2>&1 || { tail $O/pmc_write.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 20 --warmup 5 --prewarm 0 --repeats 1 $B > $O/pmc_sq.log 2>&1 || { tail $O/pmc_sq.log; exit 1; }
echo "tower profiles done"
# wide nets: BASELINE configs[2] (10x128, batch 1024, bf16) and configs[4]'s net (20x256, f16) at batch 256 and 2048
for cfg in "128 10 1024 bf16" "256 20 256 f16" "256 20 2048 f16"; do
  set -- $cfg; tag=w$1_b$3
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 tools/wide_profile.py $cfg > $O/kt_$tag.log 2>&1 || { tail $O/kt_$tag.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$tag -- python3 tools/wide_profile.py $cfg 3 > $O/pmcf_$tag.log 2>&1 || { tail $O/pmcf_$tag.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$tag -- python3 tools/wide_profile.py $cfg 3 > $O/pmcw_$tag.log 2>&1 || { tail $O/pmcw_$tag.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $SQ --output-format csv -d $O/pmcs_$tag -- python3 tools/wide_profile.py $cfg 3 > $O/pmcs_$tag.log 2>&1 || { tail $O/pmcs_$tag.log; exit 1; }
  echo "wide $tag done"
done
timeout -k 10 300 python tools/wide_variants.py > $O/wide_variants.txt 2>&1
timeout -k 10 200 python tools/host_path_bench.py > $O/host_path_bench.txt 2>&1
timeout -k 10 200 python tools/pinned_probe.py > $O/pinned_probe.txt 2>&1
timeout -k 10 300 python tools/selfplay_bench.py > $O/selfplay_bench.txt 2>&1
timeout -k 10 300 python tools/train_bench.py > $O/train_bench.txt 2>&1
KAMI_TRAIN_VALU=1 timeout -k 10 300 python tools/train_bench.py > $O/train_bench_valu.txt 2>&1
timeout -k 10 120 python tools/encode_bench.py > $O/encode_bench.txt 2>&1
# keep only the small csv summaries of the rocprof runs (the merge-back limit is 64 MiB)
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*_agent_info.csv" -delete 2>/dev/null
for f in $(find $O -name "*kernel_trace.csv"); do d=$(dirname $f); case $d in *kt) ;; *) rm -f $f;; esac; done
du -sh $O | tail -1
