#!/bin/bash
# round-4 pass C: per-block gates (rms 1e-3 / max 2e-3 + flip-floor record), exchange-contention table, rocprofv3 kernel tables of the
# t2i bench (batch 8 and batch 1) and of the training step
TAG=${1:-r4c}
R=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modules_gpu.py -m gpu -q -rP -k "tiny_forward_blockwise or full_size_logits or cfg3" > gpurun_out/${TAG}_mod_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_mod_tests.log | tail -3; grep -E "^\[parity\].*(per-block|flip floor)" gpurun_out/${TAG}_mod_tests.log | cut -c1-900; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_mod_tests.log | head -12
timeout 600 python tools/exchange_contention.py > gpurun_out/${TAG}_exchange_contention.txt 2> gpurun_out/${TAG}_exchange_contention.err; echo "contention rc=$?"; cat gpurun_out/${TAG}_exchange_contention.txt; tail -3 gpurun_out/${TAG}_exchange_contention.err
export TMPDIR=/tmp
cd /tmp
COMMON="--no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs --roofline-steps 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 1 --warmup 1 $COMMON > $R/gpurun_out/prof_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_b1 -o prof -- python $R/bench.py --batch 1 --steps 2 --warmup 1 $COMMON > $R/gpurun_out/prof_${TAG}_b1.log 2>&1
SHOWO_GEMM_TUNE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_train -o prof -- python $R/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --no-events > $R/gpurun_out/prof_${TAG}_train.log 2>&1
cd $R
find gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_b1 gpurun_out/prof_${TAG}_train -type f ! -name "*stats*" -size +2M -delete
for d in prof_$TAG prof_${TAG}_b1 prof_${TAG}_train; do echo "== $d"; f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-200; done
