#!/bin/bash
# round-3 consolidated pass: full GPU suite, smoke, the bench lines (headline with train_step + cpu baselines, train, t2i512, mmu, vq,
# batch 1), rocprofv3 kernel tables of the t2i / training / mmu benches, PMC traffic of the bench command
TAG=${1:-r3k}
R=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 2500 gpurun_out/${TAG}_bench.json | cut -c1-1200
for w in train t2i512 mmu vq; do
timeout 600 python bench.py --workload $w > gpurun_out/${TAG}_${w}_bench.json 2> gpurun_out/${TAG}_${w}_bench.err
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_${w}_bench.json"):
    if l.startswith("{"):
        d = json.loads(l); print("$w", round(d["value"], 2), d["unit"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["unit"], d["roofline"]["frac"])
PY
done
timeout 300 python bench.py --batch 1 --steps 10 --warmup 2 --no-train-leg --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/${TAG}_batch1_bench.json; cut -c 1-160 gpurun_out/${TAG}_batch1_bench.json
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --roofline-steps 0 > $R/gpurun_out/prof_$TAG.log 2>&1
SHOWO_GEMM_TUNE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_train -o prof -- python $R/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --no-events > $R/gpurun_out/prof_${TAG}_train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_mmu -o prof -- python $R/bench.py --workload mmu --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_mmu.log 2>&1
cd $R
find gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_train gpurun_out/prof_${TAG}_mmu -type f ! -name "*stats*" -size +2M -delete
head -8 gpurun_out/prof_$TAG/prof_kernel_stats.csv | cut -c1-170
head -12 gpurun_out/prof_${TAG}_train/prof_kernel_stats.csv | cut -c1-170
if [ "${SKIP_PMC:-0}" != "1" ]; then bash scripts/gpu_pmc3.sh $TAG 2>&1 | tail -3; fi
# same-box A/B of the training step: round-2 weight-gradient path (transposes + k-contiguous GEMM) and unfused GroupNorm statistics
for cfg in "SHOWO_TRAIN_TN=1" "SHOWO_TRAIN_TN=0" "SHOWO_CONV_GN_FUSE=0"; do
  env $cfg timeout 400 python bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_train_ab.log 2>&1
  echo "$cfg $(grep -h '"metric"' gpurun_out/${TAG}_train_ab.log | tail -1 | cut -c 75-130)" | tee -a gpurun_out/${TAG}_train_ab.txt
done
