#!/bin/bash
# round-2 GPU pass 3F (final): consolidated pass with the final build: full GPU suite, smoke, the four bench lines, rocprof kernel stats
TAG=${1:-r3f}
R=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests_$TAG.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 1500 gpurun_out/bench_$TAG.json | cut -c1-900
for w in train t2i512 mmu; do
timeout 600 python bench.py --workload $w > gpurun_out/bench_${TAG}_$w.json 2> gpurun_out/bench_${TAG}_$w.err
python3 - <<PY
import json
for l in open("gpurun_out/bench_${TAG}_$w.json"):
    if l.startswith("{"):
        d = json.loads(l); print("$w", round(d["value"], 2), d["unit"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["unit"], d["roofline"]["frac"])
PY
done
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --roofline-steps 0 > $R/gpurun_out/prof_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_mmu -o prof -- python $R/bench.py --workload mmu --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_mmu.log 2>&1
cd $R
find gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_mmu -type f ! -name "*stats*" -size +2M -delete
head -8 gpurun_out/prof_$TAG/prof_kernel_stats.csv | cut -c1-170
head -8 gpurun_out/prof_${TAG}_mmu/prof_kernel_stats.csv | cut -c1-170
