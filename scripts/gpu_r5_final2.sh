#!/bin/bash
# round-5 closing pass after the decode rewrite (dot2 GEMVs, attention prologue): full GPU suite, smoke, the default bench line as the
# driver runs it, the cfg4 line + its rocprofv3 kernel table.  (GEMM / training / VQ code is unchanged since scripts/gpu_r5_final.sh.)
TAG=${1:-r5zz}
R=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rP > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${TAG}_gpu_tests.log | head -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", round(d["value"], 2), "ms", round(d["ms_per_step"], 1), "frac", round(d["roofline"]["frac"], 4), "train", d["train_step"].get("ms_per_step"))
        print("accuracy", d["accuracy_mode"].get("images_per_s"), (d["accuracy_mode"].get("roofline") or {}).get("frac"))
        print("others", {k: (round(v["value"], 2) if "value" in v else v) for k, v in d["other_configs"].items()})
        print("cfg4", {k: d["other_configs"]["cfg4_mmu_decode"].get(k) for k in ("batch4", "batch1")})
PY
tail -3 gpurun_out/${TAG}_bench.err | cut -c1-200
timeout 600 python bench.py --workload mmu > gpurun_out/${TAG}_mmu_bench.json 2> gpurun_out/${TAG}_mmu_bench.err
grep '"metric"' gpurun_out/${TAG}_mmu_bench.json | cut -c1-600
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_mmu -o prof -- python $R/bench.py --workload mmu --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_mmu.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_mmu -type f ! -name "*stats*" -size +2M -delete
head -12 gpurun_out/prof_${TAG}_mmu/prof_kernel_stats.csv | cut -c1-170
