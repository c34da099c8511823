#!/bin/bash
# round-5 consolidated pass: full GPU suite, smoke, the default bench line as the driver runs it (--steps 20 --warmup 5), the other
# workload lines, rocprofv3 kernel tables (t2i, training, mmu), PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of the bench command
TAG=${1:-r5z}
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
        print("accuracy", d["accuracy_mode"].get("images_per_s"), (d["accuracy_mode"].get("roofline") or {}).get("frac"), "vq_hbm", {k: round(v) for k, v in d["vq_hbm"].items() if isinstance(v, float)})
        print("others", {k: (round(v["value"], 2) if "value" in v else v) for k, v in d["other_configs"].items()})
        print("cfg4", {k: d["other_configs"]["cfg4_mmu_decode"].get(k) for k in ("batch4", "batch1")})
        print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("one_step_seconds_by_threads"))
        print("peak", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["roofline"]["measured_peak"].items() if k != "source"})
PY
tail -4 gpurun_out/${TAG}_bench.err | cut -c1-200
for w in train t2i512 mmu vq; do
timeout 600 python bench.py --workload $w > gpurun_out/${TAG}_${w}_bench.json 2> gpurun_out/${TAG}_${w}_bench.err
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_${w}_bench.json"):
    if l.startswith("{"):
        d = json.loads(l); print("$w", round(d["value"], 2), d["unit"], round(d["ms_per_step"], 2), d["roofline"]["achieved"], d["roofline"]["unit"], d["roofline"]["frac"])
PY
done
timeout 300 python bench.py --batch 1 --steps 10 --warmup 2 --no-train-leg --no-cpu-baseline --no-accuracy-leg --no-config-legs 2>/dev/null | grep '"metric"' > gpurun_out/${TAG}_batch1_bench.json; cut -c 1-160 gpurun_out/${TAG}_batch1_bench.json
export TMPDIR=/tmp
cd /tmp
COMMON="--no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs --roofline-steps 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 2 --warmup 2 $COMMON > $R/gpurun_out/prof_$TAG.log 2>&1
SHOWO_GEMM_TUNE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_train -o prof -- python $R/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --no-events > $R/gpurun_out/prof_${TAG}_train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_mmu -o prof -- python $R/bench.py --workload mmu --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_mmu.log 2>&1
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 400 rocprofv3 --pmc $G --output-format csv -d /tmp/pmc_$i -o pmc -- python $R/bench.py --steps 1 --warmup 1 $COMMON --no-events > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc pass $i ($G) rc=$?"
done
cd $R
python tools/pmc_summary.py $TAG /tmp/pmc_1 /tmp/pmc_2 gpurun_out/pmc_${TAG}_traffic.json | tail -3
python tools/pmc_agg.py gpurun_out/pmc_${TAG}_mfma_agg.json /tmp/pmc_3 > /dev/null
python tools/pmc_mfma.py $TAG gpurun_out/pmc_${TAG}_mfma_agg.json gpurun_out/pmc_${TAG}_mfma_util.json | head -8
find gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_train gpurun_out/prof_${TAG}_mmu -type f ! -name "*stats*" -size +2M -delete
head -8 gpurun_out/prof_$TAG/prof_kernel_stats.csv | cut -c1-170
