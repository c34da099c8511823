#!/bin/bash
# round-4 last call: the default bench line on HEAD (driver form; carries the PMC traffic / MFMA fields of this build), rocprofv3 on the
# VQ path's HBM-bound kernels (kernel table + FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload vq`), a short sanity subset of the GPU tests
TAG=${1:-r4y}
R=$(pwd)
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", round(d["value"], 2), "ms", round(d["ms_per_step"], 1), "frac", round(d["roofline"]["frac"], 4), "train", d["train_step"].get("ms_per_step"))
        print("traffic", d["roofline"]["traffic"], "mfma", d["roofline"]["mfma_busy_pmc"])
        print("others", {k: (round(v["value"], 2) if "value" in v else v) for k, v in d["other_configs"].items()})
PY
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_vq -o prof -- python $R/bench.py --workload vq --steps 5 --warmup 1 > $R/gpurun_out/prof_${TAG}_vq.log 2>&1
i=0
for G in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmcvq_$i
  timeout 200 rocprofv3 --pmc $G --output-format csv -d /tmp/pmcvq_$i -o pmc -- python $R/bench.py --workload vq --steps 5 --warmup 1 > $R/gpurun_out/pmcvq_${TAG}_$i.log 2>&1; echo "vq pmc pass $i rc=$?"
done
cd $R
python tools/pmc_summary.py ${TAG}_vq /tmp/pmcvq_1 /tmp/pmcvq_2 gpurun_out/pmc_${TAG}_vq_traffic.json > /dev/null
python3 - <<PY
import json, csv, glob
t = json.load(open("gpurun_out/pmc_${TAG}_vq_traffic.json"))["kernels"]
f = glob.glob("gpurun_out/prof_${TAG}_vq/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("showo::", "").replace("void ", "").split("(")[0]
    if n in t and any(k in n for k in ("lfq", "gn_")):
        v = t[n]; by = (v["fetch_bytes_per_launch"] or 0) + (v["write_bytes_per_launch"] or 0); us = float(r["AverageNs"]) / 1e3
        print(f"{n:28s} {us:9.1f} us  fetch {v['fetch_bytes_per_launch'] / 1e6:8.1f} MB  write {v['write_bytes_per_launch'] / 1e6:8.1f} MB  -> {by / us / 1e3:7.1f} GB/s (rocprof bytes / rocprof time)")
PY
timeout 300 python -m pytest tests -m gpu -q -x -k "splitk_epilogues or magvit_256 or tiny_t2i_generate or cfg4" 2>&1 | grep -E "passed|failed|error" | tail -2
