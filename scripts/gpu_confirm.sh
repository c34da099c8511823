#!/bin/bash
# round-2 GPU pass 3L: confirmation of the last build: full GPU suite, smoke, headline line (with train_step and cpu_baseline), train line
TAG=${1:-r3l}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/tests_$TAG.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
timeout 600 python bench.py --workload train > gpurun_out/bench_${TAG}_train.json 2> gpurun_out/bench_${TAG}_train.err
python3 - <<PY
import json
for f in ("gpurun_out/bench_$TAG.json", "gpurun_out/bench_${TAG}_train.json"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f, round(d["value"], 2), d["unit"], round(d["roofline"]["achieved"]), d["roofline"]["unit"], (d.get("train_step") or {}).get("ms_per_step"), (d.get("cpu_baseline") or {}).get("value"))
PY
