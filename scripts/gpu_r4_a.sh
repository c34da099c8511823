#!/bin/bash
# round-4 pass A: full GPU suite (with the model-scale fixtures), smoke, the default bench line
TAG=${1:-r4a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rP > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -5
grep -E "^\[parity\].*(cfg3|cfg4|512|accuracy|all 24|CLIP|clip|projector)" gpurun_out/${TAG}_gpu_tests.log | cut -c1-260
grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/${TAG}_gpu_tests.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -25 gpurun_out/${TAG}_bench.err | cut -c1-220
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
        for k in ("accuracy_mode", "vq_hbm", "other_configs", "train_step", "cpu_baseline"):
            print(k, json.dumps(d.get(k))[:900])
        print("measured_peak", d["roofline"].get("measured_peak"))
PY
