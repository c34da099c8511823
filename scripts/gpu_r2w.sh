#!/bin/bash
# round-2 GPU pass W: clock / power trace next to the bench workloads (tools/power_trace.py)
TAG=${1:-r2w}
mkdir -p gpurun_out
timeout 20 rocm-smi --showbus --json 2>&1 | cut -c1-300; for d in /sys/class/drm/card*/device; do [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ] && echo "$d -> $(realpath $d)"; done
timeout 20 rocm-smi --showclocks --showpower --json 2>&1 | cut -c1-600
timeout 20 rocm-smi --showmaxpower 2>&1 | grep -i -E "power|watt" | head -3
timeout 400 python tools/power_trace.py gpurun_out/power_${TAG}_t2i.json -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/power_${TAG}_t2i.log 2>&1; tail -1 gpurun_out/power_${TAG}_t2i.log | cut -c1-900
timeout 400 python tools/power_trace.py gpurun_out/power_${TAG}_train.json -- python bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/power_${TAG}_train.log 2>&1; tail -1 gpurun_out/power_${TAG}_train.log | cut -c1-900
timeout 400 python tools/power_trace.py gpurun_out/power_${TAG}_mmu.json -- python bench.py --workload mmu --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/power_${TAG}_mmu.log 2>&1; tail -1 gpurun_out/power_${TAG}_mmu.log | cut -c1-900
