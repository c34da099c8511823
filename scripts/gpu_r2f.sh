#!/bin/bash
# round-2 GPU pass F: full parity suite, smoke, t2i bench (graph default + roofline leg + cpu baseline), train bench.
TAG=${1:-r2f}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "parity|passed|failed|Error|assert|FAILED|skipped" | cut -c1-250 > gpurun_out/tests_$TAG.log
tail -25 gpurun_out/tests_$TAG.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
grep -E '"metric"|rc=' gpurun_out/bench_$TAG.log | cut -c1-1500
timeout 900 python bench.py --workload train --steps 3 --warmup 1 > gpurun_out/bench_train_$TAG.log 2>&1; echo "train rc=$?" >> gpurun_out/bench_train_$TAG.log
grep -E '"metric"|rc=|Error' gpurun_out/bench_train_$TAG.log | cut -c1-1800
