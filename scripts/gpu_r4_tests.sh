#!/bin/bash
# full GPU suite + smoke on the last commit of the round
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r4_final_gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r4_final_gpu_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r4_final_gpu_tests.log | head -10
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
