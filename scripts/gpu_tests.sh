#!/bin/bash
# full GPU suite; the summary line is grepped out because RCCL's exit chatter follows pytest's last lines
TAG=${1:-t}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q ${2:+-k "$2"} > gpurun_out/${TAG}_gpu_tests.log 2>&1
grep -E "passed|failed|error|Error|assert" gpurun_out/${TAG}_gpu_tests.log | tail -15
