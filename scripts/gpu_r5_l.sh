#!/bin/bash
# round 5, pass L: same-box A/B of the bf16 packing (compiler conversion = shipped library vs round-4 inline asm = libshowo_hip_asm.so)
mkdir -p gpurun_out/r5l
for tag in cvt asm cvt2 asm2; do
  case $tag in cvt*) envs="A=1";; asm*) envs="SHOWO_LIB_PATH=$(pwd)/show-o_amd/libshowo_hip_asm.so";; esac
  env $envs timeout 600 python bench.py --steps 6 --warmup 2 --no-train-leg --no-config-legs --no-cpu-baseline --no-accuracy-leg > gpurun_out/r5l/bench_$tag.json 2> gpurun_out/r5l/bench_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5l/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-6s images/s %.2f  gemm frac %.3f  avg launch ms %.4f  attn TF/s %.0f  vq conv TF/s %.0f" % (sys.argv[1], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["attention"]["achieved"], d["roofline"]["vq_conv"]["achieved"]))
PY
done
for tag in cvt asm; do
  case $tag in cvt*) envs="A=1";; asm*) envs="SHOWO_LIB_PATH=$(pwd)/show-o_amd/libshowo_hip_asm.so";; esac
  env $envs timeout 600 python bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r5l/train_$tag.json 2> gpurun_out/r5l/train_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5l/train_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-6s train ms/step %.2f" % (sys.argv[1], d["ms_per_step"]))
PY
done
