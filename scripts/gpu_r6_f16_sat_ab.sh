for tag in sat nosat sat2 nosat2; do
  case $tag in sat*) envs="A=1";; nosat*) envs="SHOWO_LIB_PATH=$(pwd)/show-o_amd/libshowo_hip_nosat.so";; esac
  env $envs timeout 600 python bench.py --precision 2 --steps 6 --warmup 2 --no-train-leg --no-config-legs --no-cpu-baseline --no-accuracy-leg > gpurun_out/r6e_$tag.json 2> gpurun_out/r6e_$tag.log
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r6e_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%-6s images/s %.2f  gemm frac %.3f  avg launch ms %.4f  attn TF/s %.0f" % (sys.argv[1], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["attention"]["achieved"]))
PY
done
timeout 600 python bench.py --precision 0 --steps 6 --warmup 2 --no-train-leg --no-config-legs --no-cpu-baseline --no-accuracy-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16   images/s %.2f gemm frac %.3f' % (d['value'], d['roofline']['frac']))"
