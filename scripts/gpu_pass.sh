#!/bin/bash
# One parameterised GPU pass (round 6; replaces the one-shot scripts of round 5).  usage: scripts/gpurun_built.sh <timeout> 'bash scripts/gpu_pass.sh <tag> <step> [<step> ...]'
# steps:  tests[:<pytest -k expr>|:<file>]  smoke  bench  bench:<extra args>  train  mmu  vq  t2i512  batch1  prof-t2i  prof-train  prof-mmu  pmc  pmc-mfma
# Every step writes gpurun_out/<tag>_<step>.*; copy what is cited into profiles/.
TAG=$1; shift
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  case $name in
    tests)
      if [ -f "$arg" ]; then sel="$arg"; kexpr=""; else sel="tests"; kexpr="$arg"; fi
      timeout 2400 python -m pytest $sel -m gpu -x -q -rP ${kexpr:+-k "$kexpr"} > gpurun_out/${TAG}_tests.log 2>&1
      grep -E "^\[parity\]|passed|failed|error|Error|assert" gpurun_out/${TAG}_tests.log | tail -60 ;;
    smoke) timeout 600 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log ;;
    bench) timeout 1500 python bench.py $arg > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json ;;
    train) timeout 900 python bench.py --workload train --steps 8 --warmup 2 $arg > gpurun_out/${TAG}_train_bench.json 2> gpurun_out/${TAG}_train_bench.err; tail -c 800 gpurun_out/${TAG}_train_bench.json ;;
    mmu) timeout 900 python bench.py --workload mmu $arg > gpurun_out/${TAG}_mmu_bench.json 2> gpurun_out/${TAG}_mmu_bench.err; tail -c 1200 gpurun_out/${TAG}_mmu_bench.json ;;
    vq) timeout 900 python bench.py --workload vq $arg > gpurun_out/${TAG}_vq_bench.json 2> gpurun_out/${TAG}_vq_bench.err; tail -c 800 gpurun_out/${TAG}_vq_bench.json ;;
    t2i512) timeout 900 python bench.py --workload t2i512 $arg > gpurun_out/${TAG}_t2i512_bench.json 2> gpurun_out/${TAG}_t2i512_bench.err; tail -c 800 gpurun_out/${TAG}_t2i512_bench.json ;;
    batch1) timeout 900 python bench.py --batch 1 $arg > gpurun_out/${TAG}_batch1_bench.json 2> gpurun_out/${TAG}_batch1_bench.err; tail -c 800 gpurun_out/${TAG}_batch1_bench.json ;;
    prof-t2i|prof-train|prof-mmu)
      wl=${name#prof-}; extra="--steps 3 --warmup 2 --no-cpu-baseline --no-accuracy-leg --no-config-legs --no-train-leg"; [ $wl = train ] && extra="--workload train --steps 3 --warmup 1"; [ $wl = mmu ] && extra="--workload mmu"
      [ $wl = train ] && export SHOWO_GEMM_TUNE=0
      (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${TAG}_prof_$wl -o p -- python $OLDPWD/bench.py $extra $arg > $OLDPWD/gpurun_out/${TAG}_prof_$wl.log 2>&1)
      unset SHOWO_GEMM_TUNE
      find gpurun_out/${TAG}_prof_$wl -type f ! -name "*stats*" -size +2M -delete
      f=$(find gpurun_out/${TAG}_prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${wl}_kernel_stats.csv && head -12 "$f" ;;
    pmc) bash scripts/gpu_pmc3.sh ${TAG}
         python tools/pmc_agg.py gpurun_out/${TAG}_pmc_traffic_all_kernels.json /tmp/pmc_1 /tmp/pmc_2 | head -8 ;;
    pmc-mfma)  # MFMA busy cycles per kernel (GEMM launches, attention, conv3t of the VQ decode): one SQ / GRBM pass over the bench command
      rm -rf /tmp/pmc_mfma
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_mfma -o pmc -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg --no-accuracy-leg --no-config-legs --roofline-steps 0 --no-events > $OLDPWD/gpurun_out/${TAG}_pmc_mfma.log 2>&1)
      python tools/pmc_agg.py gpurun_out/${TAG}_pmc_mfma_agg.json /tmp/pmc_mfma > /dev/null
      python tools/pmc_mfma.py ${TAG} gpurun_out/${TAG}_pmc_mfma_agg.json gpurun_out/${TAG}_bench_mfma_util.json ;;
    *) bash -c "$step" > gpurun_out/${TAG}_cmd.log 2>&1; tail -30 gpurun_out/${TAG}_cmd.log ;;
  esac
done
