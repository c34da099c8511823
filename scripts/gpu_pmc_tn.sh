#!/bin/bash
# LDS counters of the weight-gradient harness (tools/gemm_bench 5 5): TN kernel (transposing reads) vs the k-contiguous kernel
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_tn
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_tn -o pmc -- $R/tools/gemm_bench 5 5 > $R/gpurun_out/pmc_tn.log 2>&1
echo "rc=$?"
cd $R
python3 - > gpurun_out/pmc_tn_summary.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_tn/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'gemm' not in k: continue
    n = max(1, cnt[(k, 'SQ_LDS_IDX_ACTIVE')])
    print(k, ' launches', n, ' per launch: ', {c: round(v / n) for c, v in d.items()}, ' conflict/active', round(d.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, d.get('SQ_LDS_IDX_ACTIVE', 1)), 3))
PY
grep -E "gemm_tn|2, 8, 8" gpurun_out/pmc_tn_summary.txt
