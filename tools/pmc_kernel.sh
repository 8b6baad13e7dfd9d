#!/bin/bash
# tools/pmc_kernel.sh KERNEL_SUBSTRING COUNTER... -- mean of the given PMC counters over the launches of the
# kernels whose name contains KERNEL_SUBSTRING, one rocprofv3 --pmc run of bench.py (BENCH_ARGS are passed on;
# DVDA_MLP_HIP_LIB picks a diagnostic library; PMC_SCRIPT runs another program of this repo).  Diagnostic only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_kernel/$$; mkdir -p $OUT
if [ -n "$PMC_SCRIPT" ]; then   # another program of this repo instead of bench.py: PMC_SCRIPT="tools/chain_bench.py 4096 512"
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/${PMC_SCRIPT} > $OUT/log 2>&1
else
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-sub --verify 0 ${BENCH_ARGS} > $OUT/log 2>&1
fi
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()}, "launches", {k: len(v) for k, v in acc.items()})
PY
