#!/bin/bash
# tools/pmc_valu_ab.sh VARIANT... -- SQ_INSTS_VALU / SQ_INSTS_SALU of the fast-pass kernel per prebuilt diagnostic
# library (tools/ab_build.py), one rocprofv3 --pmc run each.  Diagnostic only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset DVDA_MLP_HIP_LIB; else export DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so; fi
  OUT=$ROOT/gpurun_out/valu_ab/$v; mkdir -p $OUT
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-sub --verify 0 > $OUT/log 2>&1
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode<6, false, false, true" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: sum(v) / len(v) for k, v in acc.items()})
PY
done
