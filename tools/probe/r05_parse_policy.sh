#!/bin/bash
# tools/probe/r05_parse_policy.sh -- the parse pass's cooperative plane stores written through: time and counter traffic
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
bash tools/probe/r05_fused_time.sh 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
for so in $ROOT/libdvd-audio_amd/libdvda_mlp_hip.so $ROOT/libdvd-audio_amd/exp_psc1.so; do
  export DVDA_MLP_HIP_LIB=$so
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/r05/pp_$(basename $so .so)_$c; rm -rf $OUT; mkdir -p $OUT
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $ROOT/tools/shape_bench.py chained2 > $OUT/log 2>&1 < /dev/null
    python3 - $OUT $c $(basename $so) <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if ("k_decode<6, true, false, false, true" in k or "k_chain_fused" in k) and r["Counter_Name"] == sys.argv[2]:
            acc[k.split("(")[0][-50:]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = sum(v) / len(v) * 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1) / 1e9
    print(sys.argv[3], sys.argv[2], k, "%.3f GB" % m)
PY
  done
done
