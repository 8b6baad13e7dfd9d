#!/bin/bash
# tools/ab_traffic.sh VARIANT... -- FETCH_SIZE / WRITE_SIZE of the fast-pass k_decode per prebuilt diagnostic library
# (libdvd-audio_amd/exp_<VARIANT>.so; `base` = the shipped one), one rocprofv3 --pmc pass per counter.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset DVDA_MLP_HIP_LIB; else export DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/abt_$v/$c
    mkdir -p "$OUT"
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --no-sub --verify 0 > "$OUT.log" 2>&1 < /dev/null
  done
  python3 - "$ROOT/gpurun_out/abt_$v" "$v" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
vals = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_decode<\s*\d+\s*,\s*(\w+)\s*,\s*(\w+)", row["Kernel_Name"])
        if m and m.group(1) == "false" and m.group(2) == "false" and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
f = 2.0 * 1024 * sum(vals["FETCH_SIZE"]) / max(len(vals["FETCH_SIZE"]), 1)
w = 1024 * sum(vals["WRITE_SIZE"]) / max(len(vals["WRITE_SIZE"]), 1)
print("%-6s fetch %.3f GB  write %.3f GB  (launches %d / %d)" % (sys.argv[2], f / 1e9, w / 1e9, len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"])))
PY
done
