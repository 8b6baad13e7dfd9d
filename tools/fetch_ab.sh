#!/bin/bash
# tools/fetch_ab.sh VARIANT... -- FETCH_SIZE / WRITE_SIZE of the fast decode pass for prebuilt
# diagnostic libraries (tools/ab_build.py), one counter per rocprofv3 run.  Diagnostic only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset DVDA_MLP_HIP_LIB; else export DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/fetch_ab/$v/$c
    mkdir -p "$OUT"
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --verify 0 ${BENCH_ARGS} > "$OUT.log" 2>&1
    python3 - "$OUT" "$v" "$c" <<'PY'
import csv, glob, sys
tot, n = 0.0, 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3] and float(r["Counter_Value"]) > 1000:
            tot += float(r["Counter_Value"]); n += 1
print(sys.argv[2], sys.argv[3], "mean KiB per fast-pass launch", tot / max(n, 1), "n", n)
PY
  done
done
