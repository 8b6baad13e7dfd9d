#!/bin/bash
# tools/prof_sub.sh TAG SUB[,SUB...] -- rocprofv3 kernel trace of bench.py restricted to the named sub-records:
# per-kernel times of what those shapes run.  Summary lands in gpurun_out/sub_$TAG/kernel_stats.csv.
TAG=${1:-x}; SUBS=${2:-disc_profile}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sub_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu --verify 0 --only-sub "$SUBS" > "$OUT/trace.log" 2>&1
echo "rc=$?"
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 2e5:
        print("%-72s calls %4s avg %9.3f ms  total %9.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
