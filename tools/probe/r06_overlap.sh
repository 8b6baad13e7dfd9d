#!/bin/bash
# tools/probe/r06_overlap.sh -- the parse pass of one chained batch beside the fused chain pass of another: bench.py's
# chained_two_substreams sub-record with 2 and 3 decode contexts in flight, (a) as the bench runs it -- slot k's decode
# waits for slot k-1's -- and (b) with that edge removed (DVDA_BENCH_FREE_OVERLAP=1): the hardware schedules the two
# decodes' kernels side by side.  Then a kernel trace of both, for the kernels' own durations alone and overlapped.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06/overlap
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for F in 2 3; do
  for FREE in 0 1; do
    DVDA_BENCH_FREE_OVERLAP=$FREE python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu --no-disc --only-sub chained_two_substreams --chain-flight $F \
      > "$OUT/bench_f${F}_free${FREE}.json" 2> "$OUT/bench_f${F}_free${FREE}.err"
    python3 - "$OUT/bench_f${F}_free${FREE}.json" $F $FREE <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r = d["sub"]["chained_two_substreams"]
print("flight %s free %s: one context %.1f Msamples/s %.3f ms/step | in flight: %.1f Msamples/s %.3f ms/step" % (
    sys.argv[2], sys.argv[3], r["value"], r["ms_per_step"], r["in_flight"]["value"], r["in_flight"]["ms_per_step"]))
PY
  done
done
for FREE in 0 1; do
  DVDA_BENCH_FREE_OVERLAP=$FREE timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_free$FREE" -- \
    python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu --no-disc --verify 0 --only-sub chained_two_substreams --chain-flight 2 > "$OUT/trace_free$FREE.log" 2>&1
  echo "== kernel trace, 2 contexts in flight, free=$FREE"
  find "$OUT/trace_free$FREE" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_free$FREE.csv"
  python3 - "$OUT/kernel_stats_free$FREE.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 1e6 and "mlp::" in r["Name"]:
        print("  %-66s calls %4s avg %8.3f ms min %8.3f max %8.3f" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
done
