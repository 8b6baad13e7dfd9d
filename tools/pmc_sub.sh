#!/bin/bash
# tools/pmc_sub.sh TAG SUB[,SUB...] -- per-kernel times and HBM counter traffic of the named bench sub-records: one
# rocprofv3 kernel-trace run and two --pmc passes (FETCH_SIZE, WRITE_SIZE; each its own run, one counter per run; a --pmc run
# writes its per-dispatch rows by itself -- no trace option beside it) of
# `bench.py --only-sub ...`, then tools/pmc_sub.py: summary in gpurun_out/pmcsub_$TAG/summary.txt, entries for
# profiles/sub_traffic.json in gpurun_out/pmcsub_$TAG/sub_traffic.json.
TAG=${1:-x}; SUBS=${2:-disc_profile}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcsub_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for S in $(echo "$SUBS" | tr ',' ' '); do
  mkdir -p "$OUT/$S"
  ARGS="--steps 3 --warmup 1 --no-cpu --verify 0 --only-sub $S"
  rm -rf "$OUT/$S/trace" "$OUT/$S/fetch" "$OUT/$S/write" "$OUT/$S/bench.json"      # (no summary from an earlier run's files)
  timeout 600 python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --verify 4 --only-sub $S > "$OUT/$S/bench.json" 2> "$OUT/$S/bench.err" \
    || { echo "$S: bench failed (rc $?), sub-record skipped"; rm -f "$OUT/$S/bench.json"; continue; }
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$S/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$S/trace.log" 2>&1 \
    || { echo "$S: kernel-trace run failed (rc $?), sub-record skipped"; rm -rf "$OUT/$S/trace" "$OUT/$S/bench.json"; continue; }
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$S/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$S/fetch.log" 2>&1 \
    || { echo "$S: FETCH_SIZE run failed (rc $?)"; rm -rf "$OUT/$S/fetch"; }
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$S/write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$S/write.log" 2>&1 \
    || { echo "$S: WRITE_SIZE run failed (rc $?)"; rm -rf "$OUT/$S/write"; }
  echo "$S done"
done
python3 "$ROOT/tools/pmc_sub.py" "$OUT" "$SUBS" | tee "$OUT/summary.txt"
