#!/bin/bash
# tools/prof_chain.sh TAG [TITLES UNITS SUBSTREAMS LAYOUT] -- rocprofv3 kernel trace of tools/chain_bench.py (a batch whose segments all chain):
# per-kernel times of the chain passes.  Summary lands in gpurun_out/chain_$TAG/.
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/chain_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/chain_bench.py" ${2:-1024} ${3:-128} ${4:-1} ${5:-0} > "$OUT/trace.log" 2>&1
echo "rc=$?"; tail -3 "$OUT/trace.log"
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '"$OUT"'/kernel_stats.csv; head -30 {}'
