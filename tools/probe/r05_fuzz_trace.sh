#!/bin/bash
# tools/probe/r05_fuzz_trace.sh -- which kernels the fuzz_all batch's 31 ms are (rocprofv3 kernel trace of shape_bench).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/gpurun_out/r05"
cd /tmp && export TMPDIR=/tmp
for w in ${SHAPES:-fuzz_all}; do
  OUT=$ROOT/gpurun_out/r05/trace_$w
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout 600 rocprofv3 --kernel-trace -d "$OUT" -o t -- python3 "$ROOT/tools/shape_bench.py" $w > "$OUT/out.txt" 2> "$OUT/err.txt" < /dev/null
  grep -v amdgpu "$OUT/out.txt"
  python3 "$ROOT/tools/trace_summary.py" "$OUT" ${TL:-} | head -${ROWS:-24}
done
