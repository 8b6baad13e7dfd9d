#!/bin/bash
# tools/probe/r05_trace.sh SHAPE... -- kernel trace of tools/shape_bench.py shapes: per kernel, calls and mean / total time,
# and the gaps between kernels (launch + host round trips) of one step.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for SH in "$@"; do
  OUT=$ROOT/gpurun_out/r05/trace_$SH; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/shape_bench.py $SH > $OUT/log 2>&1
  grep -v amdgpu.ids $OUT/log | tail -2
  python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# steps: from one k_init_streams to the next; take the last full step
idx = [i for i, r in enumerate(rows) if "k_init_streams" in r["Kernel_Name"]]
if len(idx) >= 3:
    a, b = idx[-2], idx[-1]
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    print("  one step: %d kernels, %.1f us from first start to next step's start" % (len(step), (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        n = r["Kernel_Name"].replace("void ", "").replace("mlp::", "").split("(")[0][:50]
        print("    +%8.1f us  gap %7.1f  run %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n))
        prev_end = e
    print("  kernels busy %.1f us" % (busy / 1e3))
PY
done
