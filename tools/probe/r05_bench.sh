#!/bin/bash
# tools/probe/r05_bench.sh [TAG] -- the whole bench (no CPU leg) with all sub-records, one line per record.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
TAG=${1:-a}
timeout 1200 python3 bench.py --steps 20 --warmup 5 --no-cpu 2>gpurun_out/r05/bench_$TAG.err | tail -1 > gpurun_out/r05/bench_$TAG.json
python3 - gpurun_out/r05/bench_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('headline %.0f Msamples/s  %.4f ms/step  k_decode %.4f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
for k, v in d.get('sub', {}).items():
    if isinstance(v, dict) and 'value' in v:
        print('  %-32s %12.1f %-12s ms/step %-8s kernel_ms %s' % (k, v['value'] or 0, v.get('unit', ''), v.get('ms_per_step'), v.get('kernel_ms')))
PY
