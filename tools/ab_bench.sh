#!/bin/bash
# tools/ab_bench.sh VARIANT... -- runs bench.py once per prebuilt diagnostic library
# libdvd-audio_amd/exp_<VARIANT>.so (built on the dev box by tools/ab_build.py), same process
# settings, prints value / kernel_ms per variant.  Diagnostic only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  if [ "$v" = base ]; then unset DVDA_MLP_HIP_LIB; else export DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so; fi
  for rep in 1 2; do
    python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-sub --verify 0 ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$v', 'rep$rep', 'value', j['value'], 'ms_per_step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'])
"
  done
done
