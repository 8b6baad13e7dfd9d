#!/bin/bash
# tools/ab_trace.sh VARIANT... -- per-kernel average durations of one bench.py run per prebuilt diagnostic library
# (libdvd-audio_amd/exp_<VARIANT>.so; `base` = the shipped one): rocprofv3 --kernel-trace, tools/trace_summary.py.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset DVDA_MLP_HIP_LIB; else export DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so; fi
  OUT=$ROOT/gpurun_out/abtr_$v
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout 300 rocprofv3 --kernel-trace -d "$OUT" -o t -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --no-sub --verify 0 ${BENCH_ARGS} > "$OUT/bench.json" 2> "$OUT/bench.err" < /dev/null
  echo "== $v: $(python3 -c "import json,sys; j=[json.loads(l) for l in open('$OUT/bench.json') if l.startswith('{')][-1]; print('ms_per_step', j['ms_per_step'], 'bit_exact', j['config'].get('bit_exact'))")"
  python3 "$ROOT/tools/trace_summary.py" "$OUT" | head -${AB_ROWS:-12}
done
