#!/bin/bash
# tools/probe/r05_fused_time.sh -- k_chain_fused's own duration per variant library (kernel trace of shape_bench chained2)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/gpurun_out/r05"
cd /tmp && export TMPDIR=/tmp
for so in $ROOT/libdvd-audio_amd/libdvda_mlp_hip.so $ROOT/libdvd-audio_amd/exp_*.so; do
  [ -f "$so" ] || continue
  OUT=$ROOT/gpurun_out/r05/ft_$(basename $so .so)
  rm -rf "$OUT"; mkdir -p "$OUT"
  export DVDA_MLP_HIP_LIB=$so
  timeout 600 rocprofv3 --kernel-trace -d "$OUT" -o t -- python3 "$ROOT/tools/shape_bench.py" ${SHAPE:-chained2} > "$OUT/out.txt" 2> "$OUT/err.txt" < /dev/null
  echo "== $(basename $so): $(grep -v amdgpu $OUT/out.txt | tr -s ' ' | cut -c1-120)"
  python3 "$ROOT/tools/trace_summary.py" "$OUT" | grep -E "k_chain_fused|k_decodeILi6ELb1ELb0ELb0ELb1" | cut -c1-130
done
