#!/bin/bash
# tools/probe/fused_ab.sh -- chained bench batch (4096 x 512, one substream, frame-major) on the variant libraries
# libdvd-audio_amd/exp_*.so that exist (built beforehand with _build.build_hip(defines=..., out=...)).  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_*.so; do
  [ -f "$so" ] || continue
  for rep in 1 2; do
    echo -n "$(basename $so): "; DVDA_MLP_HIP_LIB=$ROOT/$so timeout 300 python tools/chain_bench.py 4096 512 ${1:-1} 1 2>&1 | tail -1
  done
done
