#!/bin/bash
# tools/probe/one_ab.sh [SHAPE] -- tools/shape_bench.py SHAPE (default one_chained: ONE chained title, one chain, one
# workgroup of k_chain_fused) on the shipped library and every variant libdvd-audio_amd/exp_*.so.  Variants that leave
# work out decode wrong PCM: only the times count ("bit-exact False" is expected for them).  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_*.so; do
  [ -f "$so" ] || continue
  echo -n "$(basename $so): "; DVDA_MLP_HIP_LIB=$ROOT/$so timeout 300 python tools/shape_bench.py ${1:-one_chained} 2>&1 | tail -1
done
