#!/bin/bash
# tools/probe/r05_ab.sh [quick] -- parity subset, then the headline and a few shapes on the shipped library and on every
# variant library libdvd-audio_amd/exp_*.so.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
if [ "$1" != "nopar" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recipe or golden or fuzz or bit_reader or code_book or two_substreams or deferred" 2>&1 | tail -3
fi
SHAPES=${SHAPES:-headline fuzz_fast chained2 two}
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_*.so; do
  [ -f "$so" ] || continue
  echo "== $(basename $so)"
  DVDA_MLP_HIP_LIB=$ROOT/$so timeout 600 python tools/shape_bench.py $SHAPES 2>&1 | grep -v "amdgpu.ids"
done
