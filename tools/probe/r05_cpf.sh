#!/bin/bash
# tools/probe/r05_cpf.sh -- the cooperative prefetch variants: parity subset on each, then shapes.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
for v in ${VARIANTS:-cpf}; do
  echo "== parity on exp_$v"
  DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recipe or golden or fuzz or bit_reader or code_book or two_substreams or deferred or chained" 2>&1 | tail -3
done
SHAPES=${SHAPES:-headline stereo fuzz_fast chained2 two}
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_*.so; do
  [ -f "$so" ] || continue
  echo "== $(basename $so)"
  DVDA_MLP_HIP_LIB=$ROOT/$so timeout 600 python tools/shape_bench.py $SHAPES 2>&1 | grep -v "amdgpu.ids"
done
