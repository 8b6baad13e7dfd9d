#!/bin/bash
# tools/probe/r05_stamp.sh -- parity subset on the shipped build, headline A/B, and where a wave's time goes (stamp build) at
# two waves per SIMD and with half the SIMDs holding one wave.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recipe or golden or fuzz or bit_reader or code_book or two_substreams or deferred" 2>&1 | tail -3
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_r4.so; do
  echo "== $(basename $so)"
  DVDA_MLP_HIP_LIB=$ROOT/$so timeout 600 python tools/shape_bench.py headline1 headline4 headline fuzz_fast two 2>&1 | grep -v "amdgpu.ids"
done
echo "== stamps, 2048 titles"; timeout 300 python tools/stamp_run.py 1 2048 2>&1 | grep -v amdgpu.ids
echo "== stamps, 512 titles"; timeout 300 python tools/stamp_run.py 1 512 2>&1 | grep -v amdgpu.ids
echo "== r4 stamps, 2048 titles"; STAMP_LIB=exp_stampr4.so timeout 300 python tools/stamp_run.py 1 2048 2>&1 | grep -v amdgpu.ids
