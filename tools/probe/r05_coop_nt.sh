#!/bin/bash
# tools/probe/r05_coop_nt.sh -- the wave storing each other's runs (whole 96-byte runs side by side in one instruction)
# with store-through policies: parity subset, time and counter traffic per variant library.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
for v in ${PARITY:-}; do
  echo "== parity on exp_$v"
  DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recipe or golden or fuzz or layout or interleaved or wav or hetero or mixed" 2>&1 | tail -2
done
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_co*.so; do
  echo "== $(basename $so)"
  DVDA_MLP_HIP_LIB=$ROOT/$so timeout 600 python tools/shape_bench.py headline headline2 2>&1 | grep -v "amdgpu.ids"
done
bash tools/ab_traffic.sh base $(ls libdvd-audio_amd/exp_co*.so | sed 's/.*exp_//; s/\.so//') 2>&1 | grep fetch
