#!/bin/bash
# tools/probe/r05_store_policy.sh -- cache policies of the PCM stores: time (shape bench) and counter traffic per variant.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_st*.so; do
  echo "== $(basename $so)"
  DVDA_MLP_HIP_LIB=$ROOT/$so timeout 600 python tools/shape_bench.py headline stereo two 2>&1 | grep -v "amdgpu.ids"
done
bash tools/ab_traffic.sh base stnt stsc1 stsc01 2>&1 | grep fetch
