#!/bin/bash
# tools/probe/r05_out_pad.sh -- does the distance between the titles' PCM regions matter?  (983 040 bytes per title in the
# bench batch: a multiple of 64 KB -- every lane of the chip writes to the same low 16 address bits at the same time)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
for pad in 0 128 256 1024 4352 8320 33024; do
  echo "== pad $pad"
  DVDA_BENCH_OUT_PAD=$pad timeout 600 python tools/shape_bench.py headline stereo two 2>&1 | grep -v "amdgpu.ids"
done
