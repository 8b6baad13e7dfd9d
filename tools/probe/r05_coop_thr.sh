#!/bin/bash
# tools/probe/r05_coop_thr.sh -- where the cooperative flush starts to pay: batches of 1..4 x 512 waves, threshold off / on
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for thr in 0 1; do
  echo "== DVDA_COOP_MIN_SEG=$thr"
  DVDA_COOP_MIN_SEG=$thr timeout 900 python tools/shape_bench.py headline1 headline2 headline3 headline4 headline6 headline 2>&1 | grep -v amdgpu
done
echo "== default threshold"
timeout 900 python tools/shape_bench.py headline2 headline3 headline 2>&1 | grep -v amdgpu
