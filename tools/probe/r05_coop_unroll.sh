#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for so in libdvd-audio_amd/exp_u*.so; do
  echo "== $(basename $so)"
  DVDA_MLP_HIP_LIB=$ROOT/$so python bench.py --no-cpu --only-sub wav24_output,heterogeneous 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
for k,v in d['sub'].items(): print(k, v.get('value'), v.get('kernel_ms'))"
done
