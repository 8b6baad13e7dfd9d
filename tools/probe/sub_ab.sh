#!/bin/bash
# tools/probe/sub_ab.sh SUB[,SUB...] -- the named bench sub-records on the shipped library and on every variant library
# libdvd-audio_amd/exp_*.so (built beforehand with _build.build_hip(defines=..., out=...)).  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
for so in libdvd-audio_amd/libdvda_mlp_hip.so libdvd-audio_amd/exp_*.so; do
  [ -f "$so" ] || continue
  echo -n "$(basename $so): "
  DVDA_MLP_HIP_LIB=$ROOT/$so timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu --verify 4 --only-sub "$1" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline %.0f (%.3f ms kernel)' % (d['value'], d['roofline']['kernel_ms']), ' '.join('%s %.0f' % (k, v['value']) for k, v in d.get('sub', {}).items() if 'host' not in k))"
done
