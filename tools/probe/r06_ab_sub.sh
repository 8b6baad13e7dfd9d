#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = base ]; then unset DVDA_MLP_HIP_LIB; else export DVDA_MLP_HIP_LIB=$ROOT/libdvd-audio_amd/exp_$v.so; fi
  python3 $ROOT/bench.py --steps 30 --warmup 3 --no-cpu --no-disc --verify 0 --only-sub two_substreams,chained_two_substreams,disc_profile 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); s=j['sub']; print('$v', 'headline', j['value'], ' '.join('%s %.0f' % (k, s[k]['value']) for k in ('two_substreams','chained_two_substreams','disc_profile')))
"
done
