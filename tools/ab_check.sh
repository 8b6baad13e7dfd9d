#!/bin/bash
# tools/ab_check.sh VARIANT... -- index kernels' times (rocprofv3 kernel trace of a short bench run) per prebuilt
# variant library libdvd-audio_amd/exp_<VARIANT>.so ("base" = the shipped one), after a quick parity run.  Diagnostic.
for v in "$@"; do
  if [ $v = base ]; then unset DVDA_MLP_HIP_LIB; else export DVDA_MLP_HIP_LIB=$GRAFT_REPO_ROOT/libdvd-audio_amd/exp_$v.so; fi
  echo "== $v"
  cd $GRAFT_REPO_ROOT
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "corruption or garbage or recipe or golden" 2>&1 < /dev/null | tail -1
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abcheck/$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-sub --verify 0 > /dev/null 2>&1 < /dev/null
  cd $GRAFT_REPO_ROOT
  python3 - $v <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/abcheck/%s/*/*kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('k_au_check', 'k_finalize', 'k_sync_mask', 'k_chase', 'k_decode<6')):
        print("  %-28s calls %3s avg %10.1f us min %10.1f us" % (n.split('(')[0][-28:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
