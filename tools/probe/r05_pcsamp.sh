#!/bin/bash
# tools/probe/r05_pcsamp.sh [SHAPE] -- PC sampling of one shape_bench.py shape (rocprofv3 beta feature): where k_decode's
# waves are when sampled.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SHAPE=${1:-headline}
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for m in stochastic host_trap; do
  OUT=$ROOT/gpurun_out/r05/pcs_$m; rm -rf $OUT; mkdir -p $OUT
  if [ $m = stochastic ]; then UNIT=cycles; IV=1048576; else UNIT=time; IV=1; fi
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $UNIT --pc-sampling-interval $IV \
      --output-format csv -d $OUT -- python3 $ROOT/tools/shape_bench.py $SHAPE > $OUT/log 2>&1
  echo "== $m rc=$?"; tail -3 $OUT/log; find $OUT -type f | head; 
  for f in $(find $OUT -name "*pc_sampling*.csv"); do wc -l $f; head -3 $f; done
done
# keep the merged output small: instruction histogram only
python3 - "$ROOT/gpurun_out/r05" <<'PY'
import csv, glob, sys, collections, os
for m in ("stochastic", "host_trap"):
    for f in glob.glob(sys.argv[1] + "/pcs_%s/**/*pc_sampling*.csv" % m, recursive=True):
        rows = csv.DictReader(open(f))
        cols = rows.fieldnames
        h = collections.Counter()
        n = 0
        for r in rows:
            n += 1
            key = tuple(r.get(c, "") for c in cols if c.lower() in ("instruction", "instruction_comment", "code_object_offset", "stall_reason", "wave_issued", "instruction_type", "code_object_id"))
            h[key] += 1
        with open(sys.argv[1] + "/pcs_%s_hist.txt" % m, "w") as o:
            o.write("columns %s\nsamples %d\n" % (cols, n))
            for k, v in h.most_common(3000):
                o.write("%8d  %s\n" % (v, " | ".join(k)))
        os.remove(f)
PY
ls -la $ROOT/gpurun_out/r05/
