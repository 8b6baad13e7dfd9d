cd $GRAFT_REPO_ROOT
for n in 512 1024 2048 4096; do
cd /tmp; export TMPDIR=/tmp; OUT=/tmp/tr_$n; rm -rf $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/shape_bench.py one_chained_$n > $OUT.log 2>&1
grep one_chained $OUT.log
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_chain_filter" in r["Name"] or "k_coop<true" in r["Name"]:
        print("   %-30s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"].split("(")[0][-30:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
