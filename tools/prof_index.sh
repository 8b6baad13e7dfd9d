cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3g/prof -o idx -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-sub --verify 0 --pipeline 1 > $GRAFT_REPO_ROOT/gpurun_out/r3g/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3g/prof_bench.err
