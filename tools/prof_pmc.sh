#!/bin/bash
# tools/prof_pmc.sh TAG [bench args] -- rocprofv3 kernel trace + PMC passes of bench.py on the GPU box.
# Counters are collected in separate runs (one --pmc set per run, never combined with
# sys/hip/hsa tracing).  Summaries land in gpurun_out/pmc_$TAG/; copy what is judged to profiles/.
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"          # (a summary over an earlier run's files too would average two builds)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu --no-sub --verify 0 $*"
run() { # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
# the kernel trace runs the default step count so that its average is the steady state bench.py reports
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --no-sub --verify 0 "$@" > "$OUT/trace.log" 2>&1
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
