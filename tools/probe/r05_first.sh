#!/bin/bash
# tools/probe/r05_first.sh -- round 5, first GPU call: issue rates of candidate instructions, the list of PMC counters,
# the instruction-cache counters of k_decode on the headline batch and on the fuzz batch, and a baseline bench.  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
timeout 300 build/valu_rate > gpurun_out/r05/valu_rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $ROOT/gpurun_out/r05/counters.txt 2>&1
cd "$ROOT"
grep -o "SQC_[A-Z0-9_]*\|SQ_INST[A-Z0-9_]*\|SQ_IFETCH[A-Z0-9_]*" gpurun_out/r05/counters.txt | sort -u > gpurun_out/r05/sqc_names.txt
IC=$(grep -i "ICACHE" gpurun_out/r05/sqc_names.txt | grep -v "TC_\|INV" | head -6 | tr '\n' ' ')
echo "icache counters: $IC" > gpurun_out/r05/icache.txt
for c in $IC; do
  tools/pmc_kernel.sh "k_decode<6, false, false, true, false>" $c >> gpurun_out/r05/icache.txt 2>&1
  PMC_SCRIPT="tools/shape_bench.py fuzz_fast" tools/pmc_kernel.sh "k_decode<6, false, false, false, false>" $c >> gpurun_out/r05/icache_fuzz.txt 2>&1
done
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH; do
  tools/pmc_kernel.sh "k_decode<6, false, false, true, false>" $c >> gpurun_out/r05/icache.txt 2>&1
done
for rep in 1 2; do
python3 bench.py --steps 10 --warmup 3 --no-cpu --no-sub --verify 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('base value %.0f ms/step %.4f kernel %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> gpurun_out/r05/base.txt
done
cat gpurun_out/r05/base.txt gpurun_out/r05/icache.txt
