#!/bin/bash
# tools/probe/r05_pmc_shapes.sh -- instruction counts and wave cycles of the kernels behind the two-substream and the chained
# shapes (one rocprofv3 --pmc run each).  Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
PMC_SCRIPT="tools/shape_bench.py two" tools/pmc_kernel.sh "k_decode<4, true, false, true, false" $C
PMC_SCRIPT="tools/shape_bench.py chained2" tools/pmc_kernel.sh "k_decode<6, true, false, false, true" $C
PMC_SCRIPT="tools/shape_bench.py chained2" tools/pmc_kernel.sh "k_chain_fused" $C
PMC_SCRIPT="tools/shape_bench.py chained1" tools/pmc_kernel.sh "k_decode<6, false, false, false, true" $C
