cd $GRAFT_REPO_ROOT
export DVDA_MLP_HIP_LIB=$GRAFT_REPO_ROOT/libdvd-audio_amd/exp_fix.so
tools/pmc_kernel.sh "k_decode<6, false, false, true, false, true>" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
tools/pmc_kernel.sh "k_decode<6, false, false, true, false, false>" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES
unset DVDA_MLP_HIP_LIB
tools/pmc_kernel.sh "k_decode<6, false, false, true, false, false>" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
