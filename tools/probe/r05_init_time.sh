#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05 build
/opt/rocm/bin/hipcc -O2 -Iinclude tools/probe/init_time.cpp -w -o build/init_time -Llibdvd-audio_amd -ldvda_mlp_hip -Wl,-rpath,$ROOT/libdvd-audio_amd || exit 1
for i in 1 2; do s=$(date +%s.%N); ./build/init_time; e=$(date +%s.%N); echo "wall $(echo "$e - $s" | bc) s"; done
