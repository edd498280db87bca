#!/bin/bash
# experiment builds of libtlxmi.so: gemm_stream.hip with -DTLXMI_EXP=n, everything else from csrc/build
set -e
cd "$(dirname "$0")/../../tlxcv_amd/csrc"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -DTLXMI_EXP=$n -c gemm_stream.hip -o /tmp/gs_exp$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probe/libtlxmi_e$n.so $(ls build/*.o | grep -v gemm_stream.o) /tmp/gs_exp$n.o
done
