#!/bin/bash
# Builds tools/gemm_lab (development harness) against the in-tree objects of libcoin_hip; run from the repo root.
set -e
make -C coin_amd/csrc -j4 >/dev/null
/opt/rocm/bin/hipcc -O2 -std=c++17 -Wno-unused-value --offload-arch=gfx950 -c tools/gemm_lab.hip -o tools/gemm_lab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/gemm_lab.o coin_amd/csrc/conv_gemm.o coin_amd/csrc/conv_gemm_p8.o -o tools/gemm_lab
echo built tools/gemm_lab
