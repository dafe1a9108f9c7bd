#!/bin/bash
# Builds the development harnesses against LAB objects of the kernels (-DCOIN_LAB: debug switches, s_memtime stamps, variant hooks);
# the product library coin_amd/csrc/libcoin_hip.so is built WITHOUT that define and carries none of them.  Run from the repo root.
#   tools/gemm_lab                       GEMM lab (links conv_gemm / conv_gemm_p8 lab objects)
#   tools/lab/libcoin_hip_lab.so         the whole library with the lab hooks (tools/roibench.py loads it)
set -e
make -C coin_amd/csrc -j4 >/dev/null
mkdir -p tools/lab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -Wall -Wno-unused-function -DCOIN_LAB"
for f in roi_align box_head conv_gemm conv_gemm_p8 conv_gemm_s4 losses optim nms batchnorm anchors augment window_attn; do
  src=coin_amd/csrc/$f.hip
  obj=tools/lab/$f.o
  if [ ! -f $obj ] || [ $src -nt $obj ] || [ coin_amd/csrc/common.h -nt $obj ] || [ coin_amd/csrc/conv_gemm_p8.h -nt $obj ] || [ coin_amd/csrc/conv_gemm_dev.h -nt $obj ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $src -o $obj &
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 tools/lab/*.o -o tools/lab/libcoin_hip_lab.so
/opt/rocm/bin/hipcc -O2 -std=c++17 -Wno-unused-value --offload-arch=gfx950 -DCOIN_LAB -c tools/gemm_lab.hip -o tools/gemm_lab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/gemm_lab.o tools/lab/conv_gemm.o tools/lab/conv_gemm_p8.o tools/lab/conv_gemm_s4.o -o tools/gemm_lab
echo built tools/gemm_lab tools/lab/libcoin_hip_lab.so
