#!/bin/bash
# Development: libscasml_hip.so variants with parts of gp_eval_f16_kernel compiled out (-DSCASML_GP_ABLATE=mask:
# 1 no MFMA, 2 no epilogue, 8 stage only the first tiles), as scasml_gp_amd/lib/abl<mask>/libscasml_hip.so.
# Use with SCASML_HIP_LIB=<path> python bench.py ...   (run after the normal build: the other objects are reused)
set -e
cd "$(dirname "$0")/.."
L=scasml_gp_amd/lib
for m in "$@"; do
  mkdir -p $L/abl$m
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Iinclude \
      -DSCASML_GP_ABLATE=$m -c scasml_gp_amd/csrc/gp_eval_f16.hip -o $L/abl$m/gp_eval_f16.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl$m/libscasml_hip.so $L/abi.o $L/picard_tree.o $L/gp_eval.o \
      $L/gp_eval_bf16.o $L/abl$m/gp_eval_f16.o $L/gp_train.o
done
