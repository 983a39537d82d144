#!/bin/bash
# Development: build libscasml_hip_<name>.so with extra -D flags on ONE translation unit (the others are reused from the
# default build), for A/B runs through SCASML_HIP_LIB=<path>.      tools/build_variant.sh prio1 gp_eval_bf16.hip -DSCASML_GP_PRIO=1
set -e
name=$1; tu=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/scasml_gp_amd/lib
python3 -c "import sys; sys.path.insert(0, '$root'); from scasml_gp_amd import _build; _build.build_library()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -I$root/include "$@" -c $root/scasml_gp_amd/csrc/$tu -o $lib/$(basename ${tu%.*})_$name.o
objs=""
for f in plan_host picard_tree picard_tree_jax picard_tree_jax_deep gp_eval gp_eval_bf16 gp_train gp_compat gp_eval_compat_mfma dist_linalg; do
  if [ "$f.hip" == "$tu" ] || [ "$f.cpp" == "$tu" ]; then objs="$objs $lib/${f}_$name.o"; else objs="$objs $lib/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/libscasml_hip_$name.so $objs
echo $lib/libscasml_hip_$name.so
