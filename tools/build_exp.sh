#!/bin/bash
# Usage (build container, repo root): tools/build_exp.sh  -> tools/bin/libvppx_exp.so
# The library with the experiment hooks compiled in (-DVPPX_EXPERIMENT: VPPX_EXP_*, vppx_exp_*, VPPX_V3_IGNORE_LOST, the
# EXP-only VPPX_VARIANT tokens), next to the shipped one and never in its place; tools load it with VPPX_LIB=tools/bin/libvppx_exp.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/exp_obj
src=vppstereo_amd/csrc
for f in vppx_api vppx_fstream rsgm_kernels vpp_kernels handoff_kernels png_kernels; do
    if [ ! -f tools/bin/exp_obj/$f.o ] || [ $src/$f.hip -nt tools/bin/exp_obj/$f.o ] || [ $src/vppx_internal.h -nt tools/bin/exp_obj/$f.o ] || [ include/vppx.h -nt tools/bin/exp_obj/$f.o ]; then
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVPPX_EXPERIMENT -Wno-unused-function -Wno-unused-variable -c $src/$f.hip -o tools/bin/exp_obj/$f.o &
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libvppx_exp.so tools/bin/exp_obj/*.o
ls -la tools/bin/libvppx_exp.so
