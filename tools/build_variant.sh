#!/bin/bash
# build_variant.sh NAME [-DFLAG=..]...  ->  gpurun_variants/libinrfit_NAME.so   (A/B builds for tools/kbench.py via INRFIT_LIB)
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -shared -fPIC -Iinclude "$@" awesome_amd/csrc/inrfit.hip -o variants/libinrfit_$name.so
echo variants/libinrfit_$name.so
